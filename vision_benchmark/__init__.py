"""Import-name shim: ``vision_benchmark.<x>`` resolves to ``pevit_amd.<x>``.

Scripts written against the reference (``from vision_benchmark.evaluation.kadaptation_clip import kadapt_clip``,
``from vision_benchmark.evaluation.model import build_model`` ...) run unchanged on the HIP engine.  The modules
are aliased, not re-executed, so there is exactly one copy of the engine cache and of the loaded library.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_PREFIX, _REAL = "vision_benchmark.", "pevit_amd."


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        try:
            mod = importlib.import_module(_REAL + fullname[len(_PREFIX):])
        except ModuleNotFoundError as e:
            if e.name and e.name.startswith(_REAL):
                return None
            raise
        return importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(mod, "__path__"))

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_PREFIX):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _Alias) for f in sys.meta_path):
    sys.meta_path.insert(0, _Alias())
