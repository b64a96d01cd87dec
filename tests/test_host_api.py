"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every symbol
the header declares; the host mirror lays parameters out exactly like the reference."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden

from pevit_amd import _lib
from pevit_amd.engine import adapter_param_spec
from pevit_amd.synth import ARCHS


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def header_functions():
    src = open(os.path.join(ROOT, "include", "pevit_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pevit_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pevit_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names            # the ctypes table mirrors the header one to one


def test_context_sizes_and_param_counts(lib):
    """Trainable-parameter counts of the engine layout == the reference's (README.md:84-87)."""
    expect = {("ViT-B/32", "kadaptation", 4): 50176, ("ViT-B/16", "kadaptation", 4): 50176,
              ("ViT-L/14", "kadaptation", 4): 126976, ("ViT-B/32", "lora", 4): 147456,
              ("ViT-B/32", "lora", 8): 294912, ("ViT-L/14", "lora", 4): 393216}
    for (arch_name, method, r), n in expect.items():
        a = ARCHS[arch_name]
        d = _lib.PevitDims(a.width, a.layers, a.patch, a.resolution, a.embed_dim, _lib.METHOD_IDS[method], r, 100)
        ctx = C.c_void_p()
        assert lib.pevit_ctx_create(C.byref(d), C.byref(ctx)) == 0, lib.pevit_last_error()
        assert lib.pevit_num_tower_params(ctx) == n
        assert lib.pevit_num_params(ctx) == n + a.embed_dim * 100 + 100
        assert lib.pevit_arena_bytes(ctx) > 0 and lib.pevit_workspace_bytes(ctx, 8) > 0
        spec_total = sum(torch.Size(s).numel() for _, s, tr in adapter_param_spec(method, a.width, a.layers, r) if tr)
        assert spec_total == n
        lib.pevit_ctx_destroy(ctx)


def test_grad_mask_marks_dead_v_adapters(lib):
    a = ARCHS["ViT-B/32"]
    d = _lib.PevitDims(a.width, a.layers, a.patch, a.resolution, a.embed_dim, 0, 4, 100)
    ctx = C.c_void_p()
    assert lib.pevit_ctx_create(C.byref(d), C.byref(ctx)) == 0
    n = lib.pevit_num_params(ctx)
    m = (C.c_ubyte * n)()
    assert lib.pevit_param_grad_mask(ctx, m, n) == 0
    dead = n - sum(m)
    assert dead == 12 * 1536            # v_proj_adapter1_left/right: 1,536 of 3,840 per layer
    assert lib.pevit_param_grad_mask(ctx, m, n - 1) != 0
    assert b"size mismatch" in lib.pevit_last_error()
    lib.pevit_ctx_destroy(ctx)


def test_bad_dims_fail_loudly(lib):
    ctx = C.c_void_p()
    for bad in (_lib.PevitDims(100, 12, 32, 224, 512, 0, 4, 100), _lib.PevitDims(768, 12, 32, 225, 512, 0, 4, 100),
                _lib.PevitDims(768, 12, 32, 224, 512, 9, 4, 100), _lib.PevitDims(768, 12, 32, 224, 512, 1, 64, 100),
                _lib.PevitDims(768, 12, 8, 224, 512, 0, 4, 100)):
        assert lib.pevit_ctx_create(C.byref(bad), C.byref(ctx)) != 0
        assert len(lib.pevit_last_error()) > 0
    # unbound context refuses to run
    d = _lib.PevitDims(128, 2, 16, 48, 64, 0, 4, 10)
    assert lib.pevit_ctx_create(C.byref(d), C.byref(ctx)) == 0
    assert lib.pevit_transformer_forward(ctx, None, None, None, 4, 1) != 0
    assert b"not bound" in lib.pevit_last_error()
    lib.pevit_ctx_destroy(ctx)


@pytest.mark.parametrize("case", ["tiny_kadaptation", "tiny_lora", "tiny_lora_r8", "tiny_adapter", "tiny_compacter"])
def test_param_spec_order_matches_reference_named_parameters(case):
    meta, t = load_golden(case)
    spec = adapter_param_spec(meta["method"], 128, 2, meta["lora_r"])
    trainable = [n for n, s, tr in spec if tr]
    assert trainable == meta["trainable_names"]
    for n, s, tr in spec:
        assert tuple(t["adapter/" + n].shape) == tuple(s), n
    # every spec name sits where the reference's named_parameters() has it
    order = [n for n in meta["all_names"] if n in {x for x, _, _ in spec}]
    assert order == [n for n, _, _ in spec]


def test_engine_requires_gpu_and_never_falls_back():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pevit_amd.engine import HipEngine
    with pytest.raises(_lib.PevitError):
        HipEngine(ARCHS["tiny-128"], "kadaptation", 10, 4)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pevit_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f
