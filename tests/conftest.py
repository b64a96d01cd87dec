import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The GPU boxes have hundreds of host cores; PyTorch's default of one intra-op thread per core
# makes the oracle's many small CPU ops crawl (measured: 166 s for one bs=32 oracle step with
# 256 threads vs 1.5 s with 8).  Bound it.
torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return torch.cuda.is_available()


def load_golden(name):
    """(meta, tensors) of a fixture written by tests/golden/make_golden.py."""
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tensors = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    return meta, tensors


def load_tiny_sd():
    z = np.load(os.path.join(GOLDEN, "tiny_sd.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])).float() for k in z.files}


def golden_param_dict(meta, tensors, which="adapter"):
    """Backbone state-dict (tiny) + the fixture's adapter tensors."""
    sd = load_tiny_sd()
    for k, v in tensors.items():
        if k.startswith(which + "/"):
            sd[k[len(which) + 1:]] = v.float().clone()
    return sd


def rel_err(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def kind_of(name):
    """a tensor's kind: its name with the block index replaced by *"""
    import re
    return re.sub(r"resblocks\.\d+\.", "resblocks.*.", name)


FLOOR_C = 3.0         # the ONE constant over the reference-recorded bf16 floor (tests/test_gpu_refinit.py has the reasoning)


def reference_floors(meta, fp8=False):
    """{quantity: floor} of a fixture that carries the reference-recorded bf16 floor (``floor`` in its .json, written by
    tests/golden/make_golden.py --bf16-weights / --floor-random: the imported reference re-run with bf16 frozen weights and with bf16
    contraction operands, deviation from its own f32 run): the larger of the two legs (fp8 engine: its fp8 leg as well); for the
    gradient kinds per TENSOR KIND (largest over the blocks).  None when the fixture has no floor."""
    f = meta.get("floor")
    if not f:
        return None
    legs = [f["weights"], f["operands"]] + ([f["fp8"]] if fp8 else [])
    out = {"logits": max(l["logits"] for l in legs), "loss0": max(l["loss0"] for l in legs),
           "loss_traj": max(max(l["loss_traj"]) for l in legs), "loss_traj_steps": [max(l["loss_traj"][i] for l in legs) for i in range(len(legs[0]["loss_traj"]))]}
    for kind in ("grad", "grad_last", "delta"):
        out[kind] = {}
        for l in legs:
            for n, v in l[kind].items():
                out[kind][kind_of(n)] = max(out[kind].get(kind_of(n), 0.0), v)
        out[kind + "_all"] = max(l[kind + "_all"] for l in legs)
    return out


def floor_gate(stated, floor, cap=None):
    """max(stated gate, FLOOR_C x reference floor), optionally capped (per-tensor gates stay below the score of a zero tensor)"""
    g = max(stated, FLOOR_C * floor)
    return min(g, cap) if cap is not None else g


def sign_projections(v, index, k=32):
    """The k seeded +-1 projections of tests/golden/make_golden.py:sign_projections (same generator, same order)."""
    g = torch.Generator(device="cpu"); g.manual_seed(100003 + index)
    s = torch.randint(0, 2, (k, v.numel()), generator=g, dtype=torch.int8).double() * 2 - 1
    return s @ v.double().flatten()


def proj_rel_err(v, index, ref_proj, ref_norm):
    """Estimate of |v - ref| / |ref| from the recorded projections of ref: E[(<s,v> - <s,ref>)^2] = |v - ref|^2."""
    d = sign_projections(v, index, ref_proj.numel()) - ref_proj.double()
    return float(torch.sqrt((d * d).mean()) / (float(ref_norm) + 1e-30))


@pytest.fixture(scope="session")
def gpu_required():
    if not has_gpu():
        pytest.skip("no GPU")
