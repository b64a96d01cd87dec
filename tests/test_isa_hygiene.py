"""Compiler-behaviour guard for the HBM-bound kernels (no GPU needed: hipcc cross-compiles gfx950 listings).

hipcc waits with ``s_waitcnt vmcnt(0)`` (a) right behind a load whose destination has a second definition at a branch join
(``v = 0; if (ok) v = load``): one memory round trip per load; (b) before the first use, after a conditional store, of a value
loaded earlier -- and on gfx950 vmcnt counts stores, so that wait is a store round trip.  The kernels below were rewritten so
that neither pattern occurs (profiles/NOTES_gemm.md (r03_gemm_experiments) 5d); this test keeps it that way by scanning the ``-S`` listing
like scripts/isa_serial_loads.py / scripts/isa_store_waits.py do."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pevit_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _kernels(listing):
    out, name, ins = {}, None, []
    for line in listing.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                out[name] = ins
            name, ins = m.group(1), []
        elif line.startswith("\t") and not line.strip().startswith((".", ";")):
            ins.append(" ".join(line.strip().split()[:2]))
    if name:
        out[name] = ins
    return out


def _scan(ins):
    loads = [i for i, x in enumerate(ins) if x.startswith(("global_load", "buffer_load")) and "lds" not in x]
    drained = sum(1 for i in loads if any(y.startswith("s_waitcnt vmcnt(0)") for y in ins[i + 1:i + 4]))
    st = [i for i, x in enumerate(ins) if x.startswith(("global_store", "buffer_store"))]
    between = [i for i, x in enumerate(ins) if st and st[0] < i < st[-1] and x.startswith("s_waitcnt vmcnt(0)")]
    loads_between = [i for i in loads if st and st[0] < i < st[-1]]
    return drained, len(between), len(loads_between)


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    out = {}
    for f in ("norm", "lowrank", "attention"):
        s = d / (f + ".s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "--cuda-device-only", "-S",
                        os.path.join(CSRC, f + ".hip"), "-o", str(s)], check=True, capture_output=True, timeout=900)
        out[f] = _kernels(s.read_text())
    return out


# (file, mangled-name fragment, loads drained right behind the request, vmcnt(0) between stores, loads between stores): upper bounds
CASES = [
    # LayerNorm: the E = 768 instances the towers run (NV = E / 256 = 3; backward: the bf16 gradient-stream form and the f32-residual
    # form) and the generic any-E ones (NV = 0)
    ("norm", "ln_fwd_kernelIDF16bLi3E", 0, 0, 0),
    ("norm", "ln_fwd_kernelIDF16bLi0E", 0, 0, 0),
    ("norm", "ln_bwd_kernelIDF16bDF16bLb1ELi3E", 1, 0, 0),
    ("norm", "ln_bwd_kernelIDF16bDF16bLb0ELi3E", 1, 0, 0),
    ("norm", "ln_bwd_kernelIDF16bDF16bLb0ELi0E", 1, 0, 0),
    ("lowrank", "delta_add_kernelIDF16b", 0, 0, 0),
    # the next slab's X panel is requested between two slabs' stores, and the one explicit vmcnt(0) that brings it in sits there too
    ("lowrank", "lowrank_grad_kernel", 0, 1, 8),
    ("attention", "attn_fwd_kernelILi2ELi4E", 0, 0, 0),
    # one vmcnt(0): hipcc places it behind the LDS barrier between pass A (dQ stores) and pass B (which reads the delta pass A wrote to
    # LDS), whatever form the barrier takes (s_waitcnt lgkmcnt(0) + s_barrier, fences, inline asm); with three workgroups per CU the
    # kernel's in-step time did not move (18.6 -> 18.5 us with 12 % fewer bytes)
    ("attention", "attn_bwd_kernelILi2ELb1ELi4E", 0, 1, 0),
]


@pytest.mark.parametrize("f,frag,max_drained,max_waits,max_loads", CASES)
def test_no_drain_behind_loads_and_no_wait_between_stores(listings, f, frag, max_drained, max_waits, max_loads):
    names = [n for n in listings[f] if frag in n]
    assert names, f"kernel {frag} not found in {f}.hip"
    drained, waits, loads_between = _scan(listings[f][names[0]])
    assert drained <= max_drained, f"{frag}: {drained} loads are drained with vmcnt(0) right behind their request"
    assert waits <= max_waits, f"{frag}: {waits} s_waitcnt vmcnt(0) between the first and the last store (a store round trip each on gfx950)"
    assert loads_between <= max_loads, f"{frag}: {loads_between} loads between the first and the last store"
