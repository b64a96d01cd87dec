"""The roofline object of bench.py reads the HBM-side traffic of the GEMM family from a rocprofv3 PMC pass committed under
profiles/ (FETCH_SIZE / WRITE_SIZE cannot be read from inside the benchmarked process).  That number is only valid for the
kernel sources it was measured on: the committed tree must never carry a pass taken with other sources (bench.py would then
print `traffic: null, traffic_stale: true`; `bench.py --strict-traffic` exits non-zero).  Refresh with
scripts/run_pmc_passes.sh on the GPU box and copy gpurun_out/hbm_traffic.json to profiles/."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_hash", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_committed_pmc_pass_was_taken_with_these_kernel_sources():
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        entry = json.load(f)["ViT-B/32|kadaptation|bs128"]
    assert entry["kernels_hash"] == bench.kernels_hash(), (
        "profiles/hbm_traffic.json was measured on other kernel sources than pevit_amd/csrc/ holds now: "
        "re-run scripts/run_pmc_passes.sh on the GPU box and commit the refreshed profiles/")
    assert entry["gemm"]["hbm_bytes_per_launch"] > 0
    traffic, how, traffic_all = bench.pmc_traffic("ViT-B/32", "kadaptation", 128)
    assert traffic == entry["gemm"]["hbm_bytes_per_launch"], how
    assert traffic_all == entry["all_kernels"]["hbm_bytes_per_step"] and traffic_all > 98 * traffic      # every dispatch of a step >= its 98 GEMMs
