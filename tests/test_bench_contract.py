"""bench.py's launch contract, the parts that need no GPU: `python bench.py --gpus N` without a launcher re-executes itself under
torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous); every rank pins itself to its own slice of the host's cores; the
roofline helpers agree with SURVEY 8d."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_without_a_launcher_reexecutes_under_torch_distributed_run(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                      # the launcher's status is the process's status
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_that_disagrees_with_gpus_is_refused(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no affinity API")
def test_every_rank_pins_itself_to_its_own_slice_of_the_cores():
    bench = _bench()
    before = os.sched_getaffinity(0)
    try:
        cores = sorted(before)
        if len(cores) < 2:
            pytest.skip("one core")
        a = bench.pin_rank_to_cores(0, 2)
        os.sched_setaffinity(0, before)
        b = bench.pin_rank_to_cores(1, 2)
        assert a and b and not (set(a) & set(b)) and set(a) | set(b) <= set(cores)
        assert os.sched_getaffinity(0) == set(b)
    finally:
        os.sched_setaffinity(0, before)
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def test_step_floors_follow_the_survey_formula():
    bench = _bench()
    g = bench.train_gflop_per_image(768, 12, 32, 224, 512, 100, 32, True)
    assert abs(g - 17.673449472) < 1e-6
    assert abs(g * 128 / bench.PEAK_TFLOPS_BF16 - 0.9048806) < 1e-5   # ms of MFMA work per bs-128 step (roofline.mfma_floor_ms)
    assert abs(bench.PEAK_TFLOPS_BF16 * 1e12 / (bench.PEAK_HBM_TBS * 1e12) - 312.5) < 1e-9   # the ridge


def test_flush_helper_and_the_dp_schedule_option(monkeypatch, capsys):
    """The JSON line must be the last line of stdout even when a C library has buffered output of its own (the collective library's
    banner): `_flush_c_stdio` empties the process's C stdio buffers and never raises; `--dp-exchange` offers the three schedules and
    defaults to timing the in-stream AND the overlapped one (and the re-exec for `--gpus N` hands the choice on to the ranks)."""
    bench = _bench()
    bench._flush_c_stdio()                                   # callable without a GPU, idempotent
    bench._flush_c_stdio()
    src = open(os.path.join(ROOT, "bench.py")).read()
    # N > 1 default: BOTH the in-stream exchange and the overlapped one are timed, the faster is `value`, both are reported
    assert 'choices=["auto", "single", "staged", "pipelined"], default="auto"' in src
    for field in ("exchange_us_per_step", "per_rank_ms_per_step", "exchange_schedules"):
        assert f'"{field}"' in src
    # the print of the line comes after the process group is gone and the buffers are flushed
    tail = src[src.rindex("destroy_process_group()"):]
    assert "_flush_c_stdio()" in tail and tail.index("_flush_c_stdio()") < tail.index("print(json.dumps(out)")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"] = cmd
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--dp-exchange", "staged"])
    with pytest.raises(SystemExit):
        bench.main()
    assert "--dp-exchange" in seen["cmd"] and "staged" in seen["cmd"]


@pytest.mark.gpu
def test_concurrent_runs_side_measurement_is_bit_identical_to_solo_runs():
    """bench.py's `concurrent_runs` object (round 6): K engine contexts on K streams at batch 64 (K = 1, 2, 3) and 128 (K = 1, 2);
    the aggregate rate per K and -- what makes the figure mean anything -- every concurrent run bit-identical to the same run alone."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pevit_amd.synth import ARCHS, synth_state_dict
    bench = _bench()
    arch = ARCHS["tiny-256"]
    out = bench.concurrent_runs_throughput(torch.device("cuda", 0), synth_state_dict(arch, seed=2, text_tower=False), arch, steps=6)
    for bs, ks in (("bs64", ("1", "2", "3")), ("bs128", ("1", "2"))):
        assert out[bs]["bit_identical_to_solo"] is True
        assert tuple(out[bs]["aggregate_images_per_sec_by_runs"]) == ks and all(v > 0 for v in out[bs]["aggregate_images_per_sec_by_runs"].values())
