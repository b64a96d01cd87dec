#!/usr/bin/env python3
"""HBM traffic of the GEMM kernel family from two rocprofv3 PMC passes of bench.py.

FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit one pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots"),
so the same command is profiled twice:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py --steps 3 --warmup 1 \
        --no-cpu-baseline --pmc-calib
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py --steps 3 --warmup 1 \
        --no-cpu-baseline --pmc-calib

Both counters are in KiB.  gfx950 correction (same guide, "HBM"): FETCH_SIZE tallies the 128-byte requests
of wide (16 B/lane) coalesced reads at 64 B, i.e. reports half the bytes; WRITE_SIZE is uncalibrated.  Rather
than assuming, ``--pmc-calib`` makes bench.py first stream a known byte count (3 launches of the library's
cast kernel over 2^28 floats: 1 GiB read, 0.5 GiB written per launch, 4x the Infinity Cache) and the factors
true_bytes / counter_bytes measured on those launches are applied to the GEMM launches of the same pass.

usage: pmc_traffic.py FETCH.db WRITE.db KEY [profiles/hbm_traffic.json]     KEY e.g. "ViT-B/32|kadaptation|bs128"
"""
import json
import os
import sqlite3
import sys

CAL_ELEMS = 256 << 20


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                      "group by kernel_name", (counter,))
    return {name: (n, total) for name, n, total in rows}


def family(stats, needles):
    hit = lambda k: any(x in k for x in needles)
    n = sum(c for k, (c, _) in stats.items() if hit(k))
    tot = sum(t for k, (_, t) in stats.items() if hit(k))
    return n, tot


GEMM_FAMILY = ("gemm_kernel<", "gemm8_kernel<", "gemm_streamk_kernel<", "gemm_ksplit_kernel<", "gemm_kphase_kernel<")   # every kernel of pevit_amd/csrc/gemm.hip


def main():
    fetch_db, write_db, key = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles", "hbm_traffic.json")
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    # calibration launches: the cast kernel dispatches with exactly CAL_ELEMS elements are the largest ones
    cal_f = max(((k, v) for k, v in fetch.items() if "cast_bf16" in k), key=lambda kv: kv[1][1])
    cal_w = max(((k, v) for k, v in write.items() if "cast_bf16" in k), key=lambda kv: kv[1][1])

    def calib(db_path, counter, name):
        db = sqlite3.connect(db_path)
        vals = [v for (v,) in db.execute("select value from counters_collection where counter_name = ? and kernel_name = ? "
                                         "order by value desc limit 3", (counter, name))]
        return sum(vals) / len(vals) * 1024.0
    f_counter = calib(fetch_db, "FETCH_SIZE", cal_f[0])
    w_counter = calib(write_db, "WRITE_SIZE", cal_w[0])
    read_factor = CAL_ELEMS * 4 / f_counter
    write_factor = CAL_ELEMS * 2 / w_counter
    gn_f, gf = family(fetch, GEMM_FAMILY)
    gn_w, gw = family(write, GEMM_FAMILY)
    assert gn_f == gn_w and gn_f > 0, (gn_f, gn_w)
    rd = gf * 1024.0 / gn_f * read_factor
    wr = gw * 1024.0 / gn_w * write_factor
    # every dispatch of the step (round 5): all kernels except the calibration casts, per train step (one sgd_kernel dispatch each)
    def whole(stats, factor, cal_name):
        steps = sum(c for k, (c, _) in stats.items() if "sgd_kernel" in k)
        mine = lambda k: k != cal_name and "Cijk_" not in k      # not the calibration casts, not bench.py's vendor-GEMM side measurement
        n = sum(c for k, (c, _) in stats.items() if mine(k))
        tot = sum(t for k, (_, t) in stats.items() if mine(k))
        return steps, n, tot * 1024.0 * factor / max(steps, 1)
    st_f, n_all, rd_all = whole(fetch, read_factor, cal_f[0])
    st_w, _, wr_all = whole(write, write_factor, cal_w[0])
    assert st_f == st_w and st_f > 0, (st_f, st_w)
    entry = {
        "all_kernels": {"steps_profiled": st_f, "dispatches_per_step": n_all / st_f, "read_bytes_per_step": rd_all,
                        "write_bytes_per_step": wr_all, "hbm_bytes_per_step": rd_all + wr_all},
        "gemm": {"launches_profiled": gn_f, "fetch_size_kib_per_launch_raw": gf / gn_f, "write_size_kib_per_launch_raw": gw / gn_w,
                 "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr},
        "calibration": {"kernel": "cast_f32_to_bf16 over 2^28 elements (1 GiB read, 0.5 GiB written)",
                        "read_factor_true_over_counter": read_factor, "write_factor_true_over_counter": write_factor},
        "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB), summed over every gemm_kernel / gemm8_kernel / gemm_ksplit_kernel / gemm_kphase_kernel / gemm_streamk_kernel "
               "dispatch of `bench.py --steps 3 --warmup 1`, / dispatches, x the true/counter factor measured in the same pass "
               "on a 1 GiB streaming cast (read %.3f, write %.3f)" % (read_factor, write_factor),
    }
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernels_hash
    entry["kernels_hash"] = kernels_hash()          # bench.py reports `traffic` only while the kernel sources still match
    try:
        with open(out_path) as f:
            allv = json.load(f)
    except (OSError, ValueError):
        allv = {}
    allv[key] = entry
    with open(out_path, "w") as f:
        json.dump(allv, f, indent=1, sort_keys=True)
    print(json.dumps(entry, indent=1))
    # the other kernels, for DESIGN.md: corrected bytes per launch
    print("\n| kernel | launches | read MB/launch | write MB/launch |\n|---|---|---|---|")
    for k in sorted(fetch, key=lambda k: -fetch[k][1])[:14]:
        n, t = fetch[k]
        wn, wt = write.get(k, (n, 0.0))
        print(f"| {k[:70]} | {n} | {t * 1024 / n * read_factor / 1e6:.2f} | {wt * 1024 / max(wn, 1) * write_factor / 1e6:.2f} |")


if __name__ == "__main__":
    main()
