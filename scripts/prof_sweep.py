import cProfile, pstats, sys, os, io
sys.argv = ["x"]
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import runpy
# run once to warm (build), then profile one reuse run
import importlib.util
src = open("/root/repo/scripts/time_sweep_reuse.py").read().split("for reuse in (True, False):")[0]
exec(compile(src, "setup", "exec"))
mod.train_task(train, val, cfg, sweep_run=True)
pr = cProfile.Profile(); pr.enable()
mod.train_task(train, val, cfg, sweep_run=True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
