#!/bin/bash
# ONE parametrised runner for a call to the GPU box (round 6; replaces the 121 one-shot gpu_r2*.sh .. gpu_r5*.sh files, whose command
# lists are recorded in INDEX.md and in the git history up to commit 4031805):
#
#   gpurun --timeout S -- 'bash scripts/experiments/run.sh <tag> <step> [<step> ...]'
#
# Every <step> is ONE quoted string "<verb> <args...>"; outputs of the call go to gpurun_out/<tag>/ (scratch -- copy what is to be
# kept into profiles/).  Verbs:
#   tests [pytest args]           GPU suite (default: tests -m gpu -q) -> pytest_<n>.log
#   bench [bench.py args]         one bench line                       -> bench_<n>.json (+ .err)
#   ab "<bench args>" "k=v" ...   three alternating rounds of the bench step per knob setting (scripts/gpu_ab.sh)
#   kstats <name> [bench args]    rocprofv3 kernel-trace summary of the bench step (scripts/gpu_kstats.sh)
#   variants [args]               kernel-trace summary for the stock library and every pevit_amd/variants/*.so (scripts/gpu_variants.sh)
#   pmc <name>                    the PMC passes behind bench.py's roofline object (scripts/run_pmc_passes.sh)
#   py <script> [args]            python <script> args                 -> py_<n>.log
#   sh <command line>             anything else, verbatim              -> sh_<n>.log
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
n=0
for step in "$@"; do
  n=$((n + 1))
  if [ "${step%% *}" = "sh" ]; then       # a raw command line (may hold ; | and quotes): taken verbatim
    set -- sh "${step#sh }"
  else
    eval "set -- $step"        # (inner quotes of a step survive: "tests -k 'a or b'")
  fi
  verb=$1; shift
  echo "=== [$TAG $n] $verb $*"
  case $verb in
    tests)    if [ $# -eq 0 ]; then set -- tests -m gpu -q; fi
              timeout ${STEP_TIMEOUT:-1500} python -m pytest -p no:cacheprovider "$@" > $O/pytest_$n.log 2>&1; echo "rc=$?" >> $O/pytest_$n.log; tail -${TAIL:-12} $O/pytest_$n.log ;;
    bench)    timeout ${STEP_TIMEOUT:-900} python bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; cut -c1-600 $O/bench_$n.json ;;
    ab)       timeout ${STEP_TIMEOUT:-1800} bash scripts/gpu_ab.sh "$@" 2>&1 | tee $O/ab_$n.log ;;
    kstats)   timeout ${STEP_TIMEOUT:-900} bash scripts/gpu_kstats.sh "$@" 2>&1 | tee $O/kstats_$n.log ;;
    variants) timeout ${STEP_TIMEOUT:-1800} bash scripts/gpu_variants.sh "$@" 2>&1 | tee $O/variants_$n.log ;;
    pmc)      timeout ${STEP_TIMEOUT:-2400} bash scripts/run_pmc_passes.sh "$@" 2>&1 | tail -60 | tee $O/pmc_$n.log ;;
    py)       timeout ${STEP_TIMEOUT:-900} python "$@" > $O/py_$n.log 2>&1; echo "rc=$?" >> $O/py_$n.log; tail -${TAIL:-20} $O/py_$n.log ;;
    sh)       timeout ${STEP_TIMEOUT:-900} bash -c "$1" > $O/sh_$n.log 2>&1; echo "rc=$?" >> $O/sh_$n.log; tail -${TAIL:-20} $O/sh_$n.log ;;
    *)        echo "unknown verb $verb"; exit 2 ;;
  esac
done
