#!/bin/bash
# kernel trace + stats of a short bench run, one batch in flight (RT_HIP_SLOTS=1) so that a kernel's duration is its own and agrees with the
# event-timed leg of bench.py; RT_HIP_SLOTS=2 scripts/prof_trace.sh <tag> traces the default two overlapping slots.  usage: scripts/prof_trace.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
RT_HIP_SLOTS=${RT_HIP_SLOTS:-1} timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --no-exact --no-strong > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
grep '^{' $OUT/run.log
cat $OUT/t_kernel_stats.csv
