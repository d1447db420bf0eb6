#!/bin/bash
# kernel trace + stats of a short bench run.  usage: scripts/prof_trace.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
grep '^{' $OUT/run.log
cat $OUT/t_kernel_stats.csv
