#!/bin/bash
# GPU box: parity tests, then a short bench.  usage: scripts/gpu_check.sh <tag> [bench args]
TAG=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_$TAG.log 2>&1; rc=$?
tail -5 gpurun_out/gpu_tests_$TAG.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 2 --warmup 1 "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; rc=$?
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
exit $rc
