#!/bin/bash
# Round-2 evidence in one GPU call: PMC traffic + kernel stats of the bench workload, PMC traffic of every config, the per-config
# roofline lines.  Everything lands under gpurun_out/r02/; copy what is to be judged into profiles/.
# usage (GPU box): scripts/r02_profiles.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r02; mkdir -p $O
# 1. bench workload (Night Spheres, 64-pass batches), one batch slot so that kernels do not overlap
RT_HIP_SLOTS=1 scripts/pmc_passes.sh r02_bench "tcc1 tcc2" > $O/pmc_bench.log 2>&1
python3 scripts/pmc_traffic.py gpurun_out/pmc_r02_bench "Night Spheres 1920x1080, 64 passes per batch, depth 10, 2 light samples" > $O/r02_traffic.json
echo "bench traffic done"
# 2. kernel trace of the bench command
scripts/prof_trace.sh r02 > $O/trace.log 2>&1
cp gpurun_out/trace_r02/t_kernel_stats.csv $O/r02_kernel_stats.csv
echo "trace done"
# 3. every config: FETCH_SIZE / WRITE_SIZE passes over the short run of scripts/run_configs.py
for c in c1 c2 c3 c4 c5; do
  RT_HIP_SLOTS=1 BENCH_CMD="python3 $R/scripts/run_configs.py $c --short" scripts/pmc_passes.sh r02_$c "tcc1 tcc2" > $O/pmc_$c.log 2>&1
  python3 scripts/pmc_traffic.py gpurun_out/pmc_r02_$c "$c (scripts/run_configs.py $c --short)" > $O/traffic_$c.json
  echo "$c traffic done"
done
python3 - <<PY
import json
out = {c: json.load(open("$O/traffic_%s.json" % c)) for c in ("c1", "c2", "c3", "c4", "c5")}
json.dump(out, open("$O/r02_traffic_configs.json", "w"), indent=1)
PY
# 4. the roofline line of every config (full passes), with that traffic
cp $O/r02_traffic_configs.json profiles/r02_traffic_configs.json
timeout -k 10 600 python3 scripts/run_configs.py c1 c2 c3 c4 c5 --traffic profiles/r02_traffic_configs.json > $O/r02_all_configs_roofline.jsonl 2> $O/run_configs.err
echo "rooflines done"
# 5. what FETCH_SIZE reports for the node-gather pattern (tools/calib_fetch.hip)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -o calib -- $R/tools/calib_fetch > $O/calib.log 2>&1
cd $R
python3 scripts/pmc_summary.py $O/calib > $O/r02_fetch_calibration.csv
cat $O/calib.log $O/r02_fetch_calibration.csv
