#!/bin/bash
# Round-3 evidence: PMC traffic + counters + kernel stats of the bench workload, PMC traffic of every config (+ the SQ counters of
# configs 4 and 5, whose trees are read through the caches), the per-config roofline lines.  Everything lands under gpurun_out/r03/;
# copy what is to be judged into profiles/.     usage (GPU box): scripts/r03_profiles.sh <stage>...     stages: bench c123 c45 roofline
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r03; mkdir -p $O
for stage in "$@"; do case $stage in
bench)
  # the bench workload (Night Spheres, 64-pass batches), one batch slot so that kernels do not overlap
  RT_HIP_SLOTS=1 scripts/pmc_passes.sh r03_bench "sq1 sq2 tcc1 tcc2" > $O/pmc_bench.log 2>&1
  python3 scripts/pmc_traffic.py gpurun_out/pmc_r03_bench "Night Spheres 1920x1080, 64 passes per batch, depth 10, 2 light samples" > $O/r03_traffic.json
  python3 scripts/pmc_report.py gpurun_out/pmc_r03_bench > $O/r03_pmc_report.txt
  scripts/prof_trace.sh r03 > $O/trace.log 2>&1
  cp gpurun_out/trace_r03/t_kernel_stats.csv $O/r03_kernel_stats.csv
  echo "bench done";;
c123|c45)
  cfgs="c1 c2 c3"; passes="tcc1 tcc2"; [ $stage = c45 ] && { cfgs="c4 c5"; passes="sq1 sq2 tcc1 tcc2"; }
  for c in $cfgs; do
    RT_HIP_SLOTS=1 BENCH_CMD="python3 $R/scripts/run_configs.py $c --short" scripts/pmc_passes.sh r03_$c "$passes" > $O/pmc_$c.log 2>&1
    python3 scripts/pmc_traffic.py gpurun_out/pmc_r03_$c "$c (scripts/run_configs.py $c --short)" > $O/traffic_$c.json
    [ $stage = c45 ] && python3 scripts/pmc_report.py gpurun_out/pmc_r03_$c > $O/r03_${c}_pmc_report.txt
    echo "$c done"
  done;;
roofline)
  python3 - <<PY
import json
out = {c: json.load(open("$O/traffic_%s.json" % c)) for c in ("c1", "c2", "c3", "c4", "c5")}
json.dump(out, open("$O/r03_traffic_configs.json", "w"), indent=1)
PY
  cp $O/r03_traffic_configs.json profiles/r03_traffic_configs.json
  timeout -k 10 600 python3 scripts/run_configs.py c1 c2 c3 c4 c5 --traffic profiles/r03_traffic_configs.json > $O/r03_all_configs_roofline.jsonl 2> $O/run_configs.err
  echo "rooflines done";;
esac; done
