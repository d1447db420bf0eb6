#!/bin/bash
# PMC passes over a short bench run (one rocprofv3 run per counter set; never combined with tracing).
# usage: scripts/pmc_passes.sh <tag>   -> gpurun_out/pmc_<tag>/<pass>/...
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
BENCH=${BENCH_CMD:-"python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --no-exact --no-strong"}
PASSES=${2:-"sq1 sq2 tcc1 tcc2 tcp1"}
run() { case " $PASSES " in *" $1 "*) ;; *) return 0;; esac; name=$1; shift; echo "== pass $name: $*"; timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BENCH > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp1 TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
echo done
