#!/bin/bash
# rocprofv3 evidence for bench.py's headline workload.  Usage (on the GPU box): bash scripts/profile.sh <tag>
# Writes raw output under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand afterwards.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multi-inprocess $BENCH_FLAGS"
# 1. per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed"
# 2. counters, each group in its own pass (no tracing flags alongside --pmc)
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_lds -- $CMD > $OUT/pmc_lds.log 2>&1 || echo "pmc_lds failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "pmc_fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || echo "pmc_write failed"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || echo "pmc_sq failed"
find $OUT -name "*.csv" | head -40
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
