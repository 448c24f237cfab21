#!/bin/bash
# rocprofv3 evidence for bench.py's headline workload.  Usage (on the GPU box): bash scripts/profile.sh <tag>
# Writes raw output under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand afterwards.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export RPF_BENCH_WATCHDOG=100
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-multi-inprocess $BENCH_FLAGS"
# every pass runs under its own limit; a pass that is killed at the limit ends the script (no further GPU step after a hang)
LIM=${PASS_LIMIT:-160}
pass() {  # pass <name> <rocprofv3 args...>
    local name=$1; shift
    local cmd="$CMD"
    # counter passes skip the 3840x2160x32 strong-scaling leg of the default workload: it has its own profile (r03_4k32slab), and
    # a counter-collection run of it hung inside the profiler twice this round (bench.py's watchdog: stuck in the filter call)
    if [ "$name" != trace ] && [[ "$cmd" != *--no-scaling-4k32* ]]; then cmd="$cmd --no-scaling-4k32"; fi
    timeout -k 10 $LIM rocprofv3 "$@" --output-format csv -d $OUT/$name -- $cmd > $OUT/$name.log 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $name hit its limit: stopping"; python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; exit 3; fi
}
# 1. per-kernel time
pass trace --kernel-trace --stats
# 2. counters, each group in its own pass (no tracing flags alongside --pmc)
pass pmc_lds --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_sq --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE
pass pmc_write --pmc WRITE_SIZE
find $OUT -name "*.csv" | head -40
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
