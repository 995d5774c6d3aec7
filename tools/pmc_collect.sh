#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel stats + separate PMC passes over a short bench.py run.
# Counters are collected in their own passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
#   bash tools/pmc_collect.sh gpurun_out/r02_pmc
set -u
OUT=$(realpath -m "${1:-gpurun_out/r02_pmc}")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# counter passes: one job at a time (the profiler serialises dispatches anyway; per-kernel attribution stays clean)
BENCH="python3 $REPO/bench.py --no-extra-configs --no-cpu-baseline --steps 3 --warmup 1 --in-flight 1 --sustain-s 0"
WANT=" ${PASSES:-stats fetch write tcp tcc ta sq mfma} "
want() { [[ "$WANT" == *" $1 "* ]]; }
if want stats; then
    timeout -k 5 240 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 $REPO/bench.py --no-extra-configs --no-cpu-baseline --no-standalone-pass --sustain-s 0 --steps 30 > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || echo "stats pass failed"
    echo "stats done $(date +%T)" | tee -a "$OUT/progress.log"
fi
pass() { # name, counters...   (a pass that asks for more counters than a block has slots aborts: keep each set small)
    local name=$1; shift
    want "$name" || return 0
    timeout -k 5 180 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -o "$name" --output-format csv -- $BENCH > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed"
    echo "pass $name done $(date +%T)" | tee -a "$OUT/progress.log"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcp TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
find "$OUT" -name "*.csv" | head -40
