#!/bin/bash
# Counter passes of the guided render (tools/guided_render.py) -> gpurun_out/pmc_guided/; GUIDED=0 for the same render unguided.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_guided${GUIDED:-1}
mkdir -p $OUT
run() {
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 tools/guided_render.py 256 > $OUT/$name.log 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD && \
run sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU && \
run tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum && \
cat $OUT/sq1.log && python3 tools/pmc_summary.py $OUT | grep -A16 "k_bounce_persist"
