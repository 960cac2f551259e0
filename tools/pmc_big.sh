#!/bin/bash
# PMC passes on the large synthetic scene, staged pipeline (scratch tool). Run from the repository root.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_big
mkdir -p $OUT
run() {
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 tools/bigscene_stats.py 2500 4 3840 2160 4 > $OUT/$name.log 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
run tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum && \
run fetch FETCH_SIZE
