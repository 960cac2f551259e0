#!/bin/bash
# PMC passes for the dominant kernel (scratch tool). Each pass is its own rocprofv3 run.
export TMPDIR=/tmp
PIPE=${1:-fused}
SPP=${2:-128}
OUT=gpurun_out/pmc_${PIPE}
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --pipeline $PIPE --spp $SPP > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY && \
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE GRBM_GUI_ACTIVE
ls -R $OUT | head -40
