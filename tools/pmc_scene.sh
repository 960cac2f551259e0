#!/bin/bash
# Counter passes of the fused pipeline on a sphere scene: tools/pmc_scene.sh N SUB -> gpurun_out/pmc_scene_N_SUB/ (L1 traffic and
# waiting of the one-launch kernel; one rocprofv3 run per counter group).
export TMPDIR=/tmp
N=$1; SUB=$2
OUT=gpurun_out/pmc_scene_${N}_${SUB}
mkdir -p $OUT
run() {
  local name=$1; shift
  PIPES=1 timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 tools/bigscene_spp.py $N $SUB 1920 1080 64 > $OUT/$name.log 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }
}
run tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum && \
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD && \
python3 tools/pmc_summary.py $OUT | grep -A12 "k_bounce_persist"
