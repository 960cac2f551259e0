#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03c
mkdir -p $OUT
set -e
timeout -k 10 120 python3 -u tools/adaptive_time.py > $OUT/adaptive.log 2> $OUT/adaptive.err
for d in 0 1; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_d$d/sq1 -o sq1 -- python3 tools/adaptive_one.py $d 0.01 > $OUT/pmc_d$d.log 2> $OUT/pmc_d$d.err
  python3 tools/pmc_summary.py $OUT/pmc_d$d > $OUT/pmc_d${d}_summary.txt
done
BENCH_TRACE=40 BENCH_BACKEND=gloo timeout -k 10 150 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench2.log 2> $OUT/bench2.err
cat $OUT/adaptive.log; cat $OUT/bench2.log
