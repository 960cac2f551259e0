#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03b
mkdir -p $OUT
step() { local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > $OUT/$name.log 2> $OUT/$name.err; local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/status.txt
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name"; exit $rc; fi; }
step tests 600 python3 -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -p no:cacheprovider
step adaptive 200 python3 -u tools/adaptive_time.py
step probe 300 python3 -u tools/threshold_probe.py
BENCH_TRACE=40 BENCH_BACKEND=gloo step bench2 150 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1
tail -3 $OUT/tests.log; cat $OUT/adaptive.log
