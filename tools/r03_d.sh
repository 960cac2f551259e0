#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03d
mkdir -p $OUT
set -e
timeout -k 5 100 python3 -u tools/occupancy.py > $OUT/occupancy.log 2>&1
timeout -k 10 120 python3 -u tools/adaptive_time.py > $OUT/adaptive.log 2> $OUT/adaptive.err
timeout -k 10 600 python3 -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread -p no:cacheprovider -k "adaptive or noise or pipelines" > $OUT/tests.log 2>&1
cat $OUT/occupancy.log $OUT/adaptive.log; tail -3 $OUT/tests.log
