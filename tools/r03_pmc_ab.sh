#!/bin/bash
# counters of the headline render under two settings of an environment variable: tools/r03_pmc_ab.sh VAR A B
export TMPDIR=/tmp
VAR=$1; shift
for v in "$@"; do
  OUT=gpurun_out/pmc_ab/$VAR$v
  rm -rf $OUT; mkdir -p $OUT
  env $VAR=$v timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq1 -o sq1 -- python3 tools/adaptive_one.py 1 0.0 > $OUT/log.txt 2> $OUT/err.txt || exit 1
  echo "== $VAR=$v"; cat $OUT/log.txt; python3 tools/pmc_summary.py $OUT | grep -A9 "k_bounce" | head -10
done
