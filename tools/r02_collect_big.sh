#!/bin/bash
# The large-scene part of tools/r02_collect.sh alone (same output files), for changes that touch only the staged pipeline.
export TMPDIR=/tmp
set -e
OUT=gpurun_out/r02
mkdir -p $OUT
timeout -k 10 300 python3 tools/bigscene.py 2500 4 3840 2160 16 > $OUT/bigscene.txt 2>&1
timeout -k 10 300 python3 tools/bigscene_stats.py 2500 4 3840 2160 16 > $OUT/bigscene_stats.txt 2>&1
timeout -k 10 300 python3 tools/bigscene.py 40000 2 3840 2160 16 > $OUT/bigscene_40000.txt 2>&1
INSTANCED=1 timeout -k 10 300 python3 tools/bigscene_spp.py 40000 2 3840 2160 16 > $OUT/bigscene_instanced.txt 2>&1
runb() { local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_big/$name -o $name -- python3 tools/bigscene_stats.py 2500 4 3840 2160 16 > $OUT/pmcb_$name.log 2> $OUT/pmcb_$name.err; }
rm -rf $OUT/pmc_big
runb sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
runb sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES
runb tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
runb tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
runb fetch FETCH_SIZE
runb grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $OUT/pmc_big > $OUT/pmc_big_summary.txt
timeout -k 10 200 python3 tools/sections_staged.py 2500 4 3840 2160 16 > $OUT/sections_staged.txt 2>&1 || true
echo collected
