#!/bin/bash
# The measurements behind profiles/r03_*: tools/r03_collect.sh a | b | c, one GPU call each (a: the default scene,
# b: the large scene, c: the rest).  Run from the repository root; tools/r03_profiles.py files the results.
# Every rocprofv3 step is its own run under its own time limit; counters never share a run with a trace.
export TMPDIR=/tmp
set -e
OUT=gpurun_out/r03
mkdir -p $OUT
PART=${1:-abc}
if [[ $PART == *a* ]]; then
# 1. kernel trace + stats of the bench command (fused, 1080p 1024 spp, one step)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f -o pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary --no-large-scene > $OUT/bench_profiled.json 2> $OUT/prof_f.err
# 2. PMC passes of the same command
run() { local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_fused/$name -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary --no-large-scene > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE GRBM_GUI_ACTIVE
run mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP32
python3 tools/pmc_summary.py $OUT/pmc_fused > $OUT/pmc_fused_summary.txt
# 3. section counters and time shares of the profiling build
timeout -k 10 200 python3 tools/sections.py 512 > $OUT/sections_fused.txt 2>&1
GUIDED=1 timeout -k 10 200 python3 tools/sections.py 128 > $OUT/sections_guided.txt 2>&1
# 4. the other configurations on one GPU
timeout -k 10 200 python3 tools/configs.py > $OUT/configs.txt 2>&1
timeout -k 10 200 python3 tools/adaptive_time.py > $OUT/adaptive_time.txt 2>&1
timeout -k 10 200 python3 tools/demosaic_time.py > $OUT/demosaic_time.txt 2>&1
# the plain bench line; two ranks on the one GPU over gloo (the N > 1 code path: strong scaling, all-reduce timing)
timeout -k 10 300 python3 bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
BENCH_BACKEND=gloo timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_two_ranks_gloo.json 2> $OUT/bench_two_ranks_gloo.err
fi
if [[ $PART == *b* ]]; then
# 12.8 M triangles two ways: 2 500 spheres of 5 120 triangles (the scene of the earlier profiles) and 40 000 of 320
timeout -k 10 300 python3 tools/bigscene.py 2500 4 3840 2160 16 > $OUT/bigscene.txt 2>&1
timeout -k 10 300 python3 tools/bigscene_stats.py 2500 4 3840 2160 16 > $OUT/bigscene_stats.txt 2>&1
timeout -k 10 300 python3 tools/bigscene.py 40000 2 3840 2160 16 > $OUT/bigscene_40000.txt 2>&1
# 5. PMC passes of the staged pipeline on the large scene
runb() { local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_big/$name -o $name -- python3 tools/bigscene_stats.py 2500 4 3840 2160 16 > $OUT/pmcb_$name.log 2> $OUT/pmcb_$name.err; }
runb sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
runb sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES
runb tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
runb tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
runb fetch FETCH_SIZE
runb grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $OUT/pmc_big > $OUT/pmc_big_summary.txt
fi
if [[ $PART == *c* ]]; then
# 7. this round's additions: thresholded test quantities, parity report, occupancy, adaptive rounds, small and mid-size scenes
timeout -k 10 300 python3 -u tools/threshold_probe.py > $OUT/threshold_probe.json 2> $OUT/threshold_probe.err
timeout -k 10 200 python3 -u tools/parity_report.py > $OUT/parity_report.json 2> $OUT/parity_report.err
timeout -k 10 100 python3 -u tools/occupancy.py > $OUT/occupancy.txt 2>&1
PT_ADAPTIVE_TRACE=1 timeout -k 10 100 python3 -u tools/adaptive_one.py 1 0.01 > $OUT/adaptive_rounds.txt 2>&1
rm -f $OUT/small_scenes.txt
for cfg in "1 0" "1 1" "2 2" "5 2" "10 2" "20 2" "50 2"; do echo "== scene $cfg" >> $OUT/small_scenes.txt; PIPES=0,1 timeout -k 10 120 python3 -u tools/bigscene_spp.py $cfg 1920 1080 256 2>&1 | grep pipe >> $OUT/small_scenes.txt; done
timeout -k 10 900 bash tools/midsize_sweep.sh > $OUT/midsize.txt 2>&1
timeout -k 10 200 python3 -u tools/config4_full.py > $OUT/config4_full.txt 2>&1
fi
echo collected $PART
