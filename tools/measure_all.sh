export TMPDIR=/tmp
set -e
timeout -k 10 300 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
bash tools/collect_profiles.sh
bash tools/pmc.sh fused 1024 > gpurun_out/pmc_run.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fused > gpurun_out/pmc_fused_summary.txt
timeout -k 10 300 python tools/configs.py > gpurun_out/configs.txt 2>&1
timeout -k 10 200 python tools/train_ring_time.py > gpurun_out/train_ring.txt 2>&1
timeout -k 10 300 python tools/bigscene.py 2500 4 3840 2160 16 > gpurun_out/bigscene.txt 2>&1
timeout -k 10 300 python tools/bigscene_stats.py 2500 4 3840 2160 16 > gpurun_out/bigscene_stats.txt 2>&1 || true
timeout -k 10 200 python tools/adaptive_time.py > gpurun_out/adaptive_time.txt 2>&1
timeout -k 10 200 python tools/parity_report.py > gpurun_out/parity_report.txt 2>&1
timeout -k 10 200 python tools/demosaic_time.py > gpurun_out/demosaic_time.txt 2>&1
