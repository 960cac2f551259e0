export TMPDIR=/tmp
set -e
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f -o pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_profiled.json 2> gpurun_out/prof_f.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -o ps -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --pipeline staged --spp 256 > gpurun_out/bench_profiled_staged.json 2> gpurun_out/prof_s.err
