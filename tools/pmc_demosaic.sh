#!/bin/bash
# Kernel trace and counter passes of the 4K demosaic (tools/demosaic_time.py) -> gpurun_out/pmc_demosaic/ and a summary on stdout.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_demosaic
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 tools/demosaic_time.py > $OUT/trace.log 2> $OUT/trace.err || { echo trace failed; exit 1; }
run() { local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python3 tools/demosaic_time.py > $OUT/$name.log 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; return 1; }; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES && \
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY && \
run mem FETCH_SIZE && run memw WRITE_SIZE || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pmc_demosaic/trace/**/tr_kernel_stats.csv", recursive=True)[0]
print("# rocprofv3 --kernel-trace --stats of tools/demosaic_time.py (30 gradient and 30 bilinear demosaics at 3840x2160)")
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"])):
    if "pt::" in r["Name"]: print(f"{r['Name'][:80]:80s} calls {r['Calls']:>4s} average {float(r['AverageNs']) / 1e3:8.1f} us")
PY
echo "# counters per kernel, totals over the 30 launches each (one rocprofv3 --pmc run per group)"
python3 tools/pmc_summary.py $OUT | grep -v "^pt::k_\(iota\|final\)" 
