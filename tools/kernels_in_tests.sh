#!/bin/bash
# Which kernels the GPU test suite launches (profiles/rNN_kernels_in_tests.md): a kernel trace of pytest in one
# process; the tests that start other processes (two ranks, the host driver) are left out of the traced run.
export TMPDIR=/tmp
set -e
OUT=gpurun_out/kt
rm -rf $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 -m pytest tests -m gpu -q -x -k "not two_ranks and not driver_writes and not shards_equal" > gpurun_out/kt.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt/**/kt_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
out = ["# Kernels that ran during `pytest tests -m gpu` (rocprofv3 --kernel-trace --stats, `tools/kernels_in_tests.sh`; the tests that start other processes left out of the traced run)", "",
       "| kernel | calls | total ms | average us |", "|---|---|---|---|"]
for r in rows:
    out.append(f"| `{r['Name']}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} |")
open("gpurun_out/kernels_in_tests.md", "w").write("\n".join(out) + "\n")
print(len(rows), "kernels")
PY
