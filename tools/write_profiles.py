"""Copies the measurements that tools/_all.sh left under gpurun_out/ into profiles/ (scratch tool)."""
import csv, json, shutil, subprocess

def table(path, top=7):
    rows = list(csv.DictReader(open(path)))[:top]
    out = "| name | total_calls | total_duration | average | percentage |\n|---|---|---|---|---|\n"
    for r in rows:
        out += f"| {r['Name']} | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.3f} | {float(r['AverageNs'])/1e3:.3f} | {float(r['Percentage']):.3f} |\n"
    return out, float(rows[0]['AverageNs']) / 1e6

b = json.load(open('gpurun_out/bench_profiled.json'))
tf, avg_f = table('gpurun_out/prof_f/pf_kernel_stats.csv')
ts, _ = table('gpurun_out/prof_s/ps_kernel_stats.csv')
open('profiles/r01_kernel_stats_final.md', 'w').write(f"""# Round 1 — kernel statistics of the final build (rocprofv3 --kernel-trace --stats; durations in microseconds)

## fused pipeline (default), 1080p 1024 spp, 1 step

command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f -o pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`

{tf}
bench.py's HIP-event average for the fused kernel in the same run: {b['roofline']['avg_launch_ms']:.4f} ms per launch over {b['roofline']['launches']} launches (`r01_bench_final_profiled.json`); rocprof average above: {avg_f:.4f} ms.

## staged pipeline, 1080p 256 spp, 1 step

command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -o ps -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --pipeline staged --spp 256`

{ts}
Earlier states of the round: `r01_kernel_stats_initial.md`, `r01_kernel_stats_chunks.md` and the history in `README.md`.
""")
shutil.copy('gpurun_out/bench_profiled.json', 'profiles/r01_bench_final_profiled.json')
shutil.copy('gpurun_out/bench_final.json', 'profiles/r01_bench_final.json')
shutil.copy('gpurun_out/pmc_fused_summary.txt', 'profiles/r01_pmc_fused.txt')
rd = lambda p: open(p).read()
open('profiles/r01_other_configs.txt', 'w').write(
    rd('gpurun_out/configs.txt') + "\n# training ring on/off, 1080p 64 spp (tools/train_ring_time.py)\n" + rd('gpurun_out/train_ring.txt')
    + "\n# adaptive sampling at 1080p, 1024 spp maximum (tools/adaptive_time.py)\n" + rd('gpurun_out/adaptive_time.txt')
    + "\n# demosaic at 4K, device-resident, steady state (tools/demosaic_time.py)\n"
    + "".join(l for l in rd('gpurun_out/demosaic_time.txt').splitlines(True) if 'rep' in l)
    + "\n# measured earlier in the round, not repeated: 4K Bayer 4096 spp whole job 23 312 Mrays/s over 268 G rays (tools/config4_full.py)\n")
open('profiles/r01_bigscene.txt', 'w').write(rd('gpurun_out/bigscene.txt') + "\n# per-kernel times, staged pipeline (tools/bigscene_stats.py)\n" + rd('gpurun_out/bigscene_stats.txt'))
s = rd('gpurun_out/parity_report.txt'); open('profiles/r01_parity_report.json', 'w').write(s[s.index('{'):])

# PMC-derived figures for bench.py
tot = {}
for line in rd('gpurun_out/pmc_fused_summary.txt').split('k_bounce_defer')[1].splitlines()[1:]:
    f = line.split()
    if len(f) < 6 or f[1] != 'total': break
    tot[f[0]] = (float(f[2]), float(f[4]), int(f[6]))
fetch, write = tot['FETCH_SIZE'][1], tot['WRITE_SIZE'][1]
T = tot['GRBM_GUI_ACTIVE'][0] / 8 / 2.4e9   # the counter sums the 8 XCDs
avail = 1024 * T * 2.4e9 / 4
fb = json.load(open('gpurun_out/bench_final.json'))
rays_per_step = fb['roofline']['rays_per_launch'] * fb['roofline']['launches'] / fb['steps']
json.dump({"fused": {
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
    "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as reported",
    "workload": f"cornell_1920x1080_1024spp_depth32_converge, {tot['FETCH_SIZE'][2]} launches of k_bounce_defer",
    # SIMD cycles that went by per VALU wave-instruction issued (GRBM_GUI_ACTIVE is summed over the 8 XCDs;
    # 1024 SIMDs).  A full 64-lane fp32 instruction occupies the SIMD for 4.
    "simd_cycles_per_valu_inst": 1024 * (tot['GRBM_GUI_ACTIVE'][0] / 8) / tot['SQ_INSTS_VALU'][0],
    "valu_lane_utilisation": tot['SQ_THREAD_CYCLES_VALU'][0] / tot['SQ_INSTS_VALU'][0] / 64,
    "valu_insts_per_ray": tot['SQ_INSTS_VALU'][0] / rays_per_step,
    "source": "profiles/r01_pmc_fused.txt (rocprofv3 --pmc, one pass per counter group)"}}, open('profiles/pmc_traffic.json', 'w'), indent=1)
print(open('profiles/pmc_traffic.json').read())
