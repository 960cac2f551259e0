"""Files the measurements tools/r03_collect.sh left under gpurun_out/r03/ as profiles/r03_* and rewrites
profiles/pmc_traffic.json, which bench.py reads for the counter-derived fields of its roofline object.
Run from the repository root after `make -C rust-pathtracer_amd asm`."""
import csv
import os
import json
import re
import shutil
import subprocess
import sys

SRC = "gpurun_out/r03"
rd = lambda p: open(f"{SRC}/{p}").read()


def table(path, top=7):
    rows = list(csv.DictReader(open(path)))[:top]
    out = "| name | total_calls | total_duration | average | percentage |\n|---|---|---|---|---|\n"
    for r in rows:
        out += (f"| {r['Name']} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e3:.3f} | "
                f"{float(r['AverageNs']) / 1e3:.3f} | {float(r['Percentage']):.3f} |\n")
    return out, float(rows[0]["AverageNs"]) / 1e6


def counters(summary, kernel):
    tot = {}
    for line in summary.split(kernel, 1)[1].splitlines()[1:]:
        f = line.split()
        if len(f) < 7 or f[1] != "total":
            break
        tot[f[0]] = (float(f[2]), float(f[4]), int(f[6]))
    return tot


def main():
    prof = json.loads(rd("bench_profiled.json"))
    final = json.loads(rd("bench_final.json"))
    tf, avg = table(f"{SRC}/prof_f/pf_kernel_stats.csv")
    r = prof["roofline"]
    open("profiles/r03_kernel_stats.md", "w").write(f"""# Kernel statistics, r03 (rocprofv3 --kernel-trace --stats; durations in microseconds)

## fused pipeline (default), 1080p 1024 spp, 1 step

command: `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_f -o pf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-boundary`

{tf}
bench.py's HIP-event average for the fused kernel in the same run: {r['avg_launch_ms']:.4f} ms per launch over
{r['launches']} launches (`r03_bench_profiled.json`); rocprof average above: {avg:.4f} ms.  The default run
(3 steps after 1 warm-up, `r03_bench_final.json`): {final['roofline']['avg_launch_ms']:.4f} ms per launch,
{final['value']:.0f} Mrays/s.
""")
    shutil.copy(f"{SRC}/bench_profiled.json", "profiles/r03_bench_profiled.json")
    shutil.copy(f"{SRC}/bench_final.json", "profiles/r03_bench_final.json")
    shutil.copy(f"{SRC}/pmc_fused_summary.txt", "profiles/r03_pmc_fused.txt")
    shutil.copy(f"{SRC}/pmc_big_summary.txt", "profiles/r03_pmc_bigscene_staged.txt")
    shutil.copy(f"{SRC}/sections_fused.txt", "profiles/r03_sections_fused.txt")
    shutil.copy(f"{SRC}/sections_guided.txt", "profiles/r03_sections_guided.txt")
    for name in ("threshold_probe.json", "parity_report.json", "bench_two_ranks_gloo.json"):
        txt = rd(name)
        open(f"profiles/r03_{name}", "w").write(txt[txt.index("{"):])
    open("profiles/r03_small_and_midsize_scenes.txt", "w").write(
        "# spheres in the room at 1080p, 256 spp (tools/bigscene_spp.py N SUBDIV): staged kernels forced (pipe 0) against the fused kernel (pipe 1).\n"
        "# Triangles: 1 0 = 32 (five nodes; in LDS, shadow rays in place), 1 1 = 92 (in LDS, in place), 2 2 = 652, 5 2 = 1 612, 10 2 = 3 212, 20 2 = 6 412, 50 2 = 16 012; the built-in scene (24 triangles, 12 items in one node) runs at 55 Grays/s\n"
        + rd("small_scenes.txt") + "\n# larger scenes under the default choice (tools/midsize_sweep.sh)\n" + rd("midsize.txt"))
    open("profiles/r03_occupancy_and_adaptive_rounds.txt", "w").write(
        "# resident blocks per CU and LDS per block of the fused kernels (tools/occupancy.py)\n" + rd("occupancy.txt")
        + "\n# rounds of an adaptive 1080p render, threshold 0.01, 1024 spp maximum (PT_ADAPTIVE_TRACE=1 tools/adaptive_one.py 1 0.01)\n"
        + rd("adaptive_rounds.txt"))
    open("profiles/r03_other_configs.txt", "w").write(
        "# one-GPU timings of the configurations (tools/configs.py)\n" + rd("configs.txt")
        + "\n# adaptive sampling at 1080p, 1024 spp maximum (tools/adaptive_time.py)\n" + rd("adaptive_time.txt")
        + "\n# demosaic at 4K, device-resident, steady state (tools/demosaic_time.py)\n"
        + "".join(l for l in rd("demosaic_time.txt").splitlines(True) if "rep" in l)
        + (("\n# configuration 4 whole on one GPU: 3840x2160 Bayer, 4096 spp, demosaic (tools/config4_full.py)\n" + rd("config4_full.txt"))
           if os.path.exists(f"{SRC}/config4_full.txt") else ""))
    open("profiles/r03_bigscene.txt", "w").write(
        "# 12.8 M triangles as 2 500 spheres of 5 120, 3840x2160, 16 spp (tools/bigscene.py 2500 4 3840 2160 16);\n"
        "# pipe -1 = default choice, 1 = fused kernel reading the scene from global memory, 0 = staged kernels\n" + rd("bigscene.txt")
        + "\n# per-kernel times, staged pipeline (tools/bigscene_stats.py)\n" + rd("bigscene_stats.txt")
        + "\n# the same triangle count as 40 000 spheres of 320 (the scene of the GPU test; tools/bigscene.py 40000 2 3840 2160 16)\n"
        + rd("bigscene_40000.txt")
        + (("\n# the room and one 320-triangle sphere placed 40 000 times (two-level tree, staged kernels; INSTANCED=1 tools/bigscene_spp.py 40000 2 3840 2160 16)\n"
            + rd("bigscene_instanced.txt")) if os.path.exists(f"{SRC}/bigscene_instanced.txt") else "")
        + (("\n# section counters of the traversal kernels at 16 spp (profiling build; tools/sections_staged.py 2500 4 3840 2160 16)\n"
            + rd("sections_staged.txt")) if os.path.exists(f"{SRC}/sections_staged.txt") else ""))

    # static issue cost of the fused kernel's instruction mix
    isa = subprocess.run([sys.executable, "tools/isa_cost.py", "rust-pathtracer_amd/build/pt_kernels.s",
                          "_ZN2pt16k_bounce_persistILb0ELb0ELi2ELb0E", "12"], capture_output=True, text=True, check=True).stdout
    open("profiles/r03_isa_cost_fused.txt", "w").write(isa)
    issue_cost_static = float(re.search(r"([\d.]+) cycles per VALU instruction", isa).group(1))

    tot = counters(rd("pmc_fused_summary.txt"), "k_bounce_persist")
    fetch, write = tot["FETCH_SIZE"][1], tot["WRITE_SIZE"][1]
    rays_per_step = final["roofline"]["rays_per_launch"] * final["roofline"]["launches"] / final["steps"]
    # the mix of vector instructions as the counters saw it (wave instructions), priced with the per-class issue costs
    # of tools/valu_ops.hip: add/mul/fma 2.4, transcendental 8.2, conversions 4.2, integer 3.3 (adds and logic 2.4,
    # shifts and multiplies 4.2), the rest -- compares, selects, min/max, moves -- 3.9
    valu = tot["SQ_INSTS_VALU"][0]
    mix = {k: tot["SQ_INSTS_VALU_" + k][0] for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "CVT")}
    fma_class = mix["ADD_F32"] + mix["MUL_F32"] + mix["FMA_F32"]
    rest = valu - fma_class - mix["TRANS_F32"] - mix["INT32"] - mix["CVT"]
    issue_cost = (2.4 * fma_class + 8.2 * mix["TRANS_F32"] + 3.3 * mix["INT32"] + 4.2 * mix["CVT"] + 3.9 * rest) / valu
    lane_util = tot["SQ_THREAD_CYCLES_VALU"][0] / valu / 64
    flops = tot["SQ_INSTS_VALU_FLOPS_FP32"][0] * 64 / rays_per_step  # the counter counts per wave instruction
    json.dump({"fused": {
        "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
        "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
        "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads); WRITE_SIZE as reported",
        "workload": f"cornell_1920x1080_1024spp_depth32_converge, {tot['FETCH_SIZE'][2]} launches of k_bounce_persist (one per step)",
        # SIMD cycles that went by per VALU wave-instruction issued (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs)
        "simd_cycles_per_valu_inst": 1024 * (tot["GRBM_GUI_ACTIVE"][0] / 8) / tot["SQ_INSTS_VALU"][0],
        "issue_cost_cycles_per_valu_inst": issue_cost,
        "issue_cost_static_cycles_per_valu_inst": issue_cost_static,
        "valu_mix": {"fma_add_mul_f32": fma_class / valu, "transcendental_f32": mix["TRANS_F32"] / valu,
                     "int32": mix["INT32"] / valu, "convert": mix["CVT"] / valu, "other": rest / valu},
        "fp32_flops_issued_per_ray": flops,
        "fp32_flops_on_active_lanes_per_ray": flops * lane_util,
        "valu_lane_utilisation": lane_util,
        "valu_insts_per_ray": valu / rays_per_step,
        "collected": "round 3, tools/r03_collect.sh on the build of this commit's parent; replayed by bench.py, not measured in its run",
        "source": "profiles/r03_pmc_fused.txt (rocprofv3 --pmc, one pass per counter group), profiles/r03_isa_cost_fused.txt"}},
        open("profiles/pmc_traffic.json", "w"), indent=1)
    print(open("profiles/pmc_traffic.json").read())

    # large scene: bytes and requests per radiance ray of the extend kernel
    big = counters(rd("pmc_big_summary.txt"), "pt::k_extend")
    m = re.search(r"rays (\d+) (\d+)", rd("pmcb_fetch.log"))
    if m:
        rr = float(m.group(1))
        line = (f"\n# k_extend per radiance ray ({rr:.0f} rays in the profiled run): "
                f"HBM fetch {big['FETCH_SIZE'][0] * 2048 / rr:.0f} B (FETCH_SIZE x 2 KiB), "
                f"L2 requests {big['TCC_REQ_sum'][0] / rr:.1f} at {big['TCC_HIT_sum'][0] / big['TCC_REQ_sum'][0] * 100:.0f} % hits, "
                f"VALU {big['SQ_THREAD_CYCLES_VALU'][0] / rr:.0f} lane-instructions "
                f"({big['SQ_THREAD_CYCLES_VALU'][0] / big['SQ_INSTS_VALU'][0] / 64 * 100:.0f} % of the lanes of the wave-instructions issued), "
                f"mean L1->L2 read latency {big['TCP_TCC_READ_REQ_LATENCY_sum'][0] / big['TCP_TCC_READ_REQ_sum'][0]:.0f} cycles\n")
        open("profiles/r03_pmc_bigscene_staged.txt", "a").write(line)
        print(line)


if __name__ == "__main__":
    main()
