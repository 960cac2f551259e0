#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (and spp/s) of the default scene at 1920x1080, 1024 spp, depth 32.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one full render pass of the hot path: every pixel of the 1080p frame takes 1024
samples through the wavefront integrator (BVH traversal + shading + NEE), sums stay in HBM.
With N ranks the step is still ONE 1024-spp frame (--scaling strong, the default for N > 1): the
samples are split by index over the ranks (sharded.split_samples: 128 per GPU at N = 8) and the
per-pixel sums are all-reduced over RCCL at the end of the step.  --scaling weak gives every rank
its own 1024 samples of the same frame instead (the image gets N*1024 spp); a strong run reports
that figure too, as "weak_value", from a second timed loop after the first.
A ray is one BVH query: every radiance ray plus every shadow ray actually traced.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rust-pathtracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# Work per ray used for the roofline figures.
# FLOP_PER_RAY: fp32 operations of one BVH query plus its share of a shading step, counted on the source
# (about 10 box tests x 20 + 6 item tests x 35 per query, 350 per opaque shading step; transcendentals as 1).
# BYTES_PER_RAY: what a wavefront formulation moves through HBM per ray (68 B path state read + written, 16 B
# hit record, 44 B shadow record and queue indices per path segment of one radiance + at most one shadow ray
# = 332 B / 2 rays).  The staged pipeline moves these bytes; the fused kernel keeps a path in registers for 16
# bounces and the scene in LDS, so for it the figure is an equivalent, not traffic, and its bound is the vector ALU.
BYTES_PER_RAY = 166.0
FLOP_PER_RAY = 600.0
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TFLOPS = 157.3


def boundary_timings(ptamd, L, w, h, spp, max_depth):
    """The reference's own entry point at the bench configuration (render + finalise + copy into pinned host
    memory from optix_alloc_host), and the 4K demosaic kernel; reported beside the headline, not part of it."""
    import ctypes as C
    out = {}
    n = w * h * 3
    host = L.optix_alloc_host(n * 4)
    buf = C.cast(host, C.POINTER(C.c_float))
    t = time.perf_counter()
    rc = L.optix_render_rgb(w, h, spp, max_depth, 0.0, 32, buf)
    out["boundary_ms"] = (time.perf_counter() - t) * 1e3
    out["boundary_device_ms"] = L.pt_last_render_ms()
    assert rc == 0
    img = np.ctypeslib.as_array(buf, shape=(n,))
    out["boundary_image_mean"] = float(img.mean())
    L.optix_free_host(host)
    dw, dh = 3840, 2160
    raw = torch.rand(dw * dh, dtype=torch.float32, device="cuda")
    rgb = torch.empty(dw * dh * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for algo, name in ((0, "demosaic_gradient_us_4k"), (1, "demosaic_bilinear_us_4k")):
        L.pt_demosaic_device(raw.data_ptr(), dw, dh, 0, algo, rgb.data_ptr())  # warm-up
        before = ptamd.kernel_stats(ptamd.K_DEMOSAIC)
        for _ in range(5):
            L.pt_demosaic_device(raw.data_ptr(), dw, dh, 0, algo, rgb.data_ptr())
        after = ptamd.kernel_stats(ptamd.K_DEMOSAIC)
        out[name] = (after["total_ms"] - before["total_ms"]) / 5 * 1e3
    return out


def large_scene_child():
    """Runs in a process of its own (a failure here must not take the headline line with it): the stress
    configuration -- the room with 2 500 spheres of 5 120 triangles, 12.8 M triangles, 1.1 GB of tree in HBM -- at
    3840x2160, 16 samples per pixel, default pipeline (staged kernels, every sample in flight), sums left on the
    device as in the headline.  Prints one JSON object."""
    import ptamd
    L = ptamd.lib()
    L.pt_set_mode(ptamd.MODE_CONVERGE)
    w, h, spp = 3840, 2160, 16
    t = time.perf_counter()
    L.pt_scene_synthetic(42, 2500, 4)
    sums = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ptamd.render_accumulate_device(sums, w, h, 0, spp, 32, 1)  # builds and uploads the tree, grows the path state
    first_s = time.perf_counter() - t
    info = ptamd.scene_info()
    ms = []
    for frame in (2, 3):
        L.pt_reset_ray_counts()
        torch.cuda.synchronize()
        t = time.perf_counter()
        ptamd.render_accumulate_device(sums, w, h, 0, spp, 32, frame)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t) * 1e3)
    rays = sum(ptamd.ray_counts())
    print(json.dumps({"workload": f"synthetic_{info['triangles']}_triangles_{w}x{h}_{spp}spp_depth32_converge",
                      "value": rays / (ms[-1] * 1e-3) / 1e6, "unit": "Mrays/s", "ms": ms[-1], "rays": rays,
                      "bvh_bytes": info["bvh_bytes"], "sample_chunks": L.pt_last_render_chunks(),
                      "scene_build_upload_first_render_s": first_s}), flush=True)


def large_scene_timing():
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--large-scene-child"], capture_output=True, text=True,
                           timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stderr or "no output")[-300:]}
    except Exception as e:  # a timeout, a missing interpreter: reported, never fatal
        return {"error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--max-depth", type=int, default=32)
    ap.add_argument("--pipeline", choices=["fused", "staged"], default="fused")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="N > 1: split one frame's samples over the ranks (strong, default) or give every rank its own --spp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=48)
    ap.add_argument("--no-boundary", action="store_true", help="skip the optix_render_rgb / demosaic timings")
    ap.add_argument("--no-large-scene", action="store_true", help="skip the 12.8 M-triangle configuration (a child process)")
    ap.add_argument("--large-scene-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.large_scene_child:
        large_scene_child()
        return

    if os.environ.get("BENCH_TRACE"):  # debugging aid: dump every thread's stack to stderr every BENCH_TRACE seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_TRACE"]), repeat=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BENCH_BACKEND=gloo rehearses the multi-rank path with several ranks on one GPU (RCCL refuses
        # two ranks per device); the driver's runs use RCCL with one GPU per rank
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % ndev
            dist.init_process_group(backend)
    assert world == max(args.gpus, 1) or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import ptamd
    from ptamd import sharded
    L = ptamd.lib()
    L.pt_set_device(local_rank)
    L.pt_set_mode(ptamd.MODE_CONVERGE)
    L.pt_set_pipeline(ptamd.PIPELINE_FUSED if args.pipeline == "fused" else ptamd.PIPELINE_STAGED)
    w, h, spp = args.width, args.height, args.spp
    sums = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    scaling = args.scaling or "strong"  # one 1024-spp frame whatever the number of GPUs (at N = 1 the two modes coincide)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_run(begin, count, steps, first_frame):
        """`steps` steps of this rank's samples [begin, begin + count), bracketed by barrier + synchronize on both
        sides; returns (max-over-ranks seconds, radiance rays of all ranks, shadow rays of all ranks, this rank's rays)."""
        def step(frame):
            sums.zero_()
            torch.cuda.synchronize(dev)
            ptamd.render_accumulate_device(sums, w, h, begin, count, args.max_depth, frame)
            if world > 1:
                dist.all_reduce(sums, op=dist.ReduceOp.SUM)

        for i in range(args.warmup):
            step(1 + i)
        L.pt_reset_ray_counts()
        L.pt_reset_kernel_stats()
        L.pt_set_kernel_timing(1)  # HIP events on the render stream around every wavefront launch
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(first_frame + i)
        sync()
        elapsed = time.perf_counter() - t0
        L.pt_set_kernel_timing(0)
        rr, rs = ptamd.ray_counts()
        if world > 1:
            stats = torch.tensor([elapsed, float(rr), float(rs)], dtype=torch.float64, device=dev)
            tmax = stats[0:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tot = stats[1:3].clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            return float(tmax[0]), float(tot[0]), float(tot[1]), (rr, rs)
        return elapsed, float(rr), float(rs), (rr, rs)

    if scaling == "strong":
        begin, count = sharded.split_samples(spp, world)[rank]
        total_spp = spp
    else:
        begin, count = sharded.chunk_of(rank, spp)
        total_spp = spp * world
    elapsed, rr_all, rs_all, (rr, rs) = timed_run(begin, count, args.steps, 101)
    rays_all = rr_all + rs_all
    kid = ptamd.K_BOUNCE if args.pipeline == "fused" else ptamd.K_EXTEND
    kst = ptamd.kernel_stats(kid)  # of the headline loop, before anything else is launched

    extra = {}
    if world > 1:
        # the collective alone: the step's one all-reduce of the per-pixel sums, timed by itself
        # (one at a time, as the step issues it: gloo, the rehearsal backend, stalls on several queued device all-reduces)
        sync()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize(dev)
        ar = torch.tensor([(time.perf_counter() - t0) / 5 * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        extra["allreduce_ms"] = float(ar[0])
        extra["allreduce_bytes"] = sums.numel() * 4
        if scaling == "strong":  # the weak figure beside it: every rank its own `spp` samples
            wb, wc = sharded.chunk_of(rank, spp)
            w_el, w_rr, w_rs, _ = timed_run(wb, wc, max(1, min(args.steps, 3)), 201)
            extra["weak_value"] = (w_rr + w_rs) / w_el / 1e6
            extra["weak_ms_per_step"] = w_el / max(1, min(args.steps, 3)) * 1e3
            extra["weak_total_spp"] = spp * world

    if rank == 0:
        avg_ms = kst["total_ms"] / max(kst["timed"], 1)
        # dominant kernel: rays this rank traced per launch of it (the staged extend kernel only traces radiance rays)
        rays_per_launch = ((rr + rs) if args.pipeline == "fused" else rr) / max(kst["launches"], 1)
        achieved_gbs = BYTES_PER_RAY * rays_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        achieved_tflops = FLOP_PER_RAY * rays_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        # HBM bytes per launch and the VALU counters come from the committed PMC passes of this same command
        # (profiles/pmc_traffic.json, written by tools/r02_profiles.py): counters cannot be read inside the timed run
        traffic, pmc, counters_from = None, {}, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and (w, h, count) == (1920, 1080, 1024):
            try:
                blob = open(tpath, "rb").read()
                pmc = json.loads(blob).get(args.pipeline, {})
                traffic = pmc.get("hbm_bytes_per_launch")
                # provenance: these figures are replayed from the tracked file, not measured in this run
                # (its git blob id, so that a reader can tell which collection they belong to)
                counters_from = "profiles/pmc_traffic.json@" + hashlib.sha1(b"blob %d\0" % len(blob) + blob).hexdigest()[:12]
            except Exception:
                traffic, pmc, counters_from = None, {}, None
        common = {"kernel": ptamd.KERNEL_NAMES[kid], "avg_launch_ms": avg_ms, "launches": kst["launches"],
                  "rays_per_launch": rays_per_launch, "traffic": traffic,
                  "measured_here": ["avg_launch_ms", "launches", "rays_per_launch", "achieved", "frac"],
                  "counters_from": counters_from, "counters_collected": pmc.get("collected")}
        hbm = dict(common, bound="hbm", achieved=achieved_gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                   frac=achieved_gbs / HBM_PEAK_GBS, bytes_per_ray=BYTES_PER_RAY)
        cyc = pmc.get("simd_cycles_per_valu_inst")
        valu = dict(common, bound="valu", achieved=achieved_tflops, peak=VALU_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=achieved_tflops / VALU_PEAK_TFLOPS, flop_per_ray=FLOP_PER_RAY,
                    flop_per_ray_basis="static count on the source (10 box tests x 20 + 6 item tests x 35 per BVH query, "
                                       "350 per opaque shading step, transcendentals as 1), not a counter",
                    lane_insts_per_ray_counted=(pmc["valu_insts_per_ray"] * 64.0 * pmc["valu_lane_utilisation"])
                    if pmc.get("valu_insts_per_ray") and pmc.get("valu_lane_utilisation") else None,
                    # SIMD cycles that pass per vector instruction issued, against what the instructions cost to
                    # issue one class at a time (tools/valu_ops.hip: 2.4 cycles for an FMA on registers, 4.2 for
                    # compares, selects, min/max, conversions, 8.2 for transcendentals).  issue_cost = those costs over
                    # the mix the counters saw (SQ_INSTS_VALU_*); in a mix the classes overlap a little, so the measured
                    # figure can fall below it and issue_util is capped at 1.  issue_floor_share = the share of all SIMD
                    # cycles the issued instructions would need if every one were the cheapest kind.
                    # lane_util = active lanes per vector instruction / 64
                    simd_cycles_per_valu_inst=cyc, issue_cost_cycles_per_valu_inst=pmc.get("issue_cost_cycles_per_valu_inst"),
                    issue_util=min(1.0, pmc.get("issue_cost_cycles_per_valu_inst") / cyc) if cyc and pmc.get("issue_cost_cycles_per_valu_inst") else None,
                    issue_floor_share=(2.4 / cyc) if cyc else None,
                    fp32_flops_issued_per_ray=pmc.get("fp32_flops_issued_per_ray"),
                    fp32_flops_on_active_lanes_per_ray=pmc.get("fp32_flops_on_active_lanes_per_ray"),
                    valu_mix=pmc.get("valu_mix"),
                    lane_util=pmc.get("valu_lane_utilisation"), valu_insts_per_ray=pmc.get("valu_insts_per_ray"))
        fused = args.pipeline == "fused"
        result = {
            "metric": "Mrays/s at 1080p default scene (radiance + shadow rays)",
            "value": rays_all / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (built-in Cornell scene of the reference, generated in-process)",
            "config": {"workload": f"cornell_{w}x{h}_{spp}spp_depth{args.max_depth}_converge", "pipeline": args.pipeline,
                       "spp_per_gpu": count, "total_spp": total_spp, "parallelism": f"sample-sharded x{world}"},
            "spp_per_s": total_spp * args.steps / elapsed,
            "paths_per_s": float(w) * h * total_spp * args.steps / elapsed,
            "rays_per_path": rays_all / (float(w) * h * total_spp * args.steps),
            # the fused kernel is bound by vector-instruction issue, the staged kernels by HBM
            "roofline": valu if fused else hbm,
            ("roofline_hbm_equiv" if fused else "roofline_valu"): hbm if fused else valu,
        }
        result.update(extra)
        if world == 1 and not args.no_boundary:
            result.update(boundary_timings(ptamd, L, w, h, spp, args.max_depth))
        if world == 1 and not args.no_cpu_baseline:
            import orc
            # threads: the CPUs this job may use.  A one-GPU box of the pool grants 16 of the machine's CPUs and says so
            # in OMP_NUM_THREADS (os.cpu_count() there is the whole machine); elsewhere, every CPU the process may run on
            host_cores = os.cpu_count() or 1
            usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
            granted = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
            cores = max(1, min(usable, granted if granted > 0 else usable, 256))
            t = time.perf_counter()
            _, _, _, crays = orc.render(w, h, args.cpu_spp, max_depth=args.max_depth, frame=101, mode=orc.CONVERGE, nthreads=cores)
            ct = time.perf_counter() - t
            result["cpu_baseline"] = {"value": sum(crays) / ct / 1e6, "unit": "Mrays/s", "cores": cores, "host_cores": host_cores,
                                      "cores_rule": "min(CPUs in the affinity mask, OMP_NUM_THREADS if set: the box's CPU grant)",
                                      "kind": "port",
                                      "sample": f"same scene and camera, {w}x{h}, {args.cpu_spp} spp ({sum(crays)} rays, {ct:.1f} s); "
                                                "rays/s does not depend on spp"}
        if world == 1 and not args.no_large_scene:
            result["large_scene"] = large_scene_timing()
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
