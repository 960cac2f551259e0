"""Sample-sharded rendering across the GPUs of one node: one process per GPU, per-GPU sums
reduced over RCCL (torch.distributed backend "nccl"), as a collective only at the end.

The reference is single-GPU (optix_wrapper.cpp:788 takes device 0 and nothing else); what is
sharded here is the sample loop of optix_wrapper.cpp:1002-1021.  Samples are independent, so
rank r renders the global sample indices [r*spp, (r+1)*spp) of every pixel and the ranks'
per-pixel sums add up to the sums of one GPU rendering the same chunks one after another.
"""
import torch
import torch.distributed as dist


def chunk_of(rank, spp_per_rank):
    """Global sample range [begin, begin+count) rendered by `rank` (weak scaling)."""
    return rank * spp_per_rank, spp_per_rank


def split_samples(total_spp, world):
    """Strong scaling: contiguous ranges covering [0,total_spp), sizes differing by at most 1."""
    base, rem = divmod(total_spp, world)
    out, begin = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((begin, n))
        begin += n
    return out


def render_sharded(render_chunk, w, h, begin, count, total_spp, group=None):
    """render_chunk(sum_tensor, begin, count) adds this rank's per-pixel RGB sums into
    sum_tensor (float32, w*h*3, on the rank's device).  Returns the mean image [h,w,3] on
    every rank (all-reduce), identical on all ranks."""
    device = render_chunk.device if hasattr(render_chunk, "device") else None
    sums = torch.zeros(w * h * 3, dtype=torch.float32, device=device)
    render_chunk(sums, begin, count)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return (sums / float(total_spp)).view(h, w, 3)


class DeviceChunkRenderer:
    """render_chunk callable backed by pt_render_accumulate_device on this rank's GPU."""

    def __init__(self, w, h, max_depth, frame, device):
        import ptamd
        self.pt = ptamd
        self.w, self.h, self.max_depth, self.frame = w, h, max_depth, frame
        self.device = torch.device(device)
        ptamd.lib().pt_set_device(self.device.index or 0)

    def __call__(self, sums, begin, count):
        torch.cuda.synchronize(self.device)  # sums was zeroed on torch's stream
        self.pt.render_accumulate_device(sums, self.w, self.h, begin, count, self.max_depth, self.frame)


def render_sharded_adaptive(render_share, w, h, begin, count, group=None, device=None):
    """Adaptive variant: render_share(sum_tensor, count_tensor, begin, count) adds this rank's per-pixel RGB
    sums (w*h*3) and sample counts (w*h); every rank stops its pixels on its own share's noise estimate.
    Two all-reduces (sums, counts); returns (mean image [h,w,3], counts [h,w]) on every rank."""
    sums = torch.zeros(w * h * 3, dtype=torch.float32, device=device)
    counts = torch.zeros(w * h, dtype=torch.float32, device=device)
    render_share(sums, counts, begin, count)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    mean = sums.view(h, w, 3) / counts.clamp(min=1.0).view(h, w, 1)
    return mean, counts.view(h, w)


def render_sharded_adaptive_shared(render_share, w, h, begin, count, total_spp, group=None, device=None):
    """Adaptive render with one noise test for all ranks.  render_share(stats_tensor, begin, count, total_spp,
    reduce) renders this rank's samples round by round and calls reduce() whenever the ranks' statistics
    (stats_tensor: w*h*8 floats per pixel: sum rgb, samples, sum of squares rgb, 0) have to be added
    together -- at every round boundary and once when it has finished, the same number of times on every
    rank.  Returns (mean image [h,w,3], samples per pixel [h,w]), identical on all ranks."""
    stats = torch.zeros(w * h * 8, dtype=torch.float32, device=device)
    many = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def reduce():
        if many:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        if stats.is_cuda:
            torch.cuda.synchronize(stats.device)

    render_share(stats, begin, count, total_spp, reduce)
    v = stats.view(h, w, 8)
    counts = v[..., 3]
    return v[..., 0:3] / counts.clamp(min=1.0).unsqueeze(-1), counts


class DeviceSharedAdaptiveRenderer:
    """render_share callable backed by pt_render_adaptive_shared_device on this rank's GPU."""

    def __init__(self, w, h, max_depth, frame, noise_threshold, min_adaptive_samples, device):
        import ptamd
        self.pt = ptamd
        self.w, self.h, self.max_depth, self.frame = w, h, max_depth, frame
        self.threshold, self.min_samples = noise_threshold, min_adaptive_samples
        self.device = torch.device(device)
        ptamd.lib().pt_set_device(self.device.index or 0)

    def __call__(self, stats, begin, count, total_spp, reduce):
        torch.cuda.synchronize(self.device)  # stats was zeroed on torch's stream
        self.pt.render_adaptive_shared_device(stats, self.w, self.h, begin, count, total_spp, self.max_depth,
                                              self.frame, self.threshold, self.min_samples, reduce)


class ShardedGuiding:
    """The guiding loop of src/guiding.rs:222-247 for several GPUs.  The trainer keeps ONE field
    (openpgl_bridge.cpp:47-58), so every rank's training ring is gathered on rank 0, which updates the field
    and takes the snapshot; the snapshot is broadcast and every rank uploads it to its own renderer.

    collect() returns this rank's records, (n, 10) float32 in the TrainSample layout; upload(regions, lobes)
    installs a snapshot; reset() empties the ring.  The defaults are the library calls; tests pass stand-ins."""

    def __init__(self, pt, scene_min, scene_max, group=None, collect=None, upload=None, reset=None):
        self.pt, self.group = pt, group
        self.many = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if self.many else 0
        self.world = dist.get_world_size(group) if self.many else 1
        # RCCL ("nccl") moves device tensors only: the records and the snapshot then pass through this rank's GPU;
        # gloo takes them from host memory as they are
        self.cdev = torch.device("cpu")
        if self.many and dist.get_backend(group) == "nccl":
            self.cdev = torch.device("cuda", torch.cuda.current_device())
        self.field = pt.GuidingField(scene_min, scene_max) if self.rank == 0 else None
        self.collect = collect or pt.train_samples
        self.upload = upload or pt.upload_snapshot
        self.reset = reset or (lambda: pt.lib().guiding_reset_train_write_idx())

    def process_batch(self):
        """One update: gather, train on rank 0, broadcast, upload.  Returns (regions, lobes), the same on all ranks."""
        local = torch.from_numpy(self.collect().astype("float32", copy=False).reshape(-1, 10).copy())
        dev = self.cdev
        if self.many:
            sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
            dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=dev), group=self.group)
            sizes = [int(x) for x in sizes]
            longest = max(sizes)
            padded = torch.zeros((longest, 10), dtype=torch.float32, device=dev)
            padded[: local.shape[0]] = local.to(dev)
            parts = [torch.zeros_like(padded) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(padded, parts, dst=0, group=self.group)
            if self.rank == 0:
                local = torch.cat([p[:n] for p, n in zip(parts, sizes)]).cpu()
        if self.rank == 0:
            self.field.update(local.numpy())
            regions, lobes = self.field.snapshot()
            shape = torch.tensor([regions.shape[0], lobes.shape[0]], dtype=torch.int64, device=dev)
        else:
            shape = torch.zeros(2, dtype=torch.int64, device=dev)
        if self.many:
            dist.broadcast(shape, src=0, group=self.group)
            nr, nl = int(shape[0]), int(shape[1])
            r = torch.from_numpy(regions).to(dev) if self.rank == 0 else torch.zeros((nr, 8), dtype=torch.float32, device=dev)
            l = torch.from_numpy(lobes).to(dev) if self.rank == 0 else torch.zeros((nl, 8), dtype=torch.float32, device=dev)
            dist.broadcast(r, src=0, group=self.group)
            dist.broadcast(l, src=0, group=self.group)
            regions, lobes = r.cpu().numpy(), l.cpu().numpy()
        self.upload(regions, lobes)
        self.reset()
        return regions, lobes

    def close(self):
        if self.field is not None:
            self.field.close()
            self.field = None


class BinnedGuiding:
    """The guiding loop with the training statistics taken on the GPUs (pt_guiding_bin_train_samples_device).

    Every rank keeps its own copy of the field.  A batch bins the rank's ring on its device into the int64 table of
    per-region, per-lobe sums, one all-reduce(SUM) adds the tables of all ranks (24 KB for a 4 x 4 x 4 field, in
    place of gathering a 10.5 MB ring per rank), and every rank applies the total to its copy.  Integer sums do not
    depend on order or on how the samples were spread over ranks, so the copies stay identical bit for bit and no
    snapshot has to be broadcast.

    bin(field) returns the rank's table as a torch int64 tensor (default: on the GPU from the renderer's ring);
    upload / reset as in ShardedGuiding.  Tests pass stand-ins."""

    def __init__(self, pt, scene_min, scene_max, group=None, bin=None, upload=None, reset=None):
        self.pt, self.group = pt, group
        self.many = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.field = pt.GuidingField(scene_min, scene_max)
        res = self.field.layout()[2]
        self.words = int(res[0]) * int(res[1]) * int(res[2]) * pt.GuidingField.BIN_LOBES * pt.GuidingField.BIN_WORDS
        self.bin = bin or self._bin_on_device
        self.upload = upload or pt.upload_snapshot
        self.reset = reset or (lambda: pt.lib().guiding_reset_train_write_idx())

    def _bin_on_device(self, field):
        table = torch.zeros(self.words, dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
        torch.cuda.synchronize()
        field.bin_ring_device(table.data_ptr())
        return table

    def process_batch(self):
        table = self.bin(self.field)
        if self.many:
            if dist.get_backend(self.group) == "nccl" and not table.is_cuda:
                table = table.cuda()
            dist.all_reduce(table, op=dist.ReduceOp.SUM, group=self.group)
        self.field.update_binned(table.cpu().numpy())
        regions, lobes = self.field.snapshot()
        self.upload(regions, lobes)
        self.reset()
        return regions, lobes

    def close(self):
        if self.field is not None:
            self.field.close()
            self.field = None

