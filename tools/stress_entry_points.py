"""Random scenes, pipelines, shard counts and guiding states through the render entry points (scratch tool):
every image finite, a repeated render gives the same bits, an instanced scene agrees with its flat copy,
in-library shards leave the image a valid render, the binned trainer keeps working across scene changes."""
import sys
sys.path.insert(0, "rust-pathtracer_amd"); sys.path.insert(0, "tests")
import numpy as np, ptamd as gpu
L = gpu.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
field = gpu.GuidingField((-1, -1, 0), (1, 1, 2))
L.guiding_set_grid_res(4, 4, 4)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    kind = int(rng.integers(0, 4))
    n, sub = int(rng.integers(1, 60)), int(rng.integers(0, 3))
    if rng.integers(0, 4) == 0: n, sub = int(rng.integers(150, 400)), 2  # beyond 2 MiB of tree: staged by default, state grows
    if kind == 0: L.pt_scene_cornell()
    elif kind == 1: L.pt_scene_synthetic(int(rng.integers(1, 99)), n, sub)
    else: L.pt_scene_synthetic_instanced(int(rng.integers(1, 99)), n, sub)
    w, h = int(rng.integers(8, 160)), int(rng.integers(8, 100))
    spp = int(rng.choice([1, 3, 16, 64, 100]))
    depth = int(rng.choice([1, 2, 4, 32]))
    pipe = int(rng.choice([-1, 0, 1]))
    shards = int(rng.choice([1, 1, 2, 3]))
    train = int(rng.choice([0, 0, 1 << 12]))
    L.pt_set_pipeline(pipe); L.pt_set_device_count(shards); L.guiding_set_train_buffer_size(train)
    frame = int(rng.integers(0, 1000))
    L.pt_set_frame(frame)
    a = gpu.render_rgb(w, h, spp, max_depth=depth)
    L.pt_set_frame(frame)
    b = gpu.render_rgb(w, h, spp, max_depth=depth)
    assert np.isfinite(a).all() and a.min() >= 0 and np.array_equal(a, b), (it, kind, w, h, spp, depth, pipe, shards)
    note = ""
    if kind >= 2 and spp >= 64 and depth > 1:
        L.pt_scene_flatten()
        L.pt_set_frame(frame)
        f = gpu.render_rgb(w, h, spp, max_depth=depth)
        rel = abs(f.mean() - a.mean()) / max(f.mean(), 1e-6)
        assert rel < 0.08, (it, rel)
        note = f" flat/instanced mean {rel:.4f}"
    if train:
        field.update_from_ring()
        regions, lobes = field.snapshot()
        gpu.upload_snapshot(regions, lobes)
        L.guiding_set_enabled(1)
        g = gpu.render_rgb(w, h, spp, max_depth=depth)
        L.guiding_set_enabled(0)
        assert np.isfinite(g).all() and g.min() >= 0
        note += f" guided/plain {g.mean() / max(a.mean(), 1e-6):.3f}"
    print(it, ["cornell", "synthetic", "instanced", "instanced"][kind], n, sub, w, h, spp, depth, pipe, shards, train, "mean", round(float(a.mean()), 4), note, flush=True)
L.pt_set_device_count(1); L.guiding_set_train_buffer_size(0); L.pt_set_pipeline(-1)
field.close()
print("ok")
