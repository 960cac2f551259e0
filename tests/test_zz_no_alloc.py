"""The allocation-free render entry point against the allocating one it is a variant of.

pt_render_accumulate_device_noalloc keeps path state, queues, counters and the scene in one static array of
the library's code object; its result must be pt_render_accumulate_device's, bit for bit, and a call must not
depend on what the previous call left in the array."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(gpu, w, h, begin, count, depth, frame):
    import torch
    L = gpu.lib()
    dev = torch.device("cuda:0")
    a = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    b = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu.render_accumulate_device(a, w, h, begin, count, depth, frame)
    rc = L.pt_render_accumulate_device_noalloc(w, h, begin, count, depth, frame, b.data_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_render_equals_allocating_render(gpu):
    import torch
    L = gpu.lib()
    # first call: 160 x 96 pixels, 64 samples from global sample 0 (two chunks of 32 side by side)
    a, b = _both(gpu, 160, 96, 0, 64, 32, 7)
    assert np.isfinite(a).all() and a.mean() > 0.05
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # second call, smaller in every way (fewer pixels, fewer samples, one chunk, a share that starts at sample 16):
    # nothing of the first call's state in the array may leak into it
    a2, b2 = _both(gpu, 64, 48, 16, 16, 32, 7)
    assert a2.mean() > 0.05
    assert np.array_equal(a2.view(np.uint32), b2.view(np.uint32))
    # and the first size again gives the first result again
    a3, b3 = _both(gpu, 160, 96, 0, 64, 32, 7)
    assert np.array_equal(a.view(np.uint32), b3.view(np.uint32))
    # it ADDS into the output like the allocating call
    dev = torch.device("cuda:0")
    acc = torch.ones(64 * 48 * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    assert L.pt_render_accumulate_device_noalloc(64, 48, 16, 16, 32, 7, acc.data_ptr()) == 0
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), 1.0 + b2, rtol=1e-6, atol=1e-6)


def test_noalloc_render_rejects_what_does_not_fit(gpu, capfd):
    import torch
    L = gpu.lib()
    dev = torch.device("cuda:0")
    out = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    assert L.pt_render_accumulate_device_noalloc(64, 48, 0, 4, 32, 1, None) == -1
    # 2^20 pixels: more path slots than the static array holds
    assert L.pt_render_accumulate_device_noalloc(1024, 1024, 0, 4, 32, 1, out.data_ptr()) == -2
    # 2^18 pixels fit with one chunk (fewer than 64 samples), not with two
    assert L.pt_render_accumulate_device_noalloc(512, 512, 0, 64, 32, 1, out.data_ptr()) == -2
    assert L.pt_render_accumulate_device_noalloc(512, 512, 0, 2, 4, 1, out.data_ptr()) == 0
    torch.cuda.synchronize()
    assert float(out[: 512 * 512 * 3].sum()) > 0.0 and float(out[512 * 512 * 3:].abs().sum()) == 0.0
    # a scene larger than the space reserved for it
    L.pt_scene_synthetic(42, 8, 2)
    assert L.pt_render_accumulate_device_noalloc(64, 48, 0, 4, 32, 1, out.data_ptr()) == -2
    L.pt_scene_cornell()
    assert L.pt_render_accumulate_device_noalloc(64, 48, 0, 4, 32, 1, out.data_ptr()) == 0
    err = capfd.readouterr().err
    assert "exceed" in err
