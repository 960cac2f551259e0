import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "rust-pathtracer_amd")
for p in (HERE, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hostcheck():
    """The device headers compiled for the host (tests/hostcheck): logic checks without a GPU."""
    import ctypes as C
    d = os.path.join(HERE, "hostcheck")
    so = os.path.join(d, "libhostcheck.so")
    srcs = [os.path.join(d, "hostcheck.cpp")] + [
        os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith((".h", ".cpp"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", d, "-B", "libhostcheck.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.hc_num_vertices.restype = C.c_int
    return lib


@pytest.fixture(scope="session")
def pt():
    """The product binding; raises if libpt_hip.so is missing (no fallback)."""
    import ptamd
    ptamd.lib()
    return ptamd


@pytest.fixture()
def gpu(pt):
    """Fresh, pinned renderer settings for a GPU test."""
    L = pt.lib()
    L.pt_scene_cornell()
    L.guiding_set_enabled(0)
    L.pt_set_mode(pt.MODE_CONVERGE)
    L.pt_set_pipeline(pt.PIPELINE_FUSED)
    L.pt_set_device_count(1)
    L.pt_set_frame(0)
    L.pt_reset_ray_counts()
    return pt
