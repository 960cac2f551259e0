"""The training-statistics table of pt_guiding_bin_train_samples_device restated with numpy (test infrastructure):
the region of a record by its position, the lobe by the largest dot(dir_in, axis) in float32 without fused
multiply-adds, eight 64-bit fixed-point sums per (region, lobe)."""
import numpy as np

LOBES, WORDS, SHIFT = 6, 8, 24
MAX_WEIGHT, MIN_WEIGHT = 2.0 ** 20, 2.0 ** -25  # PT_PGL_MAX_WEIGHT, PT_PGL_MIN_WEIGHT


def bin_records(records, bmin, bmax, res, axes):
    """records: (n, 10) float32 TrainSample rows whose last column is the plain is_delta flag (as
    guiding_map_train_samples returns them).  Returns the int64 table, flat."""
    rec = np.asarray(records, np.float32).reshape(-1, 10)
    table = np.zeros((int(np.prod(res)), LOBES, WORDS), np.int64)
    f32 = np.float32
    pos, d, rgb = rec[:, 0:3], rec[:, 3:6], rec[:, 6:9]
    ext = (bmax - bmin).astype(f32)
    inside = ((pos >= bmin) & (pos <= bmax) & (ext > 0)).all(1)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        cell = ((pos - bmin).astype(f32) / ext * res.astype(f32)).astype(f32)
        cell = np.minimum(np.nan_to_num(cell, nan=0.0, posinf=0.0, neginf=0.0).astype(np.int64), res.astype(np.int64) - 1)
        weight = ((rgb[:, 0] + rgb[:, 1]).astype(f32) + rgb[:, 2]).astype(f32) / f32(3.0)
    with np.errstate(invalid="ignore"):
        ok = inside & (weight >= f32(MIN_WEIGHT)) & (weight <= f32(MAX_WEIGHT)) & (np.abs(rgb) <= f32(MAX_WEIGHT)).all(1)
    region = (cell[:, 2] * int(res[1]) + cell[:, 1]) * int(res[0]) + cell[:, 0]
    scale = float(1 << SHIFT)
    for i in np.nonzero(ok)[0]:
        mu = axes[region[i]]
        dots = ((d[i, 0] * mu[:, 0]).astype(f32) + (d[i, 1] * mu[:, 1]).astype(f32)).astype(f32) + (d[i, 2] * mu[:, 2]).astype(f32)
        lobe = int(np.argmax(dots.astype(f32)))  # first of equal maxima, like the strict > of the trainer
        t = table[region[i], lobe]
        w = float(weight[i])
        for k in range(3):
            t[k] += int(np.rint(w * float(d[i, k]) * scale))
            t[4 + k] += int(np.rint(float(rgb[i, k]) * scale))
        t[3] += int(np.rint(w * scale))
        t[7] += 1
    return table.reshape(-1)
