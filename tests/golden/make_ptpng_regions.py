"""Region means of the reference's committed render pt.png (1920x1440, 8-bit, tone-mapped by
code that is no longer in the reference tree) -> ptpng_regions.json.  Qualitative pin of the
scene layout only: which wall is red / green, black side bars, saturated light.
Run where /root/reference exists:  python tests/golden/make_ptpng_regions.py"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
img = np.asarray(Image.open("/root/reference/pt.png").convert("RGB"), np.float64)
H, W, _ = img.shape
# regions as fractions of the image (x0, x1, y0, y1)
regions = {
    "left_bar": (0.0, 0.03, 0.1, 0.9),
    "right_bar": (0.97, 1.0, 0.1, 0.9),
    "left_wall": (0.08, 0.22, 0.3, 0.7),
    "right_wall": (0.78, 0.92, 0.3, 0.7),
    "back_wall": (0.40, 0.60, 0.30, 0.55),
    "light": (0.45, 0.55, 0.07, 0.11),
    "floor": (0.35, 0.65, 0.90, 0.97),
}
out = {"size": [W, H], "regions": {}}
for name, (x0, x1, y0, y1) in regions.items():
    patch = img[int(y0 * H):int(y1 * H), int(x0 * W):int(x1 * W)]
    out["regions"][name] = {"frac": [x0, x1, y0, y1], "mean_rgb8": patch.mean(axis=(0, 1)).round(2).tolist()}
with open(os.path.join(HERE, "ptpng_regions.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["regions"], indent=1))
