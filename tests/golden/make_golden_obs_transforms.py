#!/usr/bin/env python3
"""Golden outputs of the REAL reference's observation-transformer arithmetic (habitat_baselines/utils/common.py:
image_resize_shortest_edge :481-528, center_crop :531-557, loaded in place through oracle/ref_loader.py) on deterministic inputs.

  python tests/golden/make_golden_obs_transforms.py      # needs /root/reference; writes tests/golden/obs_transforms.npz

Inputs are re-created by oracle.functional.obs_transform_inputs(seed, n, h, w); only the reference's outputs are stored (the
640x480 case as a strided sample)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.functional import obs_transform_inputs  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402

# name -> (seed, n, h, w, resize size, crop (h, w), sample stride)
CASES = {
    "s48x64": (1, 2, 48, 64, 32, (32, 32), 1),
    "s30x50": (2, 1, 30, 50, 20, (16, 24), 1),
    "s64x64_half": (3, 2, 64, 64, 32, (32, 32), 1),
    "s37x53_up": (5, 1, 37, 53, 48, (40, 44), 1),      # upsampling: windows of one pixel, nearest scale > 1
    "objectnav_480x640": (4, 1, 480, 640, 256, (256, 256), 7),
}


def main():
    ns = load_reference()
    out = {}
    for name, (seed, n, h, w, size, crop, stride) in CASES.items():
        obs = obs_transform_inputs(seed, n, h, w)
        for k, v in obs.items():
            mode = "nearest" if k == "semantic" else "area"
            r = ns.common.image_resize_shortest_edge(v, size, channels_last=True, interpolation_mode=mode)
            c = ns.common.center_crop(r, crop, channels_last=True)
            out[f"{name}/{k}/resized_shape"] = np.array(r.shape)
            out[f"{name}/{k}/resized"] = r.numpy().reshape(-1)[::stride].copy()
            out[f"{name}/{k}/cropped"] = c.contiguous().numpy().reshape(-1)[::stride].copy()
    np.savez_compressed(os.path.join(HERE, "obs_transforms.npz"), **out)
    print("wrote obs_transforms.npz:", sum(v.nbytes for v in out.values()), "bytes uncompressed")


if __name__ == "__main__":
    main()
