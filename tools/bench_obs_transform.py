#!/usr/bin/env python3
"""HBM roofline of the fused ResizeShortestEdge + CenterCropper launch at the ObjectNav sensor geometry (640x480 -> 256x256,
rgb u8 + depth f32 + semantic i32), beside the reference's op chain run with torch on the same device and on the host cores.
Algorithmic bytes per frame = source pixels inside the crop's footprint read once + output written once.
usage: python tools/bench_obs_transform.py [frames]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "habitat-lab_amd"))
from habitat_amd.common.obs_transformers import CenterCropper, ResizeShortestEdge, apply_obs_transforms_batch  # noqa: E402
from oracle import functional as O  # noqa: E402  (the reference's op chain, used here as the thing to compare against)

HBM_PEAK_GBS = 8000.0


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    H, W, S = 480, 640, 256
    obs = {"rgb": torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device="cuda"),
           "depth": torch.rand(n, H, W, 1, device="cuda"),
           "semantic": torch.randint(0, 40, (n, H, W, 1), dtype=torch.int32, device="cuda")}
    ts = [ResizeShortestEdge(S), CenterCropper(S)]
    # the crop keeps columns [42, 298) of the 341-wide resized image = source columns [78, 560): 482 of 640 (area); the
    # nearest-neighbour sensor reads exactly one source pixel per output pixel
    bytes_of = {"rgb": n * (H * 482 + S * S) * 3, "depth": n * (H * 482 + S * S) * 4, "semantic": n * (S * S + S * S) * 4}
    t = timeit(lambda: apply_obs_transforms_batch(dict(obs), ts))
    gbs = sum(bytes_of.values()) / t / 1e6
    print(f"fused HIP launch x3 sensors : {t:7.3f} ms / {n} frames  {gbs:7.1f} GB/s algorithmic = {gbs / HBM_PEAK_GBS:.1%} of HBM peak, "
          f"{n / t * 1e3:9.0f} frames/s")
    for k in obs:
        tk = timeit(lambda: apply_obs_transforms_batch({k: obs[k]}, ts))
        print(f"    {k:9s}: {tk:7.3f} ms  {bytes_of[k] / tk / 1e6:7.1f} GB/s algorithmic")

    def torch_chain(d):
        return {k: O.center_crop(O.resize_shortest_edge(v, S, "nearest" if k == "semantic" else "area"), S).contiguous() for k, v in d.items()}
    t2 = timeit(lambda: torch_chain(obs), 5)
    print(f"reference op chain, torch on the same GPU : {t2:7.3f} ms  ({t2 / t:.1f}x slower)")
    cpu = {k: v[:8].cpu() for k, v in obs.items()}
    t0 = time.perf_counter()
    torch_chain(cpu)
    t3 = (time.perf_counter() - t0) * 1e3 * (n / 8)
    print(f"reference op chain, torch CPU ({torch.get_num_threads()} threads, scaled from 8 frames): {t3:7.1f} ms  ({t3 / t:.0f}x slower)")


if __name__ == "__main__":
    main()
