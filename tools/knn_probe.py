#!/usr/bin/env python
"""distCUDA2 (gsr_knn_mean_dist2) at 100 k and 1 M points under rocprofv3 --kernel-trace --stats: the per-kernel split
(AABB, Morton codes, radix sort, gather + box min/max, neighbour scan) behind bench.py's `knn` leg.  tools/gpu.sh knnstats."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from photo_slam_amd import rasterize_points as rp  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
for n in ([int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]):
    pts = torch.from_numpy((rng.random((n, 3), dtype=np.float32) * np.array([6, 3, 6], np.float32) - np.array([3, 1.5, 3], np.float32))).to(dev)
    for _ in range(5):
        d = rp.distCUDA2(pts)
    torch.cuda.synchronize()
    print(n, float(d.mean()))
