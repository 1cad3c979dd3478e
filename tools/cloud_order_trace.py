#!/usr/bin/env python3
"""fs_upload_landmarks on the C3-sized cloud, 20 times, with the ordering on the device — run under
`rocprofv3 --kernel-trace --stats` to see what the 1.2 ms are made of (tools/landmark_staging_probe.py times the call)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fs = importlib.import_module("fit-slam_amd")
rng = np.random.default_rng(3)
lm = np.ascontiguousarray(rng.uniform(-12.0, 12.0, size=(100_000, 3)).astype(np.float32))
s = fs.FrontierScorer(device=0)
s.set_option("cloud.order", 2)
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    s.upload_landmarks(lm)
    ts.append((time.perf_counter() - t0) * 1e3)
print("upload_landmarks ms, median of the last 15:", float(np.median(ts[5:])))
s.close()
