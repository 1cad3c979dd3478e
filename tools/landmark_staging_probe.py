#!/usr/bin/env python3
"""What a new landmark cloud costs per SLAM map update: fs_upload_landmarks (k-d ordering of 64-landmark leaves on the host,
chunk spheres, one transfer) on the C3 / C5 clouds, with the ordering on one thread (FS_KD_THREADS=1) and on the threads the
build uses by default (up to eight leaves of the recursion at once), and with the ordering on the device ("cloud.order" 1).  Each setting runs in a child process (the variable is read
per call, the child keeps the measurement clean).  One JSON object; ms, medians of 9.

    python tools/landmark_staging_probe.py
"""
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    fs = importlib.import_module("fit-slam_amd")
    out = {}
    for name in ("C3", "C5"):
        m = fs.synth.CONFIGS[name]["n_landmarks"]
        rng = np.random.default_rng(3)
        lm = np.ascontiguousarray(rng.uniform(-12.0, 12.0, size=(m, 3)).astype(np.float32))
        lm[:, 2] = rng.uniform(0.0, 2.5, size=m)
        s = fs.FrontierScorer(device=0)
        s.set_option("cloud.order", 2 if os.environ.get("FS_PROBE_DEVICE_ORDER") else 0)
        ts = []
        for _ in range(11):
            t0 = time.perf_counter()
            s.upload_landmarks(lm)
            ts.append((time.perf_counter() - t0) * 1e3)
        s.close()
        out[name] = {"landmarks": m, "upload_landmarks_ms": float(np.median(ts[2:]))}
    print(json.dumps(out))


def main():
    if os.environ.get("FS_PROBE_CHILD"):
        return child()
    res = {"what": "fs_upload_landmarks per call (host k-d ordering + spheres + transfer), ms, median of 9; uniform clouds of the configs' sizes",
           "cpus": os.cpu_count()}
    for label, t in (("one_thread", "1"), ("default_threads", None), ("device_order", "gpu")):
        env = dict(os.environ, FS_PROBE_CHILD="1")
        env.pop("FS_KD_THREADS", None)
        env.pop("FS_PROBE_DEVICE_ORDER", None)
        if t == "gpu":
            env["FS_PROBE_DEVICE_ORDER"] = "1"                       # "cloud.order" 1: the ordering on the device (fs_cloud.hip)
        elif t:
            env["FS_KD_THREADS"] = t
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, check=True)
        res[label] = json.loads(p.stdout.strip().splitlines()[-1])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
