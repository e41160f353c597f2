#!/usr/bin/env python3
"""Randomised comparison of rrt_build_bvh_method (the tree built on the GPU) with the oracle's Bvh::build restatement, node for node:
random counts (1 .. 60000), distributions (uniform, clustered, on a grid with many equal centres, along a line, a few giants), moving
spheres, negative radii, both partition methods.   python tools/fuzz_bvh.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
from oracle import OracleScene

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = rrt_amd.Context(0)
t_end = time.time() + budget
n_cases = 0
while time.time() < t_end:
    n = int(rng.choice([1, 2, 3, int(rng.integers(4, 64)), int(rng.integers(64, 2000)), int(rng.integers(2000, 60000))]))
    dist = int(rng.integers(0, 5))
    sp = np.zeros((n, 9), np.float32)
    if dist == 0:
        c = rng.uniform(-50, 50, (n, 3))
    elif dist == 1:
        c = rng.normal(0, 1, (n, 3)) + rng.integers(-3, 4, (n, 3)) * 20.0
    elif dist == 2:
        c = rng.integers(-4, 5, (n, 3)).astype(np.float64)          # many equal centres: the equal-count fallback at several depths
    elif dist == 3:
        c = np.zeros((n, 3)); c[:, int(rng.integers(0, 3))] = rng.uniform(-100, 100, n)   # along one axis
    else:
        c = rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.integers(-6, 7, (n, 1))
    sp[:, 0:3] = sp[:, 3:6] = c
    sp[:, 7] = 1.0
    sp[:, 8] = rng.uniform(0.01, 1.0, n) * rng.choice([1.0, 1.0, 1.0, 30.0], n)
    if rng.integers(0, 2):
        mv = rng.integers(0, 4, n) == 0
        sp[mv, 3:6] += rng.normal(0, 0.5, (int(mv.sum()), 3)).astype(np.float32)
    if rng.integers(0, 3) == 0:
        sp[rng.integers(0, 20, n) == 0, 8] *= -1.0
    mt = np.zeros((n, 6), np.float32); mt[:, 1:4] = 0.5
    osc = OracleScene.from_arrays(sp, np.arange(n, dtype=np.uint32), mt, np.zeros((1, 8), np.uint32))
    for method in (0, 1):
        osc.build_bvh(method)
        _, _, nd, _ = osc.export()
        got = ctx.build_bvh(sp, method)
        if got.shape != nd.shape or not np.array_equal(got, nd):
            print(f"MISMATCH n {n} dist {dist} method {method}: shapes {got.shape} {nd.shape}", flush=True)
            np.save("gpurun_out/fuzz_bvh_bad.npy", sp)
            sys.exit(1)
    n_cases += 1
    if n_cases % 50 == 0: print(n_cases, "cases", flush=True)
print(f"{n_cases} random sphere sets, both partition methods: every node of the GPU-built tree equals the oracle's", flush=True)
