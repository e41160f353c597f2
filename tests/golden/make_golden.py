#!/usr/bin/env python3
"""Writes the golden fixtures of tests/golden/ with the CPU oracle (oracle/rrt_oracle.c), per-sample mode.

The reference ships no golden images or known-answer vectors (no tests at all), and it cannot be built here, so these
fixtures pin the GPU path and the oracle to each other and against regressions — not to outputs of the reference.
Shapes: chap12 10x10x4 seed 0 (the criterion bench shape, benches/benchmark.rs:11-16), chap11 40x23x8, book2chap2
32x20x6.  Each .npz holds inputs (scene kind, nx, ny, ns, scene seed, render seed, depth) and outputs (RGB8, f32 sums,
work counters)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import CTR_NAMES, MODE_PER_SAMPLE, OracleScene  # noqa: E402

CASES = [("chap12_10x10x4", 1, 10, 10, 4, 0, 0), ("chap11_40x23x8", 0, 40, 23, 8, 0, 0), ("book2chap2_32x20x6", 2, 32, 20, 6, 0, 5)]

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, kind, nx, ny, ns, scene_seed, seed in CASES:
        osc = OracleScene.build(kind, nx, ny, seed=scene_seed)
        rgb, acc, ctr = osc.render(nx, ny, ns, seed=seed, mode=MODE_PER_SAMPLE, max_depth=50)
        np.savez_compressed(os.path.join(here, name + ".npz"), kind=kind, nx=nx, ny=ny, ns=ns, scene_seed=scene_seed, seed=seed,
                            max_depth=50, rgb=rgb, acc=acc, counters=np.array([ctr[k] for k in CTR_NAMES], np.uint64))
        print(name, rgb.mean(), ctr["rays"])
