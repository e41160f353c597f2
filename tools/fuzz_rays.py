#!/usr/bin/env python3
"""Randomised rrt_intersect-vs-oracle comparison: random scene (all kinds; sometimes with the ground sphere blown up to radius
3e19, sometimes with every sphere scaled by a power of ten), rays of random magnitudes with zero, NaN and infinite components
sprinkled in, random t_min / t_max; closest hit compared bit for bit (a NaN t as NaN).   python tools/fuzz_rays.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
from oracle import OracleScene
from tests.conftest import scene_arrays_from_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)
ctx = rrt_amd.Context(0)
t_end = time.time() + budget
n_cases = n_rays = 0
while time.time() < t_end:
    kind = int(rng.integers(0, 4))
    half = int(rng.integers(3, 45)) if kind == 3 else 11
    scene_seed = int(rng.integers(0, 100))
    osc = OracleScene.build(kind, 64, 48, seed=scene_seed, half=half)
    sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
    giant = rng.integers(0, 8) == 0
    scale = np.float32(10.0 ** int(rng.integers(-14, 9))) if rng.integers(0, 4) == 0 else np.float32(1.0)
    method = int(rng.integers(0, 2))
    if giant or scale != 1.0 or method == 1:
        sp = sp.copy()
        sp[:, 0:6] *= scale
        sp[:, 8] *= scale
        if giant:
            g = int(np.argmax(sp[:, 8]))
            sp[g, 8] = np.float32(3e19)
            sp[g, 1] = sp[g, 4] = np.float32(-3e19)
        osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), rrt_amd.Camera.from_buffer_copy(cam21.tobytes()).as_array())
        osc.build_bvh(method)
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    ctx.upload_scene(sp, mid, mt, nd)
    n = int(rng.integers(1, 30000))
    span = np.float32(half + 2.0 if kind == 3 else 13.0) * scale
    org = (rng.uniform(-1, 1, (n, 3)) * span).astype(np.float32)
    org[:, 1] = (rng.uniform(0.02, 3.0, n) * scale).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 1] *= np.float32(rng.choice([1.0, 0.2, 0.01]))
    mag = np.where(rng.integers(0, 4, n) == 0, 10.0 ** rng.integers(-25, 26, n), 1.0).astype(np.float32)   # a quarter of the rays: any magnitude
    d *= mag[:, None]
    far = rng.integers(0, 20, n) == 0
    org[far] *= (10.0 ** rng.integers(0, 30, int(far.sum()))).astype(np.float32)[:, None]                       # far-away origins
    for special in (0.0, -0.0, np.nan, np.inf, -np.inf):
        k = rng.integers(0, 3, n)
        pick = rng.integers(0, 40, n) == 0
        d[pick, k[pick]] = np.float32(special)
    nodir = rng.integers(0, 50, n) == 0
    d[nodir] = 0.0
    badorg = rng.integers(0, 200, n) == 0
    org[badorg, rng.integers(0, 3, n)[badorg]] = np.float32(np.nan)
    tm = rng.uniform(0, 1, n).astype(np.float32)
    t_min = float(rng.choice([0.001, 0.001, 0.0, -1.0, 1e-30, 0.5]))
    t_max = float(rng.choice([3.4028234663852886e38, 3.4028234663852886e38, 50.0, 1e-3, float("inf")]))
    t_o, obj_o, _, _, _ = osc.hit_rays(org, d, tm, t_min=t_min, t_max=t_max)
    t_g, obj_g = ctx.intersect(org, d, tm, t_min=t_min, t_max=t_max)
    nan2 = np.isnan(t_o) & np.isnan(t_g)
    ok = np.array_equal(obj_g, obj_o) and np.array_equal(np.where(nan2, 0, t_g.view(np.uint32)), np.where(nan2, 0, t_o.view(np.uint32)))
    if not ok:
        bad = np.flatnonzero((obj_g != obj_o) | (np.where(nan2, 0, t_g.view(np.uint32)) != np.where(nan2, 0, t_o.view(np.uint32))))
        print(f"MISMATCH kind {kind} half {half} scene_seed {scene_seed} giant {giant} scale {scale} method {method} t_min {t_min} t_max {t_max}: {len(bad)} of {n} rays", flush=True)
        for i in bad[:5]:
            print(f"   org {org[i]!r} d {d[i]!r} time {tm[i]!r}: oracle t {t_o[i]!r} obj {obj_o[i]}  gpu t {t_g[i]!r} obj {obj_g[i]}", flush=True)
        np.savez("gpurun_out/fuzz_rays_bad.npz", org=org[bad], d=d[bad], tm=tm[bad])
        sys.exit(1)
    n_cases += 1
    n_rays += n
    if n_cases % 20 == 0: print(n_cases, "cases", n_rays, "rays", flush=True)
print(f"{n_cases} random cases, {n_rays} rays, all closest hits bit-identical to the oracle", flush=True)
