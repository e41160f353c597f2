#!/usr/bin/env python3
"""Every ray of one oracle render, intersected by the GPU (rrt_intersect: the walk the render uses) and by the oracle: prints the
rays whose closest hit differs.   python tools/find_ray.py kind half nx ny ns depth seed scene_seed"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
from oracle import OracleScene
from tests.conftest import scene_arrays_from_oracle

kind, half, nx, ny, ns, depth, seed, scene_seed = (int(x) for x in sys.argv[1:9])
osc = OracleScene.build(kind, nx, ny, seed=scene_seed, half=half)
rays = []
for idx in range(nx * ny):
    for k in range(ns):
        rays.append(osc.trace_path_rays(nx, ny, seed, idx, k, max_depth=depth))
rays = np.concatenate(rays)
print(len(rays), "rays", flush=True)
ctx = rrt_amd.Context(0)
sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
ctx.upload_scene(sp, mid, mt, nd)
t_o, obj_o, _, _, _ = osc.hit_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
t_b, obj_b, _, _, _ = osc.hit_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6], brute=True)
t_g, obj_g = ctx.intersect(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
miss = np.float32(3.4028234663852886e38)
bad = np.flatnonzero((t_o.view(np.uint32) != t_g.view(np.uint32)) | ((t_o < miss) & (obj_o != obj_g)))
print(len(bad), "rays differ between GPU and oracle;", int(np.count_nonzero((t_o.view(np.uint32) != t_b.view(np.uint32)))), "between the oracle's BVH and brute force", flush=True)
np.set_printoptions(precision=9, floatmode="unique")
for i in bad[:10]:
    print(f"ray {i}: origin {rays[i, 0:3]} dir {rays[i, 3:6]}")
    print(f"   oracle t {t_o[i]!r} obj {obj_o[i]}   brute t {t_b[i]!r} obj {obj_b[i]}   gpu t {t_g[i]!r} obj {obj_g[i]}")
    for o in {int(obj_o[i]), int(obj_g[i])}:
        if o < len(sp):
            print(f"   sphere {o}: centre {sp[o, 0:3]} radius {sp[o, 8]!r} material {mt[mid[o]]}")
np.save("gpurun_out/bad_rays.npy", rays[bad])
