#!/usr/bin/env python3
"""Time of rrt_build_bvh (GPU, reference tree) against the host builder on the 1 M-sphere scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rrt_amd
t0 = time.perf_counter()
hs = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, 1920, 1080, seed=0, half=500)
t_host = time.perf_counter() - t0
sp, mt, nd, cam = hs.export()
ctx = rrt_amd.Context(0)
ctx.build_bvh(sp[:1000])
ts = []
for _ in range(3):
    t0 = time.perf_counter(); got = ctx.build_bvh(sp); ts.append(time.perf_counter() - t0)
print(f"{sp.shape[0]} spheres, {nd.shape[0]} nodes: host scene + BVH build {t_host:.2f} s; rrt_build_bvh {min(ts) * 1e3:.1f} ms "
      f"(planes up, nodes back, included); identical: {np.array_equal(got, nd)}")
ts = []
for _ in range(3):
    t0 = time.perf_counter(); ctx.upload_host_scene(hs); ts.append(time.perf_counter() - t0)
print(f"rrt_upload_scene (validation, node words, exit links, top-of-tree copy, two-children records, copies to the device): {min(ts):.2f} s")
