#!/usr/bin/env python3
"""Set-up times of the 1 M-sphere scene: the scene builder with the tree built on the host (RRT_HOST_BVH=1) and on the GPU
(Scene::build_bvh -> rrt_build_bvh_method), rrt_build_bvh_method alone for both partition methods, and rrt_upload_scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rrt_amd
ctx = rrt_amd.Context(0)
os.environ["RRT_HOST_BVH"] = "1"
t0 = time.perf_counter()
hs_host = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, 1920, 1080, seed=0, half=500)
t_host = time.perf_counter() - t0
del os.environ["RRT_HOST_BVH"]
rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, 64, 64, seed=0, half=100).close()   # the build context and its first launches
ts = []
for _ in range(3):
    t0 = time.perf_counter(); hs = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, 1920, 1080, seed=0, half=500); ts.append(time.perf_counter() - t0)
    if len(ts) < 3: hs.close()
sp, mt, nd_host, cam = hs_host.export()
_, _, nd, _ = hs.export()
print(f"{sp.shape[0]} spheres, {nd.shape[0]} nodes: scene builder with the tree built on the host {t_host:.2f} s, on the GPU {min(ts):.2f} s; same nodes: {np.array_equal(nd, nd_host)}")
ctx.build_bvh(sp[:1000])
for method, name in ((0, "Middle"), (1, "EqualCount")):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); got = ctx.build_bvh(sp, method); ts.append(time.perf_counter() - t0)
    print(f"rrt_build_bvh_method {name}: {min(ts) * 1e3:.1f} ms (planes up, nodes back, included)" + (f"; identical to the host tree: {np.array_equal(got, nd_host)}" if method == 0 else ""))
ts = []
for _ in range(3):
    t0 = time.perf_counter(); ctx.upload_host_scene(hs); ts.append(time.perf_counter() - t0)
print(f"rrt_upload_scene (validation, node words, exit links, top-of-tree copy, two-children records, copies to the device): {min(ts):.2f} s")
