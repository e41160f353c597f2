#!/usr/bin/env python3
"""The reference's own benchmark shape (benches/benchmark.rs:11-23): chap12 scene, 10x10 pixels, 4 samples per pixel,
one `tracescene` call per iteration (scene build outside the timed region).  Prints the median time per call through the
C++ mirror of the reference API, and through the bare render entry point with the scene already on the device."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd

nx, ny, ns = 10, 10, 4
hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
ts = []
for i in range(30):
    t0 = time.perf_counter()
    hs.tracescene(ns, device=0)
    ts.append(time.perf_counter() - t0)
print(f"tracescene mirror (flatten + upload + render + copy back): median {statistics.median(ts[5:]) * 1e3:.3f} ms per call")
ctx = rrt_amd.Context(0)
ctx.upload_host_scene(hs)
p = rrt_amd.Context.params(nx, ny, ns, seed=0)
ts = []
for i in range(50):
    t0 = time.perf_counter()
    ctx.render(p)
    ts.append(time.perf_counter() - t0)
print(f"rrt_render, scene resident: median {statistics.median(ts[5:]) * 1e3:.3f} ms per call")
