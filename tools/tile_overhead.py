#!/usr/bin/env python3
"""What rendering a TILE shard (pixels through a pixel map) costs against an unsharded render of as many paths:
python tools/tile_overhead.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rrt_amd
nx, ny = 1200, 800
ctx = rrt_amd.Context(0)
hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
ctx.upload_host_scene(hs)
dev = torch.device("cuda", 0)
rgb = torch.zeros(nx * ny * 3, dtype=torch.uint8, device=dev)
def run(p, n=4):
    ts, rays = [], 0
    for i in range(n + 1):
        st = ctx.render_device(p, rgb.data_ptr(), 0)
        if i: ts.append(st["gpu_ms"]); rays = st["rays"]
    return statistics.median(ts), rays
for label, p in (("unsharded, 250 spp", rrt_amd.Context.params(nx, ny, 250, seed=0)),
                 ("TILE shard 0 of 2, 500 spp", rrt_amd.Context.params(nx, ny, 500, seed=0, shard_mode=rrt_amd.SHARD_TILE, rank=0, world=2)),
                 ("SAMPLE shard 0 of 2, 500 spp", rrt_amd.Context.params(nx, ny, 500, seed=0, shard_mode=rrt_amd.SHARD_SAMPLE, rank=0, world=2))):
    ms, rays = run(p)
    print(f"{label:32s} {ms:8.2f} ms  {rays / ms / 1e3:9.1f} Mrays/s")
