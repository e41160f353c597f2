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

# ---- rrt_render_multi over 8 contexts that share this GPU (RRT_MULTI_LOOPBACK), 3840x2160: what the call costs on top of
# the rendering.  With one sample per pixel the rendering itself is a few milliseconds, so the wall time of a warm call is
# an upper bound of set-up + combine (pixel lists, staging, the shares' copies to rank 0, their placement, the copy to the
# host); the first call also makes the plan (threads, buffers, pixel lists).
import time
os.environ["RRT_MULTI_LOOPBACK"] = "1"
os.environ["RRT_MULTI_TIMING"] = "1"  # the phases of every call on stderr
nx, ny, world = 3840, 2160, 8
hs4k = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
ctxs = [rrt_amd.Context(0) for _ in range(world)]
for c in ctxs:
    c.upload_host_scene(hs4k)
for mode, name in ((rrt_amd.SHARD_TILE, "TILE"), (rrt_amd.SHARD_SAMPLE, "SAMPLE")):
    ns = 1 if mode == rrt_amd.SHARD_TILE else world
    p = rrt_amd.Context.params(nx, ny, ns, seed=0, shard_mode=mode)
    walls = []
    for i in range(6):
        t0 = time.perf_counter()
        rgb, _, st = rrt_amd.render_multi(ctxs, p, want_accum=False)
        walls.append((time.perf_counter() - t0) * 1e3)
    one = rrt_amd.Context.params(nx, ny, 1, seed=0)
    t0 = time.perf_counter(); ctxs[0].render(one, want_accum=False); t1 = time.perf_counter(); ctxs[0].render(one, want_accum=False); t2 = time.perf_counter()
    print(f"rrt_render_multi {name} x{world} loopback {nx}x{ny}, {ns} spp in all: first call {walls[0]:.1f} ms, warm calls {statistics.median(walls[1:]):.2f} ms "
          f"(one context, rrt_render of the whole image at 1 spp: {(t2 - t1) * 1e3:.2f} ms; slowest rank's render inside the warm call {st['gpu_ms']:.2f} ms)")
for c in ctxs:
    c.close()
