#!/usr/bin/env python3
"""Ray counts of the sample shards and of the tile shards of one render, GPU against oracle.
   python tools/find_shard.py kind half nx ny ns depth slots batch seed scene_seed"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
from oracle import OracleScene
from tests.conftest import scene_arrays_from_oracle

kind, half, nx, ny, ns, depth, slots, batch, seed, scene_seed = (int(x) for x in sys.argv[1:11])
os.environ["RRT_STREAMS"] = "1"
osc = OracleScene.build(kind, nx, ny, seed=scene_seed, half=half)
ctx = rrt_amd.Context(0)
sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
ctx.upload_scene(sp, mid, mt, nd)
ctx.set_camera(rrt_amd.Camera.from_buffer_copy(cam21.tobytes()))
for r in range(ns):
    _, _, ct = osc.render(nx, ny, ns, seed=seed, max_depth=depth, threads=8, k_begin=r, k_step=ns)
    p = rrt_amd.Context.params(nx, ny, ns, seed=seed, max_depth=depth, path_slots=slots, batch_samples=batch, force_hbm_bvh=True,
                               shard_mode=rrt_amd.SHARD_SAMPLE, rank=r, world=ns)
    _, _, st = ctx.render(p)
    print(f"sample shard {r}: gpu rays {st['rays']} oracle {ct['rays']}{'   <---' if st['rays'] != ct['rays'] else ''}", flush=True)
    if st["rays"] != ct["rays"]:
        # which pixel: the oracle's ray count per path of this sample against a GPU render of every pixel's tile alone is not
        # available; list the oracle's paths of this sample that end absorbed on metal near depth 21 instead
        for idx in range(nx * ny):
            rays = osc.trace_path_rays(nx, ny, seed, idx, r, max_depth=depth)
            if 15 <= len(rays) <= 30:
                print(f"   candidate path: pixel {idx} sample {r}: {len(rays)} rays", flush=True)
