#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison: random scene, image size, sample count, depth limit, pool size, batch size, pool
parts, with and without the finishing launch, every image compared bit for bit.  python tools/fuzz.py [seconds] [seed] [big]
("big": images large enough to fill the path pool, so that the workgroup-wide hand-out of the extend kernel is in use)"""
import os
import sys
import time

import numpy as np
import torch  # before rrt_amd: the process must load torch's own HIP runtime first

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
from oracle import OracleScene
from tests.conftest import scene_arrays_from_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = rrt_amd.Context(0)
t_end = time.time() + budget
big = len(sys.argv) > 3 and sys.argv[3] == "big"
n_cases = n_noalloc = 0
while time.time() < t_end:
    kind = int(rng.integers(0, 4))
    half = int(rng.integers(3, 40)) if kind == 3 else 11
    nx, ny = (int(rng.integers(300, 700)), int(rng.integers(200, 420))) if big else (int(rng.integers(1, 200)), int(rng.integers(1, 120)))
    ns = int(rng.integers(4, 24)) if big else int(rng.integers(1, 20))
    depth = int(rng.choice([0, 1, 3, 50, 50, 120, 300]))  # 300: past the narrow state's 255
    slots = int(rng.choice([256 * 64, 256 * 512, 256 * 1024, 256 * 4096])) if big else int(rng.choice([0, 256, 512, 4096, 256 * 33, 256 * 1024]))
    batch = int(rng.choice([0, 1, 3, 7]))
    parts = str(int(rng.integers(1, 5)))
    force_hbm = bool(rng.integers(0, 2))
    seed, scene_seed = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 100))
    os.environ["RRT_STREAMS"] = parts
    if rng.integers(0, 4) == 0:
        os.environ["RRT_NO_FINISH"] = "1"  # rounds to the end instead of the finishing launch
    else:
        os.environ.pop("RRT_NO_FINISH", None)
    os.environ["RRT_FINISH_MERGE"] = str(int(rng.choice([0, 1, 2, 2])))  # 2: the live lists are merged before the finishing launch whatever the pool's size
    os.environ["RRT_FINISH_DELAY"] = str(int(rng.choice([0, 1, 3, 3, 7])))
    osc = OracleScene.build(kind, nx, ny, seed=scene_seed, half=half)
    sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
    if rng.integers(0, 8) == 0:
        # spheres set in motion with shutter intervals of their own (center_at divides by each sphere's interval, sphere.rs:61-64), a few
        # of them with a negative radius or a -0.0 centre component
        sp = sp.copy()
        n_s = sp.shape[0]
        mv = rng.integers(0, 3, n_s) == 0
        mv[int(np.argmax(sp[:, 8]))] = False
        sp[mv, 3:6] = sp[mv, 0:3] + rng.normal(0.0, 0.15, (int(mv.sum()), 3)).astype(np.float32)
        if rng.integers(0, 2):
            sp[:, 6] = rng.uniform(-0.5, 0.5, n_s).astype(np.float32)
            sp[:, 7] = sp[:, 6] + rng.uniform(0.1, 2.0, n_s).astype(np.float32)
        neg = rng.integers(0, 40, n_s) == 0
        neg[int(np.argmax(sp[:, 8]))] = False
        sp[neg, 8] = -sp[neg, 8]
        sp[rng.integers(0, 50, n_s) == 0, 2] = np.float32(-0.0)
        cam = rrt_amd.Camera.from_buffer_copy(cam21.tobytes())
        if cam.time_scale == 0.0: cam.time_scale = 1.0  # (the still scenes' cameras keep the shutter closed at time 0)
        cam21 = cam.as_array()
        osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam21)
        osc.build_bvh(int(rng.integers(0, 2)))
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    if rng.integers(0, 6) == 0:
        # other materials than the scene builders make: any refraction index (below 1 too), fuzz beyond 1, albedo beyond 1, kinds reshuffled
        mt = mt.copy()
        n_m = mt.shape[0]
        if rng.integers(0, 2): mt[:, 0] = rng.integers(0, 3, n_m)
        mt[:, 1:4] = rng.uniform(0.0, 1.2, (n_m, 3))
        mt[:, 4] = np.where(mt[:, 0] == 1, rng.uniform(0.0, 1.5, n_m), 0.0)
        mt[:, 5] = np.where(mt[:, 0] == 2, rng.choice([1.5, 1.0, 0.5, 2.4, 1.33, 0.9999, 1.0001, 3.0], n_m) * rng.choice([1.0, 1.0, rng.uniform(0.9, 1.1)], n_m), 0.0)
        mt = mt.astype(np.float32)
        osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam21)
        osc.build_bvh(0)
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    if rng.integers(0, 6) == 0:
        # the whole scene and its camera scaled by a power of ten: the short reciprocal / division / square-root sequences of the kernels
        # are exact over stated ranges and hand the rest to the IEEE routines, wave by wave
        k = np.float32(10.0 ** int(rng.integers(-16, 10)))
        sp = sp.copy()
        sp[:, 0:6] *= k
        sp[:, 8] *= k
        cam = rrt_amd.Camera.from_buffer_copy(cam21.tobytes())
        for field in (cam.lower_left_corner, cam.horizontal, cam.vertical, cam.origin):
            for i in range(3): field[i] = np.float32(field[i]) * k
        cam.lens_radius = np.float32(cam.lens_radius) * k
        cam21 = cam.as_array()
        osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam21)
        osc.build_bvh(int(rng.integers(0, 2)))
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    if rng.integers(0, 12) == 0:
        # the ground sphere blown up until |o - c|^2 - r^2 is inf - inf: a NaN root, which is accepted; t_max turns NaN in the middle of
        # walks and can grow again after that.  Tree by either builder: with EqualCount the giant sits deep in the tree.
        ground = int(np.argmax(sp[:, 8]))
        sp = sp.copy()
        sp[ground, 8] = np.float32(3e19)
        sp[ground, 1] = sp[ground, 4] = np.float32(-3e19)
        osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), rrt_amd.Camera.from_buffer_copy(cam21.tobytes()).as_array())
        osc.build_bvh(int(rng.integers(0, 2)))
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    ctx.upload_scene(sp, mid, mt, nd)
    ctx.set_camera(rrt_amd.Camera.from_buffer_copy(cam21.tobytes()))
    if rng.integers(0, 16) == 0:  # a viewport without extent at the lens: every camera ray has direction (0, 0, 0), and NaN rays follow
        cam = rrt_amd.Camera()
        eye = rng.uniform(-8.0, 8.0, 3) * (1.0, 0.15, 1.0) + (0.0, 0.6, 0.0)
        cam.lower_left_corner[:] = eye
        cam.origin[:] = eye
        cam.u[:] = (1.0, 0.0, 0.0)
        cam.v[:] = (0.0, 1.0, 0.0)
        cam.time_scale = 1.0
        osc.set_camera21(cam.as_array())
        ctx.set_camera(cam)
    shard = {}
    world = rank = 0
    if rng.integers(0, 8) == 0:  # one sample shard of a multi-GPU render: samples rank, rank + world, ... (lib.rs:96-103 split over ranks)
        world = int(rng.integers(2, 6))
        rank = int(rng.integers(0, world))
        shard = dict(shard_mode=rrt_amd.SHARD_SAMPLE, rank=rank, world=world)
        rgb_o, acc_o, ct = osc.render(nx, ny, ns, seed=seed, max_depth=depth, threads=8, k_begin=rank, k_step=world)
    else:
        rgb_o, acc_o, ct = osc.render(nx, ny, ns, seed=seed, max_depth=depth, threads=8)
    p = rrt_amd.Context.params(nx, ny, ns, seed=seed, max_depth=depth, path_slots=slots, batch_samples=batch, force_hbm_bvh=force_hbm,
                               count_work=bool(rng.integers(0, 2)), **shard)
    rgb, acc, st = ctx.render(p)
    if world: rgb_o = rgb  # (a shard's own RGB8 is not what the oracle's sharded render returns: its f32 sums and counters are compared)
    tag = f"shard {rank}/{world} kind {kind} half {half} {nx}x{ny}x{ns} depth {depth} slots {slots} batch {batch} parts {parts} hbm {force_hbm} seed {seed}/{scene_seed} merge {os.environ['RRT_FINISH_MERGE']} delay {os.environ['RRT_FINISH_DELAY']} no_finish {os.environ.get('RRT_NO_FINISH', '0')}"
    # (a sum that is NaN on both sides counts as equal: the sign of a NaN that an operation makes is the processor's choice)
    nan2 = np.isnan(acc) & np.isnan(acc_o)
    if nan2.any(): print("note:", int(nan2.sum()), "NaN sums on both sides,", tag, flush=True)
    ok = np.array_equal(np.where(nan2, 0, acc.view(np.uint32)), np.where(nan2, 0, acc_o.view(np.uint32))) and np.array_equal(rgb, rgb_o) and st["rays"] == ct["rays"]
    if ok and p.count_work:
        ok = all(st[k] == ct[k] for k in ("inner_visits", "leaf_tests", "hits_lambertian", "hits_metal", "hits_dielectric", "sky_misses", "depth_cut", "absorbed"))
    if ok and not big and rng.integers(0, 8) == 0:
        # the same render through the entry point that allocates nothing (its own workspace, its own batch and pool sizes, big scenes
        # along the exit links); sizes beyond its workspace are refused, which is not a mismatch
        dev = torch.device("cuda", 0)
        d_rgb = torch.full((ny * nx * 3,), 7, dtype=torch.uint8, device=dev)
        d_acc = torch.full((ny * nx * 3,), 3.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        try:
            st_n = ctx.render_device_noalloc(p, d_rgb.data_ptr(), d_acc.data_ptr())
        except RuntimeError:
            st_n = None
        torch.cuda.synchronize()
        if st_n is not None:
            n_noalloc += 1
            acc_n = d_acc.cpu().numpy().reshape(ny, nx, 3)
            same = np.array_equal(np.where(np.isnan(acc_n) & np.isnan(acc), 0, acc_n.view(np.uint32)), np.where(np.isnan(acc_n) & np.isnan(acc), 0, acc.view(np.uint32)))
            ok = same and np.array_equal(d_rgb.cpu().numpy().reshape(ny, nx, 3), rgb) and st_n["rays"] == st["rays"]
            if not ok: tag = "NOALLOC " + tag
    if not ok:
        print("MISMATCH", tag, flush=True)
        sys.exit(1)
    n_cases += 1
    if n_cases % 20 == 0: print(n_cases, "cases", flush=True)
print(f"{n_cases} random cases, all bit-identical to the oracle ({n_noalloc} of them also through the entry point that allocates nothing)", flush=True)
