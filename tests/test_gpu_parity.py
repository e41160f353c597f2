"""GPU path vs CPU oracle, through the C ABI.  Bit-exact: f32 sums compared as bit patterns, RGB8 equal,
work counters equal.  Everything here needs an MI355X."""
import os

import numpy as np
import pytest

import rrt_amd
from oracle import MODE_PER_SAMPLE, OracleScene
from tests.conftest import scene_arrays_from_oracle

pytestmark = pytest.mark.gpu

SCENES = {"chap11": rrt_amd.SCENE_CHAP11, "chap12": rrt_amd.SCENE_CHAP12, "book2chap2": rrt_amd.SCENE_BOOK2CHAP2}


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def upload_from_oracle(ctx, osc):
    sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
    ctx.upload_scene(sp, mid, mt, nd)
    cam = rrt_amd.Camera.from_buffer_copy(cam21.tobytes())
    ctx.set_camera(cam)


def assert_same_image(gpu, ref):
    rgb_g, acc_g, st_g = gpu
    rgb_o, acc_o, ct_o = ref
    nbad = int((bits(acc_g) != bits(acc_o)).sum())
    assert nbad == 0, f"{nbad} of {acc_o.size} f32 sums differ; max abs err {np.abs(acc_g - acc_o).max()}"
    assert np.array_equal(rgb_g, rgb_o)
    assert st_g["rays"] == ct_o["rays"]
    assert st_g["paths"] == ct_o["paths"]


# ---------------------------------------------------------------- extend alone: Bvh::hit
@pytest.mark.parametrize("name", list(SCENES))
def test_intersect_primary_rays(gpu_ctx, name):
    nx, ny = 96, 64
    osc = OracleScene.build(SCENES[name], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    idx = np.tile(np.arange(nx * ny, dtype=np.uint32), 2)
    k = np.repeat(np.arange(2, dtype=np.uint32), nx * ny)
    rays, _ = osc.gen_primary(nx, ny, 7, idx, k)
    t_o, obj_o, _, _, _ = osc.hit_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
    t_g, obj_g = gpu_ctx.intersect(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
    assert np.array_equal(obj_g, obj_o)
    assert np.array_equal(bits(t_g), bits(t_o))
    assert (obj_o != 0xFFFFFFFF).sum() > 100  # the test is not vacuous


@pytest.mark.parametrize("name", list(SCENES))
def test_intersect_random_and_degenerate_rays(gpu_ctx, name):
    osc = OracleScene.build(SCENES[name], 64, 64, seed=3)
    upload_from_oracle(gpu_ctx, osc)
    rng = np.random.default_rng(5)
    n = 20000
    org = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
    org[:, 1] = rng.uniform(0.05, 3.0, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    # axis-parallel and zero components: 1/0 = inf, 0*inf = NaN in the slab test (acceleration.rs:81-82)
    d[0:2000, 0] = 0.0
    d[2000:4000, 1] = 0.0
    d[4000:6000, 2] = -0.0
    d[6000:6500, 0:2] = 0.0
    # directions and origins outside the range in which the sphere roots are divided the short way (d.d beyond 2^+-30,
    # origin components beyond 2^30): whole waves of them, and single rays inside waves that are otherwise in range
    d[6500:8000] *= np.float32(1e6)
    d[8000:9500] *= np.float32(1e-6)
    org[9500:10000, 0] += np.float32(3e9)
    d[12345] *= np.float32(1e6)
    d[15000] *= np.float32(1e-6)
    org[17777, 2] = np.float32(-2e9)
    # no direction at all (what refracted() returns past total internal reflection, ultraviolet 0.9.2 vec.rs): every root is
    # 0 / 0, and a NaN root is neither below t_min nor beyond t_max (sphere.rs:74-81): each sphere tested is accepted, t_max
    # turns NaN, f32::min drops it in the box tests that follow (acceleration.rs:81), and the last sphere in walk order wins.
    # And rays that are NaN through and through, which is what follows such a hit.
    d[18000:18400] = 0.0
    d[18400:18500] = np.float32(np.nan)
    org[18450:18500] = np.float32(np.nan)
    tm = rng.uniform(0, 1, n).astype(np.float32)
    t_o, obj_o, _, _, _ = osc.hit_rays(org, d, tm)
    t_g, obj_g = gpu_ctx.intersect(org, d, tm)
    assert np.array_equal(obj_g, obj_o)
    assert np.array_equal(bits(t_g), bits(t_o))
    # brute force (lib.rs:32-47) agrees wherever the closest t is unique
    t_b, obj_b, _, _, _ = osc.hit_rays(org, d, tm, brute=True)
    assert np.array_equal(bits(t_b[:18000]), bits(t_o[:18000]))  # (a ray without a direction is accepted by every sphere it is tested against)


def test_intersect_degenerate_rays_in_a_big_scene(gpu_ctx):
    """The same kinds of rays through the walk of scenes whose nodes stay in HBM (two children per record, far children on
    a stack): zero and negative-zero direction components, huge and tiny directions, far-away origins, and origins
    inside the field of spheres, where many boxes overlap the ray's start."""
    osc = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, 64, 64, seed=3, half=60)
    upload_from_oracle(gpu_ctx, osc)
    rng = np.random.default_rng(6)
    n = 30000
    org = rng.uniform(-70, 70, (n, 3)).astype(np.float32)
    org[:, 1] = rng.uniform(0.05, 3.0, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 1] *= np.float32(0.2)            # mostly along the ground, through many spheres' boxes
    d[0:3000, 0] = 0.0
    d[3000:6000, 1] = 0.0
    d[6000:9000, 2] = -0.0
    d[9000:9500, 0:2] = 0.0
    d[9500:11000] *= np.float32(1e6)
    d[11000:12500] *= np.float32(1e-6)
    org[12500:13000, 0] += np.float32(3e9)
    org[13000:13500, 1] = np.float32(0.2)  # at the height of the small spheres' centres
    # no direction at all: every sphere tested is accepted at t = NaN, and waiting children are popped against a NaN t_max
    # (the reference's f32::min drops it: the box is hit); NaN rays, which follow such a hit
    d[13500:15000] = 0.0
    d[15000:15200] = np.float32(np.nan)
    org[15100:15200] = np.float32(np.nan)
    # rays whose discriminants overflow for the spheres far from their origin (inf - inf: a NaN root, accepted) and not for those
    # near it: t_max is a number first and NaN later in the walk, and boxes that lay beyond the number are hit again
    d[15200:17000] *= np.float32(1e13)
    org[16000:17000] *= np.float32(1e4)
    d[17000:18000] *= np.float32(3e16)
    tm = rng.uniform(0, 1, n).astype(np.float32)
    t_o, obj_o, _, _, _ = osc.hit_rays(org, d, tm)
    t_g, obj_g = gpu_ctx.intersect(org, d, tm)
    assert np.array_equal(obj_g, obj_o)
    assert np.array_equal(bits(t_g), bits(t_o))
    assert (obj_o != 0xFFFFFFFF).sum() > 10000 and (obj_o > 0).sum() > 1000  # not only the ground
    # a t_min that is not positive (the two-children walk is for t_min > 0: these go along the exit links) and a short t_max
    for t_min, t_max in ((0.0, 3.4028234663852886e38), (-0.5, 40.0), (0.25, 15.0)):
        t_o, obj_o, _, _, _ = osc.hit_rays(org, d, tm, t_min=t_min, t_max=t_max)
        t_g, obj_g = gpu_ctx.intersect(org, d, tm, t_min=t_min, t_max=t_max)
        assert np.array_equal(obj_g, obj_o), (t_min, t_max)
        assert np.array_equal(bits(t_g), bits(t_o)), (t_min, t_max)


# ---------------------------------------------------------------- whole path: tracescene
@pytest.mark.parametrize("name,nx,ny,ns", [
    ("chap11", 40, 23, 8),       # the chap11 fixture shape
    ("chap12", 10, 10, 4),       # benches/benchmark.rs:11-13
    ("chap12", 96, 64, 16),
    ("book2chap2", 96, 64, 8),
    ("chap12", 1, 1, 3),
    ("chap11", 7, 1, 1),
])
def test_render_bit_exact(gpu_ctx, name, nx, ny, ns):
    osc = OracleScene.build(SCENES[name], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=11, mode=MODE_PER_SAMPLE, threads=8)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=11))
    assert_same_image(gpu, ref)
    assert gpu[2]["progress"] == nx * ny


@pytest.mark.parametrize("name", ["chap11", "chap12", "book2chap2"])
def test_work_counters_match(gpu_ctx, name):
    nx, ny, ns = 64, 48, 6
    osc = OracleScene.build(SCENES[name], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    _, _, ct = osc.render(nx, ny, ns, seed=5, threads=8)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=5, count_work=True))
    st = gpu[2]
    for key in ("rays", "inner_visits", "leaf_tests", "hits_lambertian", "hits_metal", "hits_dielectric", "sky_misses", "paths",
                "depth_cut", "absorbed"):
        assert st[key] == ct[key], key
    # the counting build computes the same image
    plain = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=5))
    assert np.array_equal(bits(plain[1]), bits(gpu[1]))


@pytest.mark.parametrize("max_depth", [0, 1, 2, 50])
def test_depth_limit(gpu_ctx, max_depth):
    nx, ny, ns = 48, 32, 4
    osc = OracleScene.build(SCENES["chap11"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=2, max_depth=max_depth, threads=4)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=2, max_depth=max_depth))
    assert_same_image(gpu, ref)


@pytest.mark.parametrize("name", list(SCENES))
def test_lds_walk_and_hbm_walk_agree(gpu_ctx, name):
    """Small scenes are walked from an LDS copy of the BVH, large ones from HBM: same kernels otherwise, same bits."""
    nx, ny, ns = 96, 64, 8
    osc = OracleScene.build(SCENES[name], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=21, threads=8)
    for force_hbm in (False, True):
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=21, force_hbm_bvh=force_hbm, count_work=True))
        assert_same_image(gpu, ref)
        assert gpu[2]["inner_visits"] == ref[2]["inner_visits"] and gpu[2]["leaf_tests"] == ref[2]["leaf_tests"]


def test_schedule_independence(gpu_ctx):
    """Slots in flight and batch size change the schedule, never a bit of the result."""
    nx, ny, ns = 80, 50, 12
    osc = OracleScene.build(SCENES["book2chap2"], nx, ny, seed=1)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=9, threads=8)
    for slots, batch in [(256, 1), (1024, 5), (4096, 12), (0, 0), (256 * 70, 7)]:
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=9, path_slots=slots, batch_samples=batch))
        assert_same_image(gpu, ref)


def test_pool_parts_and_streams_do_not_change_the_image(gpu_ctx, monkeypatch):
    """The path pool worked as 1..4 parts on as many streams (RRT_STREAMS), with a pool big enough for the workgroup-wide
    ray hand-out of the extend kernel to be in use: same bits every time."""
    nx, ny, ns = 320, 200, 24
    osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=4, threads=8)
    for parts in ["1", "2", "3", "4"]:
        monkeypatch.setenv("RRT_STREAMS", parts)
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=4, path_slots=256 * 1024))
        assert_same_image(gpu, ref)
    # the order in which an extend workgroup takes the rays of its pool (camera rays last, or slot order): same bits, same
    # work, and a scene with moving spheres as well
    for kind, seed in (("chap12", 0), ("book2chap2", 3)):
        osc = OracleScene.build(SCENES[kind], nx, ny, seed=seed)
        upload_from_oracle(gpu_ctx, osc)
        ref = osc.render(nx, ny, ns, seed=7, threads=8)
        for apart in ("0", "1"):
            monkeypatch.setenv("RRT_FRESH_APART", apart)
            gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=7, path_slots=256 * 1024, count_work=True))
            assert_same_image(gpu, ref)
            for k in ("rays", "inner_visits", "leaf_tests", "paths"):
                assert gpu[2][k] == ref[2][k], (kind, apart, k)


def test_finishing_launch_and_plain_rounds_agree(gpu_ctx, monkeypatch):
    """The end of a batch runs as one launch (a wave loops over extend and shade on its own list) where the scene allows
    it; RRT_NO_FINISH=1 keeps launching rounds instead.  Same image, same work counters, for a render bigger than the
    pool and for one that fits the pool whole (then the single launch is the entire render)."""
    osc = OracleScene.build(SCENES["book2chap2"], 160, 100, seed=2)
    upload_from_oracle(gpu_ctx, osc)
    for ns, slots in [(16, 64 * 1024), (3, 0)]:
        ref = osc.render(160, 100, ns, seed=6, threads=8)
        runs = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("RRT_NO_FINISH", "1")
            else:
                monkeypatch.delenv("RRT_NO_FINISH", raising=False)
            gpu = gpu_ctx.render(rrt_amd.Context.params(160, 100, ns, seed=6, path_slots=slots, count_work=True))
            assert_same_image(gpu, ref)
            for k in ("rays", "inner_visits", "leaf_tests", "paths"):
                assert gpu[2][k] == ref[2][k], (k, off)
            runs.append(gpu[2]["iterations"])
        assert runs[0] < runs[1]  # the finishing launch stands for many rounds
    monkeypatch.delenv("RRT_NO_FINISH", raising=False)


@pytest.mark.parametrize("delay", ["0", "3"])
def test_lists_merged_before_the_finishing_launch(gpu_ctx, monkeypatch, delay):
    """Once the last path of a batch is dealt the host queues a few more plain rounds, merges the thinned-out live lists into
    full ones for the first waves and runs the finishing launch on those (a list is only the slot numbers a wave works on).
    Pools as small as a test's skip the merge unless RRT_FINISH_MERGE=2 asks for it: same image and work counters with and
    without, for one batch and for several."""
    for scene, moving in (("chap12", False), ("book2chap2", True)):
        osc = OracleScene.build(SCENES[scene], 160, 100, seed=3)
        upload_from_oracle(gpu_ctx, osc)
        ref = osc.render(160, 100, 12, seed=9, threads=8)
        for batch in (0, 5):
            for merge in ("2", "0"):
                monkeypatch.setenv("RRT_FINISH_MERGE", merge)
                monkeypatch.setenv("RRT_FINISH_DELAY", delay)
                gpu = gpu_ctx.render(rrt_amd.Context.params(160, 100, 12, seed=9, path_slots=64 * 1024, batch_samples=batch, count_work=True))
                assert_same_image(gpu, ref)
                for k in ("rays", "inner_visits", "leaf_tests", "paths", "sky_misses", "depth_cut"):
                    assert gpu[2][k] == ref[2][k], (scene, batch, merge, k)
    # lists that are still full when the merge runs do not fit lists of 255: they are taken as they are (a pool of one list,
    # and one of 40, merged at once)
    osc = OracleScene.build(SCENES["chap12"], 68, 65, seed=59)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(68, 65, 6, seed=5, max_depth=120, threads=8)
    monkeypatch.setenv("RRT_FINISH_MERGE", "2")
    monkeypatch.setenv("RRT_FINISH_DELAY", delay)
    for slots in (256, 40 * 256):
        gpu = gpu_ctx.render(rrt_amd.Context.params(68, 65, 6, seed=5, max_depth=120, path_slots=slots, batch_samples=1))
        assert_same_image(gpu, ref)
    monkeypatch.delenv("RRT_FINISH_MERGE", raising=False)
    monkeypatch.delenv("RRT_FINISH_DELAY", raising=False)


def test_tile_sharding_composes_to_the_unsharded_image(gpu_ctx):
    nx, ny, ns = 100, 37, 6   # not a multiple of the 32x8 tile in either direction
    osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    full = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=4))
    for world in (2, 3, 8):
        rgb = np.zeros_like(full[0])
        acc = np.zeros_like(full[1])
        owned = np.zeros(nx * ny, np.int32)
        rays = 0
        for rank in range(world):
            r, a, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=4, shard_mode=rrt_amd.SHARD_TILE, rank=rank, world=world))
            px = rrt_amd.shard_pixels(nx, ny, rrt_amd.SHARD_TILE, rank, world)
            owned[px] += 1
            mask = np.zeros(nx * ny, bool)
            mask[px] = True
            mask = mask.reshape(ny, nx)
            assert not r[~mask].any() and not a[~mask].any()   # nothing written outside the shard
            rgb += r
            acc += a
            rays += st["rays"]
        assert (owned == 1).all()
        assert np.array_equal(rgb, full[0])
        assert np.array_equal(bits(acc), bits(full[1]))
        assert rays == full[2]["rays"]


def test_sample_sharding_matches_oracle_strides(gpu_ctx):
    nx, ny, ns, world = 60, 40, 10, 3
    osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    total = np.zeros((ny, nx, 3), np.float64)
    for rank in range(world):
        _, a, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=4, shard_mode=rrt_amd.SHARD_SAMPLE, rank=rank, world=world))
        _, a_o, ct = osc.render(nx, ny, ns, seed=4, k_begin=rank, k_step=world, threads=8)
        assert np.array_equal(bits(a), bits(a_o))
        assert st["rays"] == ct["rays"]
        total += a
    _, full, _ = osc.render(nx, ny, ns, seed=4, threads=8)
    # a different order of f32 additions: equal up to rounding of ns additions
    assert np.allclose(total, full, rtol=2e-6, atol=1e-6)


def test_custom_scene_general_motion(gpu_ctx):
    """Spheres whose shutter intervals differ (per-sphere division), a -0.0 centre component, a negative radius,
    a one-sphere scene (root is a leaf)."""
    rng = np.random.default_rng(1)
    n = 40
    sp = np.zeros((n, 9), np.float32)
    sp[:, 0:3] = rng.uniform(-3, 3, (n, 3))
    sp[:, 3:6] = sp[:, 0:3] + rng.uniform(-0.4, 0.4, (n, 3)) * (rng.uniform(size=(n, 1)) < 0.5)
    sp[:, 6] = rng.choice([0.0, 0.25], n)
    sp[:, 7] = rng.choice([1.0, 0.75, 2.0], n)
    sp[:, 8] = rng.uniform(0.2, 0.6, n)
    sp[0, 0] = -0.0
    sp[0, 3] = -0.0
    sp[1, 8] = -0.3
    mt = np.zeros((n, 6), np.float32)
    mt[:, 0] = rng.integers(0, 3, n)
    mt[:, 1:4] = rng.uniform(0.1, 0.9, (n, 3))
    mt[:, 4] = rng.uniform(0, 0.5, n)
    mt[:, 5] = 1.5
    mid = np.arange(n, dtype=np.uint32)
    cam = rrt_amd.camera_from((0, 1, 9), (0, 0, 0), (0, 1, 0), 40.0, 1.5, 0.05, 9.0, 0.0, 1.0)
    for count in (n, 1):
        osc = OracleScene.from_arrays(sp[:count], mid[:count], mt[:count], np.zeros((0, 8), np.uint32), cam.as_array())
        osc.build_bvh(0)
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
        gpu_ctx.upload_scene(sp[:count], mid[:count], mt[:count], nd)
        gpu_ctx.set_camera(cam)
        nx, ny, ns = 60, 40, 6
        ref = osc.render(nx, ny, ns, seed=3, threads=8)
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=3))
        assert_same_image(gpu, ref)


def test_normal_division_in_and_out_of_the_short_range(gpu_ctx):
    """The outward normal (p - c) / radius (sphere.rs:88) is divided through the reciprocal of the radius where every
    operand is in range, and by the division otherwise.  Camera rays that lie in the plane x = 0 hit spheres centred on
    that plane with p.x - c.x == 0 exactly (out of range: whole waves divide); after the first bounce the rays are
    general (in range).  A second scene has a radius beyond 2^30, which switches the short route off for the scene."""
    sp = np.zeros((4, 9), np.float32)
    sp[0, 0:3] = sp[0, 3:6] = (0.0, 0.0, 0.0)
    sp[1, 0:3] = sp[1, 3:6] = (0.0, -100.5, 0.0)
    sp[2, 0:3] = sp[2, 3:6] = (0.0, 0.9, 0.3)
    sp[3, 0:3] = sp[3, 3:6] = (0.0, -0.25, 1.5)
    sp[:, 7] = 1.0
    sp[:, 8] = (0.5, 100.0, -0.3, 0.2)
    mt = np.array([[0, 0.8, 0.3, 0.3, 0, 0], [0, 0.5, 0.5, 0.5, 0, 0], [2, 1, 1, 1, 0, 1.5], [1, 0.8, 0.8, 0.9, 0.2, 0]], np.float32)
    mid = np.arange(4, dtype=np.uint32)
    cam = rrt_amd.Camera()
    cam.lower_left_corner[:] = (0.0, -0.75, 3.5)
    cam.horizontal[:] = (0.0, 1.5, 0.0)     # no x component anywhere: every camera ray has direction.x == 0 and origin.x == 0
    cam.vertical[:] = (0.0, 0.0, 1.0)
    cam.origin[:] = (0.0, 0.0, 5.0)
    cam.u[:] = (0.0, 1.0, 0.0)
    cam.v[:] = (0.0, 0.0, 1.0)
    cam.lens_radius, cam.time_low, cam.time_scale = 0.0, 0.0, 0.0
    for big_radius in (False, True):
        sp2 = sp.copy()
        if big_radius:
            sp2[1, 0:3] = sp2[1, 3:6] = (0.0, -2.0e9 - 0.5, 0.0)
            sp2[1, 8] = 2.0e9
        osc = OracleScene.from_arrays(sp2, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
        osc.build_bvh(0)
        _, _, _, nd, _ = scene_arrays_from_oracle(osc)
        assert bool(rrt_amd.debug_layout(sp2, mid, mt, nd)[2] & 8) == (not big_radius)
        gpu_ctx.upload_scene(sp2, mid, mt, nd)
        gpu_ctx.set_camera(cam)
        nx, ny, ns = 64, 48, 8
        ref = osc.render(nx, ny, ns, seed=11, threads=8)
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=11, count_work=True))
        assert_same_image(gpu, ref)
        assert ref[2]["hits_lambertian"] > 1000 and ref[2]["hits_dielectric"] > 50 and ref[2]["hits_metal"] > 50, ref[2]


@pytest.mark.parametrize("kind", ["all_diffuse", "all_glass", "all_metal"])
def test_rejection_work_list_extremes(gpu_ctx, kind):
    """The shade kernel runs the in-unit-sphere rejection loop (rng.rs:32-40) off one work list per wave: the list positions
    whose slot hit a Lambertian or metal sphere.  Camera inside a closed sphere of one material: every slot of every wave
    wants a sample (a full work list of 256) nearly always, until the depth limit, or none ever does (glass: an empty list)."""
    sp = np.zeros((2, 9), np.float32)
    sp[0, 0:3] = sp[0, 3:6] = (0.0, 0.0, 0.0)
    sp[1, 0:3] = sp[1, 3:6] = (0.3, 0.1, -1.0)
    sp[:, 7] = 1.0
    sp[:, 8] = (50.0, 0.4)
    m = {"all_diffuse": [0, 0.7, 0.6, 0.5, 0, 0], "all_glass": [2, 1, 1, 1, 0, 1.5], "all_metal": [1, 0.9, 0.8, 0.7, 0.3, 0]}[kind]
    mt = np.array([m, m], np.float32)
    mid = np.arange(2, dtype=np.uint32)
    cam = rrt_amd.camera_from((0, 0, 2), (0, 0, -1), (0, 1, 0), 50.0, 1.25, 0.1, 3.0, 0.0, 0.0)
    osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    osc.build_bvh(0)
    _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    gpu_ctx.set_camera(cam)
    nx, ny, ns = 80, 64, 4   # 20480 paths over an 8192-slot pool: full lists, regenerated in place
    ref = osc.render(nx, ny, ns, seed=21, max_depth=12, threads=8)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=21, max_depth=12, count_work=True, path_slots=8192))
    assert_same_image(gpu, ref)
    hits = {k: ref[2][k] for k in ("hits_lambertian", "hits_metal", "hits_dielectric")}
    assert sum(1 for v in hits.values() if v) == 1 and max(hits.values()) > 20000, hits
    assert gpu[2]["depth_cut"] == ref[2]["depth_cut"] and ref[2]["depth_cut"] > 0   # (glass: a few paths are caught inside the small sphere)


@pytest.mark.parametrize("scale", [1e-15, 2e-8, 1.0, 1e7])
def test_scaled_scenes_reach_the_slow_paths_of_the_short_sequences(gpu_ctx, scale):
    """Sphere::hit (sphere.rs:68-85) takes a square root of the discriminant and divides by |d|^2; the kernels do both with
    short sequences that are exact over a stated range and hand the rest to the IEEE routines, wave by wave.  The same scene at
    four scales: at 1e-15 every discriminant of a camera ray is far below 2^-96 (all waves on the IEEE routines), at 2e-8 they
    straddle it (mixed waves), at 1 and 1e7 none is."""
    rng = np.random.default_rng(5)
    n = 40
    sp = np.zeros((n, 9), np.float32)
    c = rng.uniform(-3.0, 3.0, (n, 3)) * (1.0, 0.15, 1.0)
    c[0] = (0.0, -200.3, 0.0)
    rad = rng.uniform(0.1, 0.45, n)
    rad[0] = 200.0
    sp[:, 0:3] = sp[:, 3:6] = c * scale
    sp[:, 7] = 1.0
    sp[:, 8] = rad * scale
    kinds = rng.integers(0, 3, n)
    kinds[0] = 0
    mt = np.zeros((n, 6), np.float32)
    mt[:, 0] = kinds
    mt[:, 1:4] = rng.uniform(0.3, 0.95, (n, 3))
    mt[:, 4] = np.where(kinds == 1, rng.uniform(0.0, 0.4, n), 0.0)
    mt[:, 5] = np.where(kinds == 2, 1.5, 0.0)
    mid = np.arange(n, dtype=np.uint32)
    eye = np.array((4.0, 1.2, 5.0)) * scale
    cam = rrt_amd.camera_from(tuple(eye), (0.0, 0.0, 0.0), (0, 1, 0), 40.0, 1.5, 0.05 * scale, 6.0 * scale, 0.0, 0.0)
    osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    osc.build_bvh(0)
    _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    gpu_ctx.set_camera(cam)
    nx, ny, ns = 96, 64, 6
    ref = osc.render(nx, ny, ns, seed=17, threads=8)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=17, count_work=True))
    assert_same_image(gpu, ref)
    assert ref[2]["hits_lambertian"] + ref[2]["hits_metal"] + ref[2]["hits_dielectric"] > nx * ny * ns // 2, ref[2]


@pytest.mark.parametrize("max_depth", [50, 300])   # 300: beyond what the narrow per-slot state holds, so the wide layout runs on a small scene
def test_long_paths_between_two_walls(gpu_ctx, max_depth):
    """A narrow gap between two huge diffuse/metal spheres: paths scatter dozens of times before they see the sky, so the
    back-to-front product of attenuations (lib.rs:62) runs over long trails, and some paths reach the depth limit."""
    sp = np.zeros((3, 9), np.float32)
    sp[0, 0:3] = sp[0, 3:6] = (-1000.06, 0.0, 0.0)
    sp[1, 0:3] = sp[1, 3:6] = (1000.06, 0.0, 0.0)
    sp[2, 0:3] = sp[2, 3:6] = (0.0, 0.0, -3.0)
    sp[:, 7] = 1.0
    sp[:, 8] = (1000.0, 1000.0, 0.05)
    mt = np.array([[0, 0.9, 0.8, 0.7, 0, 0], [1, 0.95, 0.9, 0.85, 0.05, 0], [2, 1, 1, 1, 0, 1.5]], np.float32)
    mid = np.arange(3, dtype=np.uint32)
    cam = rrt_amd.camera_from((0, 0.5, 2), (0, 0, -3), (0, 1, 0), 30.0, 1.0, 0.0, 1.0, 0.0, 0.0)
    osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    osc.build_bvh(0)
    _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    gpu_ctx.set_camera(cam)
    nx, ny, ns = 48, 48, 8
    ref = osc.render(nx, ny, ns, seed=8, max_depth=max_depth, threads=8)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=8, max_depth=max_depth, count_work=True))
    assert_same_image(gpu, ref)
    assert ref[2]["rays"] / ref[2]["paths"] > 8 and (ref[2]["depth_cut"] > 0 or max_depth > 50) and ref[2]["sky_misses"] > 100
    assert gpu[2]["depth_cut"] == ref[2]["depth_cut"]


def test_tracescene_mirror(gpu_ctx):
    """The reference's call sequence end to end: chap12_scene -> init_pool_with_rng -> tracescene (benches/benchmark.rs:14-23)."""
    nx, ny, ns = 10, 10, 4
    hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
    px, pdc, stream_seed = hs.tracescene(ns, device=0)
    assert pdc == nx * ny
    osc = OracleScene.build(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
    rgb_o, _, _ = osc.render(nx, ny, ns, seed=stream_seed)
    assert np.array_equal(px, rgb_o)


def test_progress_counter_rises_while_rendering(gpu_ctx):
    """`progress` plays the part of `pdc` (lib.rs:81,109; polled by main.rs:78-94): zero at the start, raised while the call
    runs, nx*ny at the end.  ctypes releases the GIL for the call, so this thread can poll the word."""
    import ctypes as C
    import threading
    import time
    nx, ny, ns = 1200, 800, 160
    hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
    gpu_ctx.upload_host_scene(hs)
    p = rrt_amd.Context.params(nx, ny, ns, seed=1, batch_samples=8)
    rgb = np.zeros((ny, nx, 3), np.uint8)
    done = C.c_uint64(12345)
    rc = []
    t = threading.Thread(target=lambda: rc.append(rrt_amd.lib().rrt_render(gpu_ctx._h, C.byref(p), rgb.ctypes.data, None, None, C.byref(done))))
    t.start()
    seen = []
    while t.is_alive():
        seen.append(done.value)
        time.sleep(0.0005)
    t.join()
    assert rc == [0] and done.value == nx * ny
    seen = [v for v in seen if v != 12345]
    assert all(a <= b for a, b in zip(seen, seen[1:])), "the counter never goes back"
    assert len({v for v in seen if 0 < v < nx * ny}) >= 3, f"no intermediate values seen: {sorted(set(seen))[:10]}"


def test_errors(gpu_ctx):
    fresh = rrt_amd.Context(0)
    with pytest.raises(rrt_amd.RrtError, match="before"):
        fresh.render(rrt_amd.Context.params(4, 4, 1))
    fresh.close()
    osc = OracleScene.build(SCENES["chap11"], 8, 8)
    upload_from_oracle(gpu_ctx, osc)
    with pytest.raises(rrt_amd.RrtError):
        gpu_ctx.render(rrt_amd.Context.params(0, 4, 1))
    with pytest.raises(rrt_amd.RrtError):
        gpu_ctx.render(rrt_amd.Context.params(4, 4, 1, shard_mode=rrt_amd.SHARD_TILE, rank=2, world=2))


# ---------------------------------------------------------------- full-size configurations
def assert_matches_fullsize_digest(name, rgb, acc, st):
    """Every pixel of a whole render against the digests of the oracle's render of the same configuration
    (tests/golden/fullsize.json, written by tests/golden/make_fullsize.py): CRC-32 of every row of f32 sums, SHA-256 of the
    sums and of the RGB8 bytes, and all ten work counters that the render reports without the counting build."""
    import hashlib
    import json
    import zlib
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize.json")) as f:
        g = json.load(f)[name]
    nx, ny = g["nx"], g["ny"]
    acc = np.ascontiguousarray(acc.reshape(ny, nx, 3), np.float32)
    rows = [zlib.crc32(acc[y].tobytes()) for y in range(ny)]
    bad = [y for y in range(ny) if rows[y] != g["row_crc32"][y]]
    assert not bad, f"{len(bad)} of {ny} rows of f32 sums differ, first: {bad[:8]}"
    assert hashlib.sha256(acc.tobytes()).hexdigest() == g["sha256_acc"]
    assert hashlib.sha256(np.ascontiguousarray(rgb.reshape(ny, nx, 3), np.uint8).tobytes()).hexdigest() == g["sha256_rgb"]
    assert st["rays"] == g["counters"]["rays"] and st["paths"] == nx * ny * g["ns"]
    return g


def spot_check(gpu_ctx, kind, nx, ny, ns, seed, n_pixels, scene_seed=0, half=11):
    osc = OracleScene.build(kind, nx, ny, seed=scene_seed, half=half)
    upload_from_oracle(gpu_ctx, osc)
    rgb, acc, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=seed))
    idx = np.random.default_rng(0).choice(nx * ny, n_pixels, replace=False).astype(np.uint32)
    acc_o, rgb_o, _ = osc.render_pixels(nx, ny, ns, seed, idx)
    assert np.array_equal(bits(acc.reshape(-1, 3)[idx]), bits(acc_o))
    assert np.array_equal(rgb.reshape(-1, 3)[idx], rgb_o)
    assert st["paths"] == nx * ny * ns
    assert rgb.mean() > 20  # an image, not zeros
    return rgb, acc, st


def test_full_chap11_400x225x100(gpu_ctx):
    nx, ny, ns = 400, 225, 100
    osc = OracleScene.build(SCENES["chap11"], nx, ny)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=0, threads=16)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=0))
    assert_same_image(gpu, ref)


def test_full_chap12_1200x800x500_spot(gpu_ctx):
    rgb, acc, st = spot_check(gpu_ctx, SCENES["chap12"], 1200, 800, 500, seed=0, n_pixels=96)
    # a second schedule of the same render gives the same image
    rgb2, acc2, st2 = gpu_ctx.render(rrt_amd.Context.params(1200, 800, 500, seed=0, path_slots=1 << 19, batch_samples=50))
    assert np.array_equal(rgb, rgb2) and np.array_equal(bits(acc), bits(acc2)) and st["rays"] == st2["rays"]


def test_full_book2chap2_1200x800_spot(gpu_ctx):
    spot_check(gpu_ctx, SCENES["book2chap2"], 1200, 800, 500, seed=0, n_pixels=64)


# ---- the target configuration: chap12 at 3840x2160, 1024 samples per pixel, sharded by tile and by sample
CFG4 = (3840, 2160, 1024)


@pytest.fixture(scope="module")
def cfg4_unsharded(gpu_ctx):
    nx, ny, ns = CFG4
    osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    rgb, acc, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=0))
    idx = np.random.default_rng(4).choice(nx * ny, 96, replace=False).astype(np.uint32)
    return osc, rgb, acc, st, idx


def test_cfg4_4k_1024spp_unsharded_spot(gpu_ctx, cfg4_unsharded):
    nx, ny, ns = CFG4
    osc, rgb, acc, st, idx = cfg4_unsharded
    acc_o, rgb_o, rays_o = osc.render_pixels(nx, ny, ns, 0, idx)
    assert np.array_equal(bits(acc.reshape(-1, 3)[idx]), bits(acc_o))
    assert np.array_equal(rgb.reshape(-1, 3)[idx], rgb_o)
    assert st["paths"] == nx * ny * ns
    assert rgb.mean() > 20


def test_cfg4_4k_1024spp_whole_image_digest(gpu_ctx, cfg4_unsharded):
    """All 8.3 M pixels of the target configuration against the oracle's render (2.4e10 rays, rendered once on the CPU)."""
    osc, rgb, acc, st, idx = cfg4_unsharded
    g = assert_matches_fullsize_digest("chap12_3840x2160x1024", rgb, acc, st)
    assert (g["nx"], g["ny"], g["ns"]) == CFG4


def test_cfg4_4k_1024spp_as_8_tile_shards(gpu_ctx, cfg4_unsharded):
    """Eight TILE shards rendered one after the other on this device and combined the way the ranks of a node combine them
    (zero-filled full-size buffers added up): the composite is the unsharded image bit for bit, and equals the oracle."""
    nx, ny, ns = CFG4
    osc, rgb_full, acc_full, st_full, idx = cfg4_unsharded
    upload_from_oracle(gpu_ctx, osc)
    world = 8
    rgb = np.zeros_like(rgb_full)
    acc = np.zeros_like(acc_full)
    rays = paths = 0
    for rank in range(world):
        r, a, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=0, shard_mode=rrt_amd.SHARD_TILE, rank=rank, world=world))
        assert st["progress"] == rrt_amd.shard_pixels(nx, ny, rrt_amd.SHARD_TILE, rank, world).size
        rgb += r
        acc += a
        rays += st["rays"]
        paths += st["paths"]
    assert paths == nx * ny * ns and rays == st_full["rays"]
    assert np.array_equal(rgb, rgb_full)
    assert np.array_equal(bits(acc), bits(acc_full))
    acc_o, rgb_o, _ = osc.render_pixels(nx, ny, ns, 0, idx)
    assert np.array_equal(bits(acc.reshape(-1, 3)[idx]), bits(acc_o)) and np.array_equal(rgb.reshape(-1, 3)[idx], rgb_o)


def test_cfg4_4k_1024spp_as_4_sample_shards(gpu_ctx, cfg4_unsharded):
    """Four SAMPLE shards: each rank's partial sums equal the oracle's sums over the same samples bit for bit; their total
    (another order of f32 additions than one render) stays within rounding of the unsharded sums, and the quantised image
    within one level."""
    nx, ny, ns = CFG4
    osc, rgb_full, acc_full, st_full, idx = cfg4_unsharded
    upload_from_oracle(gpu_ctx, osc)
    world = 4
    total = np.zeros(acc_full.shape, np.float32)
    rays = paths = 0
    for rank in range(world):
        r, a, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=0, shard_mode=rrt_amd.SHARD_SAMPLE, rank=rank, world=world))
        acc_o, rgb_o, _ = osc.render_pixels(nx, ny, ns, 0, idx, k_begin=rank, k_step=world)
        assert np.array_equal(bits(a.reshape(-1, 3)[idx]), bits(acc_o))
        assert np.array_equal(r.reshape(-1, 3)[idx], rgb_o)      # the shard's own image: its sums over its 256 samples
        total += a                                                # rank order, as a reduce onto rank 0 adds them
        rays += st["rays"]
        paths += st["paths"]
    assert paths == nx * ny * ns and rays == st_full["rays"]
    # tolerance: both orders add the same ns non-negative terms; each f32 addition errs by at most 2^-24 of the running sum,
    # so either total is within ns * 2^-24 of the exact sum, relatively, and the two are within twice that of each other
    assert np.allclose(total, acc_full, rtol=2 * ns * 2.0 ** -24, atol=0.0)
    rel = np.abs(total - acc_full) / np.maximum(acc_full, 1e-30)
    print(f"sample shards vs one render: max relative difference of the f32 sums {rel.max():.3e}, rms {np.sqrt((rel ** 2).mean()):.3e}")
    from rrt_amd import multi
    rgb = multi.finalize_numpy(total, ns)
    diff = np.abs(rgb.astype(np.int16) - rgb_full.astype(np.int16))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3


def test_large_scene_walks_the_hbm_bvh(gpu_ctx):
    """14 k spheres do not fit the LDS copy: the 64-byte linked nodes are walked in HBM.  Host-built scene on the GPU,
    oracle-built scene on the CPU (the builders agree bit for bit, tests/test_host_mirror.py)."""
    nx, ny, ns, half = 96, 54, 4, 60
    hs = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=0, half=half)
    assert hs.n_nodes * 64 > 64 * 1024
    gpu_ctx.upload_host_scene(hs)
    osc = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=0, half=half)
    ref = osc.render(nx, ny, ns, seed=6, threads=8)
    # both walks of a big scene: two children per record (the default) and node by node along the exit links (what the
    # entry point that allocates nothing runs); with and without the work counters (different kernels)
    for exit_links in (False, True):
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=6, count_work=True, exit_link_walk=exit_links))
        assert_same_image(gpu, ref)
        assert gpu[2]["inner_visits"] == ref[2]["inner_visits"] and gpu[2]["leaf_tests"] == ref[2]["leaf_tests"]
        assert_same_image(gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=6, exit_link_walk=exit_links)), ref)
    # Bvh::hit alone on the same scene: the walk starts in the copy of the top of the tree and continues in the node array
    idx = np.arange(nx * ny, dtype=np.uint32)
    rays, _ = osc.gen_primary(nx, ny, 11, idx, np.zeros(nx * ny, np.uint32))
    t_o, obj_o, _, _, _ = osc.hit_rays(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
    t_g, obj_g = gpu_ctx.intersect(rays[:, 0:3], rays[:, 3:6], rays[:, 6])
    assert np.array_equal(obj_g, obj_o) and np.array_equal(bits(t_g), bits(t_o))
    assert (obj_o != 0xFFFFFFFF).sum() > 100


@pytest.mark.parametrize("uniform_times", [True, False])
def test_large_scene_with_moving_spheres(gpu_ctx, uniform_times):
    """14 k spheres, half of them moving (one shutter interval for all, or per-sphere intervals): both walks of a big scene
    in their moving-sphere variants, against the oracle."""
    nx, ny, ns, half = 80, 48, 4, 60
    base = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=0, half=half)
    sp, mid, mt, _, cam21 = scene_arrays_from_oracle(base)
    rng = np.random.default_rng(5)
    sp = sp.copy()
    n = sp.shape[0]
    move = rng.uniform(size=n) < 0.5
    move[0] = False  # the ground
    sp[move, 4] = sp[move, 1] + rng.uniform(0.0, 0.5, int(move.sum())).astype(np.float32)
    if not uniform_times:
        sp[move, 6] = rng.choice([0.0, 0.25], int(move.sum())).astype(np.float32)
        sp[move, 7] = rng.choice([1.0, 0.75, 2.0], int(move.sum())).astype(np.float32)
    cam = rrt_amd.Camera.from_buffer_copy(cam21.tobytes())
    cam.time_low, cam.time_scale = 0.0, float(np.float32(1.0) / np.float32(1.0 - 2.0 ** -23))
    osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    osc.build_bvh(0)
    _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    assert nd.shape[0] * 64 > 64 * 1024
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    gpu_ctx.set_camera(cam)
    ref = osc.render(nx, ny, ns, seed=9, threads=8)
    for exit_links in (False, True):
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=9, count_work=True, exit_link_walk=exit_links))
        assert_same_image(gpu, ref)
        assert gpu[2]["inner_visits"] == ref[2]["inner_visits"] and gpu[2]["leaf_tests"] == ref[2]["leaf_tests"]
        assert_same_image(gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=9, exit_link_walk=exit_links)), ref)


def test_deep_tree_falls_back_to_the_exit_link_walk(gpu_ctx):
    """A tree about 100 levels deep (more than the two-children walk's stack holds) in a scene too big for LDS."""
    from tests.test_host_mirror import deep_tree_scene
    sp, mid, mt, cam = deep_tree_scene()
    osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    osc.build_bvh(0)
    _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    assert rrt_amd.debug_layout(sp, mid, mt, nd)[2] & 16 == 0
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    gpu_ctx.set_camera(cam)
    # camera rays only (depth limit 0: every hit is black): a scattered ray could run down the whole row of spheres, and
    # then the reference's 64-entry stack (acceleration.rs:233) overflows and the reference panics (as does the oracle)
    nx, ny, ns = 64, 48, 4
    ref = osc.render(nx, ny, ns, seed=2, threads=8, max_depth=0)
    gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=2, max_depth=0, count_work=True))
    assert_same_image(gpu, ref)
    assert gpu[2]["inner_visits"] == ref[2]["inner_visits"] and gpu[2]["leaf_tests"] == ref[2]["leaf_tests"]
    assert ref[2]["depth_cut"] > 200 and ref[2]["sky_misses"] > 200


def test_million_sphere_scene_spot(gpu_ctx):
    """The big-scene configuration: 1000x1000 cells (~1 M spheres, ~2 M nodes), 1920x1080 at 256 spp; spot-checked pixels."""
    nx, ny, ns, half = 1920, 1080, 256, 500
    hs = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=0, half=half)
    assert hs.n_spheres > 990000
    gpu_ctx.upload_host_scene(hs)
    rgb, acc, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=0))
    sp, mt, nd, cam = hs.export()
    osc = OracleScene.from_arrays(sp, np.arange(sp.shape[0], dtype=np.uint32), mt, nd, cam.as_array())
    idx = np.random.default_rng(1).choice(nx * ny, 48, replace=False).astype(np.uint32)
    acc_o, rgb_o, _ = osc.render_pixels(nx, ny, ns, 0, idx)
    assert np.array_equal(bits(acc.reshape(-1, 3)[idx]), bits(acc_o))
    assert np.array_equal(rgb.reshape(-1, 3)[idx], rgb_o)
    assert st["paths"] == nx * ny * ns
    # and every pixel against the digests of the oracle's render of the whole image (4.4e9 rays)
    g = assert_matches_fullsize_digest("spheres1m_1920x1080x256", rgb, acc, st)
    assert (g["nx"], g["ny"], g["ns"], g["half"]) == (nx, ny, ns, half)


def test_cli_renders_the_reference_default_scene(gpu_ctx, tmp_path):
    """rrt -r WxH -s N -e SEED (main.rs:12-108): same generator -> scene -> pool -> tracescene sequence as the library path."""
    import subprocess
    nx, ny, ns, seed = 64, 40, 4, 5
    out = tmp_path / "o.ppm"
    exe = os.path.join(os.path.dirname(os.path.abspath(rrt_amd.__file__)), "rrt")
    r = subprocess.run([exe, "-r", f"{nx}x{ny}", "-s", str(ns), "-e", str(seed), "-o", str(out)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    assert f"Rendering {nx} x {ny} image using {ns} samples per pixel." in r.stderr
    raw = out.read_bytes()
    header = f"P6\n{nx} {ny}\n255\n".encode()
    assert raw.startswith(header) and len(raw) == len(header) + nx * ny * 3
    hs = rrt_amd.HostScene(rrt_amd.SCENE_BOOK2CHAP2, nx, ny, seed=seed)
    px, _, _ = hs.tracescene(ns, device=0)
    assert np.array_equal(np.frombuffer(raw[len(header):], np.uint8).reshape(ny, nx, 3), px)


@pytest.mark.parametrize("name", ["chap12_1200x800x500", "book2chap2_1200x800x500"])
def test_full_size_image_digests(gpu_ctx, name):
    """Whole images at the headline sizes against digests of the oracle's renders (tests/golden/make_fullsize.py):
    every pixel of every row, f32 sums and RGB8, and the ray count."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize.json")) as f:
        g = json.load(f)[name]
    nx, ny, ns = g["nx"], g["ny"], g["ns"]
    hs = rrt_amd.HostScene(g["kind"], nx, ny, seed=g["scene_seed"])
    gpu_ctx.upload_host_scene(hs)
    rgb, acc, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=g["seed"], max_depth=g["max_depth"]))
    assert_matches_fullsize_digest(name, rgb, acc, st)


@pytest.mark.parametrize("method", [0, 1])   # BvhPartitionMethod::Middle, ::EqualCount (acceleration.rs:98-103)
@pytest.mark.parametrize("kind,half", [(rrt_amd.SCENE_CHAP11, 0), (rrt_amd.SCENE_CHAP12, 0), (rrt_amd.SCENE_BOOK2CHAP2, 0),
                                       (rrt_amd.SCENE_RANDOM_SPHERES, 60), (rrt_amd.SCENE_RANDOM_SPHERES, 500)])
def test_gpu_bvh_build_is_the_reference_tree(gpu_ctx, kind, half, method):
    """rrt_build_bvh_method (level-synchronous, on the device) against the oracle's recursive Bvh::build_r restatement: every
    node, every bit, for both partition methods.  chap11 takes the equal-count fallback of a Middle build (two concentric
    spheres), book2chap2 has moving boxes."""
    osc = OracleScene.build(kind, 64, 64, seed=0, half=max(half, 11))
    if method:
        osc.build_bvh(method)
    sp, mt, nd, _ = osc.export()
    got = gpu_ctx.build_bvh(sp, method)
    assert got.shape == nd.shape
    assert np.array_equal(got, nd), f"{int((got != nd).any(axis=1).sum())} of {nd.shape[0]} nodes differ"
    if method and half <= 60:   # an equal-count tree renders like any other
        gpu_ctx.upload_scene(sp, np.arange(sp.shape[0], dtype=np.uint32), mt, got)
        gpu_ctx.set_camera(rrt_amd.Camera.from_buffer_copy(scene_arrays_from_oracle(osc)[4].tobytes()))
        assert_same_image(gpu_ctx.render(rrt_amd.Context.params(48, 32, 3, seed=2)), osc.render(48, 32, 3, seed=2, threads=8))


def test_gpu_bvh_build_degenerate_inputs(gpu_ctx):
    """One object, two objects at one place, many identical centroids (equal-count all the way down), negative radius."""
    rng = np.random.default_rng(3)
    cases = []
    one = np.zeros((1, 9), np.float32); one[0, 8] = 1.0
    cases.append(one)
    two = np.zeros((2, 9), np.float32); two[:, 8] = [0.5, -0.45]; two[:, 7] = 1.0
    cases.append(two)
    same = np.zeros((37, 9), np.float32); same[:, 0:3] = [1.0, 2.0, 3.0]; same[:, 3:6] = [1.0, 2.0, 3.0]; same[:, 7] = 1.0
    same[:, 8] = rng.uniform(0.1, 1.0, 37)
    cases.append(same)
    mixed = np.zeros((300, 9), np.float32)
    mixed[:, 0:3] = rng.integers(-3, 4, (300, 3))       # many repeated centres: fallbacks at several depths
    mixed[:, 3:6] = mixed[:, 0:3]
    mixed[:, 7] = 1.0
    mixed[:, 8] = 0.25
    cases.append(mixed)
    from oracle import OracleScene as O
    for sp in cases:
        n = sp.shape[0]
        mt = np.zeros((n, 6), np.float32); mt[:, 1:4] = 0.5
        osc = O.from_arrays(sp, np.arange(n, dtype=np.uint32), mt, np.zeros((1, 8), np.uint32))
        for method in (0, 1):
            osc.build_bvh(method)
            _, _, nd, _ = osc.export()
            got = gpu_ctx.build_bvh(sp, method)
            assert got.shape == nd.shape and np.array_equal(got, nd), f"n={n} method {method}: {int((got != nd).any(axis=1).sum())} nodes differ"


def test_device_buffer_entry_points(gpu_ctx):
    """rrt_render_device (image left in caller-owned device memory) and rrt_finalize (quantise summed sample shards on the
    device), as bench.py and the multi-GPU combine use them."""
    import torch
    from rrt_amd import multi
    nx, ny, ns = 150, 90, 10
    osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    ref = osc.render(nx, ny, ns, seed=8, threads=8)
    dev = torch.device("cuda", 0)
    rgb = torch.zeros(ny * nx * 3, dtype=torch.uint8, device=dev)
    acc = torch.zeros(ny * nx * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    st = gpu_ctx.render_device(rrt_amd.Context.params(nx, ny, ns, seed=8), rgb.data_ptr(), acc.data_ptr())
    assert_same_image((rgb.cpu().numpy().reshape(ny, nx, 3), acc.cpu().numpy().reshape(ny, nx, 3), st), ref)
    # two sample shards, summed on the device, quantised by rrt_finalize
    parts = []
    for rank in range(2):
        a = torch.zeros_like(acc)
        torch.cuda.synchronize()
        gpu_ctx.render_device(rrt_amd.Context.params(nx, ny, ns, seed=8, shard_mode=rrt_amd.SHARD_SAMPLE, rank=rank, world=2), 0, a.data_ptr())
        parts.append(a)
    total = parts[0] + parts[1]
    out = torch.zeros_like(rgb)
    torch.cuda.synchronize()
    gpu_ctx.finalize(total.data_ptr(), nx * ny, ns, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, multi.finalize_numpy(total.cpu().numpy(), ns))   # lib.rs:105-108, bit for bit
    diff = np.abs(got.astype(int) - ref[0].reshape(-1).astype(int))             # another order of f32 additions than one render
    assert diff.max() <= 1 and (diff != 0).mean() < 0.01


def test_render_multi_one_call_whole_image(gpu_ctx, monkeypatch):
    """rrt_render_multi (lib.rs:73-81: one call, the whole image back) over 1, 2 and 3 contexts.  This machine has one GPU, so
    the contexts share it and the shares travel by plain copies (RRT_MULTI_LOOPBACK); what is under test is the sharding,
    the packed tile shares, their placement, the sample-shard sum and the summed counters.  With one context and
    RRT_MULTI_FORCE_RCCL the share goes through real RCCL calls (a one-rank communicator: send/recv to itself, reduce)."""
    nx, ny, ns = 150, 70, 10   # not a multiple of the 32x8 tile
    osc = OracleScene.build(SCENES["book2chap2"], nx, ny, seed=0)
    ref = osc.render(nx, ny, ns, seed=3, threads=8)
    ctxs = [gpu_ctx, rrt_amd.Context(0), rrt_amd.Context(0)]
    try:
        for c in ctxs:
            upload_from_oracle(c, osc)
        # one context: the plain render
        got = rrt_amd.render_multi(ctxs[:1], rrt_amd.Context.params(nx, ny, ns, seed=3))
        assert_same_image(got, ref)
        assert got[2]["progress"] == nx * ny
        monkeypatch.setenv("RRT_MULTI_LOOPBACK", "1")
        for n in (2, 3):
            got = rrt_amd.render_multi(ctxs[:n], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_TILE, count_work=True))
            assert_same_image(got, ref)                      # tile shards: bit-identical, f32 sums included
            for key in ("inner_visits", "leaf_tests", "sky_misses"):
                assert got[2][key] == ref[2][key], key       # counters summed over the ranks
            rgb, acc, st = rrt_amd.render_multi(ctxs[:n], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_SAMPLE))
            assert st["rays"] == ref[2]["rays"] and st["paths"] == nx * ny * ns
            assert np.allclose(acc, ref[1], rtol=2 * ns * 2.0 ** -24, atol=0.0)   # another order of the same ns additions
            diff = np.abs(rgb.astype(int) - ref[0].astype(int))
            assert diff.max() <= 1 and (diff != 0).mean() < 0.01
        rgb_only = rrt_amd.render_multi(ctxs[:2], rrt_amd.Context.params(nx, ny, ns, seed=3), want_accum=False)   # NONE = tiles
        assert np.array_equal(rgb_only[0], ref[0])
        monkeypatch.delenv("RRT_MULTI_LOOPBACK")
        monkeypatch.setenv("RRT_MULTI_FORCE_RCCL", "1")
        got = rrt_amd.render_multi(ctxs[:1], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_TILE))
        assert_same_image(got, ref)
        got = rrt_amd.render_multi(ctxs[:1], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_SAMPLE))
        assert_same_image(got, ref)                          # one rank: the reduce changes nothing
        monkeypatch.delenv("RRT_MULTI_FORCE_RCCL")
        with pytest.raises(rrt_amd.RrtError, match="one GPU"):
            rrt_amd.render_multi(ctxs[:2], rrt_amd.Context.params(nx, ny, ns, seed=3))   # RCCL cannot put two ranks on one GPU
    finally:
        for c in ctxs[1:]:
            c.close()


def test_render_multi_returns_when_one_rank_fails(gpu_ctx, monkeypatch):
    """A rank that cannot render (here: its context has no scene) must make the call return its error; the other ranks must not
    be left waiting for its side of the exchange.  Every rank is joined before any of them enters the exchange."""
    nx, ny, ns = 96, 64, 4
    osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
    upload_from_oracle(gpu_ctx, osc)
    bare = rrt_amd.Context(0)   # no scene, no camera
    try:
        monkeypatch.setenv("RRT_MULTI_LOOPBACK", "1")
        for mode in (rrt_amd.SHARD_TILE, rrt_amd.SHARD_SAMPLE):
            for order in ([gpu_ctx, bare], [bare, gpu_ctx]):
                with pytest.raises(rrt_amd.RrtError, match="rank failed") as e:
                    rrt_amd.render_multi(order, rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=mode))
                assert e.value.code == rrt_amd.ERR_NOT_READY
        # the good context still renders afterwards, alone and with a good peer
        upload_from_oracle(bare, osc)
        ref = osc.render(nx, ny, ns, seed=3, threads=8)
        assert_same_image(rrt_amd.render_multi([gpu_ctx, bare], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_TILE)), ref)
    finally:
        bare.close()


def test_more_sample_shards_than_samples(gpu_ctx, monkeypatch):
    """world > ns under SAMPLE sharding: the ranks past the last sample render nothing (zero sums, zero counters, a zero
    image of their own) and the combined image is the whole render's."""
    nx, ny, ns = 64, 40, 2
    osc = OracleScene.build(SCENES["chap11"], nx, ny)
    ref = osc.render(nx, ny, ns, seed=5, threads=8)
    ctxs = [gpu_ctx, rrt_amd.Context(0), rrt_amd.Context(0)]
    try:
        for c in ctxs:
            upload_from_oracle(c, osc)
        rgb, acc, st = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=5, shard_mode=rrt_amd.SHARD_SAMPLE, rank=2, world=3))
        assert not rgb.any() and not acc.any() and st["rays"] == 0 and st["paths"] == 0
        monkeypatch.setenv("RRT_MULTI_LOOPBACK", "1")
        rgb, acc, st = rrt_amd.render_multi(ctxs, rrt_amd.Context.params(nx, ny, ns, seed=5, shard_mode=rrt_amd.SHARD_SAMPLE))
        assert st["rays"] == ref[2]["rays"] and st["paths"] == nx * ny * ns
        # two ranks with one sample each and one with none: 0 + a + b in rank order is the one-GPU sum a + b bit for bit
        assert np.array_equal(bits(acc), bits(ref[1])) and np.array_equal(rgb, ref[0])
    finally:
        for c in ctxs[1:]:
            c.close()


def test_render_multi_on_two_gpus_over_rccl(gpu_ctx):
    """The same call on two real GPUs, shares moved by RCCL.  Skipped where the machine has one GPU."""
    try:
        second = rrt_amd.Context(1)
    except rrt_amd.RrtError:
        pytest.skip("one GPU here")
    try:
        nx, ny, ns = 150, 70, 10
        osc = OracleScene.build(SCENES["chap12"], nx, ny, seed=0)
        ref = osc.render(nx, ny, ns, seed=3, threads=8)
        for c in (gpu_ctx, second):
            upload_from_oracle(c, osc)
        got = rrt_amd.render_multi([gpu_ctx, second], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_TILE))
        assert_same_image(got, ref)
        rgb, acc, st = rrt_amd.render_multi([gpu_ctx, second], rrt_amd.Context.params(nx, ny, ns, seed=3, shard_mode=rrt_amd.SHARD_SAMPLE))
        assert st["rays"] == ref[2]["rays"] and np.allclose(acc, ref[1], rtol=2 * ns * 2.0 ** -24, atol=0.0)
    finally:
        second.close()


def test_two_process_render_over_nccl(gpu_ctx):
    """One process per GPU over torch.distributed's nccl backend (bench.py --gpus 2 on the target render at reduced samples):
    fresh child processes, tile shards gathered on rank 0.  Skipped where the machine has one GPU."""
    import json
    import subprocess
    import sys
    try:
        rrt_amd.Context(1).close()
    except rrt_amd.RrtError:
        pytest.skip("one GPU here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    for shard in ("tile", "sample"):
        with socket.socket() as sock:   # a port nobody listens on right now
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--spp", "16",
                            "--shard", shard, "--no-roofline"], capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out = json.loads(line)
        assert out["n_gpus"] == 2 and out["config"]["backend"] == "nccl" and out["config"]["ranks"] == 2
        assert out["config"]["rays_per_step"] > 3840 * 2160 * 16


def test_build_bvh_argument_errors(gpu_ctx):
    import ctypes as C
    sp = np.zeros((3, 9), np.float32); sp[:, 0] = [0, 1, 2]; sp[:, 3] = [0, 1, 2]; sp[:, 8] = 0.3
    cols = [np.ascontiguousarray(sp[:, i]) for i in range(9)]
    mid = np.zeros(3, np.uint32)
    s = rrt_amd.SphereSoa(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in cols], mid.ctypes.data_as(C.POINTER(C.c_uint32)), 3)
    nodes = np.zeros((5, 8), np.uint32)
    n = C.c_uint32(0)
    L = rrt_amd.lib()
    assert L.rrt_build_bvh(gpu_ctx._h, C.byref(s), nodes.ctypes.data, 4, C.byref(n)) < 0        # 2n-1 = 5 nodes do not fit 4
    assert b"2n-1" in L.rrt_last_error()
    assert L.rrt_build_bvh(gpu_ctx._h, C.byref(s), None, 5, C.byref(n)) < 0
    empty = rrt_amd.SphereSoa(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in cols], mid.ctypes.data_as(C.POINTER(C.c_uint32)), 0)
    assert L.rrt_build_bvh(gpu_ctx._h, C.byref(empty), nodes.ctypes.data, 5, C.byref(n)) < 0
    assert L.rrt_build_bvh(gpu_ctx._h, C.byref(s), nodes.ctypes.data, 5, C.byref(n)) == 0 and n.value == 5


def test_zero_direction_ray_inside_a_path_of_a_big_scene(gpu_ctx):
    """Found by tools/fuzz.py: in this scene the path of pixel 1066, sample 14 is caught inside a glass sphere and its 22nd ray
    has direction (0, 0, 0).  The reference accepts every sphere such a ray tests (NaN roots), so which sphere the path goes on
    from is a matter of walk order and of NaN handling in the box tests; here it is a metal sphere that absorbs the path.  All
    eighteen samples, work counters included (the image alone does not show it: the path is black either way)."""
    nx, ny, ns = 91, 25, 18
    osc = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=2, half=29)
    upload_from_oracle(gpu_ctx, osc)
    rays = osc.trace_path_rays(nx, ny, 859308763, 1066, 14, max_depth=120)
    assert len(rays) == 22 and not rays[21, 3:6].any()
    ref = osc.render(nx, ny, ns, seed=859308763, max_depth=120, threads=8)
    for walk in ({}, {"exit_link_walk": True}):
        gpu = gpu_ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=859308763, max_depth=120, count_work=True, **walk))
        assert_same_image(gpu, ref)
        for k in ("rays", "hits_metal", "hits_lambertian", "absorbed", "depth_cut", "inner_visits", "leaf_tests"):
            assert gpu[2][k] == ref[2][k], (walk, k, gpu[2][k], ref[2][k])


@pytest.mark.parametrize("scene", ["chap12", "book2chap2", "big"])
def test_paths_without_a_direction(gpu_ctx, scene):
    """A camera whose viewport has no extent, at the lens: every camera ray has direction (0, 0, 0).  Such a ray accepts every
    sphere it is tested against at t = NaN (sphere.rs:74-81: a NaN root is neither below t_min nor beyond t_max), the hit point
    is NaN, and what follows is a path of NaN rays, which pass every box (f32::max / f32::min drop the NaNs,
    acceleration.rs:77-82) and go on from the last sphere in walk order until a metal sphere absorbs them or the depth limit
    cuts them.  The reference meets such paths when refracted() returns no direction; here they are the whole image, from
    several places in the scene, through the walk in LDS and both walks in HBM."""
    if scene == "big":
        osc = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, 16, 8, seed=2, half=29)
    else:
        osc = OracleScene.build(SCENES[scene], 16, 8, seed=0)
    sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    kinds_met = set()
    walks = ({}, {"exit_link_walk": True}) if scene == "big" else ({}, {"force_hbm_bvh": True})
    for k, eye in enumerate([(0.3, 0.2, 0.4), (4.0, 1.0, 0.0), (-4.0, 0.9, 0.1), (0.0, 1.0, 0.0), (2.1, 0.2, -3.3), (-7.5, 0.21, 5.2), (0.0, -900.0, 0.0), (30.0, 40.0, 5.0)]):
        cam = rrt_amd.Camera()
        cam.lower_left_corner[:] = eye
        cam.origin[:] = eye
        cam.horizontal[:] = (0.0, 0.0, 0.0)
        cam.vertical[:] = (0.0, 0.0, 0.0)
        cam.u[:] = (1.0, 0.0, 0.0)
        cam.v[:] = (0.0, 1.0, 0.0)
        cam.lens_radius, cam.time_low, cam.time_scale = 0.0, 0.0, 1.0
        osc.set_camera21(cam.as_array())
        gpu_ctx.set_camera(cam)
        ref = osc.render(16, 8, 3, seed=5 + k, max_depth=20, threads=4)
        for walk in walks:
            gpu = gpu_ctx.render(rrt_amd.Context.params(16, 8, 3, seed=5 + k, max_depth=20, count_work=True, **walk))
            # a NaN sum (the sky seen along a NaN direction) is compared as NaN: the sign and payload of a NaN that an operation
            # makes (0 * inf, 0 / 0) are the processor's choice, x86 makes it negative, the GPU positive; RGB8 is 0 either way
            both_nan = np.isnan(gpu[1]) & np.isnan(ref[1])
            assert np.array_equal(np.isnan(gpu[1]), np.isnan(ref[1]))
            assert_same_image((gpu[0], np.where(both_nan, np.float32(0), gpu[1]), gpu[2]), (ref[0], np.where(both_nan, np.float32(0), ref[1]), ref[2]))
            for key in ("rays", "hits_metal", "hits_lambertian", "hits_dielectric", "absorbed", "depth_cut", "sky_misses", "inner_visits", "leaf_tests"):
                assert gpu[2][key] == ref[2][key], (eye, walk, key, gpu[2][key], ref[2][key])
        kinds_met |= {key for key in ("hits_metal", "hits_lambertian", "hits_dielectric", "sky_misses") if ref[2][key]}
    assert len(kinds_met) >= 2, kinds_met


@pytest.mark.parametrize("scene", ["big", "chap12"])   # walked in HBM (two-children walk, walk along exit links) / in LDS (and in HBM when forced)
@pytest.mark.parametrize("method", [0, 1])   # Middle: the giant sphere is a child of the root; EqualCount: it sits deep in the tree
def test_t_max_turns_nan_in_the_middle_of_a_walk(gpu_ctx, method, scene):
    """A ground sphere of radius 3e19: |o - c|^2 and radius^2 both overflow, their difference is NaN, so is the root, and a NaN
    root is accepted (sphere.rs:74-81).  Its box holds the whole scene, so a ray meets it at some point of its walk, before or
    after spheres that gave it a finite t_max: from then on t_max is NaN, f32::min drops it from the box tests
    (acceleration.rs:81), boxes that lay beyond the finite t_max are hit again, and every sphere tested after that with a root
    above t_min replaces the hit.  Which sphere a ray ends on is a matter of exact walk order: rays and a small render, through
    both HBM walks, work counters included."""
    base = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, 48, 32, seed=4, half=29) if scene == "big" else OracleScene.build(rrt_amd.SCENE_CHAP12, 48, 32, seed=4)
    sp, mid, mt, _, cam21 = scene_arrays_from_oracle(base)
    ground = int(np.argmax(sp[:, 8]))
    assert sp[ground, 8] == 1000.0
    sp = sp.copy()
    sp[ground, 8] = np.float32(3e19)
    sp[ground, 1] = sp[ground, 4] = np.float32(-3e19)
    cam = rrt_amd.Camera.from_buffer_copy(cam21.tobytes())
    osc = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    osc.build_bvh(method)
    _, _, _, nd, _ = scene_arrays_from_oracle(osc)
    gpu_ctx.upload_scene(sp, mid, mt, nd)
    gpu_ctx.set_camera(cam)
    rng = np.random.default_rng(9)
    n = 20000
    org = rng.uniform(-25, 25, (n, 3)).astype(np.float32) * np.float32(1.0 if scene == "big" else 0.4)
    org[:, 1] = rng.uniform(0.05, 2.0, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 1] *= np.float32(0.2)
    tm = np.zeros(n, np.float32)
    t_o, obj_o, _, _, _ = osc.hit_rays(org, d, tm)
    t_g, obj_g = gpu_ctx.intersect(org, d, tm)
    assert np.isnan(t_o).sum() > 1000 and (np.isfinite(t_o) & (t_o < 1e30)).sum() > 500   # both kinds of ending
    assert np.array_equal(np.isnan(t_g), np.isnan(t_o))
    assert np.array_equal(obj_g, obj_o)
    assert np.array_equal(bits(t_g)[~np.isnan(t_o)], bits(t_o)[~np.isnan(t_o)])
    ref = osc.render(48, 32, 4, seed=3, max_depth=12, threads=8)
    for walk in (({}, {"exit_link_walk": True}) if scene == "big" else ({}, {"force_hbm_bvh": True})):
        gpu = gpu_ctx.render(rrt_amd.Context.params(48, 32, 4, seed=3, max_depth=12, count_work=True, **walk))
        both_nan = np.isnan(gpu[1]) & np.isnan(ref[1])
        assert np.array_equal(np.isnan(gpu[1]), np.isnan(ref[1]))
        assert_same_image((gpu[0], np.where(both_nan, np.float32(0), gpu[1]), gpu[2]), (ref[0], np.where(both_nan, np.float32(0), ref[1]), ref[2]))
        for key in ("rays", "hits_metal", "hits_lambertian", "hits_dielectric", "absorbed", "depth_cut", "sky_misses", "inner_visits", "leaf_tests"):
            assert gpu[2][key] == ref[2][key], (walk, key, gpu[2][key], ref[2][key])
