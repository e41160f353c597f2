"""The CPU oracle against published known answers, analytic cases, its own brute force and the committed fixtures.

The reference has no tests, fixtures or golden images (parity with the reference itself is unpinned); what pins the
oracle is listed here: xoshiro128+/SplitMix64 published vectors, rand's float conversions, closed-form sphere/sky/Schlick
cases, brute-force closest_hit (lib.rs:32-47) vs the BVH walk, and regression fixtures written by tests/golden/make_golden.py."""
import ctypes as C
import glob
import os
import struct

import numpy as np
import pytest

from oracle import CTR_NAMES, FLT_MAX, MODE_PER_SAMPLE, MODE_REF_POOL, MODE_REF_SERIAL, OracleScene, camera24_to_abi
from oracle import lib as olib
from tests.conftest import ROOT


def f32bits(x):
    return struct.unpack("I", struct.pack("f", x))[0]


# ---------------------------------------------------------------- generator (rand_xoshiro 0.6.0, rand 0.8.5)
def test_xoshiro128plus_published_vector():
    st = np.array([1, 2, 3, 4], np.uint32)
    got = [olib().orc_rng_next_u32(st.ctypes.data) for _ in range(10)]
    assert got == [5, 12295, 25178119, 27286542, 39879690, 1140358681, 3276312097, 4110231701, 399823256, 2144435200]


def test_splitmix64_and_seed_from_u64():
    s = C.c_uint64(0)
    assert olib().orc_splitmix64(C.byref(s)) == 0xE220A8397B1DCDAF
    assert olib().orc_splitmix64(C.byref(s)) == 0x6E789E6AA1B965F4
    st = np.zeros(4, np.uint32)
    olib().orc_rng_seed(0, st.ctypes.data)
    assert [int(v) for v in st] == [0x7B1DCDAF, 0xE220A839, 0xA1B965F4, 0x6E789E6A]
    assert [olib().orc_rng_next_u32(st.ctypes.data) for _ in range(4)] == [3918949401, 3103299678, 3277025221, 2547427574]
    olib().orc_rng_seed(0, st.ctypes.data)
    got = [olib().orc_rng_gen_f32(st.ctypes.data) for _ in range(4)]
    assert got == [np.float32(v) for v in (0.91245151, 0.72254324, 0.76299185, 0.59311914)]
    # 1234567 is the test vector of the xoshiro authors' splitmix64.c
    s = C.c_uint64(1234567)
    assert [olib().orc_splitmix64(C.byref(s)) for _ in range(3)] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_jump_is_linear_and_moves_the_state():
    a = np.array([1, 2, 3, 4], np.uint32)
    b = np.array([9, 9, 9, 9], np.uint32)
    x = a ^ b
    for v in (a, b, x):
        olib().orc_rng_jump(v.ctypes.data)
    assert np.array_equal(a ^ b, x)  # the jump is a GF(2)-linear map of the state
    assert not np.array_equal(a, [1, 2, 3, 4])


def test_jump_is_two_to_the_64_steps():
    """rand_xoshiro's Xoshiro128Plus::jump is documented as equivalent to 2^64 calls of next_u32 (the per-thread streams of
    threadpool.rs:10-19 are made with it).  The generator's state update is linear over GF(2): the 128 x 128 matrix of one step, squared 64
    times, must map every basis state to what the oracle's jump makes of it.  This pins the four jump constants by
    arithmetic, not by recollection."""
    def step(state128):
        st = np.array([(state128 >> (32 * i)) & 0xFFFFFFFF for i in range(4)], np.uint32)
        olib().orc_rng_next_u32(st.ctypes.data)
        return sum(int(v) << (32 * i) for i, v in enumerate(st))

    def jump(state128):
        st = np.array([(state128 >> (32 * i)) & 0xFFFFFFFF for i in range(4)], np.uint32)
        olib().orc_rng_jump(st.ctypes.data)
        return sum(int(v) << (32 * i) for i, v in enumerate(st))

    cols = [step(1 << i) for i in range(128)]          # column i: image of basis state e_i (a 128-bit integer)

    def apply(m, x):                                   # m as columns; x a state
        y, i = 0, 0
        while x:
            if x & 1:
                y ^= m[i]
            x >>= 1
            i += 1
        return y

    # linearity of one step on a few random states (so that the matrix really is the step)
    rng = np.random.default_rng(5)
    for _ in range(20):
        x = int.from_bytes(rng.bytes(16), "little")
        assert apply(cols, x) == step(x)
    m = cols
    for _ in range(64):                                # m <- m * m, 64 times: the step applied 2^64 times
        m = [apply(m, c) for c in m]
    for i in range(128):
        assert m[i] == jump(1 << i), i
    x = int.from_bytes(rng.bytes(16), "little")
    assert apply(m, x) == jump(x)


def test_uniform_scales_of_rand():
    lo, sc = C.c_float(), C.c_float()
    for (a, b), bits in {(-1.0, 1.0): 0x40000001, (0.0, 1.0): 0x3F800001, (0.0, 0.5): 0x3F000001, (0.0, 0.0): 0}.items():
        olib().orc_uniform_new_inclusive(a, b, C.byref(lo), C.byref(sc))
        assert f32bits(sc.value) == bits and lo.value == a
    # every sample consumes one draw and stays inside [low, high]
    st = np.array([5, 6, 7, 8], np.uint32)
    ref = st.copy()
    olib().orc_uniform_new_inclusive(-1.0, 1.0, C.byref(lo), C.byref(sc))
    vals = [olib().orc_uniform_sample(lo.value, sc.value, st.ctypes.data) for _ in range(1000)]
    for _ in range(1000):
        olib().orc_rng_next_u32(ref.ctypes.data)
    assert np.array_equal(st, ref) and min(vals) >= -1.0 and max(vals) <= 1.0
    # the largest draw maps to exactly high
    assert np.float32(np.float32(1.0 - 2.0**-23) * np.float32(sc.value)) + np.float32(-1.0) == np.float32(1.0)


def test_fp_constants():
    assert f32bits(olib().orc_gamma_eb(3)) == 0x34400002     # fp.rs:16-19
    assert f32bits(olib().orc_far_scale()) == 0x3F800003     # 1 + 2*gamma(3), acceleration.rs:78


def test_rejection_samplers():
    st = np.array([11, 22, 33, 44], np.uint32)
    out = np.zeros(3, np.float32)
    for _ in range(2000):
        olib().orc_gen_in_unit_sphere(st.ctypes.data, out.ctypes.data)
        assert float(out @ out) < 1.0
    d = np.zeros(2, np.float32)
    for _ in range(2000):
        olib().orc_gen_in_unit_disc(st.ctypes.data, d.ctypes.data)
        assert float(d @ d) < 1.0


def test_saturating_u8_cast():
    cases = {float("nan"): 0, -5.0: 0, -0.0: 0, 0.99: 0, 1.0: 1, 254.999: 254, 255.0: 255, 255.99: 255, 1e9: 255, float("inf"): 255}
    for x, want in cases.items():
        assert olib().orc_f32_as_u8(x) == want


def test_schlick():
    r0 = ((1 - 1.5) / (1 + 1.5)) ** 2
    assert abs(olib().orc_reflectance(1.0, 1.5, 0) - r0) < 1e-7          # cos = 1 -> r0
    assert abs(olib().orc_reflectance(0.0, 1.5, 0) - 1.0) < 1e-7         # grazing -> 1
    # the double-precision product chain and libm's powf agree to the last bit almost everywhere
    rng = np.random.default_rng(0)
    cos = rng.uniform(0, 1, 200000).astype(np.float32)
    a = np.array([olib().orc_reflectance(float(c), 1.5, 0) for c in cos[:20000]], np.float32)
    b = np.array([olib().orc_reflectance(float(c), 1.5, 1) for c in cos[:20000]], np.float32)
    ulp = np.abs(a.view(np.int32) - b.view(np.int32))
    assert ulp.max() <= 1 and (ulp != 0).mean() < 0.02


# ---------------------------------------------------------------- geometry
def one_sphere_scene(center=(0, 0, -5), radius=1.0, kind=0):
    sp = np.array([[*center, *center, 0, 1, radius]], np.float32)
    mt = np.array([[kind, 0.5, 0.5, 0.5, 0, 1.5]], np.float32)
    osc = OracleScene.from_arrays(sp, np.zeros(1, np.uint32), mt, np.zeros((0, 8), np.uint32))
    osc.build_bvh(0)
    return osc


def test_sphere_hit_closed_form():
    osc = one_sphere_scene()
    org = np.array([[0, 0, 0], [0, 0, 0], [0, 0, -5], [0, 2, 0], [0, 0, -10]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1], [0, 0, 2]], np.float32)
    t, obj, p, n, front = osc.hit_rays(org, d)
    assert t[0] == 4.0 and obj[0] == 0 and np.allclose(p[0], [0, 0, -4]) and np.allclose(n[0], [0, 0, 1]) and front[0] == 1
    assert obj[1] == 0xFFFFFFFF                                  # pointing away
    assert t[2] == 1.0 and front[2] == 0 and np.allclose(n[2], [0, 0, 1])   # from the centre: back face, normal flipped
    assert obj[3] == 0xFFFFFFFF                                  # passes above
    assert t[4] == 2.0                                           # unnormalised direction: t scales


def test_negative_radius_flips_the_normal():
    osc = one_sphere_scene(radius=-1.0)
    t, obj, p, n, front = osc.hit_rays(np.zeros((1, 3), np.float32), np.array([[0, 0, -1]], np.float32))
    assert t[0] == 4.0 and front[0] == 0 and np.allclose(n[0], [0, 0, 1])   # outward normal points inward (chap11.rs:38-42)


def test_bvh_walk_equals_brute_force():
    for kind, seed in [(0, 0), (1, 0), (1, 9), (2, 4)]:
        osc = OracleScene.build(kind, 64, 64, seed=seed)
        rng = np.random.default_rng(seed + 1)
        n = 30000
        org = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
        org[:, 1] = rng.uniform(0.01, 4, n)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        d[:3000, 0] = 0
        d[3000:6000, 2] = -0.0
        tm = rng.uniform(0, 1, n).astype(np.float32)
        t1, o1, _, _, _ = osc.hit_rays(org, d, tm)
        t2, o2, _, _, _ = osc.hit_rays(org, d, tm, brute=True)
        assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
        same = o1 == o2
        # objects may differ only on an exact tie in t (the later visited wins, sphere.rs:80-83)
        assert same.mean() > 0.9999
        assert (o1 != 0xFFFFFFFF).mean() > 0.3


def test_chap11_equal_count_fallback_and_concentric_pair():
    osc = OracleScene.build(0, 400, 225)
    sp, mt, nd, cam = osc.export()
    assert sp.shape[0] == 5 and nd.shape[0] == 9
    leaves = nd[nd[:, 7] >> 31 == 1][:, 6]
    assert sorted(leaves.tolist()) == [0, 1, 2, 3, 4]
    # the two concentric glass spheres (identical centroids) end up as the two children of one inner node
    inner = nd[nd[:, 7] >> 31 == 0]
    pairs = [{int(nd[int(r[6]), 6]), int(nd[int(r[7] & 0x1FFFFFFF), 6])} for r in inner
             if nd[int(r[6]), 7] >> 31 and nd[int(r[7] & 0x1FFFFFFF), 7] >> 31]
    assert {3, 4} in pairs


def test_sky_and_primary_rays():
    osc = OracleScene.build(0, 8, 8)
    # a one-pixel look at the sky: straight up is (0.5, 0.7, 1.0), straight down hits the ground sphere
    cam = np.zeros(21, np.float32)
    cam[0:3] = (0, 51, 0)       # lower-left corner: direction = corner - origin = straight up
    cam[9:12] = (0, 50, 0)      # origin above everything
    cam2 = cam.copy()
    osc.set_camera21(cam2)
    rgb, acc, ctr = osc.render(1, 1, 4, seed=1)
    assert np.allclose(acc[0, 0] / 4, [0.5, 0.7, 1.0], atol=1e-6) and ctr["rays"] == 4 and ctr["sky_misses"] == 4
    assert rgb[0, 0].tolist() == [int(np.sqrt(0.5) * 255.99), int(np.sqrt(np.float32(0.7)) * 255.99), 255]


def test_gen_ray_through_the_image_centre():
    osc = OracleScene.build(1, 200, 100, seed=0)          # aperture 0.1: origin jitters on the lens, the focus plane does not
    rays, states = osc.gen_primary(200, 100, 3, np.array([50 * 200 + 100] * 64, np.uint32), np.arange(64, dtype=np.uint32))
    focus = rays[:, 0:3] + rays[:, 3:6]                   # direction is (point on the focus plane) - origin
    assert np.abs(focus - focus.mean(axis=0)).max() < 0.06      # one pixel of the focus plane (fov 20 at distance 10)
    assert np.abs(rays[:, 0:3] - [13, 2, 3]).max() <= 0.05 + 1e-6   # lens radius
    assert (rays[:, 6] == 0).all()                        # shutter closed interval [0, 0] still consumes its draw
    assert len({tuple(s) for s in states}) == 64


def test_moving_spheres_need_the_ray_time():
    osc = OracleScene.build(2, 64, 64, seed=0)
    sp = osc.export()[0]
    k = int(np.nonzero((sp[:, 0:3] != sp[:, 3:6]).any(axis=1))[0][0])
    c0, c1 = sp[k, 0:3], sp[k, 3:6]
    org = np.array([c0 + [0, 5, 0]] * 2, np.float32)
    d = np.array([[0, -1, 0]] * 2, np.float32)
    t, obj, _, _, _ = osc.hit_rays(org, d, np.array([0.0, 1.0], np.float32))
    assert obj[0] == k and obj[1] == k
    assert abs((t[0] - t[1]) - (c1[1] - c0[1])) < 1e-5    # the sphere has risen by centre1.y - centre0.y at time 1


# ---------------------------------------------------------------- whole renders
def test_render_modes_agree_statistically_and_per_sample_is_schedule_free():
    osc = OracleScene.build(1, 48, 32, seed=0)
    st = osc.rng_after()
    a = osc.render(48, 32, 64, mode=MODE_REF_SERIAL, state=st)
    b = osc.render(48, 32, 64, seed=0, mode=MODE_PER_SAMPLE, threads=1)
    c = osc.render(48, 32, 64, seed=0, mode=MODE_PER_SAMPLE, threads=7)
    d = osc.render(48, 32, 64, mode=MODE_REF_POOL, state=st, threads=4)
    assert np.array_equal(b[1].view(np.uint32), c[1].view(np.uint32)) and b[2] == c[2]
    for other in (b, d):
        assert abs(a[1].mean() - other[1].mean()) / a[1].mean() < 0.02     # same estimator, different streams
    # serial mode: one stream over pixels in index order — rendering the first row alone gives the same first row
    top = osc.render(48, 1, 64, mode=MODE_REF_SERIAL, state=st)
    assert top[2]["paths"] == 48 * 64


def test_sample_strides_partition_the_samples():
    osc = OracleScene.build(0, 24, 16)
    full = osc.render(24, 16, 9, seed=2, want_per_sample=True)
    parts = [osc.render(24, 16, 9, seed=2, k_begin=r, k_step=3, want_per_sample=True) for r in range(3)]
    assert sum(p[2]["rays"] for p in parts) == full[2]["rays"]
    for r, p in enumerate(parts):
        assert np.array_equal(p[3][r::3], full[3][r::3])


def test_depth_limit_counts():
    osc = OracleScene.build(0, 32, 18)
    r0 = osc.render(32, 18, 4, seed=1, max_depth=0)
    assert r0[2]["rays"] == r0[2]["paths"] and r0[2]["depth_cut"] + r0[2]["sky_misses"] == r0[2]["paths"]
    r50 = osc.render(32, 18, 4, seed=1, max_depth=50)
    c = r50[2]
    assert c["rays"] == c["paths"] + c["hits_lambertian"] + c["hits_metal"] + c["hits_dielectric"] - c["absorbed"]
    assert c["paths"] == c["sky_misses"] + c["depth_cut"] + c["absorbed"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))))
def test_oracle_reproduces_the_committed_fixtures(path):
    g = np.load(path)
    osc = OracleScene.build(int(g["kind"]), int(g["nx"]), int(g["ny"]), seed=int(g["scene_seed"]))
    rgb, acc, ctr = osc.render(int(g["nx"]), int(g["ny"]), int(g["ns"]), seed=int(g["seed"]), max_depth=int(g["max_depth"]))
    assert np.array_equal(rgb, g["rgb"])
    assert np.array_equal(acc.view(np.uint32), g["acc"].view(np.uint32))
    assert [ctr[k] for k in CTR_NAMES] == g["counters"].tolist()


def test_fixture_files_exist():
    assert len(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))) >= 3
