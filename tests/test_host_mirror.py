"""Host-side product code (C++ mirror + layout builder) against the oracle, no GPU needed."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import rrt_amd
from oracle import OracleScene, camera24_to_abi
from oracle import lib as olib
from tests.conftest import ROOT, scene_arrays_from_oracle

LEAF, END, IDX = 0x80000000, 0x7FFFFFFF, 0x1FFFFFFF


def test_library_exports_every_declared_symbol():
    L = rrt_amd.lib()
    declared = set()
    for h in ("rrt_amd.h", "rrt_amd_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"RRT_API\s+[\w\s\*]+?\b(rrt_\w+)\s*\(", text))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} is declared in include/ but not exported by librrt_amd.so"
    assert b"gfx950" in L.rrt_version()


def test_host_rng_matches_published_vectors_and_oracle():
    L = rrt_amd.lib()
    st = (C.c_uint32 * 4)(1, 2, 3, 4)
    got = [L.rrt_host_rng_next_u32(st) for _ in range(10)]
    assert got == [5, 12295, 25178119, 27286542, 39879690, 1140358681, 3276312097, 4110231701, 399823256, 2144435200]
    for seed in (0, 1, 0xDEADBEEF, 2**63 + 12345):
        a = (C.c_uint32 * 4)()
        b = np.zeros(4, np.uint32)
        L.rrt_host_rng_seed(seed, a)
        olib().orc_rng_seed(seed, b.ctypes.data)
        assert list(a) == list(b)
        assert [L.rrt_host_rng_gen_f32(a) for _ in range(8)] == [olib().orc_rng_gen_f32(b.ctypes.data) for _ in range(8)]
        L.rrt_host_rng_jump(a)
        olib().orc_rng_jump(b.ctypes.data)
        assert list(a) == list(b)


def test_host_uniform_matches_oracle():
    L = rrt_amd.lib()
    for lo, hi in [(-1.0, 1.0), (0.0, 1.0), (0.0, 0.5), (0.0, 0.0), (2.5, 7.25), (-3.0, -1.0)]:
        l1, s1, l2, s2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        L.rrt_host_uniform_new_inclusive(lo, hi, C.byref(l1), C.byref(s1))
        olib().orc_uniform_new_inclusive(lo, hi, C.byref(l2), C.byref(s2))
        assert struct.pack("f", s1.value) == struct.pack("f", s2.value) and l1.value == l2.value
        a = (C.c_uint32 * 4)(9, 8, 7, 6)
        b = np.array([9, 8, 7, 6], np.uint32)
        for _ in range(16):
            assert L.rrt_host_uniform_sample(l1.value, s1.value, a) == olib().orc_uniform_sample(l2.value, s2.value, b.ctypes.data)


@pytest.mark.parametrize("kind,seed,half", [(0, 0, 11), (1, 0, 11), (1, 12345, 11), (2, 0, 11), (2, 77, 11), (3, 0, 30)])
def test_scene_builders_match_oracle_bit_for_bit(kind, seed, half):
    nx, ny = 1200, 800
    hs = rrt_amd.HostScene(kind, nx, ny, seed=seed, half=half)
    osc = OracleScene.build(kind, nx, ny, seed=seed, half=half)
    sp_h, mt_h, nd_h, cam_h = hs.export()
    sp_o, mt_o, nd_o, cam_o = osc.export()
    assert sp_h.shape == sp_o.shape and nd_h.shape == nd_o.shape
    assert np.array_equal(sp_h.view(np.uint32), sp_o.view(np.uint32))
    assert np.array_equal(mt_h.view(np.uint32), mt_o.view(np.uint32))
    inner = (nd_o[:, 7] >> 31) == 0
    assert np.array_equal(nd_h[inner], nd_o[inner])
    assert np.array_equal(nd_h[~inner][:, 6:8], nd_o[~inner][:, 6:8])
    assert np.array_equal(cam_h.as_array().view(np.uint32), camera24_to_abi(cam_o).view(np.uint32))
    assert np.array_equal(hs.rng_after(), osc.rng_after())


def test_reference_scene_census():
    hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, 1200, 800, seed=0)
    assert (hs.n_spheres, hs.n_nodes) == (484, 967)
    sp, mt, nd, cam = hs.export()
    kinds = mt[1:-3, 0].astype(int)
    assert (np.bincount(kinds, minlength=3) == [393, 65, 22]).all()
    hs11 = rrt_amd.HostScene(rrt_amd.SCENE_CHAP11, 400, 225)
    assert (hs11.n_spheres, hs11.n_nodes) == (5, 9)
    hb = rrt_amd.HostScene(rrt_amd.SCENE_BOOK2CHAP2, 1200, 800, seed=0)
    spb = hb.export()[0]
    moving = (spb[:, 0:3] != spb[:, 3:6]).any(axis=1)
    assert moving.sum() > 380 and np.array_equal(spb[:, 0:3], sp[:, 0:3])  # same main-stream draws as chap12 (book2chap2.rs:11-13)


def linked_visit_order(words, root, oct_, hit_fn):
    """Walk the exit-link layout the way the extend kernel does; hit_fn(node) says whether an inner node's box is hit."""
    order = []
    cur = root
    guard = 0
    while cur != END:
        guard += 1
        assert guard < 10 * len(words) + 10, "link walk does not terminate"
        i = cur & IDX
        w = words[i]
        order.append(i)
        if cur & LEAF:
            cur = int(w[8 + oct_])
        elif hit_fn(i):
            axis = (int(w[6]) >> 29) & 3
            cur = int(w[6]) & (LEAF | IDX) if (oct_ >> axis) & 1 else int(w[7])
        else:
            cur = int(w[8 + oct_])
    return order


def stack_visit_order(nd, oct_, hit_fn):
    """acceleration.rs:232-270 literally, on the interchange nodes."""
    order, stack = [], [0]
    while stack:
        i = stack.pop()
        order.append(i)
        if nd[i, 7] >> 31:
            continue
        if not hit_fn(i):
            continue
        axis = (int(nd[i, 7]) >> 29) & 3
        left, right = int(nd[i, 6]), int(nd[i, 7]) & IDX
        if (oct_ >> axis) & 1:
            stack += [left, right]
        else:
            stack += [right, left]
    return order


@pytest.mark.parametrize("kind", [0, 1])
def test_exit_links_reproduce_the_stack_order(kind):
    osc = OracleScene.build(kind, 64, 64, seed=2)
    sp, mid, mt, nd, _ = scene_arrays_from_oracle(osc)
    words, root, flags = rrt_amd.debug_layout(sp, mid, mt, nd)
    rng = np.random.default_rng(0)
    for trial in range(24):
        oct_ = trial % 8
        p = [1.0, 0.0, 0.5, 0.8][trial % 4]
        hits = rng.uniform(size=len(nd)) < p
        f = lambda i: bool(hits[i])
        assert linked_visit_order(words, root, oct_, f) == stack_visit_order(nd, oct_, f)
    # leaf payload: centre, radius^2, sphere index with the material kind in the top two bits
    leaves = np.nonzero(nd[:, 7] >> 31)[0]
    for i in leaves[:50]:
        obj = int(nd[i, 6])
        assert int(words[i, 4]) & 0x3FFFFFFF == obj
        assert int(words[i, 4]) >> 30 == int(mt[mid[obj], 0])
        assert np.array_equal(words[i, 0:3], sp[obj, 0:3].view(np.uint32))
        assert words[i, 3] == np.float32(sp[obj, 8] * sp[obj, 8]).view(np.uint32)
    # nothing moves; bit 2: every centre and radius is small enough for the short division of the sphere test; bit 3: every
    # radius is in [2^-30, 2^30] in magnitude (the shade kernel divides by it through its reciprocal)
    assert flags == 12


def test_layout_flags_moving_scene():
    osc = OracleScene.build(2, 64, 64, seed=0)
    sp, mid, mt, nd, _ = scene_arrays_from_oracle(osc)
    words, root, flags = rrt_amd.debug_layout(sp, mid, mt, nd)
    assert flags == 15  # has moving spheres, all with the same shutter interval, bounded (bit 2), radii normal (bit 3)
    sp2 = sp.copy()
    k = np.nonzero((sp[:, 0:3] != sp[:, 3:6]).any(axis=1))[0][0]
    sp2[k, 7] = 2.0
    # another shutter interval: not uniform, and then not "bounded" either; such a scene is not walked in LDS and gets the
    # records of the two-children walk (bit 4)
    assert rrt_amd.debug_layout(sp2, mid, mt, nd)[2] == 25
    # a centre or a radius out of the short division's range takes bit 2 away and nothing else
    for col, val in [(0, 3.0e9), (5, -1.0e9), (8, 2.0e9), (1, np.nan)]:
        sp3 = sp.copy()
        sp3[k, col] = val
        if col < 3:
            sp3[k, col + 3] = val  # keep centre1 - centre0 as it was in kind (the sphere stays a moving one)
        assert rrt_amd.debug_layout(sp3, mid, mt, nd)[2] & 4 == 0
    # a radius outside [2^-30, 2^30] (or NaN) takes bit 3 away
    for val in (1.0e-12, -2.0e9, np.nan):
        sp3 = sp.copy()
        sp3[k, 8] = val
        assert rrt_amd.debug_layout(sp3, mid, mt, nd)[2] & 8 == 0


def deep_tree_scene(n=1500):
    """Spheres along the x axis at geometrically growing distances (2^-40 .. 2^40): every middle split peels a few of the
    farthest off, so the reference's tree is about 80 levels deep.  A ray that walked down all of them would overflow
    the reference's 64-entry stack (acceleration.rs:233); the camera that comes with the scene looks across the far end
    of the row, where the walk stays shallow."""
    sp = np.zeros((n, 9), np.float32)
    x = 2.0 ** (-40.0 + 80.0 * np.arange(n, dtype=np.float64) / n)
    sp[:, 0] = sp[:, 3] = x.astype(np.float32)
    sp[:, 1] = sp[:, 4] = (0.01 * x).astype(np.float32) * np.where(np.arange(n) % 2, 1.0, -1.0).astype(np.float32)
    sp[:, 7] = 1.0
    sp[:, 8] = (0.012 * x).astype(np.float32)
    mt = np.zeros((n, 6), np.float32)
    mt[:, 0] = np.arange(n) % 3
    mt[:, 1:4] = 0.7
    mt[:, 4] = 0.1
    mt[:, 5] = 1.5
    far = float(2.0 ** 40)
    cam = rrt_amd.camera_from((0.7 * far, 0.0, 0.5 * far), (0.7 * far, 0.0, 0.0), (0, 1, 0), 40.0, 4.0 / 3.0, 0.0, 0.5 * far, 0.0, 0.0)
    return sp, np.arange(n, dtype=np.uint32), mt, cam


def test_layout_flag_of_the_two_children_walk():
    """Bit 4: scenes whose nodes stay in HBM get two-children records unless the tree is deeper than the walk's stack."""
    from oracle import OracleScene
    small = OracleScene.build(1, 32, 32, seed=0)
    sp, mid, mt, nd, _ = scene_arrays_from_oracle(small)
    assert rrt_amd.debug_layout(sp, mid, mt, nd)[2] & 16 == 0      # walked in LDS
    big = OracleScene.build(3, 32, 32, seed=0, half=60)
    sp, mid, mt, nd, _ = scene_arrays_from_oracle(big)
    assert rrt_amd.debug_layout(sp, mid, mt, nd)[2] & 16 == 16
    sp, mid, mt, cam = deep_tree_scene()
    deep = OracleScene.from_arrays(sp, mid, mt, np.zeros((0, 8), np.uint32), cam.as_array())
    deep.build_bvh(0)
    _, _, _, nd, _ = scene_arrays_from_oracle(deep)
    assert nd.shape[0] == 2 * sp.shape[0] - 1
    assert rrt_amd.debug_layout(sp, mid, mt, nd)[2] & 16 == 0      # deeper than 64 levels: the exit-link walk


def test_scene_validation_errors():
    osc = OracleScene.build(0, 16, 16)
    sp, mid, mt, nd, _ = scene_arrays_from_oracle(osc)
    rrt_amd.validate_scene(sp, mid, mt, nd)
    with pytest.raises(rrt_amd.RrtError, match="empty scene"):
        rrt_amd.validate_scene(sp[:0], mid[:0], mt, nd[:0])
    bad = nd.copy()
    bad[0, 6] = 99
    with pytest.raises(rrt_amd.RrtError, match="out of range"):
        rrt_amd.validate_scene(sp, mid, mt, bad)
    cyc = nd.copy()
    inner = np.nonzero((nd[:, 7] >> 31) == 0)[0]
    cyc[inner[-1], 6] = 0  # child points back at the root
    with pytest.raises(rrt_amd.RrtError, match="not a tree"):
        rrt_amd.validate_scene(sp, mid, mt, cyc)
    leafbad = nd.copy()
    leaf = np.nonzero(nd[:, 7] >> 31)[0][0]
    leafbad[leaf, 6] = 1000
    with pytest.raises(rrt_amd.RrtError, match="object index"):
        rrt_amd.validate_scene(sp, mid, mt, leafbad)
    with pytest.raises(rrt_amd.RrtError, match="material_id"):
        rrt_amd.validate_scene(sp, mid + 100, mt, nd)
    mtbad = mt.copy()
    mtbad[0, 0] = 7
    with pytest.raises(rrt_amd.RrtError, match="material kind"):
        rrt_amd.validate_scene(sp, mid, mtbad, nd)


def test_tile_shards_partition_the_image():
    for nx, ny, world in [(100, 37, 3), (1200, 800, 8), (33, 9, 4), (5, 5, 8)]:
        seen = np.zeros(nx * ny, np.int32)
        for r in range(world):
            px = rrt_amd.shard_pixels(nx, ny, rrt_amd.SHARD_TILE, r, world)
            seen[px] += 1
        assert (seen == 1).all()
    assert np.array_equal(rrt_amd.shard_pixels(7, 3, rrt_amd.SHARD_NONE, 0, 1), np.arange(21))
    # 32x8 tiles dealt round-robin: pixel (x=40, y=3) is in tile 1
    px = rrt_amd.shard_pixels(100, 37, rrt_amd.SHARD_TILE, 1, 3)
    assert 3 * 100 + 40 in px


def test_camera_from_matches_oracle():
    ci = np.array([13, 2, 3, 0, 0, 0, 0, 1, 0, 20, 1.5, 0.1, 10, 0, 1], np.float32)
    out = np.zeros(24, np.float32)
    olib().orc_camera_from(ci.ctypes.data, out.ctypes.data)
    cam = rrt_amd.camera_from(ci[0:3], ci[3:6], ci[6:9], *[float(v) for v in ci[9:15]])
    assert np.array_equal(cam.as_array().view(np.uint32), camera24_to_abi(out).view(np.uint32))


def test_ppm_writer(tmp_path):
    img = np.arange(4 * 3 * 3, dtype=np.uint8).reshape(3, 4, 3)
    path = str(tmp_path / "o.ppm")
    rrt_amd.write_ppm(path, img)
    data = open(path, "rb").read()
    assert data.startswith(b"P6\n4 3\n255\n") and data[len(b"P6\n4 3\n255\n"):] == img.tobytes()


def test_no_gpu_means_a_loud_failure():
    """The product never falls back to a CPU path: without a device, creating a context raises."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rrt_amd.RrtError, match="no HIP device|HIP"):
        rrt_amd.Context(0)


def test_cli_argument_handling():
    """The flags of main.rs:12-71; these paths return before any device is touched."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(rrt_amd.__file__)), "rrt")
    r = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and "--resolution" in r.stderr and "--samples" in r.stderr and "--seed" in r.stderr and "--random" in r.stderr
    r = subprocess.run([exe, "-r", "12y8"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and "invalid resolution" in r.stderr  # main.rs:49-52: message, then a normal return
    r = subprocess.run([exe, "-s", "many"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and "invalid sample count" in r.stderr
    r = subprocess.run([exe, "-e", "-3"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and "invalid rng seed" in r.stderr
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 2


def test_exit_links_on_a_tree_big_enough_for_the_threaded_layout_passes():
    """~47 k spheres / ~94 k nodes: past the size from which the layout builder shares its passes out to host threads."""
    hs = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, 64, 64, seed=1, half=110)
    sp, mt, nd, _ = hs.export()
    assert nd.shape[0] > (1 << 16)
    words, root, flags = rrt_amd.debug_layout(sp, np.arange(sp.shape[0], dtype=np.uint32), mt, nd)
    rng = np.random.default_rng(1)
    for oct_, p in [(0, 1.0), (5, 0.6), (7, 0.85)]:
        hits = rng.uniform(size=len(nd)) < p
        f = lambda i: bool(hits[i])
        assert linked_visit_order(words, root, oct_, f) == stack_visit_order(nd, oct_, f)
