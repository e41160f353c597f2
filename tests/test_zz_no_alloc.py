"""rrt_render_device_noalloc (the entry point that allocates no device memory) against rrt_render_device, through the
C ABI: bit-exact f32 sums, RGB8 and ray counts, on two calls in a row, the second with a smaller input, then its limits.
Needs an MI355X."""
import numpy as np
import pytest
import torch  # before rrt_amd: the process must load torch's own HIP runtime first, or torch finds no GPU afterwards

import rrt_amd
from oracle import OracleScene
from tests.conftest import scene_arrays_from_oracle

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _upload(ctx, osc):
    sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
    ctx.upload_scene(sp, mid, mt, nd)
    ctx.set_camera(rrt_amd.Camera.from_buffer_copy(cam21.tobytes()))


def _both(ctx, p):
    dev = torch.device("cuda", 0)
    n = p.nx * p.ny * 3
    out = []
    for fn in (ctx.render_device, ctx.render_device_noalloc):
        rgb = torch.full((n,), 7, dtype=torch.uint8, device=dev)        # stale contents must not show through
        acc = torch.full((n,), 3.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        st = fn(p, rgb.data_ptr(), acc.data_ptr())
        torch.cuda.synchronize()
        out.append((rgb.cpu().numpy(), acc.cpu().numpy(), st))
    return out


def test_noalloc_matches_render_device_twice_in_a_row(gpu_ctx):
    # first call: a moving-sphere scene, several batches (the workspace holds about 33 samples of 200x120 pixels)
    nx, ny, ns = 200, 120, 40
    osc = OracleScene.build(rrt_amd.SCENE_BOOK2CHAP2, nx, ny, seed=0)
    _upload(gpu_ctx, osc)
    (rgb_a, acc_a, st_a), (rgb_n, acc_n, st_n) = _both(gpu_ctx, rrt_amd.Context.params(nx, ny, ns, seed=5))
    assert np.array_equal(bits(acc_n), bits(acc_a)) and np.array_equal(rgb_n, rgb_a)
    assert st_n["rays"] == st_a["rays"] and st_n["paths"] == st_a["paths"] == nx * ny * ns
    ref = osc.render(nx, ny, ns, seed=5, threads=8)
    assert np.array_equal(bits(acc_n), bits(ref[1].reshape(-1))) and st_n["rays"] == ref[2]["rays"]
    # second call, smaller input (fewer pixels than path slots, another scene): the workspace still holds the first
    # call's paths, queues and counters
    free_before = torch.cuda.mem_get_info()[0]
    nx, ny, ns = 31, 17, 9
    osc = OracleScene.build(rrt_amd.SCENE_CHAP12, nx, ny, seed=0)
    _upload(gpu_ctx, osc)
    (rgb_a, acc_a, st_a), (rgb_n, acc_n, st_n) = _both(gpu_ctx, rrt_amd.Context.params(nx, ny, ns, seed=6, count_work=True))
    assert np.array_equal(bits(acc_n), bits(acc_a)) and np.array_equal(rgb_n, rgb_a)
    for key in ("rays", "paths", "inner_visits", "leaf_tests", "hits_lambertian", "sky_misses", "absorbed"):
        assert st_n[key] == st_a[key], key
    ref = osc.render(nx, ny, ns, seed=6, threads=4)
    assert np.array_equal(bits(acc_n), bits(ref[1].reshape(-1))) and np.array_equal(rgb_n, ref[0].reshape(-1))
    # a sample shard through the same entry point
    p = rrt_amd.Context.params(nx, ny, ns, seed=6, shard_mode=rrt_amd.SHARD_SAMPLE, rank=1, world=2)
    (rgb_a, acc_a, st_a), (rgb_n, acc_n, st_n) = _both(gpu_ctx, p)
    assert np.array_equal(bits(acc_n), bits(acc_a)) and np.array_equal(rgb_n, rgb_a) and st_n["rays"] == st_a["rays"]
    # device memory in use did not grow across the no-allocation calls of this block (torch's own tensors aside,
    # which come from its cache)
    assert torch.cuda.mem_get_info()[0] >= free_before - (4 << 20)


def test_noalloc_on_a_scene_too_big_for_lds(gpu_ctx):
    """14 k spheres: rrt_render_device walks two children per record (a kernel with a stack in scratch memory, which the
    runtime allocates), the entry point that allocates nothing walks the same tree along exit links.  Same image."""
    nx, ny, ns, half = 72, 40, 3, 60
    hs = rrt_amd.HostScene(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=0, half=half)
    gpu_ctx.upload_host_scene(hs)
    (rgb_a, acc_a, st_a), (rgb_n, acc_n, st_n) = _both(gpu_ctx, rrt_amd.Context.params(nx, ny, ns, seed=4, count_work=True))
    assert np.array_equal(bits(acc_n), bits(acc_a)) and np.array_equal(rgb_n, rgb_a)
    for key in ("rays", "paths", "inner_visits", "leaf_tests"):
        assert st_n[key] == st_a[key], key
    osc = OracleScene.build(rrt_amd.SCENE_RANDOM_SPHERES, nx, ny, seed=0, half=half)
    ref = osc.render(nx, ny, ns, seed=4, threads=8)
    assert np.array_equal(bits(acc_n), bits(ref[1].reshape(-1))) and st_n["inner_visits"] == ref[2]["inner_visits"]


def test_noalloc_first_thing_on_a_fresh_context_takes_no_device_memory():
    """The entry point as the first render call of a new context, before any other call could have grown buffers or the
    queues' scratch memory on its behalf: a moving-sphere scene (its shade kernel reserves a private segment) and a scene too
    big for LDS.  Free device memory (hipMemGetInfo, through torch) is read right before and right after each call."""
    dev = torch.device("cuda", 0)
    for kind, nx, ny, ns, half in ((rrt_amd.SCENE_BOOK2CHAP2, 160, 96, 6, 11), (rrt_amd.SCENE_RANDOM_SPHERES, 72, 40, 3, 60)):
        osc = OracleScene.build(kind, nx, ny, seed=0, half=half)
        ref = osc.render(nx, ny, ns, seed=2, threads=8)
        n = nx * ny * 3
        rgb = torch.zeros(n, dtype=torch.uint8, device=dev)
        acc = torch.zeros(n, dtype=torch.float32, device=dev)
        with rrt_amd.Context(0) as ctx:
            _upload(ctx, osc)
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            st = ctx.render_device_noalloc(rrt_amd.Context.params(nx, ny, ns, seed=2), rgb.data_ptr(), acc.data_ptr())
            free_after = torch.cuda.mem_get_info()[0]
            assert free_after >= free_before, f"{free_before - free_after} bytes of device memory taken during the call"
            assert np.array_equal(bits(acc.cpu().numpy()), bits(ref[1].reshape(-1))) and st["rays"] == ref[2]["rays"]


def test_noalloc_rejects_what_does_not_fit(gpu_ctx):
    osc = OracleScene.build(rrt_amd.SCENE_CHAP11, 8, 8)
    _upload(gpu_ctx, osc)
    dev = torch.device("cuda", 0)
    acc = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for p in (rrt_amd.Context.params(1024, 1024, 1),                      # more than 512*512 pixels
              rrt_amd.Context.params(16, 16, 1, max_depth=65),            # trail deeper than the workspace is laid out for
              rrt_amd.Context.params(16, 16, 1, path_slots=1 << 20),      # more path slots than the workspace holds
              rrt_amd.Context.params(64, 64, 2, shard_mode=rrt_amd.SHARD_TILE, rank=0, world=2)):
        with pytest.raises(rrt_amd.RrtError, match="no-allocation"):
            gpu_ctx.render_device_noalloc(p, 0, acc.data_ptr())
    # the largest image it takes
    st = gpu_ctx.render_device_noalloc(rrt_amd.Context.params(512, 512, 2, max_depth=64), 0, acc.data_ptr())
    assert st["paths"] == 512 * 512 * 2
