"""The multi-rank path (shard -> render -> framebuffer combine) with world_size 2 and 3 over gloo, on the CPU.

No GPU here, so each rank's share is produced by a stand-in built on the oracle; what is under test is the product's
sharding description (rrt_shard_pixels, the sample stride) and rrt_amd.multi's combine logic and collectives."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rrt_amd
from rrt_amd import multi

NX, NY, NS, SEED = 70, 21, 6, 3


def oracle_render_fn(osc):
    """Stand-in for Context.render_device: honours the shard fields of the params exactly as the library documents them."""
    def fn(p):
        world = p.shard_world if p.shard_mode != rrt_amd.SHARD_NONE else 1
        rank = p.shard_rank if world > 1 else 0
        if p.shard_mode == rrt_amd.SHARD_SAMPLE and world > 1:
            rgb, acc, ctr = osc.render(p.nx, p.ny, p.ns, seed=p.seed, max_depth=p.max_depth, k_begin=rank, k_step=world)
        elif p.shard_mode == rrt_amd.SHARD_TILE and world > 1:
            px = rrt_amd.shard_pixels(p.nx, p.ny, rrt_amd.SHARD_TILE, rank, world)
            a, r, rays = osc.render_pixels(p.nx, p.ny, p.ns, p.seed, px, max_depth=p.max_depth)
            rgb = np.zeros((p.ny * p.nx, 3), np.uint8)
            acc = np.zeros((p.ny * p.nx, 3), np.float32)
            rgb[px], acc[px] = r, a
            ctr = {"rays": rays}
        else:
            rgb, acc, ctr = osc.render(p.nx, p.ny, p.ns, seed=p.seed, max_depth=p.max_depth)
        return torch.from_numpy(rgb.reshape(-1).copy()), torch.from_numpy(acc.reshape(-1).copy()), ctr
    return fn


def worker(rank, world, port, mode, out_dir):
    from oracle import OracleScene

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        osc = OracleScene.build(rrt_amd.SCENE_CHAP12, NX, NY, seed=0)
        rgb, acc, stats = multi.render_sharded(oracle_render_fn(osc), NX, NY, NS, SEED, mode, rank, world)
        rays = torch.tensor([stats["rays"]], dtype=torch.int64)
        dist.all_reduce(rays)
        if rank == 0:
            np.savez(os.path.join(out_dir, f"out_{mode}_{world}.npz"), rgb=rgb.numpy(), acc=acc.numpy(), rays=rays.numpy())
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("mode,world", [(rrt_amd.SHARD_TILE, 2), (rrt_amd.SHARD_SAMPLE, 2), (rrt_amd.SHARD_TILE, 3)])
def test_sharded_render_combines_to_the_single_rank_image(tmp_path, mode, world):
    from oracle import OracleScene

    mp.spawn(worker, args=(world, free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / f"out_{mode}_{world}.npz")
    osc = OracleScene.build(rrt_amd.SCENE_CHAP12, NX, NY, seed=0)
    rgb, acc, ctr = osc.render(NX, NY, NS, seed=SEED)
    assert int(got["rays"][0]) == ctr["rays"]
    if mode == rrt_amd.SHARD_TILE:
        assert np.array_equal(got["rgb"].reshape(NY, NX, 3), rgb)
        assert np.array_equal(got["acc"].reshape(NY, NX, 3).view(np.uint32), acc.view(np.uint32))
    else:
        # same rays, another order of f32 additions: equal up to the rounding of NS additions; RGB8 within one step
        assert np.allclose(got["acc"].reshape(NY, NX, 3), acc, rtol=2e-6, atol=1e-6)
        diff = np.abs(got["rgb"].reshape(NY, NX, 3).astype(int) - rgb.astype(int))
        assert diff.max() <= 1 and (diff != 0).mean() < 0.01


def test_finalize_numpy_matches_the_oracle_quantiser():
    from oracle import OracleScene

    osc = OracleScene.build(rrt_amd.SCENE_CHAP11, 40, 23)
    rgb, acc, _ = osc.render(40, 23, 5, seed=1)
    assert np.array_equal(multi.finalize_numpy(acc, 5), rgb)
    assert multi.finalize_numpy(np.array([np.nan, -1.0, 1e9], np.float32), 1).tolist() == [0, 0, 255]
