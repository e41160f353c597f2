"""One process per GPU: shard a render over the ranks of a torch.distributed group and combine the framebuffers.

Pixels and samples are independent in the reference (lib.rs:89-110), so a render shards with no exchange during
rendering and one framebuffer combine at the end:

  * tile shards   — rank r renders the 32x8 tiles t with t % world == r into a zero-filled full-size RGB8 / f32 image;
                    images of disjoint pixel sets add up to the whole image (one reduce, bit-identical to a 1-GPU render);
  * sample shards — rank r renders samples k with k % world == r of every pixel; the f32 sums are added (one reduce),
                    then quantised once (lib.rs:105-108).  Same rays as the 1-GPU render, another order of f32 additions.

`render_fn(params) -> (rgb8 tensor [ny*nx*3] u8 or None, accum tensor [ny*nx*3] f32)` does one rank's share; on a GPU it is
Context.render_device on torch tensors (see `gpu_render_fn`), in the CPU tests a stand-in.  Collectives run on whatever
backend the process group has: "nccl" (RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

import rrt_amd


def shard_params(nx: int, ny: int, ns: int, seed: int, mode: int, rank: int, world: int, max_depth: int = 50, **kw) -> rrt_amd.RenderParams:
    if world == 1:
        return rrt_amd.Context.params(nx, ny, ns, seed=seed, max_depth=max_depth, **kw)
    return rrt_amd.Context.params(nx, ny, ns, seed=seed, max_depth=max_depth, shard_mode=mode, rank=rank, world=world, **kw)


def finalize_numpy(accum: np.ndarray, ns: int) -> np.ndarray:
    """lib.rs:105-108 on the host (used where no GPU is present): sqrt(sum / ns) * 255.99, saturating u8."""
    v = np.sqrt(accum.astype(np.float32) / np.float32(ns)) * np.float32(255.99)
    v = np.where(np.isnan(v), 0, v)
    return np.clip(np.trunc(v), 0, 255).astype(np.uint8)


def gpu_render_fn(ctx: "rrt_amd.Context", device) -> Callable:
    import torch

    def fn(p: rrt_amd.RenderParams):
        n = p.nx * p.ny * 3
        rgb = torch.zeros(n, dtype=torch.uint8, device=device)
        acc = torch.zeros(n, dtype=torch.float32, device=device)
        torch.cuda.synchronize(device)
        stats = ctx.render_device(p, rgb.data_ptr(), acc.data_ptr())
        return rgb, acc, stats

    return fn


def render_sharded(render_fn: Callable, nx: int, ny: int, ns: int, seed: int, mode: int, rank: int, world: int, dst: int = 0,
                   finalize: Optional[Callable] = None, max_depth: int = 50) -> Tuple[Optional[object], Optional[object], dict]:
    """Renders this rank's shard and combines on `dst`.  Returns (rgb8, accum, stats) — the image on `dst`, this rank's
    partial buffers elsewhere.  `finalize(accum, ns) -> rgb8` quantises summed sample shards (GPU: Context.finalize)."""
    import torch
    import torch.distributed as dist

    p = shard_params(nx, ny, ns, seed, mode, rank, world, max_depth=max_depth)
    rgb, acc, stats = render_fn(p)
    if world == 1:
        return rgb, acc, stats
    if mode == rrt_amd.SHARD_TILE:
        # disjoint pixel sets, zero elsewhere: the sum is the image
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(rgb, dst=dst, op=dist.ReduceOp.SUM)
    elif mode == rrt_amd.SHARD_SAMPLE:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
        if rank == dst:
            if finalize is None:
                rgb = torch.from_numpy(finalize_numpy(acc.cpu().numpy(), ns)).to(acc.device)
            else:
                rgb = finalize(acc, ns)
    else:
        raise ValueError("sharded render needs SHARD_TILE or SHARD_SAMPLE")
    return rgb, acc, stats
