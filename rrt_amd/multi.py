"""One process per GPU: shard a render over the ranks of a torch.distributed group and combine the framebuffers.

Pixels and samples are independent in the reference (lib.rs:89-110), so a render shards with no exchange during
rendering and one framebuffer combine at the end:

  * tile shards   — rank r renders the 32x8 tiles t with t % world == r; every rank packs the pixels it owns (RGB8 and, if
                    wanted, the f32 sums) and rank 0 gathers them: each peer sends only its share over its own link
                    (24.9 MB of RGB8 in all at 3840x2160) and the owner indices put them in place — bit-identical to a
                    1-GPU render;
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


def owned_pixels(nx: int, ny: int, mode: int, rank: int, world: int):
    """Image pixel indices rank `rank` renders under `mode` (rrt_shard_pixels), as an int64 torch tensor on the CPU."""
    import torch

    return torch.from_numpy(rrt_amd.shard_pixels(nx, ny, mode, rank, world).astype(np.int64))


_tile_cache: dict = {}


def _tile_indices(nx: int, ny: int, rank: int, world: int, dst: int, device):
    """Owner index lists of a TILE split, kept on `device` across calls (a 3840x2160 image is 8.3 M int64 indices: building
    and uploading them belongs to set-up, not to every frame's combine).  Returns (counts per rank, this rank's indices,
    every rank's indices on `dst` or None elsewhere)."""
    key = (nx, ny, rank, world, dst, str(device))
    hit = _tile_cache.get(key)
    if hit is None:
        counts = [int(rrt_amd.lib().rrt_shard_pixels(nx, ny, rrt_amd.SHARD_TILE, r, world, None)) for r in range(world)]
        mine = owned_pixels(nx, ny, rrt_amd.SHARD_TILE, rank, world).to(device)
        every = None
        if rank == dst:
            every = [mine if r == rank else owned_pixels(nx, ny, rrt_amd.SHARD_TILE, r, world).to(device) for r in range(world)]
        if len(_tile_cache) >= 8:
            _tile_cache.clear()
        hit = _tile_cache[key] = (counts, mine, every)
    return hit


def gather_tiles(buf, nx: int, ny: int, rank: int, world: int, dst: int = 0):
    """Combines TILE shards of one image plane set.  `buf` is this rank's full-size buffer [ny*nx*3] (pixels it does not own
    are ignored).  Every rank sends the pixels it owns, padded to the largest share; `dst` puts them in place and returns the
    whole image, the other ranks return their own buffer."""
    import torch
    import torch.distributed as dist

    counts, mine, every = _tile_indices(nx, ny, rank, world, dst, buf.device)
    n_max = max(counts)
    packed = torch.zeros((n_max, 3), dtype=buf.dtype, device=buf.device)
    packed[: counts[rank]] = buf.view(-1, 3)[mine]
    via_host = buf.is_cuda and dist.get_backend() == "gloo"  # rehearsals on a process group without device collectives
    if via_host:
        packed = packed.cpu()
    parts = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, parts, dst=dst)
    if rank != dst:
        return buf
    if via_host:
        parts = [t.to(buf.device) for t in parts]
    out = torch.zeros_like(buf).view(-1, 3)
    for r in range(world):
        out[every[r]] = parts[r][: counts[r]]
    return out.view(-1)


def render_sharded(render_fn: Callable, nx: int, ny: int, ns: int, seed: int, mode: int, rank: int, world: int, dst: int = 0,
                   finalize: Optional[Callable] = None, max_depth: int = 50, want_accum: bool = True) -> Tuple[Optional[object], Optional[object], dict]:
    """Renders this rank's shard and combines on `dst`.  Returns (rgb8, accum, stats) — the image on `dst`, this rank's
    partial buffers elsewhere.  `finalize(accum, ns) -> rgb8` quantises summed sample shards (GPU: Context.finalize).
    `want_accum=False` skips the f32 sums of TILE shards (the reference returns RGB8 only, lib.rs:81)."""
    import torch
    import torch.distributed as dist

    p = shard_params(nx, ny, ns, seed, mode, rank, world, max_depth=max_depth)
    rgb, acc, stats = render_fn(p)
    if world == 1:
        return rgb, acc, stats
    if mode == rrt_amd.SHARD_TILE:
        rgb = gather_tiles(rgb, nx, ny, rank, world, dst)
        if want_accum and acc is not None:
            acc = gather_tiles(acc, nx, ny, rank, world, dst)
    elif mode == rrt_amd.SHARD_SAMPLE:
        if acc.is_cuda and dist.get_backend() == "gloo":  # rehearsal without device collectives
            host = acc.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            acc.copy_(host)
        else:
            dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM)
        if rank == dst:
            if finalize is None:
                rgb = torch.from_numpy(finalize_numpy(acc.cpu().numpy(), ns)).to(acc.device)
            else:
                rgb = finalize(acc, ns)
    else:
        raise ValueError("sharded render needs SHARD_TILE or SHARD_SAMPLE")
    return rgb, acc, stats
