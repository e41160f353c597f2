#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the GPU render path on the chap12 random-spheres scene.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the workload: chap12 scene (seed 0, 484 spheres / 967 BVH nodes), 1200x800 pixels,
500 samples per pixel, depth 50 — every primary ray, BVH walk, sphere test, scatter, and the framebuffer accumulation
and quantisation.  Rays = calls of Bvh::hit (lib.rs:57), counted on the device.  The scene is uploaded before the timed
region; the image stays in HBM.

N > 1 (one process per GPU, torch.distributed over RCCL): weak scaling by sample sharding — the job is the same image at
500*N samples per pixel, rank r renders samples k = r (mod N), the f32 framebuffers are summed onto rank 0 with one
reduce over xGMI and quantised there (rrt_amd/multi.py).

Rank 0 prints one JSON line.  `roofline` prices the extend kernel (BVH walk + sphere quadratic) against the FP32 vector
peak, with launch durations from HIP events around every extend launch of the timed steps and work counters from one
extra pass of the counting build of the same kernels (same seeds, deterministic).  `cpu_baseline` is the C restatement
of the reference's threaded CPU renderer (oracle/) on a bounded sample of the same workload, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NY, NS, MAX_DEPTH, SEED = 1200, 800, 500, 50, 0
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X: 256 CU x 4 SIMD x 16 lanes x 2 (packed) x 2 (fma) x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def extend_flops(st):
    """Algorithmic FLOPs of the extend kernel: 22 per slab test (acceleration.rs:60-84) and 39 per sphere test
    (sphere.rs:68-85).  1/direction and direction.direction are computed where a ray is made, not here."""
    return 22 * st["inner_visits"] + 39 * st["leaf_tests"]


def job_flops(st):
    """Algorithmic FLOPs of the whole path: 3 divides for 1/d and 3 multiplies of the attenuation product per ray, box and
    sphere tests, hit record, scatter per material, sky, ray generation (fma = 2, min/max/compare = 1, integer work and
    selects not counted)."""
    hits = st["hits_lambertian"] + st["hits_metal"] + st["hits_dielectric"]
    return (6 * st["rays"] + 22 * st["inner_visits"] + 39 * st["leaf_tests"] + 20 * hits + 34 * st["hits_lambertian"]
            + 54 * st["hits_metal"] + 45 * st["hits_dielectric"] + 22 * st["sky_misses"] + 45 * st["paths"])


def extend_bytes(st):
    """Algorithmic bytes the extend kernel moves to and from HBM: per ray 48 B of ray state + 4 B list entry in, 8 B hit
    out.  (The BVH — 32 B per box test, 16 B per sphere test — is read from the LDS copy, not from HBM.)"""
    return 60 * st["rays"]


def cpu_baseline(ns_sample=200, ns_one_core=8):
    from oracle import MODE_REF_POOL, OracleScene

    threads = min(16, os.cpu_count() or 1)
    osc = OracleScene.build(1, NX, NY, seed=SEED)
    state = osc.rng_after()
    t0 = time.perf_counter()
    _, _, ctr1 = osc.render(NX, NY, ns_one_core, seed=SEED, mode=MODE_REF_POOL, threads=1, state=state, pow5_mode=1)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, ctr = osc.render(NX, NY, ns_sample, seed=SEED, mode=MODE_REF_POOL, threads=threads, state=state, pow5_mode=1)
    dt = time.perf_counter() - t0
    return {"value": round(ctr["rays"] / dt / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "one_core": {"value": round(ctr1["rays"] / dt1 / 1e6, 2), "unit": "Mrays/s", "sample": f"{ns_one_core} spp, {ctr1['rays']} rays in {dt1:.2f} s"},
            "sample": f"chap12 {NX}x{NY}, {ns_sample} of {NS} spp, depth {MAX_DEPTH}, seed {SEED}: {ctr['rays']} rays in {dt:.2f} s; "
                      "C restatement of the reference CPU path (recursive trace, per-thread jumped Xoshiro128+, pixels claimed "
                      "dynamically, libm powf), scene build excluded; the Rust reference itself cannot be built here"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=NS, help="samples per pixel per GPU (the headline workload is 500)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend; gloo + RRT_BENCH_ONE_DEVICE=1 rehearses N > 1 on a 1-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rrt_amd
    from rrt_amd import multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if os.environ.get("RRT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0  # rehearsal: every rank shares GPU 0 (not a scaling measurement)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    ctx = rrt_amd.Context(local_rank)
    scene = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, NX, NY, seed=SEED)
    ctx.upload_host_scene(scene)

    ns_total = args.spp * world
    accum = torch.zeros(NY * NX * 3, dtype=torch.float32, device=dev)
    rgb8 = torch.zeros(NY * NX * 3, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    flags = {"count": False, "timed": False}

    TIME_STRIDE = 4  # HIP events around every 4th extend launch: an even sample, a quarter of the event traffic

    def render_fn(p):
        p.count_work = int(flags["count"])
        p.reserved[0] = TIME_STRIDE if flags["timed"] else 0
        st = ctx.render_device(p, rgb8.data_ptr() if world == 1 else 0, accum.data_ptr())
        return rgb8, accum, st

    def finalize(acc, ns):
        torch.cuda.synchronize()
        ctx.finalize(acc.data_ptr(), NX * NY, ns, rgb8.data_ptr())
        return rgb8

    def step(count=False, timed=False):
        flags["count"], flags["timed"] = count, timed
        _, _, st = multi.render_sharded(render_fn, NX, NY, ns_total, SEED, rrt_amd.SHARD_SAMPLE, rank, world, dst=0, finalize=finalize,
                                        max_depth=MAX_DEPTH)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays_local, ext_ms, ext_launches, gpu_ms = 0, 0.0, 0, 0.0
    for _ in range(args.steps):
        st = step(timed=not args.no_roofline)
        rays_local += st["rays"]
        ext_ms += st["extend_ms"]
        ext_launches += st["extend_launches"]
        gpu_ms += st["gpu_ms"]
    fence()
    elapsed = time.perf_counter() - t0

    rays_total = rays_local
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_local], dtype=torch.int64, device=dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = int(r.item())

    roofline = None
    if not args.no_roofline:
        cst = step(count=True)  # every rank takes part (the pass ends in the same reduce); rank 0 reports its own counters
        # one more step with the path pool in one part on one stream: extend launches that overlap nothing
        os.environ["RRT_STREAMS"] = "1"
        ist = step(timed=True)
        del os.environ["RRT_STREAMS"]
        if rank == 0 and ext_ms > 0:
            flops = extend_flops(cst) * args.steps
            byts = extend_bytes(cst) * args.steps
            sec = ext_ms * 1e-3 * TIME_STRIDE          # the timed launches are every TIME_STRIDE-th one
            ext_launches *= TIME_STRIDE
            traffic = None
            try:  # HBM bytes per extend launch from separate rocprofv3 --pmc passes (profiles/README.md)
                with open(os.path.join(ROOT, "profiles", "extend_traffic.json")) as f:
                    traffic = json.load(f)["bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
            roofline = {
                "kernel": "k_extend_lds", "bound": "valu_fp32", "achieved": round(flops / sec / 1e12, 3), "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(flops / sec / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4), "traffic": traffic,
                "launches": ext_launches, "timed_launches": ext_launches // TIME_STRIDE, "avg_launch_ms": round(ext_ms * TIME_STRIDE / max(1, ext_launches), 4),
                "flop_per_launch": round(flops / max(1, ext_launches)),
                "note": "extend launches of the parts of the path pool run on streams of their own and overlap each other and the "
                        "shade kernel, so the summed launch durations exceed wall time and `achieved` is a lower bound; `isolated` is "
                        "the same kernel on the whole pool with nothing beside it (one extra step on one stream)",
                "isolated": {"achieved": round(extend_flops(cst) / (ist["extend_ms"] * TIME_STRIDE * 1e-3) / 1e12, 3),
                             "frac": round(extend_flops(cst) / (ist["extend_ms"] * TIME_STRIDE * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                             "launches": ist["extend_launches"] * TIME_STRIDE, "avg_launch_ms": round(ist["extend_ms"] / max(1, ist["extend_launches"]), 4),
                             "step_ms": round(ist["gpu_ms"], 2)},
                "whole_job": {"achieved": round(job_flops(cst) * args.steps / elapsed / 1e12, 3), "unit": "TFLOP/s",
                              "frac": round(job_flops(cst) * args.steps / elapsed / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                              "note": "all algorithmic FLOPs of a step (every kernel) over wall time, per GPU (rank 0's share)"},
                "hbm": {"achieved": round(byts / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / sec / 1e9 / HBM_PEAK_GBS, 4)},
                "per_ray": {"inner_visits": round(cst["inner_visits"] / cst["rays"], 2), "leaf_tests": round(cst["leaf_tests"] / cst["rays"], 2),
                            "rays_per_path": round(cst["rays"] / cst["paths"], 3)},
            }

    if rank == 0:
        out = {
            "metric": "Mrays/s (paths x bounces), chap12 scene", "value": round(rays_total / elapsed / 1e6, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"chap12 random-spheres scene seed {SEED} (484 spheres, 967 BVH nodes), {NX}x{NY}, "
                                   f"{args.spp} spp per GPU ({ns_total} spp total), max depth {MAX_DEPTH}",
                       "sharding": "none" if world == 1 else f"sample shards, f32 framebuffer reduce over RCCL ({world} ranks)",
                       "rays_per_step": rays_total // max(1, args.steps)},
            "device": ctx.device_info()["name"],
        }
        if roofline is not None:
            out["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
