#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the GPU render path on the chap12 random-spheres scene.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the workload — every primary ray, BVH walk, sphere test, scatter, and the framebuffer
accumulation and quantisation.  Rays = calls of Bvh::hit (lib.rs:57), counted on the device.  The scene is uploaded before
the timed region; the image stays in HBM.

  N = 1   chap12 (seed 0, 484 spheres / 967 BVH nodes), 1200x800, 500 samples per pixel, depth 50.
  N > 1   one process per GPU (torch.distributed, backend nccl = RCCL): the target configuration, chap12 at 3840x2160 with
          1024 samples per pixel IN TOTAL, strong-scaled.  Default: TILE shards (32x8 tiles dealt round-robin; every rank
          sends the RGB8 pixels it owns to rank 0, 24.9 MB in all).  --shard sample: rank r renders samples k = r (mod N),
          the f32 sums are reduced onto rank 0 (99.5 MB per rank) and quantised there.  Rank 0 checks that the ranks'
          device ray counts add up to the count of the same render on one GPU.

Rank 0 prints one JSON line.  `roofline` prices the extend kernel (BVH walk + sphere quadratic) against the FP32 vector
peak, with launch durations from HIP events around every 4th extend launch of the timed steps and work counters from one
extra pass of the counting build of the same kernels (same seeds, deterministic); counter-derived figures (VALU busy, lane
utilisation, LDS conflicts, HBM bytes per ray) come from the kept rocprofv3 summaries under profiles/.  `other_configs`
times one step each of the moving-sphere scene and the 1 M-sphere scene.  `cpu_baseline` is the C restatement of the
reference's threaded CPU renderer (oracle/) on a bounded sample of the same workload, on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NX, NY, NS, MAX_DEPTH, SEED = 1200, 800, 500, 50, 0          # N = 1: BASELINE configs[1]
NX4, NY4, NS4 = 3840, 2160, 1024                             # N > 1: BASELINE configs[3], the target
# Device-counted rays of the 3840x2160x1024 render on one GPU (tests/test_gpu_parity.py::test_cfg4_*, whose sampled pixels
# equal the oracle's bit for bit): what the ranks' counts must add up to, however the render is sharded.
RAYS_4K_1024 = 24005553045
# MI355X FP32 vector peak: 256 CUs x 4 SIMDs x 32 lanes x 2 (fma) x 2.4 GHz.  Only v_fma/v_add/v_mul/v_sub_f32 issue at
# that rate (2.3 cycles per wave64 measured); every other VALU instruction of these kernels takes 4.2 (profiles/r02_valu_rate.txt).
FP32_VECTOR_PEAK_TFLOPS = 157.3
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


def extend_bytes(st, moving=False):
    """Bytes the extend kernel moves to and from HBM as it is built: per ray the origin (12 B; 16 B with the shutter time in
    scenes with moving spheres) and the direction with the path id (16 B) in, the 8-byte hit record out, and 12 B of the
    list plane (a workgroup reads its lists, writes the order in which it takes its pool, camera rays last, and reads a
    4-byte slot number per hand-out); 1/direction and direction.direction are recomputed, not stored.  (The BVH, 32 B per
    box test and 16 B per sphere test, is read from the LDS copy, not from HBM.)  The measured figure
    (counters.bytes_per_ray.extend) is higher: a 128-byte line of a ray plane holds slots under way and fresh slots, which
    the pool's order takes some 50 us apart, so most lines are fetched twice."""
    return ((16 if moving else 12) + 16 + 8 + 12) * st["rays"]


def extend_bytes_survey(st):
    """The survey's pricing of the same kernel: a 60-byte payload per ray (origin, direction, time, throughput, pixel and
    generator state in one record).  Kept beside the figure above: this implementation moves less."""
    return 60 * st["rays"]


def job_bytes(st, npix):
    """Algorithmic bytes of the whole path: node and sphere reads, the queue payload written and read once per ray, hit
    records, and the framebuffer."""
    hits = st["hits_lambertian"] + st["hits_metal"] + st["hits_dielectric"]
    return 32 * st["inner_visits"] + 36 * st["leaf_tests"] + 16 * hits + 120 * st["rays"] + 15 * npix


def host_cores():
    """Host threads this process may use: its CPU affinity, capped by the share the box grants (OMP_NUM_THREADS)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline(ns_sample=200, ns_one_core=8):
    from oracle import MODE_REF_POOL, OracleScene

    threads = host_cores()
    osc = OracleScene.build(1, NX, NY, seed=SEED)
    state = osc.rng_after()
    t0 = time.perf_counter()
    _, _, ctr1 = osc.render(NX, NY, ns_one_core, seed=SEED, mode=MODE_REF_POOL, threads=1, state=state, pow5_mode=1)
    dt1 = time.perf_counter() - t0
    ns_sample = max(8, min(NS, 30 * threads))  # 15-25 s whatever the core count
    t0 = time.perf_counter()
    _, _, ctr = osc.render(NX, NY, ns_sample, seed=SEED, mode=MODE_REF_POOL, threads=threads, state=state, pow5_mode=1)
    dt = time.perf_counter() - t0
    return {"value": round(ctr["rays"] / dt / 1e6, 2), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "nproc": os.cpu_count(), "cores_note": "threads = this process's CPU share (affinity, capped by OMP_NUM_THREADS); nproc = "
                                                  "logical CPUs of the whole machine, most of which belong to other GPUs' jobs",
            "one_core": {"value": round(ctr1["rays"] / dt1 / 1e6, 2), "unit": "Mrays/s", "sample": f"{ns_one_core} spp, {ctr1['rays']} rays in {dt1:.2f} s"},
            "sample": f"chap12 {NX}x{NY}, {ns_sample} of {NS} spp, depth {MAX_DEPTH}, seed {SEED}: {ctr['rays']} rays in {dt:.2f} s; "
                      "C restatement of the reference CPU path (recursive trace, per-thread jumped Xoshiro128+, pixels claimed "
                      "dynamically, libm powf), scene build excluded; the Rust reference itself cannot be built here"}


def load_profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def other_configs(ctx, rrt_amd, torch, dev):
    """One timed step each of the two remaining single-GPU configurations (after one small untimed render that sizes the
    buffers): the moving-sphere scene at the headline size, and the 1 M-sphere scene whose BVH is walked in HBM."""
    out = []
    for name, kind, nx, ny, ns, half in (("book2chap2 moving spheres, 1200x800, 500 spp", rrt_amd.SCENE_BOOK2CHAP2, 1200, 800, 500, 11),
                                         ("1 M random spheres (2 M BVH nodes walked in HBM, two children per 64-byte record), 1920x1080, 256 spp", rrt_amd.SCENE_RANDOM_SPHERES, 1920, 1080, 256, 500)):
        t0 = time.perf_counter()
        scene = rrt_amd.HostScene(kind, nx, ny, seed=SEED, half=half)   # spheres on the host; the tree on the GPU for big scenes (Scene::build_bvh)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        ctx.upload_host_scene(scene)
        t_upload = time.perf_counter() - t0
        rgb = torch.zeros(ny * nx * 3, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.render_device(rrt_amd.Context.params(nx, ny, 4, seed=SEED), rgb.data_ptr(), 0)
        cst = ctx.render_device(rrt_amd.Context.params(nx, ny, 8, seed=SEED, count_work=True), rgb.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ctx.render_device(rrt_amd.Context.params(nx, ny, ns, seed=SEED), rgb.data_ptr(), 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the whole BVH is walked in an LDS copy when its compact image fits 16-bit offsets: node reads are then not memory traffic
        in_lds = (scene.n_spheres - 1) * 80 + 4 + scene.n_spheres * 52 <= 0xFFF0
        hits = cst["hits_lambertian"] + cst["hits_metal"] + cst["hits_dielectric"]
        per_ray = ((16 * hits + 120 * cst["rays"]) if in_lds else job_bytes(cst, 0)) / cst["rays"]
        gbs = per_ray * st["rays"] / dt / 1e9
        entry = {"workload": name, "spheres": scene.n_spheres, "bvh_nodes": scene.n_nodes, "ms_per_step": round(dt * 1e3, 2),
                 "value": round(st["rays"] / dt / 1e6, 1), "unit": "Mrays/s", "rays_per_step": st["rays"],
                 "setup_s": {"scene_and_bvh_build": round(t_build, 3), "layout_and_upload": round(t_upload, 3),
                             "note": "before the timed step: the scene builder (sphere list on the host, Bvh::build on the GPU for scenes of "
                                     "32 Ki spheres or more) and rrt_upload_scene (validation, traversal layout, copies to the device)"},
                 "per_ray": {"inner_visits": round(cst["inner_visits"] / cst["rays"], 2), "leaf_tests": round(cst["leaf_tests"] / cst["rays"], 2),
                             "algorithmic_bytes": round(per_ray, 1)}}
        if in_lds:
            entry["hbm"] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "note": "algorithmic bytes (16 B per shaded hit, 120 B of queue payload per ray; the BVH is walked in LDS) x rays / wall time"}
        else:
            # Most node reads are served by the LDS copy of the top of the tree, L1, L2 and the Infinity Cache: the algorithmic
            # figure is the rate at which the walk consumes node data; the HBM figure comes from measured bytes per ray.
            entry["node_data"] = {"achieved": round(gbs, 1), "unit": "GB/s",
                                  "note": "algorithmic bytes (32 B per box test, 36 B per sphere test, 16 B per shaded hit, 120 B of queue payload per "
                                          "ray; per-ray counts from a counting pass at 8 spp) x rays / wall time: what the walk consumes, from whichever "
                                          "level of the memory system serves it"}
            tr = load_profile_json("cfg5_traffic.json")
            if tr:
                mb = tr["bytes_per_ray"]["total"] * st["rays"] / dt / 1e9
                entry["hbm"] = {"achieved": round(mb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(mb / HBM_PEAK_GBS, 4),
                                "bytes_per_ray": tr["bytes_per_ray"], "extend": tr.get("extend"), "shade": tr.get("shade"),
                                "note": "measured HBM bytes per ray of extend + shade (FETCH_SIZE x 2 + WRITE_SIZE, " + tr["source"] + ") x this run's rays / wall time"}
        out.append(entry)
        scene.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of the whole job (default: 500 at N = 1, 1024 at N > 1)")
    ap.add_argument("--shard", choices=("tile", "sample"), default="tile", help="how N > 1 ranks share the image")
    ap.add_argument("--slots", type=int, default=0, help="path slots in flight (default: the library's, 3.5 Mi)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--workload", choices=("auto", "1200x800", "4k"), default="auto", help="auto: 1200x800x500 at N = 1, 3840x2160x1024 at N > 1")
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
    one_device = os.environ.get("RRT_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0  # rehearsal: every rank shares GPU 0 (not a scaling measurement)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    big = args.workload == "4k" or (args.workload == "auto" and world > 1)
    nx, ny, ns_total = (NX4, NY4, NS4) if big else (NX, NY, NS)
    if args.spp:
        ns_total = args.spp
    mode = rrt_amd.SHARD_NONE if world == 1 else (rrt_amd.SHARD_TILE if args.shard == "tile" else rrt_amd.SHARD_SAMPLE)

    ctx = rrt_amd.Context(local_rank)
    scene = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, nx, ny, seed=SEED)
    ctx.upload_host_scene(scene)

    # TILE shards and single-GPU runs produce RGB8 only, like tracescene (lib.rs:81); SAMPLE shards need the f32 sums
    want_accum = mode == rrt_amd.SHARD_SAMPLE
    accum = torch.zeros(ny * nx * 3 if want_accum else 3, dtype=torch.float32, device=dev)
    rgb8 = torch.zeros(ny * nx * 3, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    flags = {"count": False, "timed": False, "slots": args.slots}

    TIME_STRIDE = 4  # HIP events around every 4th extend launch: an even sample, a quarter of the event traffic

    def render_fn(p):
        p.count_work = int(flags["count"])
        p.reserved[0] = TIME_STRIDE if flags["timed"] else 0
        p.path_slots = flags.get("slots", 0)
        st = ctx.render_device(p, 0 if want_accum else rgb8.data_ptr(), accum.data_ptr() if want_accum else 0)
        return rgb8, (accum if want_accum else None), st

    def finalize(acc, ns):
        torch.cuda.synchronize()
        ctx.finalize(acc.data_ptr(), nx * ny, ns, rgb8.data_ptr())
        return rgb8

    def step(count=False, timed=False):
        flags["count"], flags["timed"] = count, timed
        _, _, st = multi.render_sharded(render_fn, nx, ny, ns_total, SEED, mode, rank, world, dst=0, finalize=finalize,
                                        max_depth=MAX_DEPTH, want_accum=False)
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
    rays_per_step = rays_total // max(1, args.steps)
    ray_count_check = None
    if big and ns_total == NS4 and RAYS_4K_1024 is not None:
        ray_count_check = rays_per_step == RAYS_4K_1024
        if rank == 0 and not ray_count_check:
            raise SystemExit(f"the ranks' ray counts add up to {rays_per_step}, the one-GPU render counts {RAYS_4K_1024}")

    roofline = None
    if not args.no_roofline:
        cst = step(count=True)  # every rank takes part (the pass ends in the same combine); rank 0 reports its own counters
        # one more step with the path pool in one part on one stream: extend launches that overlap nothing
        # (with 4 Mi path slots: 1024 extend workgroups, two full generations on the chip's 512 places; the default pool of 3.5 Mi
        # is sized for two parts on two streams and would leave the second generation a quarter empty here)
        os.environ["RRT_STREAMS"] = "1"
        flags["slots"] = 1 << 22
        ist = step(timed=True)
        flags["slots"] = args.slots
        del os.environ["RRT_STREAMS"]
        if rank == 0 and ext_ms > 0:
            flops = extend_flops(cst) * args.steps
            byts = extend_bytes(cst) * args.steps
            byts_survey = extend_bytes_survey(cst) * args.steps
            sec = ext_ms * 1e-3 * TIME_STRIDE          # the timed launches are every TIME_STRIDE-th one
            ext_launches *= TIME_STRIDE
            traffic = load_profile_json("extend_traffic.json")
            counters = load_profile_json("counters.json")
            roofline = {
                "kernel": "k_extend_lds", "bound": "valu_fp32", "achieved": round(flops / sec / 1e12, 3), "peak": FP32_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(flops / sec / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                "traffic": traffic.get("bytes_per_launch") if traffic else None,
                "launches": ext_launches, "timed_launches": ext_launches // TIME_STRIDE, "avg_launch_ms": round(ext_ms * TIME_STRIDE / max(1, ext_launches), 4),
                "flop_per_launch": round(flops / max(1, ext_launches)),
                "note": "extend launches of the parts of the path pool run on streams of their own and overlap each other and the "
                        "shade kernel, so the summed launch durations exceed wall time and `achieved` is a lower bound; `isolated` is "
                        "the same kernel on the whole pool with nothing beside it (one extra step on one stream); the single launch that "
                        "finishes a batch once its last path is dealt (k_finish) is not among the timed launches: it carries under 0.1 % "
                        "of this workload's rays, which the FLOP count does include",
                "isolated": {"achieved": round(extend_flops(cst) / (ist["extend_ms"] * TIME_STRIDE * 1e-3) / 1e12, 3),
                             "frac": round(extend_flops(cst) / (ist["extend_ms"] * TIME_STRIDE * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                             "launches": ist["extend_launches"] * TIME_STRIDE, "avg_launch_ms": round(ist["extend_ms"] / max(1, ist["extend_launches"]), 4),
                             "step_ms": round(ist["gpu_ms"], 2), "path_slots": 1 << 22},
                "whole_job": {"achieved": round(job_flops(cst) * args.steps / elapsed / 1e12, 3), "unit": "TFLOP/s",
                              "frac": round(job_flops(cst) * args.steps / elapsed / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                              "note": "all algorithmic FLOPs of a step (every kernel) over wall time, per GPU (rank 0's share)"},
                "hbm": {"achieved": round(byts / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / sec / 1e9 / HBM_PEAK_GBS, 4),
                        "bytes_per_ray": round(byts / args.steps / cst["rays"], 1),
                        "survey_pricing": {"bytes_per_ray": 60, "achieved": round(byts_survey / sec / 1e9, 1), "frac": round(byts_survey / sec / 1e9 / HBM_PEAK_GBS, 4)},
                        "note": "extend kernel alone, over its summed (overlapping) launch durations; bytes as the kernel moves them "
                                "(12 B origin + 16 B direction in, 8 B hit record out, 12 B of list traffic; measured: counters.bytes_per_ray.extend), the survey's "
                                "60-byte payload beside it"},
                "per_ray": {"inner_visits": round(cst["inner_visits"] / cst["rays"], 2), "leaf_tests": round(cst["leaf_tests"] / cst["rays"], 2),
                            "rays_per_path": round(cst["rays"] / cst["paths"], 3)},
            }
            if counters:
                # figures from the kept rocprofv3 --pmc summaries (profiles/counters.json says which files and how derived)
                for key in ("valu_busy", "lane_util", "lds_conflict_frac"):
                    if key in counters.get("extend", {}):
                        roofline[key] = counters["extend"][key]
                roofline["shade"] = {key: counters["shade"][key] for key in ("valu_busy", "lane_util", "lds_conflict_frac") if key in counters.get("shade", {})}
                roofline["counters"] = counters
                bpr = counters.get("bytes_per_ray")
                if bpr:
                    step_s = elapsed / args.steps
                    rays_rank0 = cst["rays"]
                    roofline["whole_step_hbm"] = {
                        "bytes_per_ray": bpr, "algorithmic_bytes_per_ray": round((job_bytes(cst, nx * ny) - 32 * cst["inner_visits"] - 36 * cst["leaf_tests"]) / cst["rays"], 1),
                        "achieved": round(bpr["total"] * rays_rank0 / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(bpr["total"] * rays_rank0 / step_s / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "measured HBM bytes per ray of extend + shade (FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes) x this "
                                "run's rays / this run's step time; algorithmic = 120 B of queue payload per ray + 16 B per shaded hit + "
                                "15 B per pixel (the BVH is walked in LDS)"}

    extra = None
    if rank == 0 and world == 1 and not big and not args.no_other_configs:
        extra = other_configs(ctx, rrt_amd, torch, dev)

    if rank == 0:
        n_sph, n_nodes = scene.n_spheres, scene.n_nodes
        if world == 1:
            sharding = "none"
        elif mode == rrt_amd.SHARD_TILE:
            sharding = f"32x8 tiles dealt round-robin over {dist.get_world_size()} ranks; RGB8 of owned tiles gathered on rank 0 over {dist.get_backend()}"
        else:
            sharding = f"sample k -> rank k mod {dist.get_world_size()}; f32 sums reduced onto rank 0 over {dist.get_backend()}, quantised there"
        out = {
            "metric": "Mrays/s (paths x bounces), chap12 scene", "value": round(rays_total / elapsed / 1e6, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"chap12 random-spheres scene seed {SEED} ({n_sph} spheres, {n_nodes} BVH nodes), {nx}x{ny}, "
                                   f"{ns_total} spp in total, max depth {MAX_DEPTH}",
                       "sharding": sharding, "rays_per_step": rays_per_step,
                       "backend": dist.get_backend() if world > 1 else None, "ranks": dist.get_world_size() if world > 1 else 1,
                       "devices_shared": bool(one_device and world > 1), "ray_count_matches_one_gpu": ray_count_check},
            "device": ctx.device_info()["name"],
        }
        if roofline is not None:
            out["roofline"] = roofline
        if extra is not None:
            out["other_configs"] = extra
        if world == 1 and not big and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
