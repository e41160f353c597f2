#!/usr/bin/env python3
"""profiles/cfg5_traffic.json from the kept counter passes of the 1 M-sphere scene (tools/cfg5_profile.sh):
    python tools/make_cfg5_json.py <tag>     e.g. r02_g  (reads profiles/<tag>_cfg5_{run,pmc_fetch,pmc_write,pmc_sq,kernel_stats}.*)
Bytes per ray = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / rays, per kernel; FETCH_SIZE doubled per the gfx950 rule for
16-byte-per-lane loads, as for the headline configuration.  The profiled command renders the scene twice at 32 spp."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_counters_json import parse

tag = sys.argv[1]
P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_cfg5_{n}")
m = re.search(r"gpu_ms\s+([0-9.]+)\s+Mrays/s\s+([0-9.]+)", open(P("run.txt")).read())
rays = float(m.group(1)) * 1e-3 * float(m.group(2)) * 1e6 * 2  # two renders under the profiler
fetch, write, sq = parse(P("pmc_fetch.txt")), parse(P("pmc_write.txt")), parse(P("pmc_sq.txt"))
out = {"source": f"profiles/{tag}_cfg5_pmc_*.txt: rocprofv3 --pmc, separate passes, tools/tune.py --scene 3 --half 500 --nx 1920 --ny 1080 --spp 32 --rounds 1",
       "rays_profiled": round(rays), "bytes_per_ray": {}}
total = 0.0
for name, key in (("extend", "k_extend"), ("shade", "k_shade")):
    kf = next(k for k in fetch if k.startswith(key))
    kw = next(k for k in write if k.startswith(key))
    b = (2.0 * fetch[kf]["FETCH_SIZE"] + write[kw]["WRITE_SIZE"]) * 1024.0 / rays
    out["bytes_per_ray"][name] = round(b, 1)
    total += b
    if name == "extend":
        ks = next(k for k in sq if k.startswith(key))
        out["extend"] = {"kernel": kf, "l2_hit_rate": round(write[kw]["TCC_HIT_sum"] / (write[kw]["TCC_HIT_sum"] + write[kw]["TCC_MISS_sum"]), 3),
                         "valu_busy": round(4.0 * sq[ks]["SQ_ACTIVE_INST_VALU"] / 1024.0 / (sq[ks]["GRBM_GUI_ACTIVE"] / 8.0), 3),
                         "valu_insts_per_ray": round(sq[ks]["SQ_INSTS_VALU"] / rays, 1),
                         "avg_l2_read_latency_cycles": round(sq[ks]["TCP_TCC_READ_REQ_LATENCY_sum"] / sq[ks]["TCP_TCC_READ_REQ_sum"])}
        if "SQ_THREAD_CYCLES_VALU" in sq[ks]:
            out["extend"]["lane_util"] = round(sq[ks]["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq[ks]["SQ_INSTS_VALU"]), 3)
        passes = os.path.join(ROOT, "profiles", f"{tag}_cfg5_extend_passes.txt")
        if os.path.exists(passes):  # tools/extend_passes.py 3 1920 1080 8 500 (counting build)
            t = open(passes).read()
            g = lambda pat: [float(x) for x in re.search(pat, t).groups()]
            d, dl = g(r"descent passes per 64 rays ([0-9.]+) lanes/pass ([0-9.]+)")
            sp, sl = g(r"sphere passes per 64 rays ([0-9.]+) lanes/pass ([0-9.]+)")
            h, hl = g(r"hand-outs per 64 rays ([0-9.]+) lanes/hand-out ([0-9.]+)")
            out["extend"]["passes_per_64_rays"] = {"descent": d, "lanes_per_descent_pass": dl, "sphere": sp, "lanes_per_sphere_pass": sl, "hand_outs": h, "lanes_per_hand_out": hl,
                                                   "note": "a descent pass of the two-children walk deals with both children of each lane's node; leaf children are tested inside it"}
    else:
        ks = next(k for k in sq if k.startswith(key))
        out["shade"] = {"kernel": kf, "valu_busy": round(4.0 * sq[ks]["SQ_ACTIVE_INST_VALU"] / 1024.0 / (sq[ks]["GRBM_GUI_ACTIVE"] / 8.0), 3),
                        "valu_insts_per_ray": round(sq[ks]["SQ_INSTS_VALU"] / rays, 1)}
        if "SQ_THREAD_CYCLES_VALU" in sq[ks]:
            out["shade"]["lane_util"] = round(sq[ks]["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq[ks]["SQ_INSTS_VALU"]), 3)
out["bytes_per_ray"]["total"] = round(total, 1)
json.dump(out, open(os.path.join(ROOT, "profiles", "cfg5_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
