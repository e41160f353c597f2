#!/usr/bin/env python3
"""Builds profiles/counters.json and profiles/extend_traffic.json from kept rocprofv3 --pmc summaries.

    python tools/make_counters_json.py <tag>        e.g. r02_b  (reads profiles/<tag>_pmc_*.txt and profiles/<tag>_1stream_pmc_*.txt)

Definitions (MI355X guide units: SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs):
  valu_busy          4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8), per kernel, from the ONE-STREAM run, where a
                     kernel's launches overlap nothing and GRBM_GUI_ACTIVE is its own time.  An upper bound: an instruction that
                     issues in 2.3 cycles (v_fma/v_add/v_mul/v_sub_f32, v_add_u32, v_and/or/xor, v_mov) still counts one quad-cycle.
  lane_util          SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)
  lds_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  bytes_per_ray      (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / rays, per kernel, default (two-stream) run; FETCH_SIZE reports half the
                     bytes of a coalesced stream at every width these kernels use (4, 8, 12, 16 bytes per lane, and 16-byte loads
                     straight into LDS: factor 2.000 each, WRITE_SIZE 1.000; tools/micro/fetch_size.hip, profiles/r03_fetch_size.txt)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAYS_PER_STEP = 1382382741  # chap12 1200x800x500 seed 0 (tests/golden/fullsize.json)


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {"dispatches": int(m.group(2))}
            continue
        m = re.match(r"^\s+(\w+)\s+([0-9.e+]+)\s+per-dispatch", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


def pick(d, prefix):
    for k, v in d.items():
        if k.startswith(prefix) and "true" not in k.split("<")[1].split(",")[1] if "<" in k else k.startswith(prefix):
            return k, v
    raise KeyError(prefix)


def main():
    tag = sys.argv[1]
    P = lambda n: os.path.join(ROOT, "profiles", n)
    one_sq, one_sq2 = parse(P(f"{tag}_1stream_pmc_sq.txt")), parse(P(f"{tag}_1stream_pmc_sq2.txt"))
    fetch, write = parse(P(f"{tag}_pmc_fetch.txt")), parse(P(f"{tag}_pmc_write.txt"))
    multi_sq, multi_sq2 = parse(P(f"{tag}_pmc_sq.txt")), parse(P(f"{tag}_pmc_sq2.txt"))
    res = {"source": f"profiles/{tag}_pmc_*.txt (default run, two pool parts on two streams) and profiles/{tag}_1stream_pmc_*.txt "
                     "(RRT_STREAMS=1); rocprofv3 --pmc, separate passes, bench.py --steps 1 --warmup 0, chap12 1200x800x500",
           "definitions": __doc__.split("Definitions")[1].strip()}
    bpr = {}
    for name, key in (("extend", "k_extend_lds"), ("shade", "k_shade")):
        k1, a = pick(one_sq, key)
        _, b = pick(one_sq2, key)
        cycles = b["GRBM_GUI_ACTIVE"] / 8.0
        res[name] = {
            "kernel": k1, "valu_busy": round(4.0 * a["SQ_ACTIVE_INST_VALU"] / 1024.0 / cycles, 3),
            "lane_util": round(a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_INSTS_VALU"]), 3),
            "lds_conflict_frac": round(b["SQ_LDS_BANK_CONFLICT"] / max(1.0, b["SQ_LDS_IDX_ACTIVE"]), 3),
            "valu_insts_per_ray": round(a["SQ_INSTS_VALU"] / RAYS_PER_STEP, 2), "salu_insts_per_ray": round(a["SQ_INSTS_SALU"] / RAYS_PER_STEP, 2),
            "lds_insts_per_ray": round(a["SQ_INSTS_LDS"] / RAYS_PER_STEP, 2), "one_stream_ms_per_step": round(cycles / 2.4e9 * 1e3, 1)}
        _, f = pick(fetch, key)
        _, w = pick(write, key)
        bpr[name] = round((2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0 / RAYS_PER_STEP, 1)
        if name == "extend":
            per = (2.0 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0 / f["dispatches"]
            with open(P("extend_traffic.json"), "w") as fo:
                json.dump({"kernel": k1, "source": f"profiles/{tag}_pmc_fetch.txt + profiles/{tag}_pmc_write.txt", "fetch_size_kb_per_launch": round(f["FETCH_SIZE"] / f["dispatches"], 1),
                           "write_size_kb_per_launch": round(w["WRITE_SIZE"] / w["dispatches"], 1), "fetch_correction": 2.0, "bytes_per_launch": round(per),
                           "note": "FETCH_SIZE doubled (calibrated per access width: profiles/r03_fetch_size.txt); a launch is one half of the path pool (3.5 Mi slots by default)"}, fo)
    bpr["total"] = round(bpr["extend"] + bpr["shade"], 1)
    res["bytes_per_ray"] = bpr
    # the whole step as it runs by default: VALU quad-cycles of both kernels over the step's wall time
    ke, e = pick(multi_sq, "k_extend_lds")
    ks, s = pick(multi_sq, "k_shade")
    res["whole_step"] = {"valu_quad_cycles": e["SQ_ACTIVE_INST_VALU"] + s["SQ_ACTIVE_INST_VALU"],
                         "note": "4 x valu_quad_cycles / 1024 SIMDs / (ms_per_step x clock) is the step's VALU-busy upper bound; with the default run's "
                                 "overlapping launches a kernel's own GRBM_GUI_ACTIVE counts time it shares with the other kernel and cannot be used per kernel"}
    with open(P("counters.json"), "w") as fo:
        json.dump(res, fo, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
