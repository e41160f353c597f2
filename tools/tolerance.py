#!/usr/bin/env python3
"""States the tolerance between the GPU path's random-stream assignment and the reference's (CPU only, oracle only).

The GPU path (== the oracle's per-sample mode with pow5 as a product chain, checked bit for bit elsewhere) cannot
replay the reference's streams: rayon decides which thread's generator a pixel consumes (threadpool.rs:11-27,
lib.rs:90-92).  The reference is reproducible only with one thread, where it is one serial stream over the pixels
(oracle MODE_REF_SERIAL; libm powf as materials.rs:68).  Both are unbiased estimates of the same image, so they differ by
Monte-Carlo noise.  This script measures that difference and the noise floor it has to be compared with:

    A  per-sample streams, pow5 product chain      (what the GPU computes)
    B  reference-serial stream, libm powf, seed s  (the reference on one thread)
    C  reference-serial stream, libm powf, seed s' (the reference again, another seed)

and prints per-channel max-abs / RMSE of the normalised radiance (sum / ns, range 0..1) and the RGB8 statistics for A-B
against B-C.  A-B within B-C's range says the stream assignment and the powf deviation change nothing but the noise.

    python tools/tolerance.py [--out profiles/r02_tolerance.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import MODE_PER_SAMPLE, MODE_REF_SERIAL, OracleScene  # noqa: E402


def stats(a, b, rgb_a, rgb_b, ns):
    d = (a.astype(np.float64) - b.astype(np.float64)) / ns
    diff8 = np.abs(rgb_a.astype(np.int16) - rgb_b.astype(np.int16))
    return {"max_abs": [float(np.abs(d[..., c]).max()) for c in range(3)], "rmse": [float(np.sqrt((d[..., c] ** 2).mean())) for c in range(3)],
            "mean_signed": [float(d[..., c].mean()) for c in range(3)],
            "rgb8_max_abs": int(diff8.max()), "rgb8_rmse": float(np.sqrt((diff8.astype(np.float64) ** 2).mean())),
            "rgb8_differing_fraction": float((diff8 != 0).any(axis=-1).mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    cases = [("cfg1 chap11 400x225x100 (in full)", 0, 400, 225, 100), ("cfg2 chap12 1200x800 at 16 of 500 spp", 1, 1200, 800, 16),
             ("cfg2 chap12 300x200x500 (full sample count, reduced size)", 1, 300, 200, 500)]
    out = []
    for name, kind, nx, ny, ns in cases:
        osc = OracleScene.build(kind, nx, ny, seed=0)
        state = osc.rng_after()
        rgb_a, acc_a, _ = osc.render(nx, ny, ns, seed=0, mode=MODE_PER_SAMPLE, threads=args.threads, pow5_mode=0)
        rgb_a1, acc_a1, _ = osc.render(nx, ny, ns, seed=0, mode=MODE_PER_SAMPLE, threads=args.threads, pow5_mode=1)
        rgb_b, acc_b, _ = osc.render(nx, ny, ns, seed=0, mode=MODE_REF_SERIAL, state=state, pow5_mode=1)
        other = OracleScene.build(kind, nx, ny, seed=0)
        rgb_c, acc_c, _ = other.render(nx, ny, ns, seed=12345, mode=MODE_REF_SERIAL, pow5_mode=1)   # a fresh base generator
        row = {"case": name, "nx": nx, "ny": ny, "ns": ns,
               "gpu_streams_vs_reference_serial": stats(acc_a, acc_b, rgb_a, rgb_b, ns),
               "noise_floor_reference_serial_two_seeds": stats(acc_b, acc_c, rgb_b, rgb_c, ns),
               "pow5_chain_vs_libm_powf_same_streams": stats(acc_a, acc_a1, rgb_a, rgb_a1, ns)}
        out.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": __doc__.split("\n\n")[0], "cases": out}, f, indent=1)


if __name__ == "__main__":
    main()
