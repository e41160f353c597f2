#!/usr/bin/env python3
"""Writes tests/golden/fullsize.json: digests of whole oracle renders at the sizes of the headline configurations
(chap12 1200x800x500 and book2chap2 1200x800x500, scene seed 0, render seed 0, depth 50), per-sample mode.

A full-size image is too big for a fixture, so what is kept is the SHA-256 of the RGB8 bytes and of the f32 sums, the
CRC-32 of every row of f32 sums (to point at the rows that differ when a digest does), and the work counters.  Takes
about a minute on 8 cores."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import CTR_NAMES, MODE_PER_SAMPLE, OracleScene  # noqa: E402

CASES = [("chap12_1200x800x500", 1, 1200, 800, 500), ("book2chap2_1200x800x500", 2, 1200, 800, 500)]


def digest(rgb, acc):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    acc = np.ascontiguousarray(acc, np.float32)
    return {"sha256_rgb": hashlib.sha256(rgb.tobytes()).hexdigest(), "sha256_acc": hashlib.sha256(acc.tobytes()).hexdigest(),
            "row_crc32": [zlib.crc32(acc[y].tobytes()) for y in range(acc.shape[0])]}


if __name__ == "__main__":
    out = {}
    for name, kind, nx, ny, ns in CASES:
        osc = OracleScene.build(kind, nx, ny, seed=0)
        rgb, acc, ctr = osc.render(nx, ny, ns, seed=0, mode=MODE_PER_SAMPLE, max_depth=50, threads=os.cpu_count() or 1)
        d = digest(rgb.reshape(ny, nx, 3), acc.reshape(ny, nx, 3))
        d.update({"kind": kind, "nx": nx, "ny": ny, "ns": ns, "scene_seed": 0, "seed": 0, "max_depth": 50,
                  "counters": {k: int(ctr[k]) for k in CTR_NAMES}})
        out[name] = d
        print(name, d["sha256_rgb"][:16], ctr["rays"], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize.json"), "w") as f:
        json.dump(out, f)
