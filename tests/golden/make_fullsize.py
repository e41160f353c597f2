#!/usr/bin/env python3
"""Writes tests/golden/fullsize.json: digests of whole oracle renders at the sizes of every benchmark configuration
(chap12 1200x800x500, book2chap2 1200x800x500, chap12 3840x2160x1024 and the 1000x1000-cell random-sphere scene at
1920x1080x256; scene seed 0, render seed 0, depth 50), per-sample mode.

A full-size image is too big for a fixture, so what is kept is the SHA-256 of the RGB8 bytes and of the f32 sums, the
CRC-32 of every row of f32 sums (to point at the rows that differ when a digest does), and the work counters.  The two
1200x800 cases take about a minute on 8 cores, the 4K and the million-sphere case ten to fifteen minutes each.

    make_fullsize.py [case ...]     renders the named cases (default: all) and keeps the other entries of the file"""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import CTR_NAMES, MODE_PER_SAMPLE, OracleScene  # noqa: E402

# name: scene kind, nx, ny, ns, half (cells per half axis of the random-sphere scene; 11 is the book's 22 x 22 grid)
CASES = {"chap12_1200x800x500": (1, 1200, 800, 500, 11), "book2chap2_1200x800x500": (2, 1200, 800, 500, 11),
         "chap12_3840x2160x1024": (1, 3840, 2160, 1024, 11), "spheres1m_1920x1080x256": (3, 1920, 1080, 256, 500)}


def digest(rgb, acc):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    acc = np.ascontiguousarray(acc, np.float32)
    return {"sha256_rgb": hashlib.sha256(rgb.tobytes()).hexdigest(), "sha256_acc": hashlib.sha256(acc.tobytes()).hexdigest(),
            "row_crc32": [zlib.crc32(acc[y].tobytes()) for y in range(acc.shape[0])]}


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fullsize.json")
    out = {}
    if os.path.exists(path):
        with open(path) as f:
            out = json.load(f)
    for name in (sys.argv[1:] or list(CASES)):
        kind, nx, ny, ns, half = CASES[name]
        osc = OracleScene.build(kind, nx, ny, seed=0, half=half)
        rgb, acc, ctr = osc.render(nx, ny, ns, seed=0, mode=MODE_PER_SAMPLE, max_depth=50, threads=int(os.environ.get("THREADS", os.cpu_count() or 1)))
        d = digest(rgb.reshape(ny, nx, 3), acc.reshape(ny, nx, 3))
        d.update({"kind": kind, "nx": nx, "ny": ny, "ns": ns, "half": half, "scene_seed": 0, "seed": 0, "max_depth": 50,
                  "counters": {k: int(ctr[k]) for k in CTR_NAMES}})
        out[name] = d
        print(name, d["sha256_rgb"][:16], ctr["rays"], flush=True)
        with open(path, "w") as f:
            json.dump(out, f)
