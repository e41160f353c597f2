#!/usr/bin/env python3
"""Renders the reference scenes on the GPU and writes PNG + PPM previews: python tools/render_png.py outdir"""
import os, sys, zlib, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rrt_amd

def write_png(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
ctx = rrt_amd.Context(0)
for name, kind, nx, ny, ns, half in [("chap11", 0, 400, 225, 100, 11), ("chap12", 1, 600, 400, 100, 11), ("book2chap2", 2, 600, 400, 100, 11), ("spheres1m", 3, 640, 360, 32, 500)]:
    hs = rrt_amd.HostScene(kind, nx, ny, seed=0, half=half)
    ctx.upload_host_scene(hs)
    rgb, _, st = ctx.render(rrt_amd.Context.params(nx, ny, ns, seed=0), want_accum=False)
    write_png(os.path.join(out, name + ".png"), rgb)
    print(name, st["rays"], round(st["gpu_ms"], 2), "ms", round(st["rays"] / st["gpu_ms"] / 1e3, 1), "Mrays/s")
