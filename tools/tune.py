#!/usr/bin/env python3
"""A/B timing of render-parameter variants, interleaved rounds in one process.
   python tools/tune.py [--spp 500] [--rounds 3] name=key:val,key:val ...   (keys: path_slots, batch_samples, retire_min, force_hbm_bvh; env_NAME:value sets an environment variable for that variant)"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rrt_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=500)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--scene", type=int, default=rrt_amd.SCENE_CHAP12)
    ap.add_argument("--nx", type=int, default=1200)
    ap.add_argument("--ny", type=int, default=800)
    ap.add_argument("--half", type=int, default=11)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    variants = {"default": {}}
    for v in a.variants:
        name, _, kv = v.partition("=")
        variants[name] = {k: int(x) for k, x in (item.split(":") for item in kv.split(",") if item)}
    ctx = rrt_amd.Context(0)
    hs = rrt_amd.HostScene(a.scene, a.nx, a.ny, seed=0, half=a.half)
    ctx.upload_host_scene(hs)
    dev = torch.device("cuda", 0)
    acc = torch.zeros(a.nx * a.ny * 3, dtype=torch.float32, device=dev)
    rgb = torch.zeros(a.nx * a.ny * 3, dtype=torch.uint8, device=dev)
    res = {k: [] for k in variants}
    for r in range(a.rounds + 1):
        for name, kw in variants.items():
            kw = dict(kw)
            for key in [k for k in kw if k.startswith("env_")]:  # env_NAME:value sets an environment variable the library reads per render (-1: unset)
                val = kw.pop(key)
                if val < 0:
                    os.environ.pop(key[4:], None)
                else:
                    os.environ[key[4:]] = str(val)
            p = rrt_amd.Context.params(a.nx, a.ny, a.spp, seed=0, time_kernels=True, **kw)
            st = ctx.render_device(p, rgb.data_ptr(), acc.data_ptr())
            if r > 0:
                res[name].append(st)
    for name, sts in res.items():
        g = statistics.median(s["gpu_ms"] for s in sts)
        e = statistics.median(s["extend_ms"] for s in sts)
        n = sts[0]["extend_launches"]
        rays = sts[0]["rays"]
        print(f"{name:28s} gpu_ms {g:8.2f}  Mrays/s {rays / g / 1e3:8.1f}  extend_ms {e:8.2f} ({e / max(1, n) * 1e3:7.1f} us x {n})  rest_ms {g - e:7.2f}  checksum {int(rgb.sum().item())}")


if __name__ == "__main__":
    main()
