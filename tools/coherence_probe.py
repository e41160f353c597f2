"""How much the extend kernel gains from rays that are ordered by where they start and where they point.
Secondary rays of the headline scene (first to third bounce, made on the CPU with the oracle: diffuse scatter off every hit)
go through rrt_intersect, which runs the render loop's extend kernel on them, in the order named on the command line:
  shuffle | octant | cell | cell_octant | global     (all but `global` sort inside runs of 4096 rays, one workgroup's pool)
Run under `rocprofv3 --kernel-trace --stats` and read the extend kernel's average duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rrt_amd
from oracle import OracleScene

order = sys.argv[1] if len(sys.argv) > 1 else "shuffle"
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # 1: the headline scene; 3: random spheres (1 M with half = 500)
half = int(sys.argv[4]) if len(sys.argv) > 4 else 11
extent = float(half) + 1.0
nx, ny = (1200, 800) if kind == 1 else (1920, 1080)
cache = f"/tmp/probe_rays_{kind}_{half}.npz"
if os.path.exists(cache):
    z = np.load(cache); org, d = z["org"], z["d"]
else:
    osc = OracleScene.build(kind, nx, ny, seed=0, half=half)
    rng = np.random.default_rng(1)
    N = 1 << 21
    idx = rng.integers(0, nx * ny, N).astype(np.uint32)
    rays, _ = osc.gen_primary(nx, ny, 0, idx, np.zeros(N, np.uint32))
    o, dd = rays[:, 0:3].copy(), rays[:, 3:6].copy()
    orgs, dirs = [], []
    for depth in range(3):
        t, obj, p, n, front = osc.hit_rays(o, dd)
        hit = obj != 0xFFFFFFFF
        p, n = p[hit], n[hit]
        u = np.zeros((len(p), 3), np.float32); k = 0
        while k < len(p):
            c = rng.uniform(-1, 1, (len(p), 3)).astype(np.float32); c = c[(c * c).sum(1) < 1]
            m = min(len(p) - k, len(c)); u[k:k + m] = c[:m]; k += m
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        o, dd = p.astype(np.float32), (n + u).astype(np.float32)
        orgs.append(o); dirs.append(dd)
    org, d = np.concatenate(orgs), np.concatenate(dirs)
    np.savez(cache, org=org, d=d)
n = len(org) // 4096 * 4096
rng = np.random.default_rng(2)
perm = rng.permutation(len(org))[:n] if order != "asmade" else np.arange(n)
org, d = org[perm], d[perm]
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
cx = np.clip(((org[:, 0] + extent) / (2 * extent) * cells).astype(np.int64), 0, cells - 1)
cz = np.clip(((org[:, 2] + extent) / (2 * extent) * cells).astype(np.int64), 0, cells - 1)
cell = cx * cells + cz
run = np.arange(n) // 4096
key = {"shuffle": None, "asmade": None, "octant": octant, "cell": cell, "cell_octant": cell * 8 + octant, "octant_cell": octant * cells * cells + cell, "global": cell * 8 + octant}[order]
if key is not None:
    if order != "global":
        key = run * (1 << 20) + key
    p = np.argsort(key, kind="stable")
    org, d = org[p], d[p]
ctx = rrt_amd.Context(0)
ctx.upload_host_scene(rrt_amd.HostScene(kind, nx, ny, seed=0, half=half))
for _ in range(4):
    t, o = ctx.intersect(org, d)
print(order, cells, kind, half, "rays", n, "hits", int((o != 0xFFFFFFFF).sum()), "checksum", int(t.view(np.uint32).astype(np.uint64).sum()))
