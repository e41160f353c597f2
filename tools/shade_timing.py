#!/usr/bin/env python3
"""Per-stage cycle counts of the shade kernel (library built with -DRRT_SHADE_TIMING): python tools/shade_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
ctx = rrt_amd.Context(0)
hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, 1200, 800, seed=0)
ctx.upload_host_scene(hs)
_, _, st = ctx.render(rrt_amd.Context.params(1200, 800, int(os.environ.get('SPP', '100')), seed=0, count_work=True), want_accum=False)
d = st["debug"]
n = max(1, d[5])
print("waves", n, "cycles/wave: classify %.0f  sample %.0f  tiles %.0f  refill %.0f  list %.0f  total %.0f" % (d[0]/n, d[1]/n, d[2]/n, d[3]/n, d[4]/n, sum(d[:5])/n))
