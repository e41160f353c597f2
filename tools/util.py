#!/usr/bin/env python3
"""Lane utilisation of the extend kernel's phases (counting build): python tools/util.py [retire_min leaf_min]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
ctx = rrt_amd.Context(0)
NX, NY = int(os.environ.get('NX', '1200')), int(os.environ.get('NY', '800'))
hs = rrt_amd.HostScene(int(os.environ.get('SCENE', str(rrt_amd.SCENE_CHAP12))), NX, NY, seed=0, half=int(os.environ.get('HALF', '11')))
ctx.upload_host_scene(hs)
args = [int(a) for a in sys.argv[1:]] or [64, 8]
for T, L in zip(args[0::2], args[1::2]):
    _, _, st = ctx.render(rrt_amd.Context.params(NX, NY, int(os.environ.get('SPP', '40')), seed=0, count_work=True, retire_min=T, leaf_min=L), want_accum=False)
    d = st["debug"]; R = st["rays"]
    print(f"T={T} L={L}: rays {R}  inner/ray {st['inner_visits']/R:.2f} leaf/ray {st['leaf_tests']/R:.2f} | descent passes/ray {d[0]*64/R:.2f} "
          f"(lanes/pass {d[1]/max(1,d[0]):.1f}) | leaf passes/ray {d[2]*64/R:.2f} (lanes/pass {d[3]/max(1,d[2]):.1f}) | retire passes per 64 rays {d[4]*64/R:.2f} (lanes/pass {d[5]/max(1,d[4]):.1f})")
