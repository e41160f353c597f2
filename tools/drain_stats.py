"""Descent passes of the extend kernel before and after a workgroup's pool of rays is used up (headline scene).
Needs the counting build with -DRRT_DRAIN_STATS:  bash tools/mkvariant.sh drainstats -DRRT_DRAIN_STATS, then run with that
library in place of rrt_amd/librrt_amd.so."""
import os,sys
sys.path.insert(0, os.getcwd())
import rrt_amd
ctx=rrt_amd.Context(0)
nx,ny=1200,800
hs=rrt_amd.HostScene(rrt_amd.SCENE_CHAP12,nx,ny,seed=0); ctx.upload_host_scene(hs)
for spp in (40, 200):
    p=rrt_amd.Context.params(nx,ny,spp,seed=0,count_work=True,retire_min=40)
    _,_,st=ctx.render(p,want_accum=False)
    d=st['debug']; rays=st['rays']
    print(f"spp {spp}: descent passes/64 rays {d[0]/rays*64:.2f} lanes/pass {d[1]/max(1,d[0]):.1f}; of which with the pool used up: {d[2]/rays*64:.2f} passes, lanes/pass {d[3]/max(1,d[2]):.1f}; before: {(d[0]-d[2])/rays*64:.2f} passes lanes/pass {(d[1]-d[3])/max(1,d[0]-d[2]):.1f}")
