import os,sys
sys.path.insert(0, os.getcwd())
import rrt_amd
ctx=rrt_amd.Context(0)
nx,ny=1200,800
hs=rrt_amd.HostScene(rrt_amd.SCENE_CHAP12,nx,ny,seed=0); ctx.upload_host_scene(hs)
p=rrt_amd.Context.params(nx,ny,40,seed=0,count_work=True,retire_min=40)
_,_,st=ctx.render(p,want_accum=False)
d=st['debug']; rays=st['rays']
print(f"rays {rays} paths {st['paths']}; rays taken at fresh positions {d[5]} ; rays taken while the wave holds fresh only {d[4]}")
print(f"descent passes/64 rays {d[0]/rays*64:.2f} lanes/pass {d[1]/max(1,d[0]):.1f}; fresh-only phase: {d[2]} passes ({d[2]/max(1,d[4])*64:.2f} per 64 of its rays) lanes/pass {d[3]/max(1,d[2]):.1f}; rest: {(d[0]-d[2])/(rays-d[4])*64:.2f} per 64 rays, lanes/pass {(d[1]-d[3])/max(1,d[0]-d[2]):.1f}")
