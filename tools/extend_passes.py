import os,sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
# usage: python tools/extend_passes.py [kind nx ny spp half]   (default: the headline scene at 40 spp; "3 1920 1080 8 500": the 1 M-sphere scene)
# RETIRE="16 24 40": the hand-out thresholds to try; MAX_DEPTH=0: camera rays only (every path ends at its first hit)
kind, nx, ny, spp, half = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (rrt_amd.SCENE_CHAP12, 1200, 800, 40, 11)
ctx=rrt_amd.Context(0)
hs=rrt_amd.HostScene(kind,nx,ny,seed=0,half=half); ctx.upload_host_scene(hs)
for rm in (int(x) for x in os.environ.get("RETIRE", "40").split()):
    p=rrt_amd.Context.params(nx,ny,spp,seed=0,count_work=True,retire_min=rm,max_depth=int(os.environ.get("MAX_DEPTH","50")))
    _,_,st=ctx.render(p,want_accum=False)
    d=st['debug']; rays=st['rays']
    print(f"retire_min {rm}: rays {rays} inner/ray {st['inner_visits']/rays:.2f} leaf/ray {st['leaf_tests']/rays:.2f}")
    print(f"  descent passes per 64 rays {d[0]/rays*64:.2f} lanes/pass {d[1]/max(1,d[0]):.1f} (ideal passes {st['inner_visits']/rays:.2f})")
    print(f"  sphere passes per 64 rays {d[2]/rays*64:.2f} lanes/pass {d[3]/max(1,d[2]):.1f} (ideal {st['leaf_tests']/rays:.2f})")
    print(f"  hand-outs per 64 rays {d[4]/rays*64:.2f} lanes/hand-out {d[5]/max(1,d[4]):.1f}")
