import os, sys
sys.path.insert(0, os.getcwd())
import rrt_amd
ctx = rrt_amd.Context(0)
hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, 1200, 800, seed=0)
ctx.upload_host_scene(hs)
_, _, st = ctx.render(rrt_amd.Context.params(1200, 800, 100, seed=0, count_work=True), want_accum=False)
d = st["debug"]; r = st["rays"]
print("rays", r, "inner/ray %.2f leaf/ray %.2f" % (st["inner_visits"]/r, st["leaf_tests"]/r))
print("descent passes %d lanes/pass %.1f | leaf passes %d lanes/pass %.1f | retire passes %d lanes/pass %.1f" % (d[0], d[1]/d[0], d[2], d[3]/d[2], d[4], d[5]/d[4]))
print("per 64 rays: descent passes %.1f leaf passes %.1f retire passes %.2f" % (d[0]*64/r, d[2]*64/r, d[4]*64/r))
