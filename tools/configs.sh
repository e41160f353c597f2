# The other configurations on one GPU (no per-launch timing): book2chap2 1200x800x500, chap12 3840x2160 at 128 of 1024 spp, 1M spheres 1920x1080x256
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, rrt_amd
ctx = rrt_amd.Context(0)
def run(name, kind, nx, ny, spp, half=11):
    hs = rrt_amd.HostScene(kind, nx, ny, seed=0, half=half)
    ctx.upload_host_scene(hs)
    acc = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda:0"); rgb = torch.zeros(nx * ny * 3, dtype=torch.uint8, device="cuda:0")
    best = None
    for _ in range(3):
        st = ctx.render_device(rrt_amd.Context.params(nx, ny, spp, seed=0), rgb.data_ptr(), acc.data_ptr())
        best = st if best is None or st["gpu_ms"] < best["gpu_ms"] else best
    print(f"{name}: {best['gpu_ms']:.1f} ms, {best['rays'] / best['gpu_ms'] / 1e3:.0f} Mrays/s, {best['rays']} rays", flush=True)
run("cfg2 chap12 1200x800x500", rrt_amd.SCENE_CHAP12, 1200, 800, 500)
run("cfg3 book2chap2 1200x800x500", rrt_amd.SCENE_BOOK2CHAP2, 1200, 800, 500)
run("cfg4 chap12 3840x2160x128 (of 1024)", rrt_amd.SCENE_CHAP12, 3840, 2160, 128)
run("cfg5 1M spheres 1920x1080x256", rrt_amd.SCENE_RANDOM_SPHERES, 1920, 1080, 256, half=500)
PY
