#!/usr/bin/env python3
"""Re-runs one case of tools/fuzz.py from its MISMATCH line and says where GPU and oracle differ.
   python tools/repro_case.py kind half nx ny ns depth slots batch parts hbm seed scene_seed merge delay no_finish [lib.so ...]
With library paths after the case, the case is rendered with each of them in turn (a fresh process per library)."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_case(a):
    import rrt_amd
    from oracle import OracleScene
    from tests.conftest import scene_arrays_from_oracle

    kind, half, nx, ny, ns, depth, slots, batch, parts, hbm, seed, scene_seed, merge, delay, no_finish = (int(x) for x in a[:15])
    os.environ["RRT_STREAMS"] = str(parts)
    if no_finish:
        os.environ["RRT_NO_FINISH"] = "1"
    os.environ["RRT_FINISH_MERGE"] = str(merge)
    os.environ["RRT_FINISH_DELAY"] = str(delay)
    ctx = rrt_amd.Context(0)
    osc = OracleScene.build(kind, nx, ny, seed=scene_seed, half=half)
    sp, mid, mt, nd, cam21 = scene_arrays_from_oracle(osc)
    ctx.upload_scene(sp, mid, mt, nd)
    ctx.set_camera(rrt_amd.Camera.from_buffer_copy(cam21.tobytes()))
    rgb_o, acc_o, ct = osc.render(nx, ny, ns, seed=seed, max_depth=depth, threads=8)
    for count_work in (False, True):
        p = rrt_amd.Context.params(nx, ny, ns, seed=seed, max_depth=depth, path_slots=slots, batch_samples=batch, force_hbm_bvh=bool(hbm), count_work=count_work)
        rgb, acc, st = ctx.render(p)
        a_g, a_o = acc.view(np.uint32).reshape(ny, nx, 3), acc_o.view(np.uint32).reshape(ny, nx, 3)
        bad = np.argwhere((a_g != a_o).any(axis=2))
        print(f"count_work {count_work}: rays gpu {st['rays']} oracle {ct['rays']}; {len(bad)} pixels differ", flush=True)
        for y, x in bad[:8]:
            print(f"   pixel row {y} col {x}: gpu {acc.reshape(ny, nx, 3)[y, x]} oracle {acc_o.reshape(ny, nx, 3)[y, x]}", flush=True)
        if count_work:
            keys = sorted(k for k in ct if k in st)
            print("   counters that differ:", {k: (st[k], ct[k]) for k in keys if st[k] != ct[k]}, flush=True)


if __name__ == "__main__":
    libs = sys.argv[16:]
    if not libs:
        run_case(sys.argv[1:16])
    else:
        import shutil
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cur = os.path.join(here, "rrt_amd", "librrt_amd.so")
        keep = cur + ".keep"
        shutil.copy(cur, keep)
        try:
            for lib in ["current"] + libs:
                shutil.copy(keep if lib == "current" else lib, cur)
                print("==", lib, flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:16], check=False, timeout=300)
        finally:
            shutil.move(keep, cur)
