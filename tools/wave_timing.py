#!/usr/bin/env python3
"""Share of the chip's wave slots that were busy inside the extend and shade kernels (library built with
-DRRT_WAVE_TIMING, one stream): RRT_STREAMS=1 python tools/wave_timing.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rrt_amd
ctx = rrt_amd.Context(0)
hs = rrt_amd.HostScene(rrt_amd.SCENE_CHAP12, 1200, 800, seed=0)
ctx.upload_host_scene(hs)
p = rrt_amd.Context.params(1200, 800, int(os.environ.get('SPP', '500')), seed=0, time_kernels=1)
ctx.render(p, want_accum=False)
_, _, st = ctx.render(p, want_accum=False)
d = st["debug"]
tick_us = 0.01  # the constant 100 MHz counter
e_busy, e_setup, e_n, s_busy, s_n = d[0] * tick_us, d[1] * tick_us, max(1, d[2]), d[3] * tick_us, max(1, d[4])
print("gpu_ms %.2f extend_ms %.2f launches %d" % (st["gpu_ms"], st["extend_ms"], st["extend_launches"]))
print("extend: %d wave runs, busy %.1f us each (%.1f us of it before the walk starts); wave-busy total %.1f ms" % (e_n, e_busy / e_n, e_setup / e_n, e_busy / 1e3))
print("extend: %.1f us per wave run between its end and its workgroup's end" % (d[5] * tick_us / e_n))
print("shade:  %d wave runs, busy %.1f us each; wave-busy total %.1f ms" % (s_n, s_busy / s_n, s_busy / 1e3))
e_slots, s_slots = 256 * 2 * 16, 256 * 2 * 8
print("extend slot utilisation %.3f (of %d wave slots over extend_ms)" % (e_busy / 1e3 / (e_slots * st["extend_ms"]), e_slots))
rest = st["gpu_ms"] - st["extend_ms"]
print("shade slot utilisation %.3f (of %d wave slots over gpu_ms - extend_ms = %.2f ms)" % (s_busy / 1e3 / (s_slots * rest), s_slots, rest))
