"""ctypes wrapper of the CPU oracle (oracle/rrt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing
under rrt_amd/ does.  PARITY UNPINNED with respect to the reference itself (see rrt_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrt_oracle.so")

MODE_REF_SERIAL, MODE_PER_SAMPLE, MODE_REF_POOL = 0, 1, 2
CTR_NAMES = ("rays", "inner_visits", "leaf_tests", "hits_lambertian", "hits_metal", "hits_dielectric", "sky_misses", "paths",
             "depth_cut", "absorbed")
FLT_MAX = 3.4028234663852886e38

_lib = None


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    L.orc_rng_seed.argtypes = [u64, vp]
    L.orc_rng_next_u32.argtypes = [vp]
    L.orc_rng_next_u32.restype = u32
    L.orc_rng_gen_f32.argtypes = [vp]
    L.orc_rng_gen_f32.restype = f32
    L.orc_rng_jump.argtypes = [vp]
    L.orc_splitmix64.argtypes = [C.POINTER(u64)]
    L.orc_splitmix64.restype = u64
    L.orc_uniform_new_inclusive.argtypes = [f32, f32, C.POINTER(f32), C.POINTER(f32)]
    L.orc_uniform_sample.argtypes = [f32, f32, vp]
    L.orc_uniform_sample.restype = f32
    L.orc_gamma_eb.argtypes = [i32]
    L.orc_gamma_eb.restype = f32
    L.orc_far_scale.restype = f32
    L.orc_sample_stream_seed.argtypes = [u64, u64, u64]
    L.orc_sample_stream_seed.restype = u64
    L.orc_reflectance.argtypes = [f32, f32, i32]
    L.orc_reflectance.restype = f32
    L.orc_f32_as_u8.argtypes = [f32]
    L.orc_f32_as_u8.restype = C.c_uint8
    L.orc_gen_in_unit_sphere.argtypes = [vp, vp]
    L.orc_gen_in_unit_disc.argtypes = [vp, vp]
    L.orc_scene_build.argtypes = [i32, u32, u32, u64, i32]
    L.orc_scene_build.restype = vp
    L.orc_scene_from_arrays.argtypes = [u32] + [vp] * 10 + [u32] + [vp] * 6 + [u32, vp]
    L.orc_scene_from_arrays.restype = vp
    L.orc_scene_build_bvh.argtypes = [vp, i32]
    L.orc_scene_free.argtypes = [vp]
    L.orc_scene_num_spheres.argtypes = [vp]
    L.orc_scene_num_spheres.restype = u32
    L.orc_scene_num_nodes.argtypes = [vp]
    L.orc_scene_num_nodes.restype = u32
    L.orc_scene_rng_after.argtypes = [vp, vp]
    L.orc_scene_export_spheres.argtypes = [vp, vp, vp]
    L.orc_scene_export_nodes.argtypes = [vp, vp]
    L.orc_scene_export_camera.argtypes = [vp, vp]
    L.orc_scene_set_camera.argtypes = [vp, vp]
    L.orc_camera_from.argtypes = [vp, vp]
    L.orc_hit_rays.argtypes = [vp, u32, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp]
    L.orc_hit_rays.restype = i32
    L.orc_gen_primary.argtypes = [vp, u32, u32, u64, u32, vp, vp, vp, vp]
    L.orc_render.argtypes = [vp, u32, u32, u32, i32, i32, u64, vp, i32, i32, u32, u32, vp, vp, vp, vp]
    L.orc_render.restype = i32
    L.orc_render_pixels.argtypes = [vp, u32, u32, u32, u64, i32, i32, u32, vp, vp, vp, vp]
    L.orc_render_pixels_shard.argtypes = [vp, u32, u32, u32, u64, i32, i32, u32, u32, u32, vp, vp, vp, vp]
    L.orc_trace_path.argtypes = [vp, u32, u32, u64, u32, u32, i32, i32, vp, vp]
    _lib = L
    return L


def camera24_to_abi(cam24: np.ndarray) -> np.ndarray:
    """oracle camera (7 vec3 + 3 scalars) -> the 21 floats of rrt_camera {llc, hor, ver, org, u, v, lens, low, scale}"""
    return np.concatenate([cam24[0:18], cam24[21:24]]).astype(np.float32)


class OracleScene:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle scene construction failed")
        self._h = handle

    @classmethod
    def build(cls, kind: int, nx: int, ny: int, seed: int = 0, half: int = 11) -> "OracleScene":
        return cls(lib().orc_scene_build(kind, nx, ny, seed, half))

    @classmethod
    def from_arrays(cls, spheres9: np.ndarray, material_id: np.ndarray, mats6: np.ndarray, nodes8: np.ndarray,
                    camera21: Optional[np.ndarray] = None) -> "OracleScene":
        sp = np.ascontiguousarray(spheres9, np.float32)
        cols = [np.ascontiguousarray(sp[:, i]) for i in range(9)]
        mid = np.ascontiguousarray(material_id, np.uint32)
        mt = np.ascontiguousarray(mats6, np.float32)
        kind = np.ascontiguousarray(mt[:, 0].astype(np.uint32))
        mc = [np.ascontiguousarray(mt[:, i]) for i in range(1, 6)]
        nd = np.ascontiguousarray(nodes8, np.uint32)
        h = lib().orc_scene_from_arrays(sp.shape[0], *[c.ctypes.data for c in cols], mid.ctypes.data, mt.shape[0], kind.ctypes.data,
                                        *[c.ctypes.data for c in mc], nd.shape[0], nd.ctypes.data)
        s = cls(h)
        if camera21 is not None:
            s.set_camera21(camera21)
        return s

    def build_bvh(self, method: int = 0) -> None:
        lib().orc_scene_build_bvh(self._h, method)

    def close(self):
        if self._h:
            lib().orc_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_spheres(self) -> int:
        return lib().orc_scene_num_spheres(self._h)

    @property
    def n_nodes(self) -> int:
        return lib().orc_scene_num_nodes(self._h)

    def rng_after(self) -> np.ndarray:
        st = np.zeros(4, np.uint32)
        lib().orc_scene_rng_after(self._h, st.ctypes.data)
        return st

    def export(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """(spheres [n,9], materials [n,6], nodes [k,8] u32, camera 24 floats)"""
        sp = np.zeros((self.n_spheres, 9), np.float32)
        mt = np.zeros((self.n_spheres, 6), np.float32)
        nd = np.zeros((self.n_nodes, 8), np.uint32)
        cam = np.zeros(24, np.float32)
        lib().orc_scene_export_spheres(self._h, sp.ctypes.data, mt.ctypes.data)
        lib().orc_scene_export_nodes(self._h, nd.ctypes.data)
        lib().orc_scene_export_camera(self._h, cam.ctypes.data)
        return sp, mt, nd, cam

    def set_camera21(self, cam21: np.ndarray) -> None:
        c = np.zeros(24, np.float32)
        c[0:18] = cam21[0:18]
        c[21:24] = cam21[18:21]
        lib().orc_scene_set_camera(self._h, c.ctypes.data)

    def hit_rays(self, org, dirs, time=None, t_min=0.001, t_max=FLT_MAX, brute=False, want_visits=False):
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        n = org.shape[0]
        tm = np.ascontiguousarray(time, np.float32) if time is not None else None
        t = np.zeros(n, np.float32)
        obj = np.zeros(n, np.uint32)
        p = np.zeros((n, 3), np.float32)
        nn = np.zeros((n, 3), np.float32)
        front = np.zeros(n, np.uint8)
        visits = np.zeros(2, np.uint64)
        rc = lib().orc_hit_rays(self._h, n, org.ctypes.data, dirs.ctypes.data, tm.ctypes.data if tm is not None else None, t_min, t_max,
                                1 if brute else 0, t.ctypes.data, obj.ctypes.data, p.ctypes.data, nn.ctypes.data, front.ctypes.data,
                                visits.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle traversal stack overflow (the reference would panic)")
        if want_visits:
            return t, obj, p, nn, front, visits
        return t, obj, p, nn, front

    def gen_primary(self, nx, ny, seed, idx, k):
        idx = np.ascontiguousarray(idx, np.uint32)
        k = np.ascontiguousarray(k, np.uint32)
        out = np.zeros((idx.shape[0], 7), np.float32)
        st = np.zeros((idx.shape[0], 4), np.uint32)
        lib().orc_gen_primary(self._h, nx, ny, seed, idx.shape[0], idx.ctypes.data, k.ctypes.data, out.ctypes.data, st.ctypes.data)
        return out, st

    def render(self, nx, ny, ns, seed=0, mode=MODE_PER_SAMPLE, max_depth=50, threads=1, pow5_mode=0, state=None, k_begin=0, k_step=1,
               want_per_sample=False):
        """Returns (rgb8 [ny,nx,3], accum [ny,nx,3] f32, counters dict[, per_sample [ns,ny,nx,3]])."""
        rgb = np.zeros((ny, nx, 3), np.uint8)
        acc = np.zeros((ny, nx, 3), np.float32)
        ctr = np.zeros(len(CTR_NAMES), np.uint64)
        ps = np.zeros((ns, ny, nx, 3), np.float32) if want_per_sample else None
        st = np.ascontiguousarray(state, np.uint32) if state is not None else None
        rc = lib().orc_render(self._h, nx, ny, ns, max_depth, mode, seed, st.ctypes.data if st is not None else None, threads, pow5_mode,
                              k_begin, k_step, rgb.ctypes.data, acc.ctypes.data, ps.ctypes.data if ps is not None else None,
                              ctr.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle traversal stack overflow (the reference would panic)")
        d = {n: int(v) for n, v in zip(CTR_NAMES, ctr)}
        if want_per_sample:
            return rgb, acc, d, ps
        return rgb, acc, d

    def render_pixels(self, nx, ny, ns, seed, idx, max_depth=50, pow5_mode=0, k_begin=0, k_step=1):
        """Per-sample-mode sums and RGB8 of chosen pixels: (acc [n,3] f32, rgb [n,3] u8, rays).  With k_step > 1: the sample
        shard k = k_begin, k_begin + k_step, ... < ns alone."""
        idx = np.ascontiguousarray(idx, np.uint32)
        acc = np.zeros((idx.shape[0], 3), np.float32)
        rgb = np.zeros((idx.shape[0], 3), np.uint8)
        rays = C.c_uint64(0)
        if k_begin == 0 and k_step == 1:
            lib().orc_render_pixels(self._h, nx, ny, ns, seed, max_depth, pow5_mode, idx.shape[0], idx.ctypes.data, acc.ctypes.data,
                                    rgb.ctypes.data, C.byref(rays))
        else:
            lib().orc_render_pixels_shard(self._h, nx, ny, ns, seed, max_depth, pow5_mode, k_begin, k_step, idx.shape[0], idx.ctypes.data,
                                          acc.ctypes.data, rgb.ctypes.data, C.byref(rays))
        return acc, rgb, rays.value

    def trace_path_rays(self, nx, ny, seed, idx, k, max_depth=50, pow5_mode=0):
        """The rays of one path, in order: [n, 7] f32 (origin, direction, time)."""
        buf = np.zeros((max_depth + 2, 7), np.float32)
        f = lib().orc_trace_path_rays
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint64]
        n = f(self._h, nx, ny, seed, idx, k, max_depth, pow5_mode, buf.ctypes.data, buf.shape[0])
        return buf[:n]

    def trace_path(self, nx, ny, seed, idx, k, max_depth=50, pow5_mode=0):
        out = np.zeros(3, np.float32)
        rays = C.c_uint64(0)
        lib().orc_trace_path(self._h, nx, ny, seed, idx, k, max_depth, pow5_mode, out.ctypes.data, C.byref(rays))
        return out, rays.value
