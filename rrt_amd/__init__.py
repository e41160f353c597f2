"""ctypes binding of librrt_amd.so — the MI355X render path behind jfroy/rrt's tracescene().

The library is the product; this module only loads it and mirrors its C ABI (include/rrt_amd.h,
include/rrt_amd_host.h).  There is no fallback: importing works without the library (so that the
build step can import the package), but every call raises if the shared object is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrt_amd.so")

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2
SHARD_NONE, SHARD_TILE, SHARD_SAMPLE = 0, 1, 2
SCENE_CHAP11, SCENE_CHAP12, SCENE_BOOK2CHAP2, SCENE_RANDOM_SPHERES = 0, 1, 2, 3
NODE_LEAF_BIT = 0x80000000
NODE_AXIS_SHIFT = 29
NODE_INDEX_MASK = 0x1FFFFFFF


ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_BAD_SCENE, ERR_OUT_OF_MEMORY, ERR_NOT_READY = -1, -2, -3, -4, -5, -6  # rrt_status


class RrtError(RuntimeError):
    """A call of the library returned a negative rrt_status (`code`)."""

    def __init__(self, message: str, code: int = 0):
        super().__init__(message)
        self.code = code


class SphereSoa(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("c0x", "c0y", "c0z", "c1x", "c1y", "c1z", "time0", "time1", "radius")] + [
        ("material_id", C.POINTER(C.c_uint32)),
        ("n", C.c_uint32),
    ]


class MaterialSoa(C.Structure):
    _fields_ = [("kind", C.POINTER(C.c_uint32))] + [
        (n, C.POINTER(C.c_float)) for n in ("albedo_r", "albedo_g", "albedo_b", "fuzz", "ref_idx")
    ] + [("n", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [
        ("lower_left_corner", C.c_float * 3),
        ("horizontal", C.c_float * 3),
        ("vertical", C.c_float * 3),
        ("origin", C.c_float * 3),
        ("u", C.c_float * 3),
        ("v", C.c_float * 3),
        ("lens_radius", C.c_float),
        ("time_low", C.c_float),
        ("time_scale", C.c_float),
    ]

    def as_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


class RenderParams(C.Structure):
    _fields_ = [
        ("nx", C.c_uint32), ("ny", C.c_uint32), ("ns", C.c_uint32), ("max_depth", C.c_uint32),
        ("seed", C.c_uint64),
        ("shard_mode", C.c_uint32), ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32),
        ("count_work", C.c_uint32), ("path_slots", C.c_uint32), ("batch_samples", C.c_uint32),
        ("reserved", C.c_uint32 * 4),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays", "inner_visits", "leaf_tests", "hits_lambertian", "hits_metal", "hits_dielectric", "sky_misses", "paths",
        "depth_cut", "absorbed", "iterations")] + [("gpu_ms", C.c_double), ("extend_ms", C.c_double), ("extend_launches", C.c_uint64),
                                                 ("debug", C.c_uint64 * 6)]

    def as_dict(self) -> dict:
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["debug"] = list(self.debug)
        return d


_lib = None


def lib() -> C.CDLL:
    """The loaded shared library.  Raises if it has not been built (python __graft_entry__.py / make)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RrtError(f"{LIB_PATH} is missing: build it with `make` (there is no CPU fallback)")
    # A process that also uses PyTorch has to load torch's HIP runtime before this library's: loaded second, torch finds
    # no GPU.  The helpers that hand device buffers around (multi.py, bench.py) use torch anyway.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    pf, pu = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.rrt_last_error.restype = C.c_char_p
    L.rrt_version.restype = C.c_char_p
    L.rrt_create.argtypes = [i32, C.POINTER(vp)]
    L.rrt_destroy.argtypes = [vp]
    L.rrt_destroy.restype = None
    L.rrt_upload_scene.argtypes = [vp, C.POINTER(SphereSoa), C.POINTER(MaterialSoa), vp, u32]
    L.rrt_set_camera.argtypes = [vp, C.POINTER(Camera)]
    L.rrt_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats), vp]
    L.rrt_render_multi.argtypes = [C.POINTER(vp), u32, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats), vp]
    L.rrt_render_device.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.rrt_render_device_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.rrt_finalize.argtypes = [vp, vp, u32, u32, vp]
    L.rrt_intersect.argtypes = [vp, u32, vp, vp, vp, f32, f32, vp, vp]
    L.rrt_validate_scene.argtypes = [C.POINTER(SphereSoa), C.POINTER(MaterialSoa), vp, u32]
    L.rrt_debug_layout.argtypes = [C.POINTER(SphereSoa), C.POINTER(MaterialSoa), vp, u32, vp, pu, pu]
    L.rrt_build_bvh.argtypes = [vp, C.POINTER(SphereSoa), vp, u32, pu]
    L.rrt_build_bvh_method.argtypes = [vp, C.POINTER(SphereSoa), u32, vp, u32, pu]
    L.rrt_shard_pixels.argtypes = [u32, u32, u32, u32, u32, vp]
    L.rrt_shard_pixels.restype = u32
    L.rrt_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32), C.POINTER(u64)]
    # host mirror shims
    L.rrt_host_scene_build.argtypes = [i32, u32, u32, u64, i32]
    L.rrt_host_scene_build.restype = vp
    L.rrt_host_scene_free.argtypes = [vp]
    L.rrt_host_scene_free.restype = None
    L.rrt_host_scene_num_spheres.argtypes = [vp]
    L.rrt_host_scene_num_spheres.restype = u32
    L.rrt_host_scene_num_nodes.argtypes = [vp]
    L.rrt_host_scene_num_nodes.restype = u32
    L.rrt_host_scene_rng_after.argtypes = [vp, pu]
    L.rrt_host_scene_rng_after.restype = None
    L.rrt_host_scene_export.argtypes = [vp, vp, vp, vp, C.POINTER(Camera)]
    L.rrt_host_scene_export.restype = None
    L.rrt_host_scene_upload.argtypes = [vp, vp]
    L.rrt_host_tracescene.argtypes = [vp, u32, u32, u32, i32, vp, C.POINTER(u64), C.POINTER(u64)]
    L.rrt_host_camera_from.argtypes = [pf, C.POINTER(Camera)]
    L.rrt_host_camera_from.restype = None
    L.rrt_host_rng_seed.argtypes = [u64, pu]
    L.rrt_host_rng_seed.restype = None
    L.rrt_host_rng_next_u32.argtypes = [pu]
    L.rrt_host_rng_next_u32.restype = u32
    L.rrt_host_rng_gen_f32.argtypes = [pu]
    L.rrt_host_rng_gen_f32.restype = f32
    L.rrt_host_rng_jump.argtypes = [pu]
    L.rrt_host_rng_jump.restype = None
    L.rrt_host_uniform_new_inclusive.argtypes = [f32, f32, pf, pf]
    L.rrt_host_uniform_new_inclusive.restype = None
    L.rrt_host_uniform_sample.argtypes = [f32, f32, pu]
    L.rrt_host_uniform_sample.restype = f32
    L.rrt_host_write_ppm.argtypes = [C.c_char_p, vp, u32, u32]
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RrtError(f"{what} failed ({rc}): {lib().rrt_last_error().decode()}", rc)


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class HostScene:
    """chapNN_scene(nx, ny[, &mut rng]) of the reference, built by the C++ host mirror."""

    def __init__(self, kind: int, nx: int, ny: int, seed: int = 0, half: int = 11):
        self._h = lib().rrt_host_scene_build(kind, nx, ny, seed, half)
        if not self._h:
            raise RrtError("rrt_host_scene_build failed")
        self.kind, self.nx, self.ny, self.seed = kind, nx, ny, seed

    def close(self) -> None:
        if self._h:
            lib().rrt_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def n_spheres(self) -> int:
        return lib().rrt_host_scene_num_spheres(self._h)

    @property
    def n_nodes(self) -> int:
        return lib().rrt_host_scene_num_nodes(self._h)

    def rng_after(self) -> np.ndarray:
        st = np.zeros(4, np.uint32)
        lib().rrt_host_scene_rng_after(self._h, st.ctypes.data_as(C.POINTER(C.c_uint32)))
        return st

    def export(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Camera]:
        """(spheres [n,9] f32, materials [n,6] f32, nodes [n_nodes,8] u32, camera)"""
        sp = np.zeros((self.n_spheres, 9), np.float32)
        mt = np.zeros((self.n_spheres, 6), np.float32)
        nd = np.zeros((self.n_nodes, 8), np.uint32)
        cam = Camera()
        lib().rrt_host_scene_export(self._h, sp.ctypes.data, mt.ctypes.data, nd.ctypes.data, C.byref(cam))
        return sp, mt, nd, cam

    def tracescene(self, ns: int, device: int = -1) -> Tuple[np.ndarray, int, int]:
        """init_pool_with_rng + tracescene: (pixels [ny,nx,3] u8, finished pixel count, stream seed)"""
        px = np.zeros((self.ny, self.nx, 3), np.uint8)
        pdc, seed = C.c_uint64(0), C.c_uint64(0)
        _check(lib().rrt_host_tracescene(self._h, self.nx, self.ny, ns, device, px.ctypes.data, C.byref(pdc), C.byref(seed)),
               "rrt_host_tracescene")
        return px, pdc.value, seed.value


class Context:
    """One GPU (rrt_ctx)."""

    def __init__(self, device: int = -1):
        h = C.c_void_p()
        _check(lib().rrt_create(device, C.byref(h)), "rrt_create")
        self._h = h

    def close(self) -> None:
        if self._h:
            lib().rrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(0), C.c_uint64(0)
        _check(lib().rrt_device_info(self._h, name, 256, C.byref(cus), C.byref(hbm)), "rrt_device_info")
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def upload_host_scene(self, scene: HostScene) -> None:
        _check(lib().rrt_host_scene_upload(scene.handle, self._h), "rrt_host_scene_upload")

    def build_bvh(self, spheres9: np.ndarray, method: int = 0) -> np.ndarray:
        """rrt_build_bvh_method: the reference's BVH over the spheres (method 0 = Middle, 1 = EqualCount), built on the GPU;
        nodes [2n-1, 8] u32."""
        sp = np.ascontiguousarray(spheres9, np.float32)
        n = sp.shape[0]
        cols = [np.ascontiguousarray(sp[:, i]) for i in range(9)]
        mid = np.zeros(max(n, 1), np.uint32)
        s = SphereSoa(*[_fptr(c) for c in cols], mid.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        nodes = np.zeros((max(1, 2 * n - 1), 8), np.uint32)
        count = C.c_uint32(0)
        _check(lib().rrt_build_bvh_method(self._h, C.byref(s), method, nodes.ctypes.data, nodes.shape[0], C.byref(count)), "rrt_build_bvh")
        return nodes[:count.value]

    def upload_scene(self, spheres9: np.ndarray, material_id: np.ndarray, mats6: np.ndarray, nodes8: np.ndarray) -> None:
        """spheres9 [n,9] {c0,c1,time0,time1,radius}; mats6 [m,6] {kind,r,g,b,fuzz,ior}; nodes8 [k,8] u32 words."""
        sp = np.ascontiguousarray(spheres9, np.float32)
        cols = [np.ascontiguousarray(sp[:, i]) for i in range(9)]
        mid = np.ascontiguousarray(material_id, np.uint32)
        mt = np.ascontiguousarray(mats6, np.float32)
        kind = np.ascontiguousarray(mt[:, 0].astype(np.uint32))
        mcols = [np.ascontiguousarray(mt[:, i]) for i in range(1, 6)]
        nd = np.ascontiguousarray(nodes8, np.uint32)
        s = SphereSoa(*[_fptr(c) for c in cols], mid.ctypes.data_as(C.POINTER(C.c_uint32)), sp.shape[0])
        m = MaterialSoa(kind.ctypes.data_as(C.POINTER(C.c_uint32)), *[_fptr(c) for c in mcols], mt.shape[0])
        _check(lib().rrt_upload_scene(self._h, C.byref(s), C.byref(m), nd.ctypes.data, nd.shape[0]), "rrt_upload_scene")

    def set_camera(self, cam: Camera) -> None:
        _check(lib().rrt_set_camera(self._h, C.byref(cam)), "rrt_set_camera")

    @staticmethod
    def params(nx: int, ny: int, ns: int, seed: int = 0, max_depth: int = 50, shard_mode: int = SHARD_NONE, rank: int = 0,
               world: int = 1, count_work: bool = False, path_slots: int = 0, batch_samples: int = 0,
               time_kernels: bool = False, force_hbm_bvh: bool = False, retire_min: int = 0, leaf_min: int = 0,
               exit_link_walk: bool = False) -> RenderParams:
        p = RenderParams()
        p.nx, p.ny, p.ns, p.max_depth, p.seed = nx, ny, ns, max_depth, seed
        p.shard_mode, p.shard_rank, p.shard_world = shard_mode, rank, world
        p.count_work, p.path_slots, p.batch_samples = int(count_work), path_slots, batch_samples
        p.reserved[0] = int(time_kernels)  # True/1: every extend launch timed; n: every n-th (n in 1, 2, 4, 8)
        # bit 0: walk the node array in HBM even if the scene fits in LDS; bit 1: big scenes node by node along the exit links
        p.reserved[1] = int(force_hbm_bvh) | (2 if exit_link_walk else 0)
        p.reserved[2] = int(retire_min)
        p.reserved[3] = int(leaf_min)
        return p

    def render(self, p: RenderParams, want_accum: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray], dict]:
        """rrt_render into host arrays: (rgb8 [ny,nx,3], accum [ny,nx,3] f32 or None, stats)"""
        rgb = np.zeros((p.ny, p.nx, 3), np.uint8)
        acc = np.zeros((p.ny, p.nx, 3), np.float32) if want_accum else None
        st = Stats()
        done = C.c_uint64(0)
        _check(lib().rrt_render(self._h, C.byref(p), rgb.ctypes.data, acc.ctypes.data if want_accum else None, C.byref(st),
                                C.byref(done)), "rrt_render")
        d = st.as_dict()
        d["progress"] = done.value
        return rgb, acc, d

    def render_device(self, p: RenderParams, d_rgb8: int, d_accum: int) -> dict:
        """rrt_render_device into caller-owned device pointers (0 = skip that output)."""
        st = Stats()
        _check(lib().rrt_render_device(self._h, C.byref(p), d_rgb8 or None, d_accum or None, C.byref(st)), "rrt_render_device")
        return st.as_dict()

    def render_device_noalloc(self, p: RenderParams, d_rgb8: int, d_accum: int) -> dict:
        """rrt_render_device_noalloc: the same render inside the library's static workspace, no device allocation."""
        st = Stats()
        _check(lib().rrt_render_device_noalloc(self._h, C.byref(p), d_rgb8 or None, d_accum or None, C.byref(st)), "rrt_render_device_noalloc")
        return st.as_dict()

    def finalize(self, d_accum: int, n_pixels: int, ns: int, d_rgb8: int) -> None:
        _check(lib().rrt_finalize(self._h, d_accum, n_pixels, ns, d_rgb8), "rrt_finalize")

    def intersect(self, org: np.ndarray, dirs: np.ndarray, time: Optional[np.ndarray] = None, t_min: float = 0.001,
                  t_max: float = 3.4028234663852886e38) -> Tuple[np.ndarray, np.ndarray]:
        org = np.ascontiguousarray(org, np.float32)
        dirs = np.ascontiguousarray(dirs, np.float32)
        n = org.shape[0]
        tm = np.ascontiguousarray(time, np.float32) if time is not None else None
        out_t = np.zeros(n, np.float32)
        out_o = np.zeros(n, np.uint32)
        _check(lib().rrt_intersect(self._h, n, org.ctypes.data, dirs.ctypes.data, tm.ctypes.data if tm is not None else None,
                                   t_min, t_max, out_t.ctypes.data, out_o.ctypes.data), "rrt_intersect")
        return out_t, out_o


def render_multi(ctxs, p: RenderParams, want_accum: bool = True):
    """rrt_render_multi over the given contexts (rank r = ctxs[r]): (rgb8 [ny,nx,3], accum or None, summed stats)."""
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    rgb = np.zeros((p.ny, p.nx, 3), np.uint8)
    acc = np.zeros((p.ny, p.nx, 3), np.float32) if want_accum else None
    st = Stats()
    done = C.c_uint64(0)
    _check(lib().rrt_render_multi(arr, len(ctxs), C.byref(p), rgb.ctypes.data, acc.ctypes.data if want_accum else None, C.byref(st),
                                  C.byref(done)), "rrt_render_multi")
    d = st.as_dict()
    d["progress"] = done.value
    return rgb, acc, d


def _soa(spheres9, material_id, mats6, nodes8):
    sp = np.ascontiguousarray(spheres9, np.float32)
    cols = [np.ascontiguousarray(sp[:, i]) for i in range(9)]
    mid = np.ascontiguousarray(material_id, np.uint32)
    mt = np.ascontiguousarray(mats6, np.float32)
    kind = np.ascontiguousarray(mt[:, 0].astype(np.uint32))
    mcols = [np.ascontiguousarray(mt[:, i]) for i in range(1, 6)]
    nd = np.ascontiguousarray(nodes8, np.uint32)
    s = SphereSoa(*[_fptr(c) for c in cols], mid.ctypes.data_as(C.POINTER(C.c_uint32)), sp.shape[0])
    m = MaterialSoa(kind.ctypes.data_as(C.POINTER(C.c_uint32)), *[_fptr(c) for c in mcols], mt.shape[0])
    keep = (cols, mid, kind, mcols, nd)
    return s, m, nd, keep


def validate_scene(spheres9, material_id, mats6, nodes8) -> None:
    """rrt_validate_scene: raises RrtError with the library's message on a malformed scene (no GPU needed)."""
    s, m, nd, _keep = _soa(spheres9, material_id, mats6, nodes8)
    _check(lib().rrt_validate_scene(C.byref(s), C.byref(m), nd.ctypes.data, nd.shape[0]), "rrt_validate_scene")


def debug_layout(spheres9, material_id, mats6, nodes8):
    """rrt_debug_layout: (words [n_nodes,16] u32, root_ref, scene_flags) — host only."""
    s, m, nd, _keep = _soa(spheres9, material_id, mats6, nodes8)
    words = np.zeros((nd.shape[0], 16), np.uint32)
    root, flags = C.c_uint32(0), C.c_uint32(0)
    _check(lib().rrt_debug_layout(C.byref(s), C.byref(m), nd.ctypes.data, nd.shape[0], words.ctypes.data, C.byref(root), C.byref(flags)),
           "rrt_debug_layout")
    return words, root.value, flags.value


def shard_pixels(nx: int, ny: int, mode: int, rank: int, world: int) -> np.ndarray:
    n = lib().rrt_shard_pixels(nx, ny, mode, rank, world, None)
    out = np.zeros(n, np.uint32)
    if n:
        lib().rrt_shard_pixels(nx, ny, mode, rank, world, out.ctypes.data)
    return out


def camera_from(look_from, look_at, up, fov, aspect, aperture, focus_dist, time0, time1) -> Camera:
    ci = np.array(list(look_from) + list(look_at) + list(up) + [fov, aspect, aperture, focus_dist, time0, time1], np.float32)
    cam = Camera()
    lib().rrt_host_camera_from(_fptr(ci), C.byref(cam))
    return cam


def write_ppm(path: str, rgb8: np.ndarray) -> None:
    a = np.ascontiguousarray(rgb8, np.uint8)
    _check(lib().rrt_host_write_ppm(path.encode(), a.ctypes.data, a.shape[1], a.shape[0]), "rrt_host_write_ppm")
