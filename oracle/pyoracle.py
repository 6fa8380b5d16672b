"""ctypes binding of oracle/librt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module (see oracle/rt_oracle.h).
The product package raytracing_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("albedo", Vec3), ("blur", C.c_float), ("ior", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_float), ("material", Material)]


class Hit(C.Structure):
    _fields_ = [("distance", C.c_float), ("point", Vec3), ("normal", Vec3), ("material", Material)]


class Camera(C.Structure):
    _fields_ = [("origin", Vec3), ("leftCorner", Vec3), ("horizontal", Vec3), ("vertical", Vec3),
                ("u", Vec3), ("v", Vec3), ("w", Vec3), ("lensRadius", C.c_float)]


class Rng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 2), ("pixel", C.c_uint32), ("sample", C.c_uint32), ("event", C.c_uint32),
                ("bare_calls", C.c_uint32), ("inject", C.POINTER(C.c_float)), ("inject_n", C.c_uint32),
                ("inject_pos", C.c_uint32), ("draws", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("term_miss", C.c_uint64), ("term_absorb", C.c_uint64), ("term_depth", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    """Compile the C restatement (and, where /root/reference exists, the reference harness)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.ro_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        L.ro_u32_to_unit.restype = C.c_float
        L.ro_u32_to_unit.argtypes = [C.c_uint32]
        L.ro_rng_begin_sample.argtypes = [P(Rng), C.c_uint64, C.c_uint32, C.c_uint32]
        L.ro_rng_inject.argtypes = [P(Rng), P(C.c_float), C.c_uint32]
        L.ro_random.restype = C.c_float
        L.ro_random.argtypes = [P(Rng)]
        L.ro_random_in_unit_sphere.restype = Vec3
        L.ro_random_in_unit_sphere.argtypes = [P(Rng)]
        L.ro_random_on_unit_disk.restype = Vec3
        L.ro_random_on_unit_disk.argtypes = [P(Rng)]
        L.ro_sphere_hit.restype = C.c_int
        L.ro_sphere_hit.argtypes = [P(Sphere), P(Ray), C.c_float, C.c_float, P(Hit)]
        L.ro_scene_hit.restype = C.c_int
        L.ro_scene_hit.argtypes = [P(Sphere), C.c_uint32, P(Ray), C.c_float, C.c_float, P(Hit), P(C.c_uint32)]
        L.ro_reflect.restype = Vec3
        L.ro_reflect.argtypes = [Vec3, Vec3]
        L.ro_refract.restype = C.c_int
        L.ro_refract.argtypes = [Vec3, Vec3, C.c_float, P(Vec3)]
        L.ro_schlick.restype = C.c_float
        L.ro_schlick.argtypes = [C.c_float, C.c_float]
        L.ro_material_scatter.restype = C.c_int
        L.ro_material_scatter.argtypes = [P(Material), P(Ray), P(Hit), P(Vec3), P(Ray), P(Rng)]
        L.ro_background.restype = Vec3
        L.ro_background.argtypes = [P(Ray)]
        L.ro_camera_init.argtypes = [P(Camera), C.c_float, C.c_float, C.c_float, C.c_float, Vec3, Vec3, Vec3]
        L.ro_camera_get_ray.restype = Ray
        L.ro_camera_get_ray.argtypes = [P(Camera), C.c_float, C.c_float, P(Rng)]
        L.ro_render.restype = C.c_int
        L.ro_render.argtypes = [P(Sphere), C.c_uint32, P(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, P(C.c_float), P(C.c_uint32), P(Stats)]
        L.ro_render_plan.restype = C.c_int
        L.ro_render_plan.argtypes = [P(Sphere), C.c_uint32, P(Camera), C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_uint32), C.c_uint32,
                                     C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, P(C.c_float), P(C.c_uint32), P(Stats)]
        L.ro_resolve.argtypes = [P(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_uint32)]
        L.ro_random_scene_capacity.restype = C.c_uint32
        L.ro_random_scene_capacity.argtypes = [C.c_int]
        L.ro_random_scene.restype = C.c_uint32
        L.ro_random_scene.argtypes = [C.c_uint64, C.c_int, C.c_float, C.c_float, P(Sphere)]
        _lib = L
    return _lib


# ---- helpers -------------------------------------------------------------------------------------------------

def vec3(a):
    return Vec3(float(a[0]), float(a[1]), float(a[2]))


def v3np(v):
    return np.array([v.x, v.y, v.z], dtype=np.float32)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (C.c_uint32 * 4)()
    lib().ro_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def table_to_spheres(table):
    """[n,10] float32 rows (cx cy cz r type ar ag ab blur ior) -> ctypes Sphere array."""
    table = np.asarray(table, dtype=np.float32)
    arr = (Sphere * max(1, len(table)))()
    for i, r in enumerate(table):
        arr[i] = Sphere(vec3(r[0:3]), float(r[3]), Material(int(r[4]), vec3(r[5:8]), float(r[8]), float(r[9])))
    return arr


def spheres_to_table(arr, n):
    out = np.zeros((n, 10), dtype=np.float32)
    for i in range(n):
        s = arr[i]
        out[i] = [s.center.x, s.center.y, s.center.z, s.radius, s.material.type, s.material.albedo.x,
                  s.material.albedo.y, s.material.albedo.z, s.material.blur, s.material.ior]
    return out


def random_scene(seed=1, grid_half=11, p_diffuse=0.8, p_metal=0.95):
    L = lib()
    cap = L.ro_random_scene_capacity(grid_half)
    arr = (Sphere * cap)()
    n = L.ro_random_scene(seed, grid_half, p_diffuse, p_metal, arr)
    return spheres_to_table(arr, n)


def camera(vfov, aspect, aperture, focus, pos, up, lookat):
    cam = Camera()
    lib().ro_camera_init(C.byref(cam), vfov, aspect, aperture, focus, vec3(pos), vec3(up), vec3(lookat))
    return cam


def book_camera(cols, rows):
    """RayTracing.cpp:107-117 with aspect = cols/rows."""
    return camera(20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, (13, 2, 3), (0, 1, 0), (0, 0, 0))


def camera_to_array(cam):
    return np.concatenate([v3np(cam.origin), v3np(cam.leftCorner), v3np(cam.horizontal), v3np(cam.vertical),
                           v3np(cam.u), v3np(cam.v), v3np(cam.w), np.array([cam.lensRadius], dtype=np.float32)])


def camera_from_array(a):
    a = np.asarray(a, dtype=np.float32)
    cam = Camera(vec3(a[0:3]), vec3(a[3:6]), vec3(a[6:9]), vec3(a[9:12]), vec3(a[12:15]), vec3(a[15:18]), vec3(a[18:21]), float(a[21]))
    return cam


def render(table, cam, rows, cols, spp_begin, spp_count, max_depth, seed, chunk_spp=0, recursive=False,
           nthreads=None, tile=64, want_primary=False, chunk_starts=None):
    """Returns (accum[rows*cols*3] float32, primary or None, stats dict).  chunk_starts: explicit chunk boundaries
    (relative to spp_begin, first 0, last spp_count) instead of the uniform chunk_spp."""
    L = lib()
    sph = table_to_spheres(table)
    accum = np.zeros(rows * cols * 3, dtype=np.float32)
    primary = np.zeros(spp_count * rows * cols, dtype=np.uint32) if want_primary else None
    st = Stats()
    if nthreads is None:
        nthreads = min(8, os.cpu_count() or 1)
    if chunk_starts is not None:
        cs = np.ascontiguousarray(chunk_starts, dtype=np.uint32)
        assert cs[0] == 0 and cs[-1] == spp_count
        rc = L.ro_render_plan(sph, len(table), C.byref(cam), rows, cols, spp_begin, cs.ctypes.data_as(C.POINTER(C.c_uint32)), len(cs) - 1,
                              max_depth, seed, 1 if recursive else 0, nthreads, tile, accum.ctypes.data_as(C.POINTER(C.c_float)),
                              primary.ctypes.data_as(C.POINTER(C.c_uint32)) if want_primary else None, C.byref(st))
    else:
        rc = L.ro_render(sph, len(table), C.byref(cam), rows, cols, spp_begin, spp_count, chunk_spp, max_depth, seed,
                         1 if recursive else 0, nthreads, tile, accum.ctypes.data_as(C.POINTER(C.c_float)),
                         primary.ctypes.data_as(C.POINTER(C.c_uint32)) if want_primary else None, C.byref(st))
    if rc != 0:
        raise RuntimeError("ro_render failed")
    return accum, primary, st.as_dict()


def resolve(accum, rows, cols, spp_total, debug_tile=0):
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    fb = np.zeros(rows * cols, dtype=np.uint32)
    lib().ro_resolve(accum.ctypes.data_as(C.POINTER(C.c_float)), rows, cols, spp_total, debug_tile,
                     fb.ctypes.data_as(C.POINTER(C.c_uint32)))
    return fb


def write_scene_bin(path, table):
    table = np.ascontiguousarray(table, dtype=np.float32)
    with open(path, "wb") as f:
        f.write(np.uint32(len(table)).tobytes())
        f.write(table.tobytes())


def kat_draws(row, stream, n, mode):
    """The injected-draw lists of the function KATs (same generator as oracle/ref_harness.cpp make_draws)."""
    out = np.zeros(n, dtype=np.float32)
    L = lib()
    for b in range((n + 3) // 4):
        r = philox((b, row, stream, 0x4B4154), (0, 0))
        for j in range(4):
            if 4 * b + j < n:
                out[4 * b + j] = L.ro_u32_to_unit(r[j])
    if mode == 1:
        out = (np.float32(0.97) + np.float32(0.0299) * out).astype(np.float32)
    elif mode == 2:
        for i in range(min(n, 30)):
            out[i] = np.float32(0.999) if i % 3 == 0 else np.float32(0.001)
    return out


def psnr_rgb8(fb_a, fb_b):
    """PSNR (dB) and per-channel RMSE between two packed 0xRRGGBB00 framebuffers."""
    a = np.asarray(fb_a, dtype=np.uint32)
    b = np.asarray(fb_b, dtype=np.uint32)
    ch = []
    for sh in (24, 16, 8):
        ca = ((a >> sh) & 0xFF).astype(np.float64)
        cb = ((b >> sh) & 0xFF).astype(np.float64)
        ch.append(np.mean((ca - cb) ** 2))
    mse = float(np.mean(ch))
    rmse = [float(np.sqrt(m)) for m in ch]
    return (float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)), rmse
