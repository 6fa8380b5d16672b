"""ctypes binding of librtx.so (include/rtx.h) and the host-side mirror of the reference's interface."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTX_LIB_PATH selects another build of the same library (A/B experiments); the default is the in-tree one.
LIB_PATH = os.environ.get("RTX_LIB_PATH") or os.path.join(_HERE, "librtx.so")

MATERIAL_NONE, MATERIAL_DIFFUSE, MATERIAL_METAL, MATERIAL_GLASS = 0, 1, 2, 3   # material.h:9-14
FLAG_ACCUMULATE, FLAG_RECURSIVE, FLAG_VALIDATE_FILTER = 1, 2, 4
FILTER_VALU, FILTER_MFMA, FILTER_GRID, FILTER_AUTO = 0, 1, 2, 3

# every symbol include/rtx.h declares
EXPORTS = ["rtx_create", "rtx_destroy", "rtx_last_error", "rtx_upload_scene", "rtx_set_filter", "rtx_set_tuning", "rtx_last_chunk_plan", "rtx_last_launch_ordered", "rtx_set_camera", "rtx_render",
           "rtx_render_noalloc", "rtx_render_async", "rtx_sync", "rtx_last_kernel_ms", "rtx_last_trace_ms", "rtx_use_accum_buffer", "rtx_accum_device", "rtx_read_accum", "rtx_write_accum",
           "rtx_save_accum", "rtx_load_accum", "rtx_resolve_rgbx8", "rtx_host_alloc", "rtx_host_free", "rtx_camera_init", "rtx_scene_capacity", "rtx_random_scene", "rtx_write_ppm", "rtx_write_ppm_p6", "rtx_plan_chunks", "rtx_build_grid",
           "rtx_version", "rtx_group_create", "rtx_group_destroy", "rtx_group_size", "rtx_group_member", "rtx_group_upload_scene", "rtx_group_set_camera",
           "rtx_group_render", "rtx_group_allreduce", "rtx_group_last_exchange_ms", "rtx_group_render_frame", "rtx_group_resolve_rgbx8", "rtx_spp_range"]


class RtxError(RuntimeError):
    pass


class CCamera(C.Structure):
    """rtx_camera: the derived camera of camera.h:94-101."""
    _fields_ = [("origin", C.c_float * 3), ("left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64), ("term_miss", C.c_uint64),
                ("term_absorb", C.c_uint64), ("term_depth", C.c_uint64), ("candidates", C.c_uint64), ("filter_misses", C.c_uint64), ("kernel_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class GridInfo(C.Structure):
    """rtx_grid_info: the uniform grid of FILTER_GRID (rtx_build_grid)."""
    _fields_ = [("gmin", C.c_float * 3), ("inv_h", C.c_float * 3), ("h", C.c_float * 3), ("dim", C.c_uint32 * 3),
                ("pad", C.c_float), ("n_big", C.c_uint32), ("n_items", C.c_uint64), ("tier_s", C.c_float * 3),
                ("tier_l", C.c_double * 3), ("tier_grow", C.c_double * 3), ("box_diag", C.c_float), ("r_min", C.c_float)]


_lib = None


def _share_torch_hip_runtime():
    """torch's ROCm wheels bundle their own HIP runtime as torch/lib/libamdhip64.so (soname libamdhip64.so.7, file name
    without the version).  librtx.so needs libamdhip64.so.7: loaded after torch it binds to torch's copy by soname, but
    loaded BEFORE torch it brings in /opt/rocm's copy, `import torch` then adds the bundled one as a second runtime in
    the process, and torch finds no GPU.  Loading the bundled copy first gives one runtime whichever import comes first.
    RTX_SYSTEM_HIP=1 skips this (processes that never use torch on the GPU)."""
    if "torch" in sys.modules or os.environ.get("RTX_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        path = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return


def lib():
    """Loads librtx.so.  Raises if it has not been built -- there is no other implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtxError("%s is missing: build it with `make -C raytracing_amd/csrc` (or __graft_entry__.build()); "
                       "raytracing_amd has no CPU fallback" % LIB_PATH)
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    P, u32, u64, f32 = C.POINTER, C.c_uint32, C.c_uint64, C.c_float
    fp, up = P(f32), P(u32)
    L.rtx_last_error.restype = C.c_char_p
    L.rtx_version.restype = C.c_char_p
    L.rtx_create.argtypes = [C.c_int, P(C.c_void_p)]
    L.rtx_destroy.argtypes = [C.c_void_p]
    L.rtx_upload_scene.argtypes = [C.c_void_p, u32, fp, fp, fp, fp, up, fp, fp, fp, fp, fp]
    L.rtx_set_filter.argtypes = [C.c_void_p, C.c_int]
    L.rtx_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.rtx_last_chunk_plan.argtypes = [C.c_void_p, up, up, up, u32]
    L.rtx_last_launch_ordered.argtypes = [C.c_void_p, P(C.c_int)]
    L.rtx_set_camera.argtypes = [C.c_void_p, P(CCamera)]
    L.rtx_render.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u64, u32, u32, up, P(Stats)]
    L.rtx_render_async.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u64, u32, u32]
    L.rtx_render_noalloc.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u64, u32, u32, C.c_void_p]
    L.rtx_sync.argtypes = [C.c_void_p]
    L.rtx_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_double)]
    L.rtx_last_trace_ms.argtypes = [C.c_void_p, P(C.c_double)]
    L.rtx_use_accum_buffer.argtypes = [C.c_void_p, u32, u32, C.c_void_p]
    L.rtx_accum_device.argtypes = [C.c_void_p, P(C.c_void_p), P(C.c_size_t)]
    L.rtx_read_accum.argtypes = [C.c_void_p, fp, C.c_size_t]
    L.rtx_write_accum.argtypes = [C.c_void_p, u32, u32, fp, C.c_size_t]
    L.rtx_resolve_rgbx8.argtypes = [C.c_void_p, u32, u32, up]
    L.rtx_host_alloc.argtypes = [C.c_void_p, C.c_size_t, P(C.c_void_p)]
    L.rtx_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.rtx_camera_init.argtypes = [P(CCamera), f32, f32, f32, f32, fp, fp, fp]
    L.rtx_scene_capacity.restype = u32
    L.rtx_scene_capacity.argtypes = [C.c_int]
    L.rtx_random_scene.argtypes = [u64, C.c_int, f32, f32, up, fp, fp, fp, fp, up, fp, fp, fp, fp, fp]
    L.rtx_plan_chunks.argtypes = [u32, u64, u64, u32, u32, u32, u64, up, u32, up]
    L.rtx_build_grid.argtypes = [u32, fp, fp, fp, fp, u32, P(GridInfo), up, u64, up, u64, up, u64, up, u64]
    L.rtx_write_ppm.argtypes = [C.c_char_p, u32, u32, up]
    L.rtx_write_ppm_p6.argtypes = [C.c_char_p, u32, u32, up]
    L.rtx_save_accum.argtypes = [C.c_void_p, C.c_char_p, u32, u64, u32, u32]
    L.rtx_load_accum.argtypes = [C.c_void_p, C.c_char_p, up, up, up, P(u64), up, up]
    L.rtx_group_create.argtypes = [P(C.c_int), C.c_int, P(C.c_void_p)]
    L.rtx_group_destroy.argtypes = [C.c_void_p]
    L.rtx_group_size.argtypes = [C.c_void_p]
    L.rtx_group_member.restype = C.c_void_p
    L.rtx_group_member.argtypes = [C.c_void_p, C.c_int]
    L.rtx_group_upload_scene.argtypes = [C.c_void_p, u32, fp, fp, fp, fp, up, fp, fp, fp, fp, fp]
    L.rtx_group_set_camera.argtypes = [C.c_void_p, P(CCamera)]
    L.rtx_group_render.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u64, u32, u32, P(Stats)]
    L.rtx_group_allreduce.argtypes = [C.c_void_p]
    L.rtx_group_last_exchange_ms.argtypes = [C.c_void_p, P(C.c_double)]
    L.rtx_group_render_frame.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u64, u32, u32, u32, u32, up]
    L.rtx_group_resolve_rgbx8.argtypes = [C.c_void_p, u32, u32, up]
    L.rtx_spp_range.restype = None
    L.rtx_spp_range.argtypes = [u32, u32, u32, u32, up, up]
    for name in EXPORTS:
        getattr(L, name)
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RtxError("rtx error %d: %s" % (rc, lib().rtx_last_error().decode()))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


# ---- the reference's scene-construction interface, mirrored ----------------------------------------------------

class Material:
    """material_t, material.h:17-38 (same defaults)."""

    def __init__(self, type=MATERIAL_NONE, albedo=(1.0, 0.0, 1.0), blur=1.0, refractionIndex=1.0):
        self.type, self.albedo, self.blur, self.refractionIndex = int(type), tuple(albedo), float(blur), float(refractionIndex)


class Sphere:
    """Sphere, sphere.h:10-21."""

    def __init__(self, pos=(0.0, 0.0, 0.0), radius=1.0, material=None):
        self.center, self.radius, self.material = tuple(pos), float(radius), material


class Scene:
    """Scene, scene.h:22-47: a list of objects.  `table()` is the flatten step of raytracer.cpp:76-90, to SoA."""

    def __init__(self, objects=None):
        self.objects = list(objects or [])

    @classmethod
    def from_table(cls, table):
        t = np.asarray(table, dtype=np.float32).reshape(-1, 10)
        return cls([Sphere(r[0:3], r[3], Material(int(r[4]), r[5:8], r[8], r[9])) for r in t])

    def table(self):
        t = np.zeros((len(self.objects), 10), dtype=np.float32)
        for i, s in enumerate(self.objects):
            if not isinstance(s, Sphere) or s.material is None:
                raise RtxError("object %d is not a Sphere with a material (raytracer.cpp:82 dynamic_casts to Sphere)" % i)
            m = s.material
            t[i] = [s.center[0], s.center[1], s.center[2], s.radius, m.type, m.albedo[0], m.albedo[1], m.albedo[2], m.blur, m.refractionIndex]
        return t


class Camera:
    """Camera, camera.h:15-102: same constructor arguments and defaults; the derivation runs in librtx's host code."""

    def __init__(self, vfov=50.0, aspect=2.0, aperture=1.0, focusDistance=3.402823466e+38, pos=(0, 0, 0), up=(0, 1, 0), lookat=(0, 0, -1)):
        self.c = CCamera()
        a = [np.asarray(v, dtype=np.float32) for v in (pos, up, lookat)]
        _check(lib().rtx_camera_init(C.byref(self.c), np.float32(vfov), np.float32(aspect), np.float32(aperture), np.float32(focusDistance), _fp(a[0]), _fp(a[1]), _fp(a[2])))

    def as_array(self):
        c = self.c
        return np.array(list(c.origin) + list(c.left_corner) + list(c.horizontal) + list(c.vertical) + list(c.u) + list(c.v) + list(c.w) + [c.lens_radius], dtype=np.float32)


def book_camera(cols, rows):
    """The camera main() builds, RayTracing.cpp:107-117, with aspect = cols / rows."""
    return Camera(20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, (13, 2, 3), (0, 1, 0), (0, 0, 0))


def random_scene(seed=1, grid_half=11, p_diffuse=0.8, p_metal=0.95):
    """_randomScene(), RayTracing.cpp:182-213, seeded.  Returns the [n,10] table (cx cy cz r type ar ag ab blur ior)."""
    L = lib()
    cap = L.rtx_scene_capacity(grid_half)
    f = [np.zeros(cap, dtype=np.float32) for _ in range(9)]
    t = np.zeros(cap, dtype=np.uint32)
    n = C.c_uint32(0)
    _check(L.rtx_random_scene(seed, grid_half, p_diffuse, p_metal, C.byref(n), _fp(f[0]), _fp(f[1]), _fp(f[2]), _fp(f[3]), _up(t),
                              _fp(f[4]), _fp(f[5]), _fp(f[6]), _fp(f[7]), _fp(f[8])))
    n = n.value
    return np.stack([f[0][:n], f[1][:n], f[2][:n], f[3][:n], t[:n].astype(np.float32), f[4][:n], f[5][:n], f[6][:n], f[7][:n], f[8][:n]], axis=1)


def plan_chunks(spp_count, n_pixels, lanes, items_per_lane=1024, max_chunks=63, tail_factor=30, max_items=1 << 26):
    """The automatic sample-chunk plan (rtx_plan_chunks): boundaries relative to spp_begin, first 0, last spp_count."""
    starts = np.zeros(64, dtype=np.uint32)
    n = C.c_uint32(0)
    _check(lib().rtx_plan_chunks(spp_count, n_pixels, lanes, items_per_lane, max_chunks, tail_factor, max_items, _up(starts), 64, C.byref(n)))
    return starts[:n.value + 1].copy()


def build_grid(table, density_pct=35):
    """The uniform grid that upload_scene builds for FILTER_GRID (rtx_build_grid; host arithmetic only): a dict with gmin, h,
    inv_h, dim, pad, the arrays cell_start, items, big and the registration tiers (tier_end per cell, tier_s, tier_l, tier_grow)."""
    t = np.ascontiguousarray(table, dtype=np.float32).reshape(-1, 10)
    cols = [np.ascontiguousarray(t[:, k]) for k in range(4)]
    info = GridInfo()
    null = C.POINTER(C.c_uint32)()
    _check(lib().rtx_build_grid(len(t), _fp(cols[0]), _fp(cols[1]), _fp(cols[2]), _fp(cols[3]), density_pct, C.byref(info), null, 0, null, 0, null, 0, null, 0))
    ncell = int(info.dim[0]) * int(info.dim[1]) * int(info.dim[2])
    cell_start = np.zeros(ncell + 1, dtype=np.uint32)
    items = np.zeros(max(1, info.n_items), dtype=np.uint32)
    big = np.zeros(max(1, info.n_big), dtype=np.uint32)
    tier_end = np.zeros(max(1, 3 * ncell), dtype=np.uint32)
    _check(lib().rtx_build_grid(len(t), _fp(cols[0]), _fp(cols[1]), _fp(cols[2]), _fp(cols[3]), density_pct, C.byref(info),
                                _up(cell_start), cell_start.size, _up(items), items.size, _up(big), big.size, _up(tier_end), tier_end.size))
    return {"gmin": np.array(info.gmin[:], dtype=np.float32), "h": np.array(info.h[:], dtype=np.float32), "inv_h": np.array(info.inv_h[:], dtype=np.float32),
            "dim": np.array(info.dim[:], dtype=np.int64), "pad": float(info.pad), "cell_start": cell_start, "items": items[:info.n_items], "big": big[:info.n_big],
            "tier_end": tier_end[:3 * ncell].reshape(-1, 3), "tier_s": np.array(info.tier_s[:], dtype=np.float32),
            "tier_l": np.array(info.tier_l[:]), "tier_grow": np.array(info.tier_grow[:]),
            "box_diag": np.float32(info.box_diag), "r_min": np.float32(info.r_min)}


def write_ppm(path, cols, rows, framebuffer, binary=False):
    """RayTracing.cpp:152-165 ("P3" text); binary=True writes the same pixels as "P6"."""
    fb = np.ascontiguousarray(framebuffer, dtype=np.uint32)
    fn = lib().rtx_write_ppm_p6 if binary else lib().rtx_write_ppm
    _check(fn(str(path).encode(), cols, rows, _up(fb)))


# ---- the backend ---------------------------------------------------------------------------------------------------

class Context:
    """One rtx_ctx: one GPU, one stream."""

    def __init__(self, device=0, _borrowed=None):
        self._h = C.c_void_p()
        self._owned = _borrowed is None
        if _borrowed is None:
            _check(lib().rtx_create(device, C.byref(self._h)))
        else:
            self._h = C.c_void_p(_borrowed)          # a member of a Group: the group destroys it
        self.cols = self.rows = 0

    def close(self):
        if self._h:
            if self._owned:
                lib().rtx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, table):
        t = np.ascontiguousarray(np.asarray(table, dtype=np.float32).reshape(-1, 10))
        cols = [np.ascontiguousarray(t[:, i]) for i in range(10)]
        mt = np.ascontiguousarray(t[:, 4].astype(np.uint32))
        _check(lib().rtx_upload_scene(self._h, len(t), _fp(cols[0]), _fp(cols[1]), _fp(cols[2]), _fp(cols[3]), _up(mt),
                                      _fp(cols[5]), _fp(cols[6]), _fp(cols[7]), _fp(cols[8]), _fp(cols[9])))
        self.n = len(t)

    def set_filter(self, kind):
        _check(lib().rtx_set_filter(self._h, kind))

    def set_tuning(self, key, value):
        """A launch knob of include/rtx.h (rtx_set_tuning): "block", "stream", "max_chunks", ..."""
        _check(lib().rtx_set_tuning(self._h, key.encode(), int(value)))

    def last_chunk_plan(self):
        """(chunk_spp, None) for uniform chunks, or (0, boundaries) of the automatic plan of the most recent render."""
        cs, n = C.c_uint32(), C.c_uint32()
        starts = np.zeros(64, dtype=np.uint32)
        _check(lib().rtx_last_chunk_plan(self._h, C.byref(cs), C.byref(n), _up(starts), 64))
        return (cs.value, None) if cs.value else (0, starts[:n.value + 1].copy())

    def last_launch_ordered(self):
        """True when the most recent launch ran a cost-ordered work list (rtx_last_launch_ordered)."""
        v = C.c_int(0)
        _check(lib().rtx_last_launch_ordered(self._h, C.byref(v)))
        return bool(v.value)

    def set_camera(self, cam):
        c = cam.c if isinstance(cam, Camera) else cam
        _check(lib().rtx_set_camera(self._h, C.byref(c)))

    def render(self, cols, rows, spp_begin, spp_count, max_depth, seed, flags=0, chunk_spp=0, want_primary=False, want_stats=True):
        primary = np.zeros(spp_count * rows * cols, dtype=np.uint32) if want_primary else None
        st = Stats()
        _check(lib().rtx_render(self._h, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp,
                                _up(primary) if want_primary else None, C.byref(st) if want_stats else None))
        self.cols, self.rows = cols, rows
        return primary, (st.as_dict() if want_stats else None)

    def render_noalloc(self, cols, rows, spp_begin, spp_count, max_depth, seed, accum_dev_ptr, flags=0, chunk_spp=0):
        """rtx_render_noalloc: renders into the caller's device buffer (integer address of cols*rows*3 floats) without
        allocating; scratch is the library's static device array."""
        _check(lib().rtx_render_noalloc(self._h, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, C.c_void_p(accum_dev_ptr)))

    def render_async(self, cols, rows, spp_begin, spp_count, max_depth, seed, flags=0, chunk_spp=0):
        _check(lib().rtx_render_async(self._h, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp))
        self.cols, self.rows = cols, rows

    def sync(self):
        _check(lib().rtx_sync(self._h))

    def last_kernel_ms(self):
        ms = C.c_double(0)
        _check(lib().rtx_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_trace_ms(self):
        ms = C.c_double(0)
        _check(lib().rtx_last_trace_ms(self._h, C.byref(ms)))
        return ms.value

    def use_accum_buffer(self, cols, rows, dev_ptr):
        """dev_ptr: integer device address of cols*rows*3 floats on this context's GPU (0/None: library-owned)."""
        _check(lib().rtx_use_accum_buffer(self._h, cols, rows, C.c_void_p(dev_ptr or None)))
        if dev_ptr:
            self.cols, self.rows = cols, rows

    def accum_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().rtx_accum_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_accum(self):
        a = np.zeros(self.cols * self.rows * 3, dtype=np.float32)
        _check(lib().rtx_read_accum(self._h, _fp(a), a.size))
        return a

    def write_accum(self, cols, rows, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        _check(lib().rtx_write_accum(self._h, cols, rows, _fp(a), a.size))
        self.cols, self.rows = cols, rows

    def save_accum(self, path, spp_done, seed, max_depth, flags=0):
        """Checkpoint of a progressive render: the accumulation buffer plus (size, samples done, seed, depth, flags)."""
        _check(lib().rtx_save_accum(self._h, os.fsencode(path), spp_done, seed, max_depth, flags))

    def load_accum(self, path):
        """Loads a checkpoint into the accumulation buffer; returns dict(cols, rows, spp_done, seed, max_depth, flags).
        Continue with render(..., spp_begin=spp_done, flags=flags | FLAG_ACCUMULATE)."""
        v = [C.c_uint32() for _ in range(5)]
        seed = C.c_uint64()
        _check(lib().rtx_load_accum(self._h, os.fsencode(path), C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(seed), C.byref(v[3]), C.byref(v[4])))
        self.cols, self.rows = v[0].value, v[1].value
        return {"cols": v[0].value, "rows": v[1].value, "spp_done": v[2].value, "seed": seed.value, "max_depth": v[3].value, "flags": v[4].value}

    def resolve(self, spp_total, debug_tile=0, out=None):
        fb = out if out is not None else np.zeros(self.cols * self.rows, dtype=np.uint32)
        _check(lib().rtx_resolve_rgbx8(self._h, spp_total, debug_tile, _up(fb)))
        return fb

    def host_framebuffer(self, n_pixels):
        """rtx_host_alloc: a uint32 array of n_pixels in pinned host memory mapped into the GPU; resolve(out=array) then needs no
        staging copy.  The memory belongs to the context: do not use the array after host_free(array) or close()."""
        p = C.c_void_p()
        _check(lib().rtx_host_alloc(self._h, n_pixels * 4, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n_pixels,))

    def host_free(self, array):
        _check(lib().rtx_host_free(self._h, C.c_void_p(array.ctypes.data)))


def spp_range(rank, world, spp_begin, spp_count):
    """rtx_spp_range: the contiguous share of [spp_begin, spp_begin + spp_count) of `rank` (remainder to the lowest ranks)."""
    b, c = C.c_uint32(), C.c_uint32()
    lib().rtx_spp_range(rank, world, spp_begin, spp_count, C.byref(b), C.byref(c))
    return b.value, c.value


class Group:
    """rtx_group: several GPUs of one node in one process -- samples split by global sample index, one RCCL all-reduce
    of the accumulation buffers, resolve on member 0."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        _check(lib().rtx_group_create(devs, len(devices), C.byref(self._h)))
        self.cols = self.rows = 0

    def close(self):
        if self._h:
            lib().rtx_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return lib().rtx_group_size(self._h)

    def member(self, rank):
        h = lib().rtx_group_member(self._h, rank)
        if not h:
            raise RtxError(lib().rtx_last_error().decode())
        m = Context(_borrowed=h)
        m.cols, m.rows = self.cols, self.rows
        return m

    def upload_scene(self, table):
        t = np.ascontiguousarray(np.asarray(table, dtype=np.float32).reshape(-1, 10))
        cols = [np.ascontiguousarray(t[:, i]) for i in range(10)]
        mt = np.ascontiguousarray(t[:, 4].astype(np.uint32))
        _check(lib().rtx_group_upload_scene(self._h, len(t), _fp(cols[0]), _fp(cols[1]), _fp(cols[2]), _fp(cols[3]), _up(mt),
                                            _fp(cols[5]), _fp(cols[6]), _fp(cols[7]), _fp(cols[8]), _fp(cols[9])))

    def set_camera(self, cam):
        c = cam.c if isinstance(cam, Camera) else cam
        _check(lib().rtx_group_set_camera(self._h, C.byref(c)))

    def render(self, cols, rows, spp_begin, spp_count, max_depth, seed, flags=0, chunk_spp=0, want_stats=True):
        st = Stats()
        _check(lib().rtx_group_render(self._h, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, C.byref(st) if want_stats else None))
        self.cols, self.rows = cols, rows
        return st.as_dict() if want_stats else None

    def allreduce(self):
        _check(lib().rtx_group_allreduce(self._h))

    def last_exchange_ms(self):
        ms = C.c_double(0)
        _check(lib().rtx_group_last_exchange_ms(self._h, C.byref(ms)))
        return ms.value

    def render_frame(self, cols, rows, spp_begin, spp_count, max_depth, seed, spp_total, flags=0, chunk_spp=0, debug_tile=0, out=None):
        """rtx_group_render_frame: every member's share, the exchange and the resolve queued behind one another; returns the
        8-bit framebuffer (written into `out` when given)."""
        fb = out if out is not None else np.zeros(cols * rows, dtype=np.uint32)
        _check(lib().rtx_group_render_frame(self._h, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, spp_total, debug_tile, _up(fb)))
        self.cols, self.rows = cols, rows
        return fb

    def resolve(self, spp_total, debug_tile=0):
        fb = np.zeros(self.cols * self.rows, dtype=np.uint32)
        _check(lib().rtx_group_resolve_rgbx8(self._h, spp_total, debug_tile, _up(fb)))
        return fb


def render_scene_hip(scene, camera, rows, cols, frameBuffer, num_aa_samples=4, max_ray_depth=50, numThreads=1, blockSize=64,
                     debug=False, recursive=True, seed=1, device=0):
    """renderSceneHIP: the fourth backend behind the signature of raytracer.h:17-19.

    scene/camera are borrowed, frameBuffer (rows*cols uint32, 0xRRGGBB00) is filled, 0 is returned on success --
    raytracer.cpp:60-143's contract.  numThreads is advisory (as for renderSceneCUDA); blockSize only positions the
    -d borders.  `recursive` selects the -r semantics; NOTE that main() passes recursive=false unless -r is given
    (RayTracing.cpp:90-93), whatever the header's default says.
    """
    ctx = Context(device)
    try:
        ctx.upload_scene(scene.table() if isinstance(scene, Scene) else scene)
        ctx.set_camera(camera)
        ctx.render(cols, rows, 0, num_aa_samples, max_ray_depth, seed, FLAG_RECURSIVE if recursive else 0, want_stats=False)
        fb = ctx.resolve(num_aa_samples, blockSize if debug else 0)
        np.asarray(frameBuffer).reshape(-1)[:] = fb
        return 0
    finally:
        ctx.close()
