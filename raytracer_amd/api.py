"""Host-side mirror of the reference's interface for the hot path, over the C ABI of librtmi.so.

The reference's seams for this path are C++ members; the classes below keep their
names and argument meaning so that tests read like calls into the reference:

    Scene.BuildBSP()                      Render2/scene.cpp:463   -> rtmi_scene_create
    Scene.Cast(rays) -> hit records       Render2/scene.cpp:159   -> rtmi_closest_hit[_dev]
    Scene.Occluded(rays)                  Render2/shader.cpp:131  -> rtmi_any_hit[_dev]
    Rasterizer.RasterDetails / Rasterize  Render2/rasterizer.cpp  -> rtmi_camera_setup, rtmi_primary_*
    Shader(scene,...).Trace(rays)         Render2/scene.cpp:127 + shader.cpp:395 -> rtmi_trace_dev
    Shader(scene,...).Rasterize(camera)   Render2/rasterizer.cpp:148-241        -> rtmi_render_frame_dev
    Radiosity.isVisibleXiToXj             Render2/radiosity.cpp:231
    Radiosity.findFormFactor              Render2/radiosity.cpp:335
    Radiosity.progressiveRefinement       Render2/radiosity.cpp:483 (same fixed point, Jacobi sweeps)
    Radiosity.setRadiosityForAllElements  Render2/radiosity.cpp:122

Everything that computes goes through the HIP library: there is no CPU fallback, and importing this
module raises if librtmi.so is missing.  numpy is used for host buffers, torch (optional) only as
the owner of device memory and streams.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTMI_LIB", os.path.join(_HERE, "librtmi.so"))   # RTMI_LIB: kernel-variant experiments

MISS = 0xFFFFFFFF
INFINITY = 1.0e20
FF_LIGHT_MEAN = 1

# include/rtmi.h: rtmi_ray (64 B), rtmi_hit (24 B)
RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("tmax", "<f8"), ("reserved", "<f8")])
HIT_DTYPE = np.dtype([("t", "<f8"), ("u", "<f4"), ("v", "<f4"), ("tri", "<u4"), ("object", "<u4")])
assert RAY_DTYPE.itemsize == 64 and HIT_DTYPE.itemsize == 24


class rtmi_camera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("O", C.c_double * 3), ("dR", C.c_double * 3), ("dU", C.c_double * 3),
                ("x_res", C.c_uint32), ("y_res", C.c_uint32)]


class rtmi_primary_sample(C.Structure):
    _fields_ = [("sx", C.c_double), ("sy", C.c_double), ("tile_first", C.c_uint32), ("reserved", C.c_uint32), ("d_out", C.c_void_p)]


MAX_SAMPLES_PER_LAUNCH = 16


class rtmi_scene_info(C.Structure):
    _fields_ = [("n_tris", C.c_uint64), ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64), ("max_depth", C.c_uint32),
                ("max_leaf_size", C.c_uint32), ("build_seconds", C.c_double), ("device_bytes", C.c_uint64),
                ("scene_lo", C.c_float * 3), ("scene_hi", C.c_float * 3)]


class rtmi_solve_stats(C.Structure):
    _fields_ = [("sweeps", C.c_uint32), ("rel_change", C.c_double), ("max_change", C.c_double), ("seconds", C.c_double),
                ("bytes_per_sweep", C.c_double), ("accelerated_sweeps", C.c_uint32), ("spectral_radius", C.c_double)]


class rtmi_raster(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("lens_dim", C.c_uint32), ("sx", C.c_double), ("sy", C.c_double),
                ("lookat", C.c_double * 3), ("R", C.c_double * 3), ("U", C.c_double * 3), ("x_gradation", C.c_double * 3),
                ("y_gradation", C.c_double * 3), ("vpdist", C.c_double), ("xmin", C.c_double), ("ymax", C.c_double)]


RASTER_NORMAL, RASTER_ANTI_ALIASING, RASTER_DEPTH_OF_FIELD = 0, 1, 2


class rtmi_trace_params(C.Structure):
    _fields_ = [("max_tree_depth", C.c_uint32), ("ao_rays", C.c_uint32), ("ao_radius", C.c_double), ("seed", C.c_uint64),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


SHADE_RADIOSITY_BRANCH = 1   # include/rtmi.h: RTMI_SHADE_RADIOSITY_BRANCH


# include/rtmi.h: rtmi_material (18 doubles), rtmi_light (6 doubles)
MATERIAL_DOUBLES = 18
LIGHT_DOUBLES = 6

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)

# every symbol include/rtmi.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "rtmi_last_error", "rtmi_device_count", "rtmi_triangle_intersection", "rtmi_scene_create", "rtmi_scene_destroy", "rtmi_scene_get_info", "rtmi_scene_check",
    "rtmi_closest_hit", "rtmi_closest_hit_dev", "rtmi_camera_setup", "rtmi_primary_closest_hit_dev", "rtmi_primary_closest_hit_samples_dev",
    "rtmi_primary_rays_dev", "rtmi_primary_work_dev", "rtmi_shadow_rays_dev", "rtmi_ao_rays_dev", "rtmi_ao_occlusion_dev", "rtmi_shadow_occlusion_dev", "rtmi_debug_check_hierarchy", "rtmi_debug_form_factor_kernel", "rtmi_debug_form_factor_planes", "rtmi_debug_div3_dev", "rtmi_debug_unit_dev", "rtmi_radiosity_vertex_values",
    "rtmi_occlusion_fraction_dev", "rtmi_any_hit", "rtmi_any_hit_dev", "rtmi_patches_create", "rtmi_patches_destroy",
    "rtmi_visible_points", "rtmi_form_factors_dev", "rtmi_jacobi_sweep_dev", "rtmi_radiosity_solve_dev", "rtmi_radiosity_solve_noalloc_dev",
    "rtmi_comm_unique_id", "rtmi_comm_create", "rtmi_comm_destroy", "rtmi_radiosity_solve_sharded_dev",
    "rtmi_shader_create", "rtmi_shader_destroy", "rtmi_trace_dev", "rtmi_render_frame_dev", "rtmi_tone_map_dev",
    "rtmi_raster_setup", "rtmi_render_frame_sampled_dev", "rtmi_soft_shadow_rays_dev", "rtmi_soft_shadow_factor_dev", "rtmi_soft_shadow_dev",
    "rtmi_scene_file_load", "rtmi_scene_file_load_ex", "rtmi_scene_file_free", "rtmi_scene_file_info", "rtmi_scene_file_get",
    "rtmi_radiosity_trace_dev", "rtmi_radiosity_raster_dev", "rtmi_tone_map_radiosity_dev",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("librtmi.so is not built: run `make -C raytracer_amd/csrc` (or __graft_entry__.build()). "
                          "There is no CPU fallback for this path.")
    try:
        # PyTorch-ROCm bundles its own HIP runtime under the same soname as /opt/rocm's, which librtmi.so links: the
        # process must load torch's copy first, or torch later finds "No HIP GPUs" through the other one.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double
    L.rtmi_last_error.restype = C.c_char_p
    L.rtmi_triangle_intersection.argtypes = [u64, vp, vp, vp, vp, vp, vp]
    L.rtmi_scene_create.restype = vp
    L.rtmi_scene_create.argtypes = [vp, vp, u64]
    L.rtmi_scene_destroy.argtypes = [vp]
    L.rtmi_scene_destroy.restype = None
    L.rtmi_scene_get_info.argtypes = [vp, C.POINTER(rtmi_scene_info)]
    L.rtmi_scene_check.argtypes = [vp]
    L.rtmi_closest_hit.argtypes = [vp, vp, u64, vp]
    L.rtmi_closest_hit_dev.argtypes = [vp, vp, u64, vp, vp]
    L.rtmi_any_hit.argtypes = [vp, vp, u64, vp]
    L.rtmi_any_hit_dev.argtypes = [vp, vp, u64, vp, vp]
    L.rtmi_camera_setup.argtypes = [vp, vp, vp, dbl, vp, vp, u32, u32, C.POINTER(rtmi_camera)]
    L.rtmi_primary_closest_hit_dev.argtypes = [vp, C.POINTER(rtmi_camera), dbl, dbl, u32, u32, u32, u32, vp, vp]
    L.rtmi_primary_closest_hit_samples_dev.argtypes = [vp, C.POINTER(rtmi_camera), u32, u32, u32, C.POINTER(rtmi_primary_sample), u32, vp]
    L.rtmi_primary_rays_dev.argtypes = [C.POINTER(rtmi_camera), dbl, dbl, vp, vp]
    L.rtmi_primary_work_dev.argtypes = [vp, C.POINTER(rtmi_camera), dbl, dbl, vp, vp]
    L.rtmi_shadow_rays_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp]
    L.rtmi_ao_rays_dev.argtypes = [vp, vp, vp, vp, u64, u32, dbl, u64, vp, vp]
    L.rtmi_ao_occlusion_dev.argtypes = [vp, vp, vp, vp, u64, u32, dbl, u64, vp, vp]
    L.rtmi_shadow_occlusion_dev.argtypes = [vp, vp, vp, vp, u64, vp, vp, vp]
    L.rtmi_debug_check_hierarchy.argtypes = [vp, u64, vp]
    L.rtmi_debug_form_factor_kernel.argtypes = [vp]
    L.rtmi_debug_form_factor_planes.argtypes = [vp]
    L.rtmi_radiosity_vertex_values.argtypes = [vp, u32, vp, vp]
    L.rtmi_debug_div3_dev.argtypes = [vp, vp, u64, vp, vp, vp]
    L.rtmi_debug_unit_dev.argtypes = [vp, u64, vp, vp, vp]
    L.rtmi_occlusion_fraction_dev.argtypes = [vp, u64, u32, vp, vp]
    L.rtmi_soft_shadow_rays_dev.argtypes = [vp, vp, vp, vp, u64, vp, u32, vp, vp]
    L.rtmi_soft_shadow_factor_dev.argtypes = [vp, vp, u64, u32, vp, vp]
    L.rtmi_soft_shadow_dev.argtypes = [vp, vp, vp, vp, u64, vp, u32, vp, vp, vp]
    L.rtmi_patches_create.restype = vp
    L.rtmi_patches_create.argtypes = [vp, vp, vp]
    L.rtmi_patches_destroy.argtypes = [vp]
    L.rtmi_patches_destroy.restype = None
    L.rtmi_visible_points.argtypes = [vp, u64, vp, vp, vp, vp]
    L.rtmi_form_factors_dev.argtypes = [vp, vp, u32, u32, u64, u32, u32, u32, vp, u64, vp]
    L.rtmi_jacobi_sweep_dev.argtypes = [vp, u64, u32, u32, u32, vp, vp, vp, vp, vp, vp]
    L.rtmi_radiosity_solve_dev.argtypes = [vp, u64, u32, u32, u32, vp, vp, vp, C.c_float, u32, u32, vp, vp,
                                           C.POINTER(rtmi_solve_stats), vp]
    L.rtmi_radiosity_solve_noalloc_dev.argtypes = [vp, u64, u32, vp, vp, vp, C.c_float, u32, u32, C.POINTER(rtmi_solve_stats), vp]
    L.rtmi_comm_unique_id.argtypes = [vp]
    L.rtmi_comm_create.restype = vp
    L.rtmi_comm_create.argtypes = [vp, C.c_int, C.c_int]
    L.rtmi_comm_destroy.argtypes = [vp]
    L.rtmi_comm_destroy.restype = None
    L.rtmi_radiosity_solve_sharded_dev.argtypes = [vp, vp, u64, u32, u32, u32, vp, vp, vp, C.c_float, u32, u32, C.POINTER(rtmi_solve_stats), vp]
    L.rtmi_shader_create.restype = vp
    L.rtmi_shader_create.argtypes = [vp, vp, vp, u32, vp, u32]
    L.rtmi_shader_destroy.argtypes = [vp]
    L.rtmi_shader_destroy.restype = None
    L.rtmi_trace_dev.argtypes = [vp, vp, u64, u32, u64, C.POINTER(rtmi_trace_params), vp, vp]
    L.rtmi_render_frame_dev.argtypes = [vp, C.POINTER(rtmi_camera), dbl, dbl, u32, u32, u32, u32,
                                        C.POINTER(rtmi_trace_params), vp, vp, vp]
    L.rtmi_tone_map_dev.argtypes = [vp, u64, vp, vp]
    L.rtmi_raster_setup.argtypes = [u32, vp, vp, vp, dbl, vp, vp, C.POINTER(rtmi_raster)]
    L.rtmi_render_frame_sampled_dev.argtypes = [vp, C.POINTER(rtmi_camera), C.POINTER(rtmi_raster), u32, u32, u32, u32,
                                                C.POINTER(rtmi_trace_params), vp, vp, vp]
    L.rtmi_radiosity_trace_dev.argtypes = [vp, vp, vp, u64, vp, vp]
    L.rtmi_radiosity_raster_dev.argtypes = [vp, vp, C.POINTER(rtmi_camera), dbl, dbl, u32, u32, u32, u32, vp, vp, vp]
    L.rtmi_tone_map_radiosity_dev.argtypes = [vp, u64, vp, vp]
    L.rtmi_scene_file_load.restype = vp
    L.rtmi_scene_file_load.argtypes = [C.c_char_p, C.c_char_p]
    L.rtmi_scene_file_load_ex.restype = vp
    L.rtmi_scene_file_load_ex.argtypes = [C.c_char_p, C.c_char_p, u32]
    L.rtmi_scene_file_free.argtypes = [vp]
    L.rtmi_scene_file_free.restype = None
    L.rtmi_scene_file_info.argtypes = [vp, vp, vp, vp]
    L.rtmi_scene_file_get.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    return L


lib = _load()


class RtmiError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise RtmiError(lib.rtmi_last_error().decode())


def _np(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a.reshape(shape) if shape is not None else a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_rays(o, d, tmax=None):
    """Pack origins / directions (n×3 doubles) into rtmi_ray records."""
    o = _np(o, np.float64, (-1, 3)); d = _np(d, np.float64, (-1, 3))
    r = np.zeros(len(o), RAY_DTYPE)
    r["o"] = o; r["d"] = d
    r["tmax"] = INFINITY if tmax is None else tmax
    return r


def triangle_intersection(o, d, v0, v1, v2, u0=0.0, v0_=0.0):
    """triangle_intersection (object.cpp:29-60) on a batch; returns (t, u, v), t == 0.0 is a miss."""
    o = _np(o, np.float64, (-1, 3)); d = _np(d, np.float64, (-1, 3))
    tri = np.ascontiguousarray(np.concatenate([_np(v0, np.float64, (-1, 3)), _np(v1, np.float64, (-1, 3)),
                                               _np(v2, np.float64, (-1, 3))], axis=1))
    n = len(o)
    t = np.zeros(n); u = np.full(n, float(u0)); v = np.full(n, float(v0_))
    _check(lib.rtmi_triangle_intersection(n, _ptr(o), _ptr(d), _ptr(tri), _ptr(t), _ptr(u), _ptr(v)))
    return t, u, v


def check_hierarchy(tri_xyz):
    """Host only: builds the hierarchies of Scene(tri_xyz) and checks the invariants the kernels rely on.
    Returns (violations, wide_nodes, stack_entries, binary_depth)."""
    t = _np(tri_xyz, np.float64, (-1, 9))
    stats = np.zeros(4, np.uint64)
    _check(lib.rtmi_debug_check_hierarchy(_ptr(t), len(t), _ptr(stats)))
    return tuple(int(v) for v in stats)


SCENE_LENIENT = 1


def BuildScene(scenes_dir, stem, lenient=False):
    """bool Scene::BuildScene(file, fileObj, camera) (scene.cpp:209-460): reads <scenes_dir>/<stem>.sdf and its .obj files
    (host only).  Returns a dict: tris n x 9, tri_obj n, normals n x 3, mats n_obj x 18, lights n_l x 6, cam14, res.
    lenient: without the reference's limits (512-byte lines, `o <name>` first, triangles and quads only)."""
    h = lib.rtmi_scene_file_load_ex(str(scenes_dir).encode(), str(stem).encode(), SCENE_LENIENT if lenient else 0)
    if not h:
        raise RtmiError(lib.rtmi_last_error().decode())
    h = C.c_void_p(h)
    try:
        n, no, nl = C.c_uint64(), C.c_uint32(), C.c_uint32()
        _check(lib.rtmi_scene_file_info(h, C.byref(n), C.byref(no), C.byref(nl)))
        out = dict(tris=np.zeros((n.value, 9)), tri_obj=np.zeros(n.value, np.uint32), normals=np.zeros((n.value, 3)),
                   mats=np.zeros((no.value, MATERIAL_DOUBLES)), lights=np.zeros((nl.value, LIGHT_DOUBLES)), cam14=np.zeros(14),
                   res=np.zeros(2, np.uint32))
        _check(lib.rtmi_scene_file_get(h, _ptr(out["tris"]), _ptr(out["tri_obj"]), _ptr(out["normals"]), _ptr(out["mats"]),
                                       _ptr(out["lights"]), _ptr(out["cam14"]), _ptr(out["res"])))
        return out
    finally:
        lib.rtmi_scene_file_free(h)


class Camera:
    """Render2/ray_tracer.h:113-122 + the RasterDetails basis (rasterizer.cpp:40-56)."""

    def __init__(self, eye, lookat, up, vpdist, x_win, y_win, x_res, y_res):
        self.c = rtmi_camera()
        eye, lookat, up = (_np(v, np.float64, (3,)) for v in (eye, lookat, up))
        x_win = _np(x_win, np.float64, (2,)); y_win = _np(y_win, np.float64, (2,))
        self._args = (eye, lookat, up, float(vpdist), x_win, y_win)
        _check(lib.rtmi_camera_setup(_ptr(eye), _ptr(lookat), _ptr(up), float(vpdist), _ptr(x_win), _ptr(y_win),
                                     int(x_res), int(y_res), C.byref(self.c)))

    @classmethod
    def from_cam14(cls, cam14, res):
        c = np.asarray(cam14, np.float64)
        return cls(c[0:3], c[3:6], c[6:9], c[9], c[10:12], c[12:14], int(res[0]), int(res[1]))

    @property
    def x_res(self):
        return self.c.x_res

    @property
    def y_res(self):
        return self.c.y_res

    def raster(self, mode):
        """The RasterDetails fields (rasterizer.cpp:40-68) the Anti_Aliasing / Depth_Of_Field_Effect loops read."""
        eye, lookat, up, vpdist, x_win, y_win = self._args
        out = rtmi_raster()
        _check(lib.rtmi_raster_setup(int(mode), _ptr(eye), _ptr(lookat), _ptr(up), vpdist, _ptr(x_win), _ptr(y_win), C.byref(out)))
        return out

    def raster_details(self):
        """(O, dR, dU) as computed on the host side of the library."""
        return np.array(self.c.O), np.array(self.c.dR), np.array(self.c.dU)


class Scene:
    """The ray-query side of the reference's Scene (Render2/ray_tracer.h:144-171)."""

    def __init__(self, tri_xyz, tri_object=None):
        self.tri_xyz = _np(tri_xyz, np.float64, (-1, 9))
        self.tri_object = None if tri_object is None else _np(tri_object, np.uint32)
        self.handle = None
        self.BuildBSP()

    def BuildBSP(self):
        """Scene::BuildBSP (scene.cpp:463): here a SAH BVH over the original triangles, uploaded to HBM."""
        self.close()
        h = lib.rtmi_scene_create(_ptr(self.tri_xyz), _ptr(self.tri_object), len(self.tri_xyz))
        if not h:
            raise RtmiError(lib.rtmi_last_error().decode())
        self.handle = C.c_void_p(h)
        return True

    def check(self):
        """rtmi_scene_check: raises if a launch on this scene gave up (call after synchronising)."""
        _check(lib.rtmi_scene_check(self.handle))

    def info(self):
        out = rtmi_scene_info()
        _check(lib.rtmi_scene_get_info(self.handle, C.byref(out)))
        return out

    def close(self):
        if getattr(self, "handle", None):
            lib.rtmi_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer calls (synchronous) --
    def Cast(self, rays):
        """bool Scene::Cast(ray, hitinfo) for a batch: returns HIT_DTYPE records (tri == MISS: no hit)."""
        rays = _np(rays, RAY_DTYPE)
        out = np.zeros(len(rays), HIT_DTYPE)
        _check(lib.rtmi_closest_hit(self.handle, _ptr(rays), len(rays), _ptr(out)))
        return out

    def Occluded(self, rays):
        """Scene::Cast used as a predicate (shader.cpp:131,226): uint8 per ray."""
        rays = _np(rays, RAY_DTYPE)
        out = np.zeros(len(rays), np.uint8)
        _check(lib.rtmi_any_hit(self.handle, _ptr(rays), len(rays), _ptr(out)))
        return out

    # -- device-buffer calls (asynchronous on `stream`) --
    def cast_dev(self, d_rays, n, d_out, stream=0):
        _check(lib.rtmi_closest_hit_dev(self.handle, d_rays, n, d_out, stream))

    def occluded_dev(self, d_rays, n, d_out, stream=0):
        _check(lib.rtmi_any_hit_dev(self.handle, d_rays, n, d_out, stream))

    def primary_dev(self, cam, d_out, sx=0.5, sy=0.5, tile=(0, 0), tile_first=0, tile_stride=1, stream=0):
        """The ray loop of Rasterizer::Serial_Normal_Raster (rasterizer.cpp:219-233), rays generated on the GPU."""
        _check(lib.rtmi_primary_closest_hit_dev(self.handle, C.byref(cam.c), sx, sy, tile[0], tile[1], tile_first,
                                                tile_stride, d_out, stream))

    def primary_samples_dev(self, cam, samples, tile=(0, 0), tile_stride=1, stream=0):
        """rtmi_primary_closest_hit_samples_dev: samples = [(sx, sy, tile_first, d_out), ...] (at most MAX_SAMPLES_PER_LAUNCH),
        one launch and one work queue for all of them; same records as one primary_dev call per sample."""
        arr = (rtmi_primary_sample * len(samples))()
        for k, (sx, sy, first, d_out) in enumerate(samples):
            arr[k].sx, arr[k].sy, arr[k].tile_first, arr[k].reserved, arr[k].d_out = sx, sy, first, 0, d_out
        _check(lib.rtmi_primary_closest_hit_samples_dev(self.handle, C.byref(cam.c), tile[0], tile[1], tile_stride, arr, len(samples), stream))

    # -- wavefront stages (device buffers): secondary rays from hit records --
    def shadow_rays_dev(self, d_normals, d_rays, d_hits, n, light_center, d_out, stream=0):
        """Shader::generateNormalShadows' ray (shader.cpp:116-127) for every hit record."""
        lc = _np(light_center, np.float64, (3,))
        _check(lib.rtmi_shadow_rays_dev(self.handle, d_normals, d_rays, d_hits, n, _ptr(lc), d_out, stream))

    def ao_rays_dev(self, d_normals, d_rays, d_hits, n, spp, radius, seed, d_out, stream=0):
        """Shader::findOcclusionNew's rays (shader.cpp:140-232), spp per hit record."""
        _check(lib.rtmi_ao_rays_dev(self.handle, d_normals, d_rays, d_hits, n, spp, float(radius), seed, d_out, stream))

    def soft_shadow_dev(self, d_normals, d_rays, d_hits, n, light_center, n_points, d_contrib, d_factor, stream=0):
        """Shader::generateSoftShadows for a batch of hit records in one launch (+ reduction): the factor per hit."""
        lc = _np(light_center, np.float64, (3,))
        _check(lib.rtmi_soft_shadow_dev(self.handle, d_normals, d_rays, d_hits, n, _ptr(lc), n_points, d_contrib, d_factor, stream))

    def shadow_occlusion_dev(self, d_normals, d_rays, d_hits, n, light_center, d_occluded, stream=0):
        """Shader::generateNormalShadows for a batch of hit records in one launch: shadow_rays_dev + occluded_dev."""
        lc = _np(light_center, np.float64, (3,))
        _check(lib.rtmi_shadow_occlusion_dev(self.handle, d_normals, d_rays, d_hits, n, _ptr(lc), d_occluded, stream))

    def ao_occlusion_dev(self, d_normals, d_rays, d_hits, n, spp, radius, seed, d_occluded, stream=0):
        """The loop of Shader::findOcclusionNew in one launch: ao_rays_dev + occluded_dev without the rays in memory."""
        _check(lib.rtmi_ao_occlusion_dev(self.handle, d_normals, d_rays, d_hits, n, spp, float(radius), seed, d_occluded, stream))

    def soft_shadow_rays_dev(self, d_normals, d_rays, d_hits, n, light_center, n_points, d_out, stream=0):
        """Shader::generateSoftShadows' rays (shader.cpp:41-103), n_points per hit record (reference: 100)."""
        lc = _np(light_center, np.float64, (3,))
        _check(lib.rtmi_soft_shadow_rays_dev(self.handle, d_normals, d_rays, d_hits, n, _ptr(lc), n_points, d_out, stream))

    def primary_work_dev(self, cam, d_counts, sx=0.5, sy=0.5, stream=0):
        """Diagnostic: adds (node visits, triangle tests) of the frame's primary rays to d_counts[0..1] (uint64)."""
        _check(lib.rtmi_primary_work_dev(self.handle, C.byref(cam.c), sx, sy, d_counts, stream))


def occlusion_fraction_dev(d_occluded, n, spp, d_out, stream=0):
    """1 − occlusions/spp per hit (shader.cpp:231)."""
    _check(lib.rtmi_occlusion_fraction_dev(d_occluded, n, spp, d_out, stream))


def soft_shadow_factor_dev(d_hits, d_shadow_hits, n, n_points, d_out, stream=0):
    """1 − contributors/n_points per hit (shader.cpp:93-100)."""
    _check(lib.rtmi_soft_shadow_factor_dev(d_hits, d_shadow_hits, n, n_points, d_out, stream))


def primary_rays_dev(cam, d_rays, sx=0.5, sy=0.5, stream=0):
    _check(lib.rtmi_primary_rays_dev(C.byref(cam.c), sx, sy, d_rays, stream))


class Shader:
    """Scene::Trace + Shader::Shade (scene.cpp:127-155, shader.cpp:395-645) and the pixel loop of
    Rasterizer::Rasterize (rasterizer.cpp:148-241), as one wavefront on the device.
    normals: n_tris x 3 (triNormal); materials: n_objects x 18 (diffuse, specular, ambient, reflectivity, translucency,
    Phong_exp, ref_index, roughness); lights: n_lights x 6 (Center(GetBox(light)), emission)."""

    def __init__(self, scene, normals, materials, lights, max_tree_depth=1, ao_rays=10, ao_radius=2.0, seed=1, radiosity_branch=False):
        """radiosity_branch: Shade as it runs with scene.radiosity != NULL (shader.cpp:431-478)."""
        self.scene = scene
        nrm = _np(normals, np.float64, (-1, 3))
        mats = _np(materials, np.float64, (-1, MATERIAL_DOUBLES))
        lts = _np(lights, np.float64, (-1, LIGHT_DOUBLES))
        if len(nrm) != len(scene.tri_xyz):
            raise ValueError("one normal per triangle of the scene")
        self.handle = None
        h = lib.rtmi_shader_create(scene.handle, _ptr(nrm), _ptr(mats), len(mats), _ptr(lts), len(lts))
        if not h:
            raise RtmiError(lib.rtmi_last_error().decode())
        self.handle = C.c_void_p(h)
        self.params = rtmi_trace_params(int(max_tree_depth), int(ao_rays), float(ao_radius), int(seed),
                                        SHADE_RADIOSITY_BRANCH if radiosity_branch else 0, 0)

    def close(self):
        if getattr(self, "handle", None):
            lib.rtmi_shader_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_dev(self, d_rays, n, d_rgb, generation=1, first_index=0, stream=0):
        _check(lib.rtmi_trace_dev(self.handle, d_rays, n, generation, first_index, C.byref(self.params), d_rgb, stream))

    def render_frame_dev(self, cam, d_rgb, d_pixels, sx=0.5, sy=0.5, tile=(0, 0), tile_first=0, tile_stride=1, stream=0):
        _check(lib.rtmi_render_frame_dev(self.handle, C.byref(cam.c), sx, sy, tile[0], tile[1], tile_first, tile_stride,
                                         C.byref(self.params), d_rgb, d_pixels, stream))

    # -- host-buffer conveniences (torch owns the device memory) --
    def Trace(self, rays, generation=1, first_index=0):
        """Color Scene::Trace(ray) for a batch of RAY_DTYPE records -> n x 3 doubles (before ToneMap)."""
        import torch
        rays = _np(rays, RAY_DTYPE)
        n = len(rays)
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, 64).copy()).cuda()
        d_rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        self.trace_dev(d_rays.data_ptr(), n, d_rgb.data_ptr(), generation, first_index, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return d_rgb.cpu().numpy()

    def render_frame_sampled_dev(self, cam, raster, d_rgb, d_pixels, tile=(0, 0), tile_first=0, tile_stride=1, stream=0):
        _check(lib.rtmi_render_frame_sampled_dev(self.handle, C.byref(cam.c), C.byref(raster), tile[0], tile[1], tile_first,
                                                 tile_stride, C.byref(self.params), d_rgb, d_pixels, stream))

    def Rasterize(self, cam, sx=0.5, sy=0.5, tile=(0, 0), tile_first=0, tile_stride=1, mode=RASTER_NORMAL):
        """Rasterizer::Rasterize: returns (rgb y_res x x_res x 3 doubles, pixels y_res x x_res x 3 uint8 = ToneMap).
        mode: RASTER_NORMAL (Serial_Normal_Raster), RASTER_ANTI_ALIASING (Anti_Aliasing), RASTER_DEPTH_OF_FIELD."""
        import torch
        n = cam.x_res * cam.y_res
        d_rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        d_pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
        if mode != RASTER_NORMAL:
            self.render_frame_sampled_dev(cam, cam.raster(mode), d_rgb.data_ptr(), d_pix.data_ptr(), tile, tile_first, tile_stride,
                                          torch.cuda.current_stream().cuda_stream)
        else:
            self.render_frame_dev(cam, d_rgb.data_ptr(), d_pix.data_ptr(), sx, sy, tile, tile_first, tile_stride,
                                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return (d_rgb.cpu().numpy().reshape(cam.y_res, cam.x_res, 3), d_pix.cpu().numpy().reshape(cam.y_res, cam.x_res, 3))


def tone_map_dev(d_rgb, n, d_pixels, stream=0):
    """ToneMap (rasterizer.cpp:79-88)."""
    _check(lib.rtmi_tone_map_dev(d_rgb, n, d_pixels, stream))


def write_ppm(path, pixels):
    """PPM_Image::Write (ppm_image.cpp): binary P6, maxval 255."""
    px = np.ascontiguousarray(pixels, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (px.shape[1], px.shape[0]))
        f.write(px.tobytes())


class Radiosity:
    """Render2/ray_tracer.h:366-390.  Elements = the scene's triangles; B, FF as in radiosity.cpp:8-14."""

    REFLECTIVITY_INDEX = 0.70                 # base.h:42
    SOLAR_RADIOSITY_POWER = 1370.0 * 50.0     # base.h:44

    def __init__(self, scene, normals, light_center=None):
        self.scene = scene
        self.normals = _np(normals, np.float64, (-1, 3))
        assert len(self.normals) == len(scene.tri_xyz)
        self.numOfElements = len(self.normals)
        self.light_center = None if light_center is None else _np(light_center, np.float64, (3,))
        self.n = self.numOfElements + (1 if self.light_center is not None else 0)
        h = lib.rtmi_patches_create(scene.handle, _ptr(scene.tri_xyz), _ptr(self.normals))
        if not h:
            raise RtmiError(lib.rtmi_last_error().decode())
        self.handle = C.c_void_p(h)
        self.FF = None   # torch.float32 [n_rows, ld] on the device
        self.B = None
        self.radiosityValue = None

    def close(self):
        if getattr(self, "handle", None):
            lib.rtmi_patches_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def isVisibleXiToXj(self, ej, xi, xj):
        """radiosity.cpp:231-262 on injected point pairs (batch)."""
        ej = _np(ej, np.int32); xi = _np(xi, np.float64, (-1, 3)); xj = _np(xj, np.float64, (-1, 3))
        vis = np.zeros(len(ej), np.uint8)
        _check(lib.rtmi_visible_points(self.handle, len(ej), _ptr(ej), _ptr(xi), _ptr(xj), _ptr(vis)))
        return vis

    def findFormFactor(self, samples=10, light_samples=30, seed=1, flags=0, row0=0, n_rows=None, stream=None):
        """radiosity.cpp:335-403: fills FF rows [row0, row0+n_rows) on the device (fp32, ld = n rounded up to 32)."""
        import torch
        n_rows = self.n - row0 if n_rows is None else n_rows
        ld = (self.n + 31) // 32 * 32     # rows start on 128-byte lines: the sweeps read such a matrix 8 % faster than one at a multiple of 4
        self.ld, self.row0, self.n_rows = ld, row0, n_rows
        self.FF = torch.zeros((n_rows, ld), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(lib.rtmi_form_factors_dev(self.handle, _ptr(self.light_center), samples, light_samples, seed, flags,
                                         row0, n_rows, self.FF.data_ptr(), ld, st))
        return self.FF

    def emission(self):
        """E of radiosity.cpp:493-499 as [n][3] float32: 0 except the light."""
        E = np.zeros((self.n, 3), np.float32)
        if self.light_center is not None:
            E[self.n - 1] = self.SOLAR_RADIOSITY_POWER
        return E

    def progressiveRefinement(self, rho=None, E=None, tol_rel=1e-7, max_sweeps=1000, check_every=4):
        """Solves (I − ρF)·B = E — the system radiosity.cpp:483-563 iterates on — by Jacobi sweeps on the GPU.
        With torch.distributed initialised (one process per GPU) FF holds this rank's rows and the ranks
        all-gather B after every sweep (multi.solve_sharded); otherwise the whole loop runs inside the library."""
        import torch
        import torch.distributed as dist
        rho = _np([self.REFLECTIVITY_INDEX] * 3 if rho is None else rho, np.float64, (3,))
        E = self.emission() if E is None else _np(E, np.float32, (self.n, 3))
        dE = torch.from_numpy(E).cuda()
        st = torch.cuda.current_stream().cuda_stream
        stats = rtmi_solve_stats()
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        in_library = (sharded and dist.get_backend() == "nccl") or os.environ.get("RTMI_SOLVE_IN_LIBRARY") == "1"
        if in_library:
            # production: the loop and its RCCL exchange run inside the library (a communicator of one rank works too)
            from . import multi
            try:
                multi.library_comm()
            except RtmiError as e:       # no RCCL to dlopen on this host (every rank sees the same): the torch.distributed loop does the same iteration
                if not sharded:
                    raise
                print("raytracer_amd: the library's RCCL exchange is unavailable (%s): solving through torch.distributed" % e, file=sys.stderr, flush=True)
                in_library = False
        if in_library:
            self.B, stats = multi.solve_sharded_library(self.FF.data_ptr(), self.ld, self.n, self.row0, self.n_rows, rho, dE, tol_rel,
                                                        max_sweeps, check_every, stream=st)
        elif sharded:
            from . import multi
            row0, n_rows = self.row0, self.n_rows

            def sweep(B_old, B_new, norms):
                jacobi_sweep_dev(self.FF.data_ptr(), self.ld, self.n, row0, n_rows, rho, dE.data_ptr(), B_old.data_ptr(),
                                 B_new.data_ptr(), norms.data_ptr(), torch.cuda.current_stream().cuda_stream)
            r = multi.solve_sharded(sweep, dE, self.n, tol_rel, max_sweeps, check_every)
            self.B = r.B
            stats.sweeps, stats.rel_change, stats.max_change = r.sweeps, r.rel_change, r.max_change
            stats.bytes_per_sweep = 4.0 * n_rows * self.n
        else:
            dB = torch.zeros_like(dE)
            _check(lib.rtmi_radiosity_solve_dev(self.FF.data_ptr(), self.ld, self.n, self.row0, self.n_rows, _ptr(rho),
                                                dE.data_ptr(), dB.data_ptr(), tol_rel, max_sweeps, check_every,
                                                None, None, C.byref(stats), st))
            self.B = dB
        self.stats = stats
        return self.B

    def setRadiosityForAllElements(self):
        """radiosity.cpp:122-134: QuadTreeNode::radiosityValue[i] = B[i] (monochrome = channel 0), as doubles."""
        self.radiosityValue = self.B[: self.numOfElements, 0].double().cpu().numpy()
        return self.radiosityValue

    # -- display path (Rasterizer::Radiosity_Raster, rasterizer.cpp:408-470)
    def _vertex_radiosity(self, vertex_radiosity):
        import torch
        if isinstance(vertex_radiosity, str) and vertex_radiosity == "shared":   # returnQuadTreeColor's shared-vertex averages
            vertex_radiosity = vertex_values(self.scene.tri_xyz, self.radiosityValue)
        if vertex_radiosity is None:      # flat: returnQuadTreeColor without shared-vertex averages = radiosityValue at every vertex
            vertex_radiosity = np.repeat(np.asarray(self.radiosityValue, np.float64)[:, None], 3, 1)
        vr = _np(vertex_radiosity, np.float64, (-1, 3))
        if len(vr) != self.numOfElements:
            raise ValueError("one row of three vertex radiosities per element")
        return torch.from_numpy(vr.copy()).cuda()

    def trace_ray(self, rays, vertex_radiosity=None):
        """Radiosity_Helper::trace_ray (radiosity_helper.cpp:318-365) for a batch of RAY_DTYPE records -> n x 3 colours."""
        import torch
        rays = _np(rays, RAY_DTYPE)
        n = len(rays)
        d_vr = self._vertex_radiosity(vertex_radiosity)
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, 64).copy()).cuda()
        d_rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        _check(lib.rtmi_radiosity_trace_dev(self.handle, d_vr.data_ptr(), d_rays.data_ptr(), n, d_rgb.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return d_rgb.cpu().numpy()

    def Radiosity_Raster(self, cam, vertex_radiosity=None, sx=0.5, sy=0.5, tile=(0, 0), tile_first=0, tile_stride=1):
        """Rasterizer::Radiosity_Raster: (rgb y x x x 3 doubles, pixels y x x x 3 uint8 = ToneMapRadiosity)."""
        import torch
        n = cam.x_res * cam.y_res
        d_vr = self._vertex_radiosity(vertex_radiosity)
        d_rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        d_pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
        _check(lib.rtmi_radiosity_raster_dev(self.handle, d_vr.data_ptr(), C.byref(cam.c), sx, sy, tile[0], tile[1], tile_first,
                                             tile_stride, d_rgb.data_ptr(), d_pix.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        return (d_rgb.cpu().numpy().reshape(cam.y_res, cam.x_res, 3), d_pix.cpu().numpy().reshape(cam.y_res, cam.x_res, 3))


def vertex_values(verts, B):
    """returnQuadTreeColor's vertex values (radiosity_helper.h:156-213): n x 3, for elements verts [n][9] with radiosities B [n]
    (host arrays; rtmi_radiosity_vertex_values)."""
    verts = _np(verts, np.float64, (-1, 9))
    B = _np(B, np.float64)
    if len(B) != len(verts):
        raise ValueError("one radiosity per element")
    out = np.zeros((len(verts), 3))
    _check(lib.rtmi_radiosity_vertex_values(_ptr(verts), len(verts), _ptr(B), _ptr(out)))
    return out


NOALLOC_MAX_N = 131072            # include/rtmi.h: RTMI_NOALLOC_MAX_N
NOALLOC_MAX_CHECK_EVERY = 64      # RTMI_NOALLOC_MAX_CHECK_EVERY


def radiosity_solve_dev(d_F, ld, n, rho, d_E, d_B, tol_rel, max_sweeps, check_every, stream=0, noalloc=False):
    """rtmi_radiosity_solve_dev over all rows (no exchange), or its allocation-free twin; returns rtmi_solve_stats."""
    rho = _np(rho, np.float64, (3,))
    stats = rtmi_solve_stats()
    if noalloc:
        _check(lib.rtmi_radiosity_solve_noalloc_dev(d_F, ld, n, _ptr(rho), d_E, d_B, tol_rel, max_sweeps, check_every, C.byref(stats), stream))
    else:
        _check(lib.rtmi_radiosity_solve_dev(d_F, ld, n, 0, n, _ptr(rho), d_E, d_B, tol_rel, max_sweeps, check_every, None, None,
                                            C.byref(stats), stream))
    return stats


def jacobi_sweep_dev(d_F, ld, n, row0, n_rows, rho, d_E, d_B_old, d_B_new, d_norms, stream=0):
    rho = _np(rho, np.float64, (3,))
    _check(lib.rtmi_jacobi_sweep_dev(d_F, ld, n, row0, n_rows, _ptr(rho), d_E, d_B_old, d_B_new, d_norms, stream))
