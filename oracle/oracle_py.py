"""TEST INFRASTRUCTURE — not part of the product.

ctypes bindings for the two CPU checkers:
  * ``Oracle``   -> oracle/liboracle.so           (our plain-C restatement, rt_oracle.c)
  * ``Reference``-> oracle/_ref/libref_render2.so (the compiled reference + ref_driver.cpp)

Only tests/, oracle/gen_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Nothing under raytracer_amd/ does.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref_render2.so")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Oracle:
    """rt_oracle.h"""

    def __init__(self, path=ORACLE_SO):
        self.lib = L = C.CDLL(path)
        L.rto_mt.restype = C.c_double
        L.rto_area.restype = C.c_double
        L.rto_ff_term_points.restype = C.c_double
        L.rto_light_term_points.restype = C.c_double
        L.rto_progressive_refinement.restype = C.c_long
        L.rto_jacobi_rgb.restype = C.c_long
        L.rto_bvh_build.restype = C.c_void_p
        L.rto_bvh_closest_hit.restype = C.c_double
        L.rto_hash.restype = C.c_uint64

    def mt_batch(self, o, d, v0, v1, v2):
        o, d, v0, v1, v2 = (_f64(x, (-1, 3)) for x in (o, d, v0, v1, v2))
        n = len(o)
        t, u, v = np.zeros(n), np.zeros(n), np.zeros(n)
        self.lib.rto_mt_batch(C.c_long(n), _p(o), _p(d), _p(v0), _p(v1), _p(v2), _p(t), _p(u), _p(v))
        return t, u, v

    def closest_hit(self, tris, o, d, tmax=None):
        tris = _f64(tris, (-1, 9)); o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        n = len(o)
        tm = _f64(tmax) if tmax is not None else None
        tri = np.zeros(n, np.int32); t = np.zeros(n); u = np.zeros(n); v = np.zeros(n)
        self.lib.rto_closest_hit(C.c_long(len(tris)), _p(tris), C.c_long(n), _p(o), _p(d), _p(tm), _p(tri), _p(t),
                                 _p(u), _p(v))
        return tri, t, u, v

    def any_hit(self, tris, o, d, tmax=None):
        tris = _f64(tris, (-1, 9)); o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        n = len(o)
        tm = _f64(tmax) if tmax is not None else None
        occ = np.zeros(n, np.uint8)
        self.lib.rto_any_hit(C.c_long(len(tris)), _p(tris), C.c_long(n), _p(o), _p(d), _p(tm), _p(occ))
        return occ

    def raster_details(self, cam14, x_res, y_res):
        cam14 = _f64(cam14)
        out = np.zeros(18)
        self.lib.rto_raster_details(_p(cam14), C.c_uint32(int(x_res)), C.c_uint32(int(y_res)), _p(out))
        return out

    def primary_dirs(self, rd18, pix):
        rd18 = _f64(rd18); pix = np.ascontiguousarray(pix, np.uint32).reshape(-1, 2)
        d = np.zeros((len(pix), 3))
        self.lib.rto_primary_dirs(_p(rd18), C.c_long(len(pix)), _p(pix), _p(d))
        return d

    def raster_rays(self, mode, cam14, res, pix):
        """(o, d, S): the S generation-1 rays per listed pixel of Serial_Normal_Raster (0), Anti_Aliasing (1),
        Depth_Of_Field_Effect (2), in the reference's loop order"""
        S = {0: 1, 1: 4, 2: 49}[mode]
        cam14 = _f64(cam14); pix = np.ascontiguousarray(pix, np.uint32).reshape(-1, 2)
        o = np.zeros((len(pix) * S, 3)); d = np.zeros((len(pix) * S, 3))
        self.lib.rto_raster_rays(C.c_int(mode), _p(cam14), C.c_uint32(int(res[0])), C.c_uint32(int(res[1])), C.c_long(len(pix)),
                                 _p(pix), _p(o), _p(d))
        return o, d, S

    def hit_point(self, o, d, v1, nrm):
        o, d, v1, nrm = (_f64(x, (-1, 3)) for x in (o, d, v1, nrm))
        out = np.zeros_like(o)
        for k in range(len(o)):
            self.lib.rto_hit_point(_p(o[k]), _p(d[k]), _p(v1[k]), _p(nrm[k]), _p(out[k]))
        return out

    def shadow_rays(self, P, N, Cc):
        P = _f64(P, (-1, 3)); N = _f64(N, (-1, 3)); Cc = _f64(Cc)
        o = np.zeros_like(P); d = np.zeros_like(P)
        self.lib.rto_shadow_rays(C.c_long(len(P)), _p(P), _p(N), _p(Cc), _p(o), _p(d))
        return o, d

    def soft_shadow_rays(self, P, Cc, n_points=100):
        P = _f64(P, (-1, 3)); Cc = _f64(Cc)
        o = np.zeros((len(P) * n_points, 3)); d = np.zeros((len(P) * n_points, 3))
        self.lib.rto_soft_shadow_rays(C.c_long(len(P)), _p(P), _p(Cc), C.c_uint32(n_points), _p(o), _p(d))
        return o, d

    def ao_rays(self, P, N, R, spp, seed):
        P = _f64(P, (-1, 3)); N = _f64(N, (-1, 3))
        n = len(P)
        o = np.zeros((n * spp, 3)); d = np.zeros((n * spp, 3))
        self.lib.rto_ao_rays(C.c_long(n), _p(P), _p(N), C.c_double(R), C.c_uint32(spp), C.c_uint64(seed), _p(o), _p(d))
        return o, d

    def ao_rays_points(self, P, N, R, ab):
        P = _f64(P, (-1, 3)); N = _f64(N, (-1, 3)); ab = _f64(ab, (-1, 2))
        o = np.zeros_like(P); d = np.zeros_like(P)
        self.lib.rto_ao_rays_points(C.c_long(len(P)), _p(P), _p(N), C.c_double(R), _p(ab), _p(o), _p(d))
        return o, d

    def area(self, verts):
        verts = _f64(verts, (-1, 9))
        return np.array([self.lib.rto_area(_p(v[0:3]), _p(v[3:6]), _p(v[6:9])) for v in verts])

    def visible(self, verts, normals, ei, ej, xi, xj):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3))
        ei = np.ascontiguousarray(ei, np.int32); ej = np.ascontiguousarray(ej, np.int32)
        xi = _f64(xi, (-1, 3)); xj = _f64(xj, (-1, 3))
        vis = np.zeros(len(ei), np.uint8)
        self.lib.rto_visible(C.c_long(len(verts)), _p(verts), _p(normals), C.c_long(len(ei)), _p(ei), _p(ej), _p(xi),
                             _p(xj), _p(vis))
        return vis

    def light_blocked(self, verts, normals, ei, xi, Cc):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3))
        ei = np.ascontiguousarray(ei, np.int32); xi = _f64(xi, (-1, 3)); Cc = _f64(Cc)
        out = np.zeros(len(ei), np.uint8)
        self.lib.rto_light_blocked(C.c_long(len(verts)), _p(verts), _p(normals), C.c_long(len(ei)), _p(ei), _p(xi),
                                   _p(Cc), _p(out))
        return out

    def sample_point(self, verts9, seed, a, b, k, which):
        verts9 = _f64(verts9)
        p = np.zeros(3)
        self.lib.rto_sample_point(_p(verts9), C.c_uint64(seed), C.c_uint32(a), C.c_uint32(b), C.c_uint32(k),
                                  C.c_uint32(which), _p(p))
        return p

    def form_factors(self, verts, normals, samples, seed, row0=0, n_rows=None):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3))
        n = len(verts)
        n_rows = n - row0 if n_rows is None else n_rows
        F = np.zeros((n_rows, n))
        self.lib.rto_form_factors(C.c_long(n), _p(verts), _p(normals), C.c_uint32(samples), C.c_uint64(seed),
                                  C.c_long(row0), C.c_long(n_rows), C.c_long(n), _p(F))
        return F

    def form_factors_bvh(self, verts, normals, samples, seed, row0=0, n_rows=None, bvh=None):
        """form_factors with the closest hits taken through the oracle's own hierarchy (same records, for large n)"""
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3))
        n = len(verts)
        n_rows = n - row0 if n_rows is None else n_rows
        F = np.zeros((n_rows, n))
        h = self.bvh_build(verts) if bvh is None else bvh
        self.lib.rto_form_factors_bvh(h, C.c_long(n), _p(verts), _p(normals), C.c_uint32(samples), C.c_uint64(seed),
                                      C.c_long(row0), C.c_long(n_rows), C.c_long(n), _p(F))
        if bvh is None:
            self.bvh_free(h)
        return F

    def ff_term_points(self, verts, normals, i, j, xi, xj):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3)); xi = _f64(xi, (-1, 3)); xj = _f64(xj, (-1, 3))
        return self.lib.rto_ff_term_points(C.c_long(len(verts)), _p(verts), _p(normals), C.c_int32(i), C.c_int32(j),
                                           C.c_uint32(len(xi)), _p(xi), _p(xj))

    def light_terms(self, verts, normals, Cc, samples, seed, mirror_bug=1, row0=0, n_rows=None):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3)); Cc = _f64(Cc)
        n = len(verts)
        n_rows = n - row0 if n_rows is None else n_rows
        f = np.zeros(n_rows)
        self.lib.rto_light_terms(C.c_long(n), _p(verts), _p(normals), _p(Cc), C.c_uint32(samples), C.c_uint64(seed),
                                 C.c_int(mirror_bug), C.c_long(row0), C.c_long(n_rows), _p(f))
        return f

    def light_term_points(self, verts, normals, Cc, i, xi, mirror_bug=1):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3)); Cc = _f64(Cc); xi = _f64(xi, (-1, 3))
        return self.lib.rto_light_term_points(C.c_long(len(verts)), _p(verts), _p(normals), _p(Cc), C.c_int32(i),
                                              C.c_uint32(len(xi)), _p(xi), C.c_int(mirror_bug))

    def progressive_refinement(self, F, area, rho=0.70, power=68500.0, max_shots=1000000):
        F = _f64(F); n = F.shape[0]; area = _f64(area)
        B = np.zeros(n)
        shots = self.lib.rto_progressive_refinement(C.c_long(n), _p(F), _p(area), C.c_double(rho), C.c_double(power),
                                                    C.c_long(max_shots), _p(B))
        return B, shots

    def direct_solve(self, F, rho=0.70, power=68500.0):
        F = _f64(F); n = F.shape[0]
        B = np.zeros(n)
        rc = self.lib.rto_direct_solve(C.c_long(n), _p(F), C.c_double(rho), C.c_double(power), _p(B))
        assert rc == 0
        return B

    def jacobi_rgb(self, F32, rho3, E, tol_rel=1e-7, max_sweeps=10000):
        F32 = np.ascontiguousarray(F32, np.float32); n = F32.shape[0]
        rho3 = _f64(rho3); E = _f64(E, (n, 3))
        B = np.zeros((n, 3))
        s = self.lib.rto_jacobi_rgb(C.c_long(n), _p(F32), _p(rho3), _p(E), C.c_double(tol_rel), C.c_long(max_sweeps),
                                    _p(B))
        return B, s

    def trace(self, tris, normals, tri_object, mats18, lights6, o, d, max_depth=1, ao_rays=10, ao_radius=2.0, seed=1,
              generation=1, first_index=0, bvh=None, radiosity_branch=False):
        """rto_trace_flags: Scene::Trace + Shader::Shade on n rays -> n × 3 colours (radiosity_branch: shader.cpp:431-478)"""
        tris = _f64(tris, (-1, 9)); normals = _f64(normals, (-1, 3)); mats18 = _f64(mats18, (-1, 18))
        lights6 = _f64(lights6, (-1, 6)); o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        tob = np.ascontiguousarray(tri_object, np.uint32)
        rgb = np.zeros((len(o), 3))
        self.lib.rto_trace_flags(C.c_long(len(tris)), _p(tris), _p(normals), _p(tob), _p(mats18), _p(lights6),
                                 C.c_int(len(lights6)), bvh, C.c_uint32(max_depth), C.c_uint32(ao_rays), C.c_double(ao_radius),
                                 C.c_uint64(seed), C.c_uint32(1 if radiosity_branch else 0), C.c_long(len(o)), _p(o), _p(d),
                                 C.c_uint32(generation), C.c_uint64(first_index), _p(rgb))
        return rgb

    def tone_map(self, rgb):
        rgb = _f64(rgb, (-1, 3))
        out = np.zeros((len(rgb), 3), np.uint8)
        self.lib.rto_tone_map(C.c_long(len(rgb)), _p(rgb), _p(out))
        return out

    def radiosity_trace(self, verts, normals, vertex_rad, o, d):
        verts = _f64(verts, (-1, 9)); normals = _f64(normals, (-1, 3)); vr = _f64(vertex_rad, (-1, 3))
        o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        rgb = np.zeros((len(o), 3))
        self.lib.rto_radiosity_trace(C.c_long(len(verts)), _p(verts), _p(normals), _p(vr), C.c_long(len(o)), _p(o), _p(d), _p(rgb))
        return rgb

    def vertex_values(self, verts, B):
        verts = _f64(verts, (-1, 9)); B = _f64(B)
        out = np.zeros((len(verts), 3))
        self.lib.rto_vertex_values(C.c_long(len(verts)), _p(verts), _p(B), _p(out))
        return out

    def pinv(self, verts):
        verts = _f64(verts, (-1, 9))
        A = np.zeros((len(verts), 3)); B = np.zeros((len(verts), 3)); ok = np.zeros(len(verts), np.uint8)
        for i in range(len(verts)):
            ok[i] = self.lib.rto_pinv(_p(verts[i]), _p(A[i]), _p(B[i]))
        return A, B, ok

    def tone_map_radiosity(self, rgb):
        rgb = _f64(rgb, (-1, 3))
        out = np.zeros((len(rgb), 3), np.uint8)
        self.lib.rto_tone_map_radiosity(C.c_long(len(rgb)), _p(rgb), _p(out))
        return out

    def num_threads(self):
        return self.lib.rto_num_threads()

    def bvh_build(self, tris):
        tris = _f64(tris, (-1, 9))
        return C.c_void_p(self.lib.rto_bvh_build(C.c_long(len(tris)), _p(tris)))

    def bvh_free(self, h):
        self.lib.rto_bvh_free(h)

    def bvh_closest_hit(self, h, o, d, tmax=None):
        o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        n = len(o)
        tm = _f64(tmax) if tmax is not None else None
        tri = np.zeros(n, np.int32); t = np.zeros(n); u = np.zeros(n); v = np.zeros(n)
        secs = self.lib.rto_bvh_closest_hit(h, C.c_long(n), _p(o), _p(d), _p(tm), _p(tri), _p(t), _p(u), _p(v))
        return (tri, t, u, v), secs


class Reference:
    """ref_driver.cpp over the compiled reference (libref_render2.so)."""

    def __init__(self, path=REF_SO):
        self.lib = L = C.CDLL(path)
        for f in ("ref_scene_load", "ref_scene_from_triangles", "ref_scene_one_tri_objects"):
            getattr(L, f).restype = C.c_void_p
        for f in ("ref_cast", "ref_brute", "ref_trace", "ref_rad_ff_term", "ref_rad_ff_light_term", "ref_rad_find_form_factor",
                  "ref_rad_solve", "ref_rad_residual"):
            getattr(L, f).restype = C.c_double
        L.ref_num_triangles.restype = C.c_long

    @staticmethod
    def available(path=REF_SO):
        return os.path.exists(path)

    def mt_batch(self, o, d, v0, v1, v2):
        o, d, v0, v1, v2 = (_f64(x, (-1, 3)) for x in (o, d, v0, v1, v2))
        n = len(o)
        t, u, v = np.zeros(n), np.zeros(n), np.zeros(n)
        self.lib.ref_mt_batch(C.c_long(n), _p(o), _p(d), _p(v0), _p(v1), _p(v2), _p(t), _p(u), _p(v))
        return t, u, v

    def scene_load(self, workdir, stem):
        h = self.lib.ref_scene_load(workdir.encode(), stem.encode())
        if not h:
            raise RuntimeError("reference BuildScene failed for %s" % stem)
        return C.c_void_p(h)

    def scene_from_triangles(self, tris, obj=None, light=None):
        tris = _f64(tris, (-1, 9))
        n = len(tris)
        obj = np.zeros(n, np.uint32) if obj is None else np.ascontiguousarray(obj, np.uint32)
        n_obj = int(obj.max()) + 1 if n else 1
        lt = _f64(light) if light is not None else None
        return C.c_void_p(self.lib.ref_scene_from_triangles(_p(tris), _p(obj), C.c_long(n), C.c_int(n_obj), _p(lt)))

    def scene_one_tri_objects(self, tris, normals, mat_of, mats18, lights6):
        """n one-triangle Objects (Scene::Cast = brute force over the original triangles), see ref_driver.cpp"""
        tris = _f64(tris, (-1, 9)); mats18 = _f64(mats18, (-1, 18)); lights6 = _f64(lights6, (-1, 6))
        nr = _f64(normals, (-1, 3)) if normals is not None else None
        mo = np.ascontiguousarray(mat_of, np.uint32)
        return C.c_void_p(self.lib.ref_scene_one_tri_objects(_p(tris), _p(nr), C.c_long(len(tris)), _p(mo), _p(mats18),
                                                             C.c_int(len(mats18)), _p(lights6), C.c_int(len(lights6))))

    def materials(self, h):
        out = np.zeros((self.lib.ref_num_objects(h), 18))
        self.lib.ref_get_materials(h, _p(out))
        return out

    def lights(self, h):
        out = np.zeros((self.lib.ref_num_lights(h), 6))
        self.lib.ref_get_lights(h, _p(out))
        return out

    def set_materials(self, h, mats18):
        mats18 = _f64(mats18, (-1, 18))
        assert len(mats18) == self.lib.ref_num_objects(h)
        self.lib.ref_set_materials(h, _p(mats18))

    def set_camera(self, h, cam14, res):
        self.lib.ref_set_camera14(h, _p(_f64(cam14)), C.c_uint32(int(res[0])), C.c_uint32(int(res[1])))

    def raster(self, h, mode, workdir, res):
        """the reference's Normal_Raster (0) / Anti_Aliasing (1) / Depth_Of_Field_Effect (2) -> y x x x 3 ToneMap bytes"""
        px = np.zeros((int(res[1]), int(res[0]), 3), np.uint8)
        rc = self.lib.ref_raster(h, C.c_int(mode), workdir.encode(), _p(px))
        if rc != 0:
            raise RuntimeError("ref_raster: x_res must divide by 8 and y_res by 6")
        return px

    def set_max_depth(self, h, depth):
        self.lib.ref_set_max_depth(h, C.c_uint(depth))

    def trace(self, h, o, d, generation=1):
        """Scene::Trace per ray -> (n × 3 colours, seconds)"""
        o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        rgb = np.zeros((len(o), 3))
        secs = self.lib.ref_trace(h, C.c_long(len(o)), _p(o), _p(d), C.c_uint(generation), _p(rgb))
        return rgb, secs

    def tone_map(self, rgb):
        rgb = _f64(rgb, (-1, 3))
        out = np.zeros((len(rgb), 3), np.uint8)
        self.lib.ref_tone_map(C.c_long(len(rgb)), _p(rgb), _p(out))
        return out

    def build_bsp(self, h, workdir):
        return self.lib.ref_build_bsp(h, workdir.encode())

    def triangles(self, h):
        n = self.lib.ref_num_triangles(h)
        xyz = np.zeros((n, 9)); obj = np.zeros(n, np.uint32); nrm = np.zeros((n, 3))
        self.lib.ref_get_triangles(h, _p(xyz), _p(obj), _p(nrm))
        return xyz, obj, nrm

    def camera(self, h):
        c = np.zeros(14); res = np.zeros(2, np.uint32)
        self.lib.ref_get_camera(h, _p(c), _p(res))
        return c, res

    def light_center(self, h):
        c = np.zeros(3)
        if self.lib.ref_light_center(h, _p(c)) != 0:
            return None
        return c

    def raster_details(self, h):
        out = np.zeros(18)
        self.lib.ref_raster_details(h, _p(out))
        return out

    def primary_dirs(self, rd18, pix):
        rd18 = _f64(rd18); pix = np.ascontiguousarray(pix, np.uint32).reshape(-1, 2)
        d = np.zeros((len(pix), 3))
        self.lib.ref_primary_dirs(_p(rd18), C.c_long(len(pix)), _p(pix), _p(d))
        return d

    def cast(self, h, o, d):
        o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3)); n = len(o)
        hit = np.zeros(n, np.uint8); dist = np.zeros(n); pt = np.zeros((n, 3)); nr = np.zeros((n, 3))
        ob = np.zeros(n, np.int32); bu = np.zeros(n, np.float32); bv = np.zeros(n, np.float32)
        secs = self.lib.ref_cast(h, C.c_long(n), _p(o), _p(d), _p(hit), _p(dist), _p(pt), _p(nr), _p(ob), _p(bu), _p(bv))
        return dict(hit=hit, dist=dist, point=pt, normal=nr, obj=ob, bu=bu, bv=bv, seconds=secs)

    def ao_occlusion(self, h, P, N, radius, nrays):
        P = _f64(P, (-1, 3)); N = _f64(N, (-1, 3))
        out = np.zeros(len(P))
        self.lib.ref_ao_occlusion(h, C.c_long(len(P)), _p(P), _p(N), C.c_double(radius), C.c_int(nrays), _p(out))
        return out

    def soft_shadow(self, h, P, obj_index):
        """Shader::generateSoftShadows at n surface points whose hit.object is sceneObjects[obj_index[k]] -> the factor
        the reference multiplies the colour by"""
        P = _f64(P, (-1, 3)); oi = np.ascontiguousarray(obj_index, np.int32)
        out = np.zeros(len(P))
        self.lib.ref_soft_shadow(h, C.c_long(len(P)), _p(P), _p(oi), _p(out))
        return out

    def brute(self, h, o, d):
        o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3)); n = len(o)
        tri = np.zeros(n, np.int32); t = np.zeros(n); u = np.zeros(n); v = np.zeros(n)
        secs = self.lib.ref_brute(h, C.c_long(n), _p(o), _p(d), _p(tri), _p(t), _p(u), _p(v))
        return dict(tri=tri, t=t, u=u, v=v, seconds=secs)

    def shadow_rays(self, P, N, Cc):
        P = _f64(P, (-1, 3)); N = _f64(N, (-1, 3)); Cc = _f64(Cc)
        o = np.zeros_like(P); d = np.zeros_like(P)
        self.lib.ref_shadow_rays(C.c_long(len(P)), _p(P), _p(N), _p(Cc), _p(o), _p(d))
        return o, d

    def rad_setup(self, h):
        return self.lib.ref_rad_setup(h)

    def rad_attach(self, h, mode):
        """scene.radiosity for Shader::Shade's radiosity branch: 0 = NULL, 1 = the Radiosity of rad_setup, 2 = a bare one."""
        if self.lib.ref_rad_attach(h, C.c_int(mode)) != 0:
            raise RuntimeError("ref_rad_attach: no Radiosity to attach (rad_setup first)")

    def rad_elements(self, h, n_el):
        verts = np.zeros((n_el, 9)); nrm = np.zeros((n_el, 3)); obj = np.zeros(n_el, np.int32)
        self.lib.ref_rad_get_elements(h, _p(verts), _p(nrm), _p(obj))
        return verts, nrm, obj

    def rad_visible(self, h, ei, ej, xi, xj):
        ei = np.ascontiguousarray(ei, np.int32); ej = np.ascontiguousarray(ej, np.int32)
        xi = _f64(xi, (-1, 3)); xj = _f64(xj, (-1, 3))
        vis = np.zeros(len(ei), np.uint8)
        self.lib.ref_rad_visible(h, C.c_long(len(ei)), _p(ei), _p(ej), _p(xi), _p(xj), _p(vis))
        return vis

    def rad_light_blocked(self, h, ei, xi):
        ei = np.ascontiguousarray(ei, np.int32); xi = _f64(xi, (-1, 3))
        out = np.zeros(len(ei), np.uint8)
        self.lib.ref_rad_light_blocked(h, C.c_long(len(ei)), _p(ei), _p(xi), _p(out))
        return out

    def rad_ff_term(self, h, i, j):
        return self.lib.ref_rad_ff_term(h, C.c_int(i), C.c_int(j))

    def rad_ff_light_term(self, h, i):
        return self.lib.ref_rad_ff_light_term(h, C.c_int(i))

    def rad_find_form_factor(self, h):
        return self.lib.ref_rad_find_form_factor(h)

    def rad_n(self, h):
        return self.lib.ref_rad_n(h)

    def rad_get_FF(self, h):
        n = self.rad_n(h)
        F = np.zeros((n, n))
        self.lib.ref_rad_get_FF(h, _p(F))
        return F

    def rad_set_FF(self, h, F):
        F = _f64(F)
        self.lib.ref_rad_set_FF(h, _p(F))

    def rad_solve(self, h):
        return self.lib.ref_rad_solve(h)

    def rad_B(self, h):
        B = np.zeros(self.rad_n(h))
        self.lib.ref_rad_get_B(h, _p(B))
        return B

    def rad_scatter_B(self, h, n_el):
        out = np.zeros(n_el)
        self.lib.ref_rad_scatter_B(h, _p(out))
        return out

    def rad_set_element_radiosity(self, h, B):
        self.lib.ref_rad_set_element_radiosity(h, _p(_f64(B)))

    def rad_trace_ray(self, h, o, d):
        """Radiosity_Helper::trace_ray per ray -> n x 3"""
        o = _f64(o, (-1, 3)); d = _f64(d, (-1, 3))
        rgb = np.zeros((len(o), 3))
        self.lib.ref_rad_trace_ray(h, C.c_long(len(o)), _p(o), _p(d), _p(rgb))
        return rgb

    def rad_set_adjacency(self, h, ei, ej, pts):
        ei, ej, pts = np.ascontiguousarray(ei, np.int32), np.ascontiguousarray(ej, np.int32), _f64(pts)
        self.lib.ref_rad_set_adjacency(h, C.c_long(len(ei)), _p(ei), _p(ej), _p(pts))

    def rad_vertex_radiosity(self, h, n_el):
        out = np.zeros((n_el, 3))
        self.lib.ref_rad_vertex_radiosity(h, _p(out))
        return out

    def tone_map_radiosity(self, rgb):
        rgb = _f64(rgb, (-1, 3))
        out = np.zeros((len(rgb), 3), np.uint8)
        self.lib.ref_tone_map_radiosity(C.c_long(len(rgb)), _p(rgb), _p(out))
        return out

    def rad_residual(self, h):
        return self.lib.ref_rad_residual(h)
