"""Drop-in proof on the GPU box: the product's C++ host glue (include/rtmi_adapter.hpp) instantiated with the
COMPILED REFERENCE's own Scene / Ray / HitInfo / QuadTreeNode types (oracle/ref_adapter.cpp, built into
oracle/_ref/libref_adapter.so where /root/reference exists) runs beside the reference's own Scene::Cast and
Radiosity on the same objects.  The scenes are rebuilt from the golden fixtures' triangle arrays — the reference's
scene files do not travel."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
ADAPTER_SO = os.path.join(ROOT, "oracle", "_ref", "libref_adapter.so")


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@pytest.fixture(scope="module")
def ref():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    if not os.path.exists(ADAPTER_SO):
        pytest.skip("oracle/_ref/libref_adapter.so not built (needs /root/reference at build time)")
    from oracle_py import Reference
    r = Reference(ADAPTER_SO)
    r.lib.ref_adapter_bind.restype = C.c_void_p
    r.lib.ref_adapter_error.restype = C.c_char_p
    return r


def adapter_cast(ref, h, b, o, d):
    n = len(o)
    hit = np.zeros(n, np.uint8); dist = np.zeros(n); pt = np.zeros((n, 3)); nr = np.zeros((n, 3))
    ob = np.zeros(n, np.int32); bu = np.zeros(n, np.float32); bv = np.zeros(n, np.float32); tri = np.zeros(n, np.int32)
    rc = ref.lib.ref_adapter_cast(h, b, C.c_long(n), _p(o), _p(d), _p(hit), _p(dist), _p(pt), _p(nr), _p(ob), _p(bu), _p(bv), _p(tri))
    assert rc == 0, ref.lib.ref_adapter_error()
    return dict(hit=hit, dist=dist, point=pt, normal=nr, obj=ob, bu=bu, bv=bv, tri=tri)


@pytest.mark.parametrize("stem", ["test1", "test5", "scene_room_bed_sample"])
def test_adapter_cast_fills_the_references_hitinfo(ref, oracle, stem, tmp_path):
    g = load_golden("g2_%s.npz" % stem)
    h = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["light_center"])
    b = C.c_void_p(ref.lib.ref_adapter_bind(h))
    assert b, ref.lib.ref_adapter_error()
    rd = oracle.raster_details(g["cam"], g["res"][0], g["res"][1])
    pix = g["pix"][::3]
    d = np.ascontiguousarray(oracle.primary_dirs(rd, pix))
    o = np.ascontiguousarray(np.tile(g["cam"][:3], (len(d), 1)))
    a = adapter_cast(ref, h, b, o, d)
    # (1) against the golden vectors of the reference's own triangle_intersection brute force: exact
    tri = g["tri"][::3]
    assert np.array_equal(a["hit"].astype(bool), tri >= 0)
    assert np.array_equal(a["tri"], tri)
    assert np.array_equal(a["dist"], g["t"][::3])
    assert np.array_equal(a["bu"], g["u"][::3].astype(np.float32)) and np.array_equal(a["bv"], g["v"][::3].astype(np.float32))
    hm = tri >= 0
    assert np.array_equal(a["obj"][hm], g["tri_obj"][tri[hm]].astype(np.int32))
    assert np.array_equal(a["normal"][hm], g["tri_normal"][tri[hm]]) or stem == "scene_room_bed_sample"
    # hit.point by the plane formula of object.cpp:23 — bit-exact against the oracle's restatement with the scene's normals
    xyz, _, nrm = ref.triangles(h)
    p = oracle.hit_point(o[hm], d[hm], xyz[tri[hm], 0:3], nrm[tri[hm]])
    assert np.array_equal(a["point"][hm], p)
    # (2) beside the reference's own Scene::Cast on the same Scene object (its BSP of split fragments, rebuilt with
    #     an unseeded RNG on every run): same hit/miss and object, distance and point to rounding
    if ref.build_bsp(h, str(tmp_path)) == 0:
        c = ref.cast(h, o, d)
        assert np.array_equal(c["hit"], a["hit"])
        assert np.array_equal(c["obj"][hm], a["obj"][hm])
        assert np.max(np.abs(c["dist"][hm] - a["dist"][hm]) / a["dist"][hm]) < 1e-12
        assert np.max(np.abs(c["point"][hm] - a["point"][hm])) < 1e-9
        assert np.max(np.abs(c["normal"][hm] - a["normal"][hm])) < 1e-12
    # (3) the per-ray signature bool Scene::Cast(const Ray&, HitInfo&)
    for k in (0, len(d) // 2, len(d) - 1):
        dist = C.c_double(0)
        f = ref.lib.ref_adapter_cast_one(b, _p(o[k]), _p(d[k]), C.byref(dist))
        assert bool(f) == bool(a["hit"][k]) and dist.value == a["dist"][k]
    ref.lib.ref_adapter_unbind(b)


def test_adapter_shadow_predicate(ref):
    g = load_golden("g2_test1.npz")
    h = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["light_center"])
    b = C.c_void_p(ref.lib.ref_adapter_bind(h))
    so = np.ascontiguousarray(g["shadow_o"]); sd = np.ascontiguousarray(g["shadow_d"])
    occ = np.zeros(len(so), np.uint8)
    assert ref.lib.ref_adapter_occluded(b, C.c_long(len(so)), _p(so), _p(sd), None, _p(occ)) == 0
    assert np.array_equal(occ, g["shadow_occluded"])                      # shader.cpp:131, from the reference itself
    ao_o = np.ascontiguousarray(g["ao_o"]); ao_d = np.ascontiguousarray(g["ao_d"]); tm = np.full(len(ao_o), 2.0)
    assert ref.lib.ref_adapter_occluded(b, C.c_long(len(ao_o)), _p(ao_o), _p(ao_d), _p(tm), _p(occ)) == 0
    assert np.array_equal(occ, g["ao_occluded"])                          # shader.cpp:226
    ref.lib.ref_adapter_unbind(b)


def test_adapter_radiosity_on_the_references_elements(ref, oracle, tmp_path):
    """The reference builds its own element list (BSP leaves, radiosity.cpp:137-163); the adapter computes F and B on
    the GPU and writes QuadTreeNode::radiosityValue; the reference's own progressiveRefinement, run on that same F,
    must land on the same B."""
    g = load_golden("g2_test1.npz")
    h = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["light_center"])
    if ref.build_bsp(h, str(tmp_path)) != 0:
        pytest.skip("the reference's BSP build threw on this run (unseeded RNG)")
    ne = ref.rad_setup(h)
    assert ne >= len(g["tris"])
    verts, nrm, _ = ref.rad_elements(h, ne)
    n = ne + 1
    FF = np.zeros((n, n)); B = np.zeros(n); er = np.zeros(ne); sweeps = C.c_uint32(0)
    lc = np.ascontiguousarray(g["light_center"])
    rc = ref.lib.ref_adapter_radiosity(h, _p(lc), C.c_uint32(10), C.c_uint32(30), C.c_uint64(5), _p(FF), _p(B), _p(er), C.byref(sweeps))
    assert rc == n, ref.lib.ref_adapter_error()
    assert np.array_equal(er, B[:ne]) and B[ne] == 68500.0                # setRadiosityForAllElements
    # F is what the oracle's restatement of radiosity.cpp:305-333 / 264-303 gives on the same elements and seed
    F2 = oracle.form_factors(verts, nrm, 10, 5)
    fl = oracle.light_terms(verts, nrm, lc, 30, 5, mirror_bug=1)
    assert np.array_equal(FF[:ne, :ne], F2.astype(np.float32).astype(np.float64))
    assert np.array_equal(FF[:ne, ne], fl.astype(np.float32).astype(np.float64))
    # the reference's own solver on that F
    ref.rad_find_form_factor(h)            # allocates the reference's FF / K / B globals (its F is random: overwritten)
    ref.rad_set_FF(h, FF)
    ref.rad_solve(h)
    Bref = ref.rad_B(h)
    assert ref.rad_residual(h) <= 1e-5
    assert np.linalg.norm(B[:ne] - Bref[:ne]) / np.linalg.norm(Bref[:ne]) < 1e-4


def adapter_rasterize(ref, h, b, res, ao_rays, seed, o, d):
    px = np.zeros((int(res[1]), int(res[0]), 3), np.uint8)
    rgb = np.zeros((len(o), 3))
    rc = ref.lib.ref_adapter_rasterize(h, b, C.c_uint32(ao_rays), C.c_uint64(seed), _p(px), C.c_long(len(o)), _p(o), _p(d), _p(rgb))
    assert rc == 0, ref.lib.ref_adapter_error()
    return px, rgb


def test_adapter_rasterize_beside_the_references_trace(ref, oracle, tmp_path):
    """rtmi_host::ShaderBinding on the reference's own Scene / Camera / PPM_Image beside its own Scene::Trace + ToneMap
    loop (rasterizer.cpp:219-233) on the same objects.
    (1) test1 as shipped, through the reference's BSP: on that scene no AO ray is ever occluded, so the reference's
        unseeded draw cannot show and the two images agree byte for byte;
    (2) the studio scene (every Shade branch, max_tree_depth 3, ambient 0): colours within 1e-11, bytes equal except
        where 256·c sits on an integer."""
    from conftest import frame_rays
    g = load_golden("g6_shade_test1.npz")
    h = None
    for _ in range(10):
        h = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["lights"][0, :3])
        if ref.build_bsp(h, str(tmp_path)) == 0:
            break
    else:
        pytest.skip("the reference's BSP build kept throwing")
    ref.set_materials(h, g["mats"])
    ref.set_camera(h, g["cam14"], g["res"])
    b = C.c_void_p(ref.lib.ref_adapter_bind(h))
    assert b, ref.lib.ref_adapter_error()
    o, d, _ = frame_rays(oracle, g["cam14"], g["res"])
    px, probe = adapter_rasterize(ref, h, b, g["res"], 10, 1, o[:5000], d[:5000])
    want, _ = ref.trace(h, o, d)
    assert np.array_equal(px.reshape(-1, 3), ref.tone_map(want))
    assert np.array_equal(px.reshape(-1, 3), g["pixels"])
    assert np.abs(probe - want[:5000]).max() < 1e-11

    s = load_golden("g6_shade_studio.npz")
    h2 = ref.scene_one_tri_objects(s["tris"], s["normals"], s["tri_obj"], s["mats"], s["lights"])
    ref.set_max_depth(h2, 3)
    ref.set_camera(h2, s["cam14"], s["res"])
    b2 = C.c_void_p(ref.lib.ref_adapter_bind(h2))
    assert b2, ref.lib.ref_adapter_error()
    o, d, _ = frame_rays(oracle, s["cam14"], s["res"])
    px2, probe2 = adapter_rasterize(ref, h2, b2, s["res"], 10, 1, o, d)
    assert np.abs(probe2 - s["rgb_d3"]).max() < 1e-11
    wp = ref.tone_map(s["rgb_d3"])
    edge = np.abs(256 * s["rgb_d3"] - np.round(256 * s["rgb_d3"])) < 1e-8
    assert not ((px2.reshape(-1, 3) != wp) & ~edge).any()

    # (3) the same Scene with a Radiosity attached: Shade takes its radiosity branch (shader.cpp:431-478), the binding sees
    #     scene.radiosity != NULL and asks for the same branch; colours beside the reference's own Trace on that object
    ref.rad_attach(h2, 2)
    b3 = C.c_void_p(ref.lib.ref_adapter_bind(h2))
    px3, probe3 = adapter_rasterize(ref, h2, b3, s["res"], 10, 1, o[:6000], d[:6000])
    want3, _ = ref.trace(h2, o[:6000], d[:6000])
    ref.rad_attach(h2, 0)
    assert np.abs(probe3 - want3).max() < 1e-11
    assert not np.array_equal(px3, px2)


def test_adapter_radiosity_raster_beside_the_references_trace_ray(ref, oracle, tmp_path):
    """rtmi_host::RadiosityBinding::Radiosity_Raster on the reference's own QuadTreeNode list, Camera and PPM_Image beside
    its own ToneMapRadiosity(trace_ray(ray)) loop (rasterizer.cpp:436-455): the same bytes on every pixel"""
    from conftest import frame_rays
    g = load_golden("g8_radiosity_display.npz")
    gs = np.load(os.path.join(ROOT, "tests", "golden", "g7_scene_files.npz"))
    stem = "practice_room_enclosed"
    tris, tobj = gs[stem + "/tris"], gs[stem + "/tri_obj"]
    h = None
    for _ in range(10):
        h = ref.scene_from_triangles(tris, tobj, gs[stem + "/lights"][0, :3])
        if ref.build_bsp(h, str(tmp_path)) == 0:
            break
    else:
        pytest.skip("the reference's BSP build kept throwing")
    cam14, res = g[stem + "/cam14"], (120, 90)
    ref.set_camera(h, cam14, res)
    ne = ref.rad_setup(h)
    B = 0.05 + 4.0 * ((np.arange(ne) * 7919) % ne) / ne
    ref.rad_set_element_radiosity(h, B)
    px = np.zeros((res[1], res[0], 3), np.uint8)
    assert ref.lib.ref_adapter_radiosity_raster(h, _p(px)) == 0, ref.lib.ref_adapter_error()
    o, d, _ = frame_rays(oracle, cam14, res)
    want = ref.tone_map_radiosity(ref.rad_trace_ray(h, o, d))
    assert np.array_equal(px.reshape(-1, 3), want)
    assert (want[:, 0] != want[:, 1]).any() and len(np.unique(want[:, 0])) > 20     # Green pixels and a range of greys
