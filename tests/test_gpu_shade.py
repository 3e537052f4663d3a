"""Scene::Trace + Shader::Shade on the device (rtmi_trace_dev / rtmi_render_frame_dev) against the oracle's
restatement (same seeded AO stream: hit ids, shadow and occlusion bits identical; colours within TOL because pow / exp /
sin / cos come from the device library) and against the compiled reference's own colours (tests/golden/g6_*)."""
import numpy as np
import pytest

from conftest import load_golden, frame_rays, ao_steps

pytestmark = pytest.mark.gpu

# floating point: the device's pow/exp/sin/cos differ from glibc's by a few ulp; colours are O(1)
TOL = 1e-11


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    import raytracer_amd
    return raytracer_amd


def _shader(rt, g, mats="mats", **kw):
    scene = rt.Scene(g["tris"], g["tri_obj"])
    return scene, rt.Shader(scene, g["normals"], g[mats], g["lights"], **kw)


@pytest.mark.parametrize("depth", [1, 3])
def test_studio_vs_reference_golden(rt, oracle, depth):
    """ambient 0: the compiled reference's colours are reproducible; every branch of Shade is on the frame"""
    g = load_golden("g6_shade_studio.npz")
    scene, sh = _shader(rt, g, max_tree_depth=depth, ao_rays=10)
    cam = rt.Camera.from_cam14(g["cam14"], g["res"])
    rgb, pix = sh.Rasterize(cam)
    ref = g["rgb_d%d" % depth].reshape(rgb.shape)
    assert np.abs(rgb - ref).max() < TOL
    if depth == 1:
        # ToneMap bytes: identical except where 256·c sits within TOL of an integer
        diff = (pix != g["pixels_d1"].reshape(pix.shape))
        edge = np.abs(256 * ref - np.round(256 * ref)) < 256 * TOL
        assert not (diff & ~edge).any()


def test_studio_ambient_same_stream_as_oracle(rt, oracle):
    """ambient 0.1, 10 AO rays, depth 3: same seeded stream as the oracle -> same occlusion counts -> colours within TOL;
    also in two halves of the tiles (multi-GPU sharding) and through rtmi_trace_dev on a ray batch"""
    g = load_golden("g6_shade_studio.npz")
    o, d, _ = frame_rays(oracle, g["cam14"], g["res"])
    want = oracle.trace(g["tris"], g["normals"], g["tri_obj"], g["mats_amb"], g["lights"], o, d, max_depth=3, ao_rays=10, seed=5)
    scene, sh = _shader(rt, g, "mats_amb", max_tree_depth=3, ao_rays=10, seed=5)
    cam = rt.Camera.from_cam14(g["cam14"], g["res"])
    rgb, _ = sh.Rasterize(cam)
    assert np.abs(rgb.reshape(-1, 3) - want).max() < TOL
    # tiles 32 x 24, rank 0 of 2 and rank 1 of 2: the union is the frame, each pixel written once
    a, _ = sh.Rasterize(cam, tile=(32, 24), tile_first=0, tile_stride=2)
    b, _ = sh.Rasterize(cam, tile=(32, 24), tile_first=1, tile_stride=2)
    assert np.array_equal(np.where(a != 0, a, b), rgb) and not ((a != 0) & (b != 0)).any()
    # ragged tiles (not multiples of 8, not dividing the frame)
    c, _ = sh.Rasterize(cam, tile=(50, 37))
    assert np.array_equal(c, rgb)
    # the same rays as a batch, in two calls with first_index
    rays = rt.make_rays(o, d)
    h = len(rays) // 3
    got = np.concatenate([sh.Trace(rays[:h]), sh.Trace(rays[h:], first_index=h)])
    assert np.array_equal(got, rgb.reshape(-1, 3))
    # the reference's colours: same up to its own random occlusion count (integer steps)
    occ1 = oracle.trace(g["tris"], g["normals"], g["tri_obj"], g["mats_amb"], g["lights"], o, d, max_depth=1, ao_rays=0)
    sh1 = rt.Shader(scene, g["normals"], g["mats_amb"], g["lights"], max_tree_depth=1, ao_rays=10, seed=9)
    mine, _ = sh1.Rasterize(cam)
    k_ref = 10 * ao_steps(g["rgb_amb"], occ1, g["tris"], g["tri_obj"], g["mats_amb"], g["lights"], oracle, o, d)
    k_gpu = 10 * ao_steps(mine.reshape(-1, 3), occ1, g["tris"], g["tri_obj"], g["mats_amb"], g["lights"], oracle, o, d)
    lit = ~np.isnan(k_ref[:, 0]) & (np.abs(occ1 - 0.07).max(1) > 1e-12)
    assert np.abs(k_gpu[lit] - np.round(k_gpu[lit])).max() < 1e-6
    assert abs(k_gpu[lit, 0].mean() - k_ref[lit, 0].mean()) < 0.1


def test_generation_beyond_depth_and_small_workspace(rt, oracle, monkeypatch):
    g = load_golden("g6_shade_studio.npz")
    o, d, _ = frame_rays(oracle, g["cam14"], g["res"])
    scene, sh = _shader(rt, g, max_tree_depth=2, ao_rays=0)
    rays = rt.make_rays(o[:1000], d[:1000])
    assert np.array_equal(sh.Trace(rays, generation=3), np.zeros((1000, 3)))     # scene.cpp:134-137: termination colour
    want = oracle.trace(g["tris"], g["normals"], g["tri_obj"], g["mats"], g["lights"], o[:1000], d[:1000], max_depth=2, ao_rays=0,
                        generation=2)
    assert np.abs(sh.Trace(rays, generation=2) - want).max() < TOL
    # a 1 MiB workspace forces many chunks; the frame must not change
    cam = rt.Camera.from_cam14(g["cam14"], g["res"])
    full, _ = sh.Rasterize(cam)
    monkeypatch.setenv("RTMI_SHADE_WS_MB", "1")
    scene2, sh2 = _shader(rt, g, max_tree_depth=2, ao_rays=0)
    small, _ = sh2.Rasterize(cam)
    assert np.array_equal(full, small)
    # the same with ambient occlusion on (AO frames and occlusion bytes are carved from the workspace per chunk; the
    # sample streams are keyed by the ray, not by the slot, so chunks must not change a pixel)
    monkeypatch.delenv("RTMI_SHADE_WS_MB")
    scene3, sh3 = _shader(rt, g, "mats_amb", max_tree_depth=3, ao_rays=10, seed=5)
    full_ao, _ = sh3.Rasterize(cam)
    monkeypatch.setenv("RTMI_SHADE_WS_MB", "1")
    scene4, sh4 = _shader(rt, g, "mats_amb", max_tree_depth=3, ao_rays=10, seed=5)
    small_ao, _ = sh4.Rasterize(cam)
    assert np.array_equal(full_ao, small_ao) and not np.array_equal(full_ao, full)


def test_test1_frame_as_shipped(rt, oracle):
    """scenes/test1 at the reference's settings (500 x 400, ambient 0.1, 10 AO rays, depth 1)"""
    g = load_golden("g6_shade_test1.npz")
    scene, sh = _shader(rt, g, max_tree_depth=1, ao_rays=10, seed=1)
    cam = rt.Camera.from_cam14(g["cam14"], g["res"])
    rgb, pix = sh.Rasterize(cam)
    rgb = rgb.reshape(-1, 3)
    o, d, _ = frame_rays(oracle, g["cam14"], g["res"])
    want = oracle.trace(g["tris"], g["normals"], g["tri_obj"], g["mats"], g["lights"], o, d, max_depth=1, ao_rays=10, seed=1)
    assert np.abs(rgb - want).max() < TOL
    wp = oracle.tone_map(want)
    edge = (np.abs(256 * want - np.round(256 * want)) < 256 * TOL)
    assert not ((pix.reshape(-1, 3) != wp) & ~edge).any()
    # against the reference's own frame: equal where AO cannot act, integer occlusion steps elsewhere
    sub = g["sub"]
    occ1 = oracle.trace(g["tris"], g["normals"], g["tri_obj"], g["mats"], g["lights"], o[sub], d[sub], max_depth=1, ao_rays=0)
    k_ref = 10 * ao_steps(g["rgb_sub"], occ1, g["tris"], g["tri_obj"], g["mats"], g["lights"], oracle, o[sub], d[sub])
    k_gpu = 10 * ao_steps(rgb[sub], occ1, g["tris"], g["tri_obj"], g["mats"], g["lights"], oracle, o[sub], d[sub])
    flat = np.isnan(k_ref[:, 0])
    assert np.abs(rgb[sub][flat] - g["rgb_sub"][flat]).max() < 1e-9
    lit = ~flat & (np.abs(occ1 - 0.07).max(1) > 1e-12)
    assert abs(k_gpu[lit, 0].mean() - k_ref[lit, 0].mean()) < 0.1
    # the image as a whole: the reference's ToneMap bytes differ from ours only through those occlusion steps
    dpx = np.abs(pix.reshape(-1, 3).astype(int) - g["pixels"].astype(int))
    assert dpx.max() <= int(256 * 0.85 * 0.1) + 1


def test_radiosity_branch_of_shade(rt, oracle):
    """RTMI_SHADE_RADIOSITY_BRANCH = Shade with scene.radiosity != NULL (shader.cpp:431-478) against the compiled reference's
    own colours (tests/golden/g9_shade_radiosity.npz): whole studio frames at depth 1 and 3 (the branch never recurses),
    ToneMap bytes, a ray batch through rtmi_trace_dev, tiles of two ranks, and test1 with the reference's Radiosity."""
    g = load_golden("g9_shade_radiosity.npz")
    scene = rt.Scene(g["studio_tris"], g["studio_tri_obj"])
    cam = rt.Camera.from_cam14(g["studio_cam14"], g["studio_res"])
    frames = []
    for depth in (1, 3):
        sh = rt.Shader(scene, g["studio_normals"], g["studio_mats"], g["studio_lights"], max_tree_depth=depth, ao_rays=10, radiosity_branch=True)
        rgb, pix = sh.Rasterize(cam)
        ref = g["studio_rgb_d%d" % depth]
        assert np.abs(rgb.reshape(-1, 3) - ref).max() < TOL
        edge = np.abs(256 * ref - np.round(256 * ref)) < 256 * TOL
        assert not ((pix.reshape(-1, 3) != oracle.tone_map(ref)) & ~edge).any()
        frames.append(rgb)
    assert np.array_equal(frames[0], frames[1])
    a, _ = sh.Rasterize(cam, tile=(32, 24), tile_first=0, tile_stride=2)
    b, _ = sh.Rasterize(cam, tile=(32, 24), tile_first=1, tile_stride=2)
    assert np.array_equal(np.where(a != 0, a, b), frames[1]) and not ((a != 0) & (b != 0)).any()
    o, d, _ = frame_rays(oracle, g["studio_cam14"], g["studio_res"])
    assert np.array_equal(sh.Trace(rt.make_rays(o[:3000], d[:3000])), frames[1].reshape(-1, 3)[:3000])
    assert np.array_equal(sh.Trace(rt.make_rays(o[:100], d[:100]), generation=4), np.zeros((100, 3)))   # scene.cpp:134-137 comes first
    # the other branch on the same shader inputs gives another picture
    other, _ = rt.Shader(scene, g["studio_normals"], g["studio_mats"], g["studio_lights"], max_tree_depth=1, ao_rays=10).Rasterize(cam)
    assert not np.array_equal(other, frames[0])
    # test1 with the reference's own Radiosity object attached
    s1 = rt.Scene(g["test1_tris"], g["test1_tri_obj"])
    sh1 = rt.Shader(s1, g["test1_normals"], g["test1_mats"], g["test1_lights"], max_tree_depth=1, ao_rays=10, radiosity_branch=True)
    d1 = oracle.primary_dirs(oracle.raster_details(g["test1_cam14"], int(g["test1_res"][0]), int(g["test1_res"][1])), g["test1_pix"])
    o1 = np.tile(g["test1_cam14"][:3], (len(d1), 1))
    assert np.abs(sh1.Trace(rt.make_rays(o1, d1)) - g["test1_rgb"]).max() < TOL
    # unknown flags are refused
    sh1.params.flags = 6
    with pytest.raises(rt.RtmiError):
        sh1.Trace(rt.make_rays(o1[:4], d1[:4]))


@pytest.mark.parametrize("name,mode", [("aa", 1), ("dof", 2)])
def test_sampled_raster_loops(rt, oracle, name, mode):
    """Anti_Aliasing / Depth_Of_Field_Effect on the device: colours against the oracle's restated loops (same sample
    order, same mean), bytes against the reference's own Rasterizer functions (golden), and in two tile halves"""
    g = load_golden("g6_shade_studio.npz")
    r = load_golden("g6_raster_studio.npz")
    res = r["res_" + name]
    depth = int(r["depth_" + name])
    scene, sh = _shader(rt, g, max_tree_depth=depth, ao_rays=10)
    cam = rt.Camera.from_cam14(g["cam14"], res)
    rgb, pix = sh.Rasterize(cam, mode=mode)
    pl = np.array([(i, j) for i in range(int(res[1])) for j in range(int(res[0]))], np.uint32)
    o, d, S = oracle.raster_rays(mode, g["cam14"], res, pl)
    c = oracle.trace(g["tris"], g["normals"], g["tri_obj"], g["mats"], g["lights"], o, d, max_depth=depth, ao_rays=10).reshape(len(pl), S, 3)
    acc = np.zeros((len(pl), 3))
    for s in range(S):
        acc = acc + c[:, s]
    acc = acc / float(S)
    assert np.abs(rgb.reshape(-1, 3) - acc).max() < TOL
    edge = (np.abs(256 * acc - np.round(256 * acc)) < 256 * TOL).reshape(pix.shape)
    assert not ((pix != r["pixels_" + name]) & ~edge).any()
    a, _ = sh.Rasterize(cam, mode=mode, tile=(16, 12), tile_first=0, tile_stride=2)
    b, _ = sh.Rasterize(cam, mode=mode, tile=(16, 12), tile_first=1, tile_stride=2)
    assert np.array_equal(np.where(a != 0, a, b), rgb)


def test_scene_files_to_image_like_main_cpp(rt, tmp_path):
    """The reference's main.cpp end to end without the reference: BuildScene (rtmi_scene_file_load) -> BuildBSP ->
    Rasterize -> PPM; the test1 image equals the one the compiled reference's own pipeline produced (golden)."""
    import os
    from conftest import GOLDEN
    g = load_golden("g6_shade_test1.npz")
    s = rt.BuildScene(os.path.join(GOLDEN, "scenes"), "test1")
    scene = rt.Scene(s["tris"], s["tri_obj"])
    sh = rt.Shader(scene, s["normals"], s["mats"], s["lights"], max_tree_depth=1, ao_rays=10)
    cam = rt.Camera.from_cam14(s["cam14"], s["res"])
    _, pix = sh.Rasterize(cam)
    assert np.array_equal(pix.reshape(-1, 3), g["pixels"])
    out = tmp_path / "test1.ppm"
    rt.write_ppm(str(out), pix)
    raw = out.read_bytes()
    assert raw.startswith(b"P6\n500 400\n255\n") and len(raw) == 15 + 600000      # scenes/out.ppm of the reference: 600 015 bytes


def test_generated_100k_face_mesh_loads_and_renders(rt, oracle, tmp_path):
    """SURVEY.md §8f rank 3 "at scale": a 122 881-face OBJ (triangles, quads, pentagons, one 96-gon; tools/gen_obj.py) goes
    through the lenient reader, BuildBSP and the frame path; its primary hits equal brute force on a sample of pixels."""
    import sys, os
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_obj
    want = gen_obj.generate(str(tmp_path), "terrain", 320)
    s = rt.BuildScene(str(tmp_path), "terrain", lenient=True)
    assert len(s["tris"]) == 225374 and np.array_equal(s["tris"], want)
    scene = rt.Scene(s["tris"], s["tri_obj"])
    sh = rt.Shader(scene, s["normals"], s["mats"], s["lights"], max_tree_depth=1, ao_rays=4, seed=3)
    cam = rt.Camera.from_cam14(s["cam14"], s["res"])
    rgb, pix = sh.Rasterize(cam)
    scene.check()
    lit = (pix.reshape(-1, 3).astype(int).sum(1) > 0)
    assert lit.mean() > 0.5 and len(np.unique(pix.reshape(-1, 3), axis=0)) > 1000          # a picture, not a constant
    # primary hits of every 97th pixel against brute force over the 225 374 triangles
    o, d, pixels = frame_rays(oracle, s["cam14"], s["res"])
    sel = np.arange(0, len(d), 97 * 13)
    tri, t, _, _ = oracle.closest_hit(s["tris"], o[sel], d[sel])
    hits = scene.Cast(rt.make_rays(o[sel], d[sel]))
    assert np.array_equal(hits["tri"], np.where(tri < 0, rt.MISS, tri).astype(np.uint32)) and np.array_equal(hits["t"], t)
    assert (tri >= 0).mean() > 0.5


@pytest.mark.parametrize("stem,n_tris", [("real_full_room", 14676), ("real_buddha", 3367)])
def test_the_references_large_assets_render(rt, oracle, stem, n_tris):
    """the reference's own meshes full_room.obj and Golden_Buddha_low.obj (tests/golden/scenes, data
    fixtures) through the lenient reader, BuildBSP and the frame path at 640x480: primary hits of a pixel sample equal brute
    force over all triangles, colours of the same sample equal the oracle's Trace (seeded AO stream) within TOL."""
    import os
    from conftest import GOLDEN
    s = rt.BuildScene(os.path.join(GOLDEN, "scenes"), stem, lenient=True)
    assert len(s["tris"]) == n_tris
    scene = rt.Scene(s["tris"], s["tri_obj"])
    sh = rt.Shader(scene, s["normals"], s["mats"], s["lights"], max_tree_depth=2, ao_rays=10, seed=4)
    cam = rt.Camera.from_cam14(s["cam14"], s["res"])
    rgb, pix = sh.Rasterize(cam)
    scene.check()
    o, d, _ = frame_rays(oracle, s["cam14"], s["res"])
    sel = np.arange(0, len(d), 61)
    tri, t, _, _ = oracle.closest_hit(s["tris"], o[sel], d[sel])
    hits = scene.Cast(rt.make_rays(o[sel], d[sel]))
    assert np.array_equal(hits["tri"], np.where(tri < 0, rt.MISS, tri).astype(np.uint32)) and np.array_equal(hits["t"], t)
    assert (tri >= 0).mean() > 0.2                                                         # the mesh is in the picture
    # the AO stream is keyed by the ray's index in the frame: trace the sample with the same indices
    want = np.stack([oracle.trace(s["tris"], s["normals"], s["tri_obj"], s["mats"], s["lights"], o[k:k + 1], d[k:k + 1], max_depth=2,
                                  ao_rays=10, seed=4, first_index=int(k))[0] for k in sel[::8]])
    assert np.abs(rgb.reshape(-1, 3)[sel[::8]] - want).max() < TOL
    assert len(np.unique(pix.reshape(-1, 3), axis=0)) > 200                                # a picture, not a constant


def test_frame_and_ray_queries_inside_a_graph_capture(rt):
    """Every stage of the frame path is a launch whose ray count lives in device memory, and the ray queries reset
    their queue with a memset on the stream: both can be captured into a graph (torch.cuda.graph) once a first,
    uncaptured call has sized the workspace, and the replayed graph gives the plain calls' bits."""
    import torch
    g = load_golden("g6_shade_studio.npz")
    scene, sh = _shader(rt, g, "mats_amb", max_tree_depth=3, ao_rays=10, seed=5)
    cam = rt.Camera.from_cam14(g["cam14"], g["res"])
    n = cam.x_res * cam.y_res
    rgb, pix = sh.Rasterize(cam)                                  # plain call: the workspace exists from here on
    d_rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    d_pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
    d_hits = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    d_plain = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    scene.primary_dev(cam, d_plain.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        st = torch.cuda.current_stream().cuda_stream
        scene.primary_dev(cam, d_hits.data_ptr(), stream=st)
        sh.render_frame_dev(cam, d_rgb.data_ptr(), d_pix.data_ptr(), stream=st)
    assert int(d_pix.sum()) == 0 and int(d_hits.sum()) == 0      # captured, not run
    for _ in range(2):
        d_rgb.zero_(); d_pix.zero_(); d_hits.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(d_rgb.cpu().numpy().reshape(rgb.shape), rgb)
        assert np.array_equal(d_pix.cpu().numpy().reshape(pix.shape), pix)
        assert torch.equal(d_hits, d_plain)
    scene.check()


def test_shader_errors(rt):
    g = load_golden("g6_shade_studio.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    with pytest.raises(rt.RtmiError, match="light"):
        rt.Shader(scene, g["normals"], g["mats"], np.zeros((0, 6)))
    with pytest.raises(rt.RtmiError, match="materials"):
        rt.Shader(scene, g["normals"], g["mats"][:3], g["lights"])
    sh = rt.Shader(scene, g["normals"], g["mats"], g["lights"], ao_rays=10, ao_radius=0.0)
    with pytest.raises(rt.RtmiError, match="ao_radius"):
        sh.Rasterize(rt.Camera.from_cam14(g["cam14"], g["res"]))
