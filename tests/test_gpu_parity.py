"""GPU parity: the HIP path (through the C ABI of librtmi.so) against the oracle and the golden vectors
produced by the compiled reference.  Bit-exact (==) for hit ids, t, u, v, visibility bits and form
factors (fp32 storage of the oracle's double); the solve within 1e-4 relative L2 (north star)."""
import zlib

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

SCENES = ["test1", "test2", "test5", "test6", "test7", "scene_room_bed_sample"]


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    import raytracer_amd
    return raytracer_amd


def primary_frame(rt, scene, cam, tile=(64, 64), tile_first=0, tile_stride=1, fill=None):
    import torch
    n = cam.x_res * cam.y_res
    out = torch.zeros(n * rt.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    if fill is not None:
        out.fill_(fill)
    scene.primary_dev(cam, out.data_ptr(), tile=tile, tile_first=tile_first, tile_stride=tile_stride,
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(rt.HIT_DTYPE)


def ids(tri):
    """oracle ids (-1 = miss) -> rtmi ids"""
    return np.where(tri < 0, 0xFFFFFFFF, tri).astype(np.uint32)


def test_g1_triangle_intersection_bit_exact(rt):
    g = load_golden("g1_mt.npz")
    t, u, v = rt.triangle_intersection(g["o"], g["d"], g["v0"], g["v1"], g["v2"])
    assert np.array_equal(t, g["t"]) and np.array_equal(u, g["u"]) and np.array_equal(v, g["v"])


@pytest.mark.parametrize("stem", SCENES)
def test_g2_primary_closest_hit_bit_exact(rt, oracle, stem):
    g = load_golden("g2_%s.npz" % stem)
    scene = rt.Scene(g["tris"], g["tri_obj"])
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    O, dR, dU = cam.raster_details()
    assert np.array_equal(np.concatenate([O, dR, dU]), g["raster"][9:18])     # RasterDetails, rasterizer.cpp:40-56
    hits = primary_frame(rt, scene, cam)
    k = g["pix"][:, 0].astype(np.int64) * cam.x_res + g["pix"][:, 1]
    got = hits[k]
    # golden = the reference's own triangle_intersection, brute force over the original triangles
    assert np.array_equal(got["tri"], ids(g["tri"]))
    assert np.array_equal(got["t"], g["t"])
    assert np.array_equal(got["u"], g["u"].astype(np.float32)) and np.array_equal(got["v"], g["v"].astype(np.float32))
    hit = g["tri"] >= 0
    assert np.array_equal(got["object"][hit], g["tri_obj"][g["tri"][hit]])
    # Scene::Cast through the reference's BSP: same hit/miss, distance to rounding
    if int(g["bsp_rc"]) == 0:
        hm = g["cast_hit"].astype(bool)
        assert np.array_equal(hm, got["tri"] != rt.MISS)
        assert np.max(np.abs(g["cast_dist"][hm] - got["t"][hm]) / got["t"][hm]) < 1e-12
    # the whole frame, by checksum of the reference's full-frame brute force
    tri_full = np.where(hits["tri"] == rt.MISS, -1, hits["tri"].astype(np.int64)).astype("<i4")
    assert int((tri_full >= 0).sum()) == int(g["full_hits"])
    assert zlib.crc32(tri_full.tobytes()) == int(g["full_tri_crc32"])
    assert zlib.crc32(hits["t"].astype("<f8").tobytes()) == int(g["full_t_crc32"])


def test_g2_ray_buffer_path_equals_generated_primaries(rt):
    """rtmi_primary_rays_dev + rtmi_closest_hit_dev == rtmi_primary_closest_hit_dev, and the host-pointer
    entry equals the device-pointer entry."""
    import torch
    g = load_golden("g2_scene_room_bed_sample.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    n = cam.x_res * cam.y_res
    ref = primary_frame(rt, scene, cam)
    rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
    scene.cast_dev(rays.data_ptr(), n, out.data_ptr(), stream=st)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(rt.HIT_DTYPE)
    assert got.tobytes() == ref.tobytes()
    hrays = rays.cpu().numpy().view(rt.RAY_DTYPE)
    got2 = scene.Cast(hrays[::7])
    assert got2.tobytes() == ref[::7].tobytes()
    occ = scene.Occluded(hrays[::7])
    assert np.array_equal(occ.astype(bool), ref[::7]["tri"] != rt.MISS)


def test_g2_tiles_shard_without_overlap(rt):
    """Image tiles dealt round-robin to 3 'ranks' (tile_first = rank, tile_stride = 3): every pixel is written by
    exactly one rank, and the union equals the unsharded frame — the multi-GPU path for ray queries."""
    g = load_golden("g2_test1.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    full = primary_frame(rt, scene, cam, tile=(0, 0))
    parts = [primary_frame(rt, scene, cam, tile=(48, 40), tile_first=r, tile_stride=3, fill=0xAB) for r in range(3)]
    raw = [p.view(np.uint8).reshape(-1, 24) for p in parts]
    written = np.stack([(r != 0xAB).any(axis=1) for r in raw])
    assert np.array_equal(written.sum(0), np.ones(len(full), np.int64))
    union = np.where(written[0][:, None], raw[0], np.where(written[1][:, None], raw[1], raw[2]))
    assert union.tobytes() == full.tobytes()


def test_samples_in_one_launch_equal_one_launch_per_sample(rt):
    """rtmi_primary_closest_hit_samples_dev: 5 samples with their own pixel offsets and first tiles (ragged tiles, one sample
    starting beyond the last tile, two samples sharing one frame on disjoint tiles) write, byte for byte, what one
    rtmi_primary_closest_hit_dev call per sample writes; untouched pixels stay untouched."""
    import torch
    g = load_golden("g2_test1.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    n = cam.x_res * cam.y_res
    st = torch.cuda.current_stream().cuda_stream
    tile, stride = (48, 40), 4
    n_tiles = -(-cam.x_res // tile[0]) * -(-cam.y_res // tile[1])
    spec = [(0.5, 0.5, 0, 0), (0.25, 0.75, 3, 1), (0.9, 0.1, 2, 2), (0.5, 0.5, 1, 0), (0.3, 0.3, n_tiles + 1, 3)]   # (sx, sy, first, frame)
    one = [torch.full((n * 24,), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(4)]
    per = [torch.full((n * 24,), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(4)]
    for sx, sy, first, f in spec:
        scene.primary_dev(cam, per[f].data_ptr(), sx=sx, sy=sy, tile=tile, tile_first=first, tile_stride=stride, stream=st)
    scene.primary_samples_dev(cam, [(sx, sy, first, one[f].data_ptr()) for sx, sy, first, f in spec], tile=tile, tile_stride=stride, stream=st)
    torch.cuda.synchronize()
    scene.check()
    for a, b in zip(one, per):
        assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes()
    assert (per[0].cpu().numpy() != 0xAB).any() and (per[3].cpu().numpy() == 0xAB).all()
    # whole frames: 3 samples x all tiles (tile_stride 1, tile (0, 0) = the frame as one tile)
    outs = [torch.zeros(n * 24, dtype=torch.uint8, device="cuda") for _ in range(3)]
    offs = [(0.5, 0.5), (0.1, 0.2), (0.8, 0.6)]
    scene.primary_samples_dev(cam, [(sx, sy, 0, o.data_ptr()) for (sx, sy), o in zip(offs, outs)], stream=st)
    for (sx, sy), o in zip(offs, outs):
        ref = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        scene.primary_dev(cam, ref.data_ptr(), sx=sx, sy=sy, stream=st)
        torch.cuda.synchronize()
        assert o.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes()
    with pytest.raises(rt.RtmiError):
        scene.primary_samples_dev(cam, [(0.5, 0.5, 0, outs[0].data_ptr())] * 17, stream=st)
    with pytest.raises(rt.RtmiError):
        scene.primary_samples_dev(cam, [(0.5, 0.5, 0, None)], stream=st)


def test_g3_shadow_and_ao_any_hit(rt):
    g = load_golden("g2_test1.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    occ = scene.Occluded(rt.make_rays(g["shadow_o"], g["shadow_d"]))                  # shader.cpp:131
    assert np.array_equal(occ, g["shadow_occluded"])
    ao = scene.Occluded(rt.make_rays(g["ao_o"], g["ao_d"], tmax=2.0))                 # shader.cpp:226
    assert np.array_equal(ao, g["ao_occluded"])
    hit = scene.Occluded(rt.make_rays(g["ao_o"], g["ao_d"]))
    assert np.array_equal(hit, g["ao_hit"])
    ch = scene.Cast(rt.make_rays(g["ao_o"], g["ao_d"]))
    assert np.array_equal(ch["t"][g["ao_hit"] == 1], g["ao_dist"][g["ao_hit"] == 1])


@pytest.mark.parametrize("stem", ["radiosity_test", "practice_room_enclosed"])
def test_g4_visibility_and_light_bits(rt, stem):
    g = load_golden("g4_%s.npz" % stem)
    scene = rt.Scene(g["verts"])
    rad = rt.Radiosity(scene, g["normals"], g["light_center"])
    vis = rad.isVisibleXiToXj(g["ej"], g["xi"], g["xj"])                              # radiosity.cpp:231-262
    assert np.array_equal(vis, g["vis"])
    # radiosity.cpp:279-290: ray xi + 1e-5·n_i -> Unit(C − xi), blocked ⇔ any hit
    x = g["lx"]; n_i = g["normals"][g["li"]]
    dv = g["light_center"][None, :] - x
    l2 = dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1] + dv[:, 2] * dv[:, 2]
    d = dv / np.sqrt(l2)[:, None]
    o = x + 1e-5 * n_i
    assert np.array_equal(scene.Occluded(rt.make_rays(o, d)), g["light_blocked"])


@pytest.mark.parametrize("stem", ["radiosity_test", "practice_room_enclosed"])
def test_g4_form_factors_and_solve(rt, oracle, stem):
    g = load_golden("g4_%s.npz" % stem)
    ne = len(g["verts"])
    scene = rt.Scene(g["verts"])
    rad = rt.Radiosity(scene, g["normals"], g["light_center"])
    FF = rad.findFormFactor(samples=int(g["S"]), light_samples=int(g["SL"]), seed=int(g["seed"]))
    F = FF.cpu().numpy()[:, : ne + 1]
    # the fixture's F is the formula of radiosity.cpp:305-333 / 264-303 in double on the seeded points; the
    # device evaluates the same doubles and rounds once to fp32
    assert np.array_equal(F, g["F"].astype(np.float32))
    rad.progressiveRefinement(tol_rel=1e-7, check_every=1)
    assert rad.stats.sweeps < 200 and rad.stats.rel_change <= 1e-7
    B = rad.setRadiosityForAllElements()
    # vs the reference's own progressiveRefinement output on that F, and the exact fixed point
    Bref = g["B_ref"][:ne]
    assert np.linalg.norm(B - Bref) / np.linalg.norm(Bref) < 1e-4
    Bd = oracle.direct_solve(g["F"])[:ne]
    assert np.linalg.norm(B - Bd) / np.linalg.norm(Bd) < 2e-6     # fp32 storage of F and B
    assert float(rad.B[ne, 0]) == 68500.0
    # sharded rows give the same rows
    half = (ne + 1) // 2
    top = rad.findFormFactor(samples=int(g["S"]), light_samples=int(g["SL"]), seed=int(g["seed"]), row0=0, n_rows=half)
    assert np.array_equal(top.cpu().numpy()[:, : ne + 1], F[:half])
    bot = rad.findFormFactor(samples=int(g["S"]), light_samples=int(g["SL"]), seed=int(g["seed"]), row0=half)
    assert np.array_equal(bot.cpu().numpy()[:, : ne + 1], F[half:])


def test_form_factors_light_mean_flag(rt, oracle):
    g = load_golden("g4_practice_room_enclosed.npz")
    ne = len(g["verts"])
    scene = rt.Scene(g["verts"])
    rad = rt.Radiosity(scene, g["normals"], g["light_center"])
    F = rad.findFormFactor(samples=1, light_samples=int(g["SL"]), seed=7, flags=rt.FF_LIGHT_MEAN).cpu().numpy()
    fl = oracle.light_terms(g["verts"], g["normals"], g["light_center"], int(g["SL"]), 7, mirror_bug=0)
    assert np.array_equal(F[:ne, ne], fl.astype(np.float32))
    assert np.all(F[ne] == 0)


def test_closed_box_form_factors_rows(rt, oracle):
    """C2-type geometry (closed box, 12·k² patches): a few rows of F against the oracle, bit-exact in fp32."""
    from raytracer_amd import workloads
    verts, normals, light = workloads.closed_box(6)              # 432 patches
    n = len(verts)
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, light)
    rows = [0, 5, 217, n - 1]
    for r in rows:
        got = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=r, n_rows=1).cpu().numpy()[0]
        want = oracle.form_factors(verts, normals, 10, 1, row0=r, n_rows=1)[0]
        fl = oracle.light_terms(verts, normals, light, 30, 1, mirror_bug=1, row0=r, n_rows=1)
        assert np.array_equal(got[:n], want.astype(np.float32))
        assert got[n] == np.float32(fl[0])
        assert want.sum() > 0.3                                    # a closed box: most of the hemisphere is covered


def _ff_scenes():
    """(name, verts, normals): geometry that exercises every branch of the form-factor kernel's path walk — occluders
    between the patches (studio, box in a box), random orientations and depths (soup), tiny scenes (1, 2, 3 patches),
    rows across the 64-patch groups."""
    from raytracer_amd import workloads
    out = []
    v, _, _, _ = workloads.studio(3)
    out.append(("studio", v, workloads.face_normals(v)))
    vb, nb, _ = workloads.closed_box(5, hole=0.4)
    inner = workloads._box((3.0, 2.0, 0.0), (6.0, 5.5, 4.0), 2)
    out.append(("box_in_box", np.concatenate([vb, inner]), np.concatenate([nb, workloads.face_normals(inner)])))
    sp = workloads.soup(700, seed=3, edge=0.35)
    out.append(("soup", sp, workloads.face_normals(sp)))
    for m in (1, 2, 3):
        out.append(("tiny%d" % m, sp[:m], workloads.face_normals(sp[:m])))
    return out


@pytest.mark.parametrize("name", ["studio", "box_in_box", "soup", "tiny1", "tiny2", "tiny3"])
def test_form_factor_path_walk_equals_root_traversal_and_oracle(rt, oracle, monkeypatch, name):
    """rtmi_form_factors_dev finds each ray's closest hit from the two patches' paths (rtmi_formfactor.hip); the round-1
    kernel (RTMI_FF_KERNEL=flat) descends from the root.  Same F bit for bit, and both equal the oracle's rows."""
    verts, normals = next((v, n) for nm, v, n in _ff_scenes() if nm == name)
    n = len(verts)
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, None)
    monkeypatch.setenv("RTMI_FF_KERNEL", "path")
    F_path = rad.findFormFactor(samples=10, seed=11).cpu().numpy()[:, :n]
    monkeypatch.setenv("RTMI_FF_KERNEL", "flat")
    F_flat = rad.findFormFactor(samples=10, seed=11).cpu().numpy()[:, :n]
    monkeypatch.setenv("RTMI_FF_KERNEL", "path")
    assert np.array_equal(F_path.view(np.uint32), F_flat.view(np.uint32))
    rows = sorted(set([0, n // 3, n - 1]))
    for r in rows:
        want = oracle.form_factors(verts, normals, 10, 11, row0=r, n_rows=1)[0].astype(np.float32)
        assert np.array_equal(F_path[r], want)
    if n > 100:
        assert (F_path > 0).mean() > 0.02 and (F_path == 0).mean() > 0.2     # both visible and hidden / back-facing pairs
    # a band that starts and ends inside 64-patch groups, with a band size that does not divide it
    if n > 200:
        monkeypatch.setenv("RTMI_FF_BAND", "7")
        part = rad.findFormFactor(samples=10, seed=11, row0=61, n_rows=75).cpu().numpy()[:, :n]
        monkeypatch.delenv("RTMI_FF_BAND", raising=False)
        assert np.array_equal(part, F_path[61:136])


@pytest.mark.parametrize("samples", [1, 2, 3, 7, 10, 21, 33, 64, 65])
def test_form_factor_samples_dealt_over_the_wave(rt, oracle, monkeypatch, samples):
    """The path-walk kernel lets the lanes whose pair kept a sibling for its rays sit out the sample loop and deals THEIR
    samples over the wave afterwards (lane L: sample L mod S of the (L div S)-th such pair), adding the terms back in sample
    order.  Every S from one pair per pass (33, 64) to 32 (2), S = 1 and S > 64 (no dealing): the same F as the root
    traversal, which has no such loop, bit for bit; rows equal the oracle's."""
    from raytracer_amd import workloads
    vb, nb, _ = workloads.closed_box(10)                             # 1200 patches: the bench's geometry, 19 groups
    inner = workloads._box((3.0, 2.0, 0.0), (6.0, 5.5, 4.0), 3)
    for verts, normals in ((vb, nb), (np.concatenate([vb, inner]), np.concatenate([nb, workloads.face_normals(inner)]))):
        n = len(verts)
        rad = rt.Radiosity(rt.Scene(verts), normals, None)
        monkeypatch.setenv("RTMI_FF_KERNEL", "path")
        F_path = rad.findFormFactor(samples=samples, seed=5).cpu().numpy()[:, :n]
        monkeypatch.setenv("RTMI_FF_KERNEL", "flat")
        F_flat = rad.findFormFactor(samples=samples, seed=5).cpu().numpy()[:, :n]
        monkeypatch.setenv("RTMI_FF_KERNEL", "path")
        assert np.array_equal(F_path.view(np.uint32), F_flat.view(np.uint32))
        for r in (3, n - 2):
            want = oracle.form_factors(verts, normals, samples, 5, row0=r, n_rows=1)[0].astype(np.float32)
            assert np.array_equal(F_path[r], want)


def _multi_plane_scenes():
    """Face-normal geometry (so the host's plane flags are ON) whose 64-patch groups of the form-factor kernel span
    several planes and have occluders that lie behind one patch's plane and in front of its group mates': a staircase
    (treads and risers of 8-16 triangles alternate in the leaf order; a lower step hides part of an upper riser from the
    floor), an L-shaped concave block (one arm hides the other's inner face), a tessellated sphere with a small box
    beside it — each inside a coarse room."""
    from raytracer_amd import workloads
    room, rn, _ = workloads.closed_box(6)
    out = []
    steps = [workloads._box((1.0 + 0.7 * s, 3.0, 0.0), (1.7 + 0.7 * s, 7.0, 4.0 - 0.45 * s), 2) for s in range(8)]
    st = np.concatenate(steps)
    out.append(("stairs", np.concatenate([room, st]), np.concatenate([rn, workloads.face_normals(st)])))
    ell = np.concatenate([workloads._box((2.0, 2.0, 0.0), (7.0, 3.5, 3.0), 3), workloads._box((2.0, 3.5, 0.0), (3.5, 7.5, 3.0), 3)])
    out.append(("L_block", np.concatenate([room, ell]), np.concatenate([rn, workloads.face_normals(ell)])))
    nt, nph = 14, 28
    th = np.linspace(0.0, np.pi, nt + 1); ph = np.linspace(0.0, 2.0 * np.pi, nph + 1)
    c, rad = np.array([5.0, 5.0, 4.0]), 2.2
    P = lambda a, b: c + rad * np.array([np.sin(th[a]) * np.cos(ph[b]), np.sin(th[a]) * np.sin(ph[b]), np.cos(th[a])])
    sph = []
    for a in range(nt):
        for b in range(nph):
            p00, p10, p11, p01 = P(a, b), P(a + 1, b), P(a + 1, b + 1), P(a, b + 1)
            if a > 0: sph.append(np.concatenate([p00, p10, p01]))          # outward: (p10 − p00) x (p01 − p00) points along +r
            if a < nt - 1: sph.append(np.concatenate([p10, p11, p01]))
    sph = np.concatenate([np.array(sph), workloads._box((7.4, 4.2, 3.0), (8.2, 5.8, 5.0), 2)])
    out.append(("sphere", np.concatenate([room, sph]), np.concatenate([rn, workloads.face_normals(sph)])))
    return out


@pytest.mark.parametrize("name", ["stairs", "L_block", "sphere"])
def test_form_factor_plane_flags_belong_to_the_patch_not_the_group(rt, oracle, monkeypatch, name):
    """Round 2's path walk read "this sibling lies behind the patch's plane" for the levels a 64-patch group shares from the
    table of the group's FIRST patch: on geometry whose groups span several planes it skipped occluders in front of a
    lane's own plane.  The flag is per patch now (PatchMeta::behind).  Whole F: path walk with the
    flags on == root traversal == path walk with the flags off, bit for bit; rows equal the oracle's."""
    verts, normals = next((v, n) for nm, v, n in _multi_plane_scenes() if nm == name)
    n = len(verts)
    scene = rt.Scene(verts)

    def build(kernel, planes):
        monkeypatch.setenv("RTMI_FF_KERNEL", kernel)
        monkeypatch.setenv("RTMI_FF_PLANES", planes)
        rad = rt.Radiosity(scene, normals, None)                 # (the flags are fixed when a patch set's tables are built)
        F = rad.findFormFactor(samples=10, seed=31).cpu().numpy()[:, :n]
        on = int(rt.lib.rtmi_debug_form_factor_planes(rad.handle))
        rad.close()
        return F, on

    F_on, on = build("path", "1")
    F_off, off = build("path", "0")
    F_flat, _ = build("flat", "1")
    monkeypatch.delenv("RTMI_FF_PLANES", raising=False)
    monkeypatch.setenv("RTMI_FF_KERNEL", "path")
    assert on == 1 and off == 0                      # the flags really were on / off
    bad = np.argwhere(F_on.view(np.uint32) != F_flat.view(np.uint32))
    assert len(bad) == 0, (len(bad), bad[:8].tolist())
    assert np.array_equal(F_off.view(np.uint32), F_flat.view(np.uint32))
    for r in (0, 432 + 5, n - 7):                                # a room patch, two patches of the object inside
        want = oracle.form_factors(verts, normals, 10, 31, row0=r, n_rows=1)[0].astype(np.float32)
        assert np.array_equal(F_on[r], want)
    assert (F_on > 0).mean() > 0.05 and (F_on == 0).mean() > 0.3
    scene.check()


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.002, 0.0), (130.0, 0.0), (1.0, 900.0), (0.05, -37.5), (1.0, 2.0e4), (250.0, 0.0)])
def test_form_factor_grazing_thresholds_follow_the_scene_scale(rt, oracle, monkeypatch, scale, offset):
    """The cosines below which a ray counts as grazing a patch's plane (and tests what lies in or behind that plane after
    all) scale with the tolerance of the host's plane flags, 2^-40·(1 + |coord|max) — rtmi_formfactor.hip.  A long low
    corridor (8 x 1 x 0.25, finely meshed floor and ceiling: thousands of rays per row leave or arrive at less than a
    thousandth of a radian) at several sizes and far from the origin: the path walk with the flags on, with the flags
    off, and the root traversal give the same bits; rows equal the oracle's."""
    from raytracer_amd import workloads
    vb, nb, _ = workloads.closed_box(12)
    squash = np.array([0.8, 0.1, 0.025] * 3)                    # the 10-box becomes 8 x 1 x 0.25
    verts = vb * squash * scale + offset
    normals = nb.copy()
    n = len(verts)
    scene = rt.Scene(verts)

    def build(kernel, planes):
        monkeypatch.setenv("RTMI_FF_KERNEL", kernel)
        monkeypatch.setenv("RTMI_FF_PLANES", planes)
        rad = rt.Radiosity(scene, normals, None)                 # (the flags are fixed when a patch set's tables are built)
        F = rad.findFormFactor(samples=10, seed=21).cpu().numpy()[:, :n]
        rad.close()
        return F

    F_on = build("path", "1")
    F_off = build("path", "0")
    F_flat = build("flat", "1")
    monkeypatch.delenv("RTMI_FF_PLANES", raising=False)
    monkeypatch.setenv("RTMI_FF_KERNEL", "path")
    assert np.array_equal(F_on.view(np.uint32), F_flat.view(np.uint32))
    assert np.array_equal(F_off.view(np.uint32), F_flat.view(np.uint32))
    for r in (0, n // 2 + 3, n - 1):
        want = oracle.form_factors(verts, normals, 10, 21, row0=r, n_rows=1)[0].astype(np.float32)
        assert np.array_equal(F_on[r], want)
    assert (F_on > 0).mean() > 0.3
    scene.check()


@pytest.mark.parametrize("stem,row0", [("real_full_room", 7000), ("real_buddha", 1500)])
def test_form_factors_on_the_references_meshes(rt, oracle, monkeypatch, stem, row0):
    """The path walk on real meshes (the reference's full_room.obj: furniture inside a room, 78 zero-area faces, vertex
    normals that are not the geometric ones — so no plane flags; Golden_Buddha_low.obj: a closed curved surface, seen from the inside): 384 rows
    equal the root-traversal kernel's bit for bit, two of them the oracle's (through its own hierarchy)."""
    import os
    from conftest import GOLDEN
    s = rt.BuildScene(os.path.join(GOLDEN, "scenes"), stem, lenient=True)
    verts, normals = s["tris"], s["normals"]
    if stem == "real_buddha":
        normals = -normals            # the statue from the inside: a closed cavity whose folds hide its patches from each other
    n = len(verts)
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, None)
    monkeypatch.setenv("RTMI_FF_KERNEL", "path")
    F_path = rad.findFormFactor(samples=10, seed=5, row0=row0, n_rows=384).cpu().numpy()[:, :n]
    scene.check()
    monkeypatch.setenv("RTMI_FF_KERNEL", "flat")
    F_flat = rad.findFormFactor(samples=10, seed=5, row0=row0, n_rows=384).cpu().numpy()[:, :n]
    monkeypatch.delenv("RTMI_FF_KERNEL", raising=False)
    assert np.array_equal(F_path.view(np.uint32), F_flat.view(np.uint32))
    assert (F_path > 0).mean() > 0.01 and (F_path == 0).mean() > 0.3
    for r in (row0 + 3, row0 + 380):
        want = oracle.form_factors_bvh(verts, normals, 10, 5, row0=r, n_rows=1)[0].astype(np.float32)
        assert np.array_equal(F_path[r - row0], want)
    # left to itself the library times both kernels on the first rows of the first sizeable call (256 rows or more) and keeps the faster:
    # on these meshes (every ray passes much geometry) that is the root traversal; the matrix is the same
    assert rt.lib.rtmi_debug_form_factor_kernel(rad.handle) == 0
    rad.findFormFactor(samples=10, seed=5, row0=0, n_rows=8)                       # too small a call to measure on
    assert rt.lib.rtmi_debug_form_factor_kernel(rad.handle) == 0
    F_auto = rad.findFormFactor(samples=10, seed=5, row0=row0, n_rows=384).cpu().numpy()[:, :n]
    assert rt.lib.rtmi_debug_form_factor_kernel(rad.handle) == 2
    assert np.array_equal(F_auto.view(np.uint32), F_path.view(np.uint32))
    # a first call LONGER than the sample the choice is made on (1024 rows): the sample's rows are kept as written — by the path
    # walk, with slices of them written again by the root traversal — and only the rest is launched, with the kernel chosen
    if stem == "real_buddha":
        rad2 = rt.Radiosity(scene, normals, None)
        F_long = rad2.findFormFactor(samples=10, seed=5, row0=row0 - 900, n_rows=1300).cpu().numpy()[:, :n]
        assert rt.lib.rtmi_debug_form_factor_kernel(rad2.handle) == 2
        assert np.array_equal(F_long[900:1284].view(np.uint32), F_path.view(np.uint32))
        monkeypatch.setenv("RTMI_FF_KERNEL", "flat")
        F_long_flat = rad2.findFormFactor(samples=10, seed=5, row0=row0 - 900, n_rows=1300).cpu().numpy()[:, :n]
        monkeypatch.delenv("RTMI_FF_KERNEL", raising=False)
        assert np.array_equal(F_long.view(np.uint32), F_long_flat.view(np.uint32))
        rad2.close()


def test_form_factors_inside_a_graph_capture(rt):
    """A caller that captures its launches into a graph (torch.cuda.graph = hipStreamBeginCapture on the stream): a
    form-factor call of measurable size neither times the two kernels (that waits for events, which a capture
    forbids) nor allocates — the path tables exist since the first, uncaptured call — and the replayed graph writes
    the same bits as a plain call."""
    import torch
    from raytracer_amd import workloads
    verts, normals, _ = workloads.closed_box(16)      # 3072 patches
    n = len(verts)
    assert n >= 1024
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, None)
    plain = rad.findFormFactor(samples=4, seed=9, row0=0, n_rows=8).clone()       # builds the tables; too small to measure on
    ld = rad.ld
    out = torch.zeros((512, ld), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        rc = rt.lib.rtmi_form_factors_dev(rad.handle, None, 4, 30, 9, 0, 0, 512, out.data_ptr(), ld, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rt.lib.rtmi_last_error().decode()
    assert rt.lib.rtmi_debug_form_factor_kernel(rad.handle) == 0                   # nothing was measured
    assert float(out.abs().sum()) == 0.0                                            # captured, not run
    graph.replay()
    torch.cuda.synchronize()
    want = rad.findFormFactor(samples=4, seed=9, row0=0, n_rows=512)                # a plain call (this one measures)
    assert rt.lib.rtmi_debug_form_factor_kernel(rad.handle) in (1, 2)
    assert torch.equal(out.view(torch.int32), want.view(torch.int32))
    assert torch.equal(out[:8].view(torch.int32), plain.view(torch.int32))
    scene.check()


def test_shared_reciprocal_division_is_the_plain_division(rt):
    """csrc/rtmi_device.hpp div3 (every `Vec3 / double` of the kernels: one reciprocal for three numerators) against the
    compiler's three IEEE divisions: the same bits on 6 M random operand sets over the whole exponent range — including
    the ends, where the hardware rescales operands and the shared form must step aside — and on the special values."""
    import torch
    rng = np.random.default_rng(11)

    def wide(n):      # random sign, mantissa and exponent (denormals, zeros, infinities and NaN included)
        bits = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63))
        return bits.view(np.float64)

    n = 1 << 21
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                         1e-300, 1e300, 3.0, 1.0 / 3.0, 0.1, 2.0 ** -1022, 2.0 ** 1023, 2.0 ** -1074, 1e-5, 0.135])
    sp = np.array(np.meshgrid(specials, specials, indexing="ij")).reshape(2, -1).T      # all (numerator, denominator) pairs
    sets = []
    old_err = np.seterr(all="ignore")
    # ordinary magnitudes (what the kernels see), wide exponents, near-equal exponents at both ends, the special pairs
    sets.append((rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 4, size=(n, 1)), rng.normal(size=n) * 10.0 ** rng.integers(-3, 4, size=n)))
    sets.append((wide(3 * n).reshape(n, 3), wide(n)))
    big = 2.0 ** rng.integers(960, 1024, size=(n, 1)).astype(np.float64)
    sets.append((rng.normal(size=(n, 3)) * big * 2.0 ** rng.integers(-60, 1, size=(n, 3)).astype(np.float64), rng.normal(size=n) * 2.0 ** rng.integers(-1074, -900, size=n).astype(np.float64)))
    sets.append((rng.normal(size=(n, 3)) * 2.0 ** rng.integers(-1074, -900, size=(n, 3)).astype(np.float64), rng.normal(size=n) * big[:, 0]))
    sets.append((np.stack([sp[:, 0], np.roll(sp[:, 0], 1), np.roll(sp[:, 0], 7)], 1), sp[:, 1]))
    np.seterr(**old_err)
    with np.errstate(all="ignore"):
        for num, den in sets:
            d_num = torch.from_numpy(np.ascontiguousarray(num, dtype=np.float64)).cuda()
            d_den = torch.from_numpy(np.ascontiguousarray(den, dtype=np.float64)).cuda()
            a = torch.zeros_like(d_num)
            b = torch.zeros_like(d_num)
            rc = rt.lib.rtmi_debug_div3_dev(d_num.data_ptr(), d_den.data_ptr(), len(den), a.data_ptr(), b.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rt.lib.rtmi_last_error().decode()
            torch.cuda.synchronize()
            ga, gb = a.cpu().numpy(), b.cpu().numpy()
            want = num / den[:, None]                                    # the host's IEEE division
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(gb), nan) and np.array_equal(np.isnan(ga), nan)
            assert np.array_equal(gb.view(np.uint64)[~nan], want.view(np.uint64)[~nan])      # the compiler's division is IEEE
            assert np.array_equal(ga.view(np.uint64)[~nan], want.view(np.uint64)[~nan])      # ... and so is the shared form


def test_unit_vector_fast_path_is_the_plain_root_and_division(rt):
    """csrc/rtmi_device.hpp unit_of_len2 (the direction of every form-factor sample: a / sqrt(a·a)) runs the compiler's square
    root and the three divisions WITHOUT their range handling when every lane of the wave is far from the ends of the exponent
    range, and the plain forms otherwise.  Same bits as the plain forms and as the host's IEEE root and division: on waves
    that are all in range (ordinary magnitudes: the bare sequence runs), on waves at and across the guard's bounds, on the
    whole exponent range, zeros, infinities and NaNs (the plain forms run)."""
    import torch
    rng = np.random.default_rng(23)
    n = 1 << 21

    def wide(k):
        bits = rng.integers(0, 1 << 63, size=k, dtype=np.uint64) | (rng.integers(0, 2, size=k, dtype=np.uint64) << np.uint64(63))
        return bits.view(np.float64)

    sets = []
    with np.errstate(all="ignore"):
        sets.append(rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 4, size=(n, 1)))                                   # what the kernel sees
        sets.append(rng.normal(size=(n, 3)) * 2.0 ** rng.integers(-40, 41, size=(n, 3)).astype(np.float64))              # components far apart
        for e in (-302, -300, -298, 298, 300, 302, -151, 151):                                                             # around the guard's bounds
            sets.append((1.0 + rng.random(size=(n // 8, 3))) * rng.choice([-1.0, 1.0], size=(n // 8, 3)) * 2.0 ** e)
        v = rng.normal(size=(n // 8, 3)); v[:, 1] *= 2.0 ** -310; sets.append(v)                                          # one component below the bound
        v = rng.normal(size=(n // 8, 3)); v[rng.integers(0, len(v), 5000), rng.integers(0, 3, 5000)] = 0.0; sets.append(v)  # zeros among ordinary lanes
        sets.append(wide(3 * n).reshape(n, 3))                                                                             # anything
        specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 1e-200, 1e200, 3.0, 1e-5])
        sets.append(np.array(np.meshgrid(specials, specials, specials, indexing="ij")).reshape(3, -1).T.copy())
        for vec in sets:
            d = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.float64)).cuda()
            a, b = torch.zeros_like(d), torch.zeros_like(d)
            rc = rt.lib.rtmi_debug_unit_dev(d.data_ptr(), len(vec), a.data_ptr(), b.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rt.lib.rtmi_last_error().decode()
            torch.cuda.synchronize()
            ga, gb = a.cpu().numpy(), b.cpu().numpy()
            d2 = vec[:, 0] * vec[:, 0] + vec[:, 1] * vec[:, 1] + vec[:, 2] * vec[:, 2]      # vec3.h:14, left to right
            want = vec / np.sqrt(d2)[:, None]
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(gb), nan) and np.array_equal(np.isnan(ga), nan)
            assert np.array_equal(gb.view(np.uint64)[~nan], want.view(np.uint64)[~nan])      # the compiler's root and division are IEEE
            assert np.array_equal(ga.view(np.uint64)[~nan], want.view(np.uint64)[~nan])      # ... and the bare sequence gives the same bits


def test_patches_must_be_the_scenes_triangles(rt):
    from raytracer_amd import workloads
    v = workloads.soup(50, seed=2, edge=0.3)
    scene = rt.Scene(v)
    other = v.copy()
    other[7, 4] += 1e-9
    with pytest.raises(rt.RtmiError, match="not the array the scene was built from"):
        scene_tri = scene.tri_xyz
        scene.tri_xyz = other
        try:
            rt.Radiosity(scene, workloads.face_normals(v), None)
        finally:
            scene.tri_xyz = scene_tri


@pytest.mark.parametrize("n_tris,n_rays", [(1, 4096), (5, 4096), (3000, 60000)])
def test_soup_closest_and_any_hit_vs_brute_force(rt, oracle, n_tris, n_rays):
    from raytracer_amd import workloads
    tris = workloads.soup(n_tris, seed=3, edge=0.2)
    rng = np.random.default_rng(5)
    o = rng.uniform(-1.5, 1.5, (n_rays, 3))
    d = rng.normal(size=(n_rays, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    tmax = np.where(rng.random(n_rays) < 0.5, 1e20, rng.uniform(0.1, 3.0, n_rays))
    scene = rt.Scene(tris)
    got = scene.Cast(rt.make_rays(o, d, tmax))
    tri, t, u, v = oracle.closest_hit(tris, o, d, tmax)
    assert np.array_equal(got["tri"], ids(tri))
    assert np.array_equal(got["t"], t)
    assert np.array_equal(got["u"], u.astype(np.float32)) and np.array_equal(got["v"], v.astype(np.float32))
    assert np.array_equal(scene.Occluded(rt.make_rays(o, d, tmax)), oracle.any_hit(tris, o, d, tmax))
    if n_tris >= 1000:
        assert (tri >= 0).sum() > 1000


def test_ragged_launch_sizes_through_the_shared_queue(rt, oracle):
    """The traversal queue hands out items in claims that the waves of a workgroup share, 64 at a time, with sizes that follow
    the measured rate (rtmi_device.hpp: trace_queue): every ray of launches of 1 ... 70 001 rays — counts around the multiples of
    64, 256 and the claim sizes — gets exactly its own record (closest and any hit against the oracle's scan), and frames of odd sizes
    dealt over three ranks are written exactly once.  (Claim sizes pinned to 64 ... 2048 with RTMI_CHUNK gave the same frame
    digest in the sweeps of profiles/ab_r03.md; the variable is read once per process, so it is not varied here.)"""
    from raytracer_amd import workloads
    tris = workloads.soup(2000, seed=9, edge=0.15)
    rng = np.random.default_rng(17)
    n_max = 70001
    o = rng.uniform(-1.5, 1.5, (n_max, 3))
    d = rng.normal(size=(n_max, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    tmax = np.where(rng.random(n_max) < 0.5, 1e20, rng.uniform(0.1, 3.0, n_max))
    tri, t, u, v = oracle.closest_hit(tris, o, d, tmax)
    occ = oracle.any_hit(tris, o, d, tmax)
    scene = rt.Scene(tris)
    for n in (1, 2, 63, 64, 65, 127, 129, 255, 256, 257, 511, 1023, 1025, 4097, 16385, 65537, n_max):
        got = scene.Cast(rt.make_rays(o[:n], d[:n], tmax[:n]))
        assert np.array_equal(got["tri"], ids(tri[:n])) and np.array_equal(got["t"], t[:n]), n
        assert np.array_equal(scene.Occluded(rt.make_rays(o[:n], d[:n], tmax[:n])), occ[:n]), n
    cam14 = np.array([0, 0, -3, 0, 0, 0, 0, 1, 0, 2.75, -1.6, 1.6, -0.9, 0.9], np.float64)
    for res in ((61, 37), (130, 67), (257, 129)):
        cam = rt.Camera.from_cam14(cam14, res)
        full = primary_frame(rt, scene, cam, tile=(0, 0))
        parts = [primary_frame(rt, scene, cam, tile=(16, 16), tile_first=r, tile_stride=3, fill=0xAB) for r in range(3)]
        raw = [p.view(np.uint8).reshape(-1, 24) for p in parts]
        written = np.stack([(r != 0xAB).any(axis=1) for r in raw])
        assert np.array_equal(written.sum(0), np.ones(len(full), np.int64)), res
        union = np.where(written[0][:, None], raw[0], np.where(written[1][:, None], raw[1], raw[2]))
        assert union.tobytes() == full.tobytes(), res


@pytest.mark.parametrize("case", ["n2", "n3", "n4", "n6", "n7", "n9", "n17", "offset", "flat", "mixed_scales", "axis_rays"])
def test_wide_node_layouts_vs_brute_force(rt, oracle, case):
    """The 4-wide hierarchy's corner cases (rtmi_device.hpp WNode): nodes with 2 or 3 children, leaves beside inner
    children, a grid origin far from 0, nodes of zero extent on an axis, children a million times smaller than their
    node's 8-bit grid step, rays parallel to an axis.  Closest and any hit, bit for bit against the brute-force oracle."""
    from raytracer_amd import workloads
    rng = np.random.default_rng(11)
    n_rays = 20000
    o = rng.uniform(-1.5, 1.5, (n_rays, 3))
    d = rng.normal(size=(n_rays, 3))
    if case.startswith("n"):
        tris = workloads.soup(int(case[1:]), seed=7, edge=0.5)
    elif case == "offset":
        tris = workloads.soup(500, seed=8, edge=0.3) + np.tile([4096.0, -8192.0, 1000.0], 3)
        o += [4096.0, -8192.0, 1000.0]
    elif case == "flat":
        tris = workloads.soup(400, seed=9, edge=0.3)
        tris[:, [2, 5, 8]] = 0.25                                   # every vertex in the plane z = 0.25
    elif case == "mixed_scales":
        small = workloads.soup(300, seed=10, edge=1e-5)
        big = workloads.soup(20, seed=12, edge=50.0)
        tris = np.concatenate([small, big])
        aim = small[rng.integers(0, len(small), n_rays), 0:3] + rng.uniform(-2e-5, 2e-5, (n_rays, 3))
        d = aim - o                                                 # aimed at the small triangles
    else:
        tris = workloads.soup(600, seed=13, edge=0.4)
        ax = rng.integers(0, 3, n_rays)
        d[:] = 0.0
        d[np.arange(n_rays), ax] = rng.choice([-1.0, 1.0], n_rays)
    d /= np.linalg.norm(d, axis=1)[:, None]
    tmax = np.where(rng.random(n_rays) < 0.5, 1e20, rng.uniform(0.1, 3.0, n_rays))
    scene = rt.Scene(tris)
    got = scene.Cast(rt.make_rays(o, d, tmax))
    tri, t, u, v = oracle.closest_hit(tris, o, d, tmax)
    assert np.array_equal(got["tri"], ids(tri))
    assert np.array_equal(got["t"], t)
    assert np.array_equal(got["u"], u.astype(np.float32)) and np.array_equal(got["v"], v.astype(np.float32))
    assert np.array_equal(scene.Occluded(rt.make_rays(o, d, tmax)), oracle.any_hit(tris, o, d, tmax))
    if case not in ("n2", "n3", "n4"):
        assert (tri >= 0).sum() > 20


def test_wide_nodes_fuzz_many_small_scenes(rt, oracle):
    """150 random scenes of 1..60 triangles (every shape of a small 4-wide hierarchy: leaf roots, nodes of 2, 3 and 4
    children, leaves beside inner children), edge lengths over five decades, 400 rays each: closest and any hit bit for bit
    against the brute-force oracle."""
    rng = np.random.default_rng(2024)
    total_hits = 0
    for it in range(150):
        n = int(rng.integers(1, 61))
        edge = 10.0 ** rng.uniform(-3, 1.5)
        c = rng.uniform(-1.0, 1.0, (n, 3)) * 10.0 ** rng.uniform(-1, 2)
        tris = np.concatenate([c, c + rng.uniform(-edge, edge, (n, 3)), c + rng.uniform(-edge, edge, (n, 3))], axis=1)
        m = 400
        o = rng.uniform(-1.5, 1.5, (m, 3)) * (np.abs(tris).max() + 1.0)
        aim = tris[rng.integers(0, n, m)].reshape(m, 3, 3)
        w = rng.dirichlet([1.0, 1.0, 1.0], m)
        d = (aim * w[:, :, None]).sum(1) - o                                  # aimed at a point of some triangle ...
        d[m // 2:] += rng.normal(size=(m - m // 2, 3)) * edge                 # ... half of them only roughly
        d /= np.linalg.norm(d, axis=1)[:, None]
        tmax = np.where(rng.random(m) < 0.7, 1e20, rng.uniform(0.1, 3.0, m) * np.linalg.norm(o, axis=1))
        scene = rt.Scene(tris)
        got = scene.Cast(rt.make_rays(o, d, tmax))
        tri, t, u, v = oracle.closest_hit(tris, o, d, tmax)
        assert np.array_equal(got["tri"], ids(tri)), (it, n)
        assert np.array_equal(got["t"], t), (it, n)
        assert np.array_equal(scene.Occluded(rt.make_rays(o, d, tmax)), oracle.any_hit(tris, o, d, tmax)), (it, n)
        scene.close()
        total_hits += int((tri >= 0).sum())
    assert total_hits > 150 * 400 // 5


def test_ties_on_shared_edges_and_duplicates(rt, oracle):
    """Rays aimed exactly at shared vertices / edges of a grid and at duplicated triangles: equal t ->
    smallest original id, as the reference's in-order scan (object.cpp:143-168)."""
    k = 8
    g = np.arange(k + 1, dtype=np.float64)
    tris = []
    for i in range(k):
        for j in range(k):
            p00, p10, p11, p01 = (g[i], g[j], 0.0), (g[i + 1], g[j], 0.0), (g[i + 1], g[j + 1], 0.0), (g[i], g[j + 1], 0.0)
            tris.append(p00 + p10 + p11)
            tris.append(p00 + p11 + p01)
    tris = np.array(tris + tris)                                    # every triangle twice
    gx, gy = np.meshgrid(np.arange(0, k + 0.01, 0.5), np.arange(0, k + 0.01, 0.5), indexing="ij")
    o = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, 3.0)], 1)
    d = np.tile([0.0, 0.0, -1.0], (len(o), 1))
    scene = rt.Scene(tris)
    got = scene.Cast(rt.make_rays(o, d))
    tri, t, u, v = oracle.closest_hit(tris, o, d)
    assert (tri >= 0).all() and np.array_equal(got["tri"], ids(tri)) and np.array_equal(got["t"], t)
    assert (tri < len(tris) // 2).all()


def test_empty_and_degenerate_inputs(rt):
    scene = rt.Scene(np.zeros((0, 9)))
    rays = rt.make_rays([[0, 0, 0]], [[0, 0, 1]])
    h = scene.Cast(rays)
    assert h["tri"][0] == rt.MISS and h["t"][0] == 0.0
    assert scene.Occluded(rays)[0] == 0
    assert len(scene.Cast(rays[:0])) == 0
    # zero-area triangles never hit (det window), and do not break the build
    deg = np.zeros((7, 9)); deg[:, 0] = 1.0
    s2 = rt.Scene(deg)
    assert s2.Cast(rays)["tri"][0] == rt.MISS
    with pytest.raises(rt.RtmiError):
        bad = np.zeros((1, 9)); bad[0, 0] = np.nan
        rt.Scene(bad)
    with pytest.raises(rt.RtmiError):
        rt.Radiosity(scene, np.zeros((0, 3)))


def test_a_wave_that_gives_up_is_reported(rt, monkeypatch):
    """RTMI_GUARD bounds the iterations a wave may spend on a ray (default 2^26: never reached on valid input).  With a
    tiny budget the waves of a real frame give up: the scene's error word must say so — through rtmi_scene_check and
    through every later entry point — instead of returning success with unwritten records."""
    import torch
    from raytracer_amd import workloads
    tris = workloads.soup(3000, seed=5, edge=0.2)
    cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, (256, 128))
    ok_scene = rt.Scene(tris)
    good = primary_frame(rt, ok_scene, cam)
    ok_scene.check()
    monkeypatch.setenv("RTMI_GUARD", "3")
    scene = rt.Scene(tris)                      # the budget is read when the scene is created
    monkeypatch.delenv("RTMI_GUARD")
    out = torch.zeros(cam.x_res * cam.y_res * rt.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    with pytest.raises(rt.RtmiError, match="gave up"):
        scene.check()
    with pytest.raises(rt.RtmiError, match="gave up"):     # sticky: later launches on the scene are refused
        scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=torch.cuda.current_stream().cuda_stream)
    assert (good["tri"] != rt.MISS).any()
    ok_scene.check()                             # other scenes are not affected


def test_many_launches_in_flight_on_two_streams(rt):
    """Every stream has its own work-queue counter, so any number of launches of one scene may be in flight on
    different streams: 2 x 300 queued launches (more than the 256 counters a scene holds) give the records of a
    single-stream run."""
    import torch
    from raytracer_amd import workloads
    tris = workloads.soup(20000, seed=6, edge=0.05)
    scene = rt.Scene(tris)
    cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, (320, 200))
    want = primary_frame(rt, scene, cam)
    n = cam.x_res * cam.y_res
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.zeros(n * rt.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(4)] for _ in streams]
    torch.cuda.synchronize()
    for k in range(300):
        for si, st in enumerate(streams):
            scene.primary_dev(cam, outs[si][k % 4].data_ptr(), tile=(64, 64), stream=st.cuda_stream)
    torch.cuda.synchronize()
    scene.check()
    for si in range(2):
        for o in outs[si]:
            got = o.cpu().numpy().view(rt.HIT_DTYPE)
            assert np.array_equal(got["tri"], want["tri"]) and np.array_equal(got["t"], want["t"])


def test_more_streams_than_one_block_of_queue_counters(rt):
    """A long-lived scene used by a host that makes a stream per request: 600 distinct live streams launch
    on one scene — more than the 256 counters of its first block; the scene adds blocks instead of failing, and every
    launch writes the single-stream records."""
    import torch
    from raytracer_amd import workloads
    tris = workloads.soup(5000, seed=8, edge=0.08)
    scene = rt.Scene(tris)
    cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, (160, 100))
    want = primary_frame(rt, scene, cam)
    n = cam.x_res * cam.y_res
    streams = [torch.cuda.Stream() for _ in range(600)]       # (torch hands out pooled handles: only the distinct ones count)
    distinct = len({s.cuda_stream for s in streams})
    hip = rt.api.C.CDLL("libamdhip64.so")
    raw = []
    for _ in range(max(0, 600 - distinct)):                    # top up with raw HIP streams so that > 256 distinct handles are live at once
        h = rt.api.C.c_void_p()
        assert hip.hipStreamCreate(rt.api.C.byref(h)) == 0
        raw.append(h)
    handles = sorted({s.cuda_stream for s in streams} | {h.value for h in raw})
    assert len(handles) > 256
    outs = [torch.zeros(n * rt.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    for k, h in enumerate(handles):
        scene.primary_dev(cam, outs[k % 8].data_ptr(), tile=(64, 64), stream=h)
        if k % 8 == 7:
            torch.cuda.synchronize()                           # (8 launches in flight at a time, each into its own buffer)
    torch.cuda.synchronize()
    scene.check()
    for o in outs:
        got = o.cpu().numpy().view(rt.HIT_DTYPE)
        assert np.array_equal(got["tri"], want["tri"]) and np.array_equal(got["t"], want["t"])
    for h in raw:
        assert hip.hipStreamDestroy(h) == 0


@pytest.mark.parametrize("n", [11, 257, 1000])
def test_jacobi_sweep_vs_oracle_rgb(rt, oracle, n):
    import torch
    rng = np.random.default_rng(n)
    F = rng.random((n, n)).astype(np.float32)
    np.fill_diagonal(F, 0)
    F *= (0.9 / F.sum(1))[:, None].astype(np.float32)
    rho = np.array([0.84, 0.37, 0.7])
    E = (rng.random((n, 3)) * 100).astype(np.float32)
    ld = (n + 3) // 4 * 4
    Fp = np.zeros((n, ld), np.float32); Fp[:, :n] = F
    dF = torch.from_numpy(Fp).cuda(); dE = torch.from_numpy(E).cuda()
    Bold = torch.from_numpy((rng.random((n, 3)) * 50).astype(np.float32)).cuda()
    Bnew = torch.zeros_like(Bold)
    norms = torch.zeros(3, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    # two half-sweeps (row shards) = one sweep
    h = n // 2
    rt.jacobi_sweep_dev(dF.data_ptr(), ld, n, 0, h, rho, dE.data_ptr(), Bold.data_ptr(), Bnew.data_ptr(), norms.data_ptr(), st)
    rt.jacobi_sweep_dev(dF[h:].data_ptr(), ld, n, h, n - h, rho, dE.data_ptr(), Bold.data_ptr(), Bnew.data_ptr(),
                        norms.data_ptr(), st)
    torch.cuda.synchronize()
    want = E.astype(np.float64) + rho[None, :] * (F.astype(np.float64) @ Bold.cpu().numpy().astype(np.float64))
    got = Bnew.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - want) / np.abs(want)) < 2e-7          # one fp32 rounding of a double accumulation
    nm = norms.cpu().numpy()
    diff = got - Bold.cpu().numpy()
    assert abs(nm[0] - (diff ** 2).sum()) / nm[0] < 1e-12 and abs(nm[1] - ((got - E) ** 2).sum()) / nm[1] < 1e-12
    assert nm[2] == np.abs(diff).max()
    # whole solve vs the oracle's double Jacobi
    dB = torch.zeros_like(dE)
    stats = rt.api.rtmi_solve_stats()
    rc = rt.lib.rtmi_radiosity_solve_dev(dF.data_ptr(), ld, n, 0, n, rho.ctypes.data, dE.data_ptr(), dB.data_ptr(), 1e-7,
                                         500, 1, None, None, rt.api.C.byref(stats), st)
    assert rc == 0
    Bo, sweeps = oracle.jacobi_rgb(F, rho, E, tol_rel=1e-12)
    got = dB.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(got - Bo) / np.linalg.norm(Bo) < 1e-6
    assert 5 < stats.sweeps < 500 and stats.rel_change <= 1e-7


def test_sharded_solve_inside_the_library_one_rank(rt, monkeypatch):
    """rtmi_radiosity_solve_sharded_dev with a communicator of ONE rank (RCCL loaded at run time): the exchange path —
    in-place all-gather per sweep, all-reduced norms per check — on hardware.  A fixed number of sweeps gives the bits of
    rtmi_radiosity_solve_dev (same kernel launches); to tolerance the same sweep count and B; Radiosity.progressiveRefinement
    takes that path with RTMI_SOLVE_IN_LIBRARY=1; and the Python loop multi.solve_sharded drives the real sweep too."""
    import torch
    from raytracer_amd import multi
    n = 777
    rng = np.random.default_rng(9)
    F = rng.random((n, n)).astype(np.float32)
    np.fill_diagonal(F, 0)
    F *= (0.9 / F.sum(1))[:, None].astype(np.float32)
    ld = (n + 3) // 4 * 4
    Fp = np.zeros((n, ld), np.float32); Fp[:, :n] = F
    rho = np.array([0.84, 0.37, 0.7])
    E = (rng.random((n, 3)) * 100).astype(np.float32)
    dF = torch.from_numpy(Fp).cuda(); dE = torch.from_numpy(E).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for tol, sweeps_max, ce in ((0.0, 21, 4), (1e-6, 500, 3)):
        ref = torch.zeros_like(dE)
        s0 = rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), ref.data_ptr(), tol, sweeps_max, ce, st)
        B, s1 = multi.solve_sharded_library(dF.data_ptr(), ld, n, 0, n, rho, dE, tol_rel=tol, max_sweeps=sweeps_max, check_every=ce)
        torch.cuda.synchronize()
        assert s0.sweeps == s1.sweeps and np.array_equal(B.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))
        if tol > 0:      # (tolerance 0 = a fixed number of sweeps: the norms are never read back)
            assert s1.max_change == s0.max_change and abs(s1.rel_change - s0.rel_change) <= 1e-12 * max(s0.rel_change, 1e-300)
            assert s1.accelerated_sweeps == s0.accelerated_sweeps > 0 and s1.spectral_radius == s0.spectral_radius
        else:
            assert np.isnan(s0.rel_change) and np.isnan(s1.rel_change) and s1.accelerated_sweeps == 0
    # the Python loop over the same sweep (a group of one): the callback path on hardware
    def sweep(B_old, B_new, norms):
        rt.jacobi_sweep_dev(dF.data_ptr(), ld, n, 0, n, rho, dE.data_ptr(), B_old.data_ptr(), B_new.data_ptr(), norms.data_ptr(), st)
    r = multi.solve_sharded(sweep, dE, n, tol_rel=1e-6, max_sweeps=500, check_every=3)      # (plain Jacobi)
    monkeypatch.setenv("RTMI_SOLVER", "jacobi")
    Bj, sj = multi.solve_sharded_library(dF.data_ptr(), ld, n, 0, n, rho, dE, tol_rel=1e-6, max_sweeps=500, check_every=3)
    monkeypatch.delenv("RTMI_SOLVER")
    assert r.sweeps == sj.sweeps > s1.sweeps and np.array_equal(r.B.cpu().numpy(), Bj.cpu().numpy())
    # a row range that is not the rank's block is refused when there is more than one rank; with one rank any range runs
    B2, s2 = multi.solve_sharded_library(dF.data_ptr(), ld, n, 0, 100, rho, dE, tol_rel=0.0, max_sweeps=2, check_every=1)
    assert s2.sweeps == 2
    # the member-function mirror
    g = load_golden("g4_practice_room_enclosed.npz")
    rad = rt.Radiosity(rt.Scene(g["verts"]), g["normals"], g["light_center"])
    rad.findFormFactor(samples=int(g["S"]), light_samples=int(g["SL"]), seed=int(g["seed"]))
    rad.progressiveRefinement(tol_rel=1e-7)
    Bplain = rad.B.cpu().numpy().copy()
    monkeypatch.setenv("RTMI_SOLVE_IN_LIBRARY", "1")
    rad.progressiveRefinement(tol_rel=1e-7)
    assert np.array_equal(rad.B.cpu().numpy(), Bplain)


def test_jacobi_argument_errors(rt):
    import torch
    x = torch.zeros(64, device="cuda")
    rho = np.array([0.7, 0.7, 0.7])
    with pytest.raises(rt.RtmiError):
        rt.jacobi_sweep_dev(x.data_ptr(), 3, 3, 0, 3, rho, x.data_ptr(), x.data_ptr(), x.data_ptr() + 16, x.data_ptr(), 0)  # ld % 4
    with pytest.raises(rt.RtmiError):
        rt.jacobi_sweep_dev(x.data_ptr(), 4, 3, 2, 3, rho, x.data_ptr(), x.data_ptr(), x.data_ptr() + 16, x.data_ptr(), 0)  # rows > n
    with pytest.raises(rt.RtmiError):
        rt.jacobi_sweep_dev(x.data_ptr(), 4, 3, 0, 3, rho, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 0)       # in place


@pytest.mark.parametrize("stem", ["radiosity_test", "practice_room_enclosed"])
def test_radiosity_display_path(rt, oracle, stem):
    """rtmi_radiosity_trace_dev / rtmi_radiosity_raster_dev against the compiled reference's trace_ray + ToneMapRadiosity
    (golden, every 4th pixel) and against the oracle on the whole frame: no transcendental on the device, so bit-exact"""
    g = load_golden("g8_radiosity_display.npz")
    G = lambda k: g[stem + "/" + k]
    scene = rt.Scene(G("verts"))
    rad = rt.Radiosity(scene, G("normals"))
    rad.radiosityValue = G("B")
    cam = rt.Camera.from_cam14(G("cam14"), G("res"))
    rgb, pix = rad.Radiosity_Raster(cam, G("vertex_rad"))
    k = G("pix")[:, 0].astype(np.int64), G("pix")[:, 1].astype(np.int64)
    assert np.array_equal(rgb[k], G("rgb"))
    assert np.array_equal(pix[k], G("pixels"))
    # whole frame vs the oracle, and the batch entry point on the same rays
    allpix = np.array([(i, j) for i in range(cam.y_res) for j in range(cam.x_res)], np.uint32)
    d = oracle.primary_dirs(oracle.raster_details(G("cam14"), G("res")[0], G("res")[1]), allpix)
    o = np.tile(G("cam14")[:3], (len(d), 1))
    want = oracle.radiosity_trace(G("verts"), G("normals"), G("vertex_rad"), o, d)
    assert np.array_equal(rgb.reshape(-1, 3), want)
    assert np.array_equal(pix.reshape(-1, 3), oracle.tone_map_radiosity(want))
    assert np.array_equal(rad.trace_ray(rt.make_rays(o[::3], d[::3]), G("vertex_rad")), want[::3])
    # tiles of two ranks; flat shading = vertex radiosity B, B, B
    a, _ = rad.Radiosity_Raster(cam, G("vertex_rad"), tile=(50, 40), tile_first=0, tile_stride=2)
    b, _ = rad.Radiosity_Raster(cam, G("vertex_rad"), tile=(50, 40), tile_first=1, tile_stride=2)
    assert np.array_equal(np.where(a != 0, a, b), rgb)
    flat, _ = rad.Radiosity_Raster(cam)
    assert np.array_equal(flat.reshape(-1, 3), oracle.radiosity_trace(G("verts"), G("normals"), np.repeat(G("B")[:, None], 3, 1), o, d))
    # smooth shading without the reference host: the library's own shared-vertex averages (rtmi_radiosity_vertex_values,
    # pinned by g10 to the reference's returnQuadTreeColor) through the same display kernels
    smooth, _ = rad.Radiosity_Raster(cam, "shared")
    assert np.array_equal(smooth.reshape(-1, 3), oracle.radiosity_trace(G("verts"), G("normals"), oracle.vertex_values(G("verts"), G("B")), o, d))
    assert (smooth != flat).any()


def test_radiosity_display_bvh_equals_the_full_loop(rt, oracle, monkeypatch):
    """The display query through the elements' BVH gives the very colours of the reference's loop over all elements:
    forced on for the golden scenes (10 / 48 elements) against the reference's vectors, and on the 3 888-patch box from
    inside, from outside, and from an eye lying exactly on a wall (the distance-0 rule of trace_ray)."""
    from raytracer_amd import workloads
    g = load_golden("g8_radiosity_display.npz")
    monkeypatch.setenv("RTMI_DISPLAY_QUERY", "2")
    for stem in ("radiosity_test", "practice_room_enclosed"):
        G = lambda k: g[stem + "/" + k]
        rad = rt.Radiosity(rt.Scene(G("verts")), G("normals"))
        rgb, pix = rad.Radiosity_Raster(rt.Camera.from_cam14(G("cam14"), G("res")), G("vertex_rad"))
        k = G("pix")[:, 0].astype(np.int64), G("pix")[:, 1].astype(np.int64)
        assert np.array_equal(rgb[k], G("rgb")) and np.array_equal(pix[k], G("pixels"))
    verts, normals, _ = workloads.closed_box(18)
    rad = rt.Radiosity(rt.Scene(verts), normals)
    vr = np.random.default_rng(5).random((len(verts), 3))
    # trace_ray takes the nearest element on the LINE of the ray, also behind the eye (and then answers Black for its back
    # face): the eyes look toward the walls they are nearer to
    cams = [np.array([3.0, 4.0, 6.0, 0.0, 2.0, 3.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.75, 0.75]),
            np.array([-6.0, -9.0, 14.0, 5.0, 5.0, 5.0, 0, 0, 1, 2.0, -1.0, 1.0, -0.75, 0.75]),
            np.array([0.0, 2.5, 2.5, 9.0, 7.0, 2.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.75, 0.75])]      # eye on the wall x = 0
    for c in cams:
        cam = rt.Camera.from_cam14(c, (320, 240))
        monkeypatch.setenv("RTMI_DISPLAY_QUERY", "2")
        a, _ = rad.Radiosity_Raster(cam, vr)
        monkeypatch.setenv("RTMI_DISPLAY_QUERY", "1")
        b, _ = rad.Radiosity_Raster(cam, vr)
        assert np.array_equal(a, b)
    inside, _ = rad.Radiosity_Raster(rt.Camera.from_cam14(cams[0], (320, 240)), vr)
    assert len(np.unique(inside[..., 0])) > 1000
