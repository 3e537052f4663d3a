"""GPU parity at BASELINE.json's full sizes: a sample against the oracle plus size-independent properties
(closest-ness by re-tracing with tmax, any-hit == closest-hit != miss, tile shards == whole frame,
known-answer fixed point for the 64 K-patch solve)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    import raytracer_amd
    return raytracer_amd


@pytest.fixture(scope="module")
def soup(rt):
    from raytracer_amd import workloads
    tris = workloads.soup(1 << 20, seed=1)
    scene = rt.Scene(tris)
    cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, workloads.SOUP_RES)
    return tris, scene, cam


def frame(rt, scene, cam, **kw):
    import torch
    out = torch.zeros(cam.x_res * cam.y_res * 24, dtype=torch.uint8, device="cuda")
    scene.primary_dev(cam, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, **kw)
    torch.cuda.synchronize()
    return out


def test_c3_soup_frame_vs_oracle_sample_and_properties(rt, oracle, soup):
    import torch
    from raytracer_amd import workloads
    tris, scene, cam = soup
    info = scene.info()
    assert info.n_tris == 1 << 20 and info.max_depth <= 30
    d_hits = frame(rt, scene, cam, tile=(64, 64))
    hits = d_hits.cpu().numpy().view(rt.HIT_DTYPE)
    n = len(hits)
    hit = hits["tri"] != rt.MISS
    assert 0.5 < hit.mean() < 0.95
    # (1) 300 000 random pixels against the oracle (its own BVH, the reference's Möller–Trumbore): bit-exact
    rng = np.random.default_rng(11)
    pix = np.stack([rng.integers(0, cam.y_res, 300000), rng.integers(0, cam.x_res, 300000)], 1).astype(np.uint32)
    d = oracle.primary_dirs(oracle.raster_details(workloads.SOUP_CAM14, cam.x_res, cam.y_res), pix)
    o = np.tile(workloads.SOUP_CAM14[:3], (len(d), 1))
    h = oracle.bvh_build(tris)
    (tri, t, u, v), _ = oracle.bvh_closest_hit(h, o, d)
    oracle.bvh_free(h)
    got = hits[pix[:, 0].astype(np.int64) * cam.x_res + pix[:, 1]]
    assert np.array_equal(got["tri"], np.where(tri < 0, rt.MISS, tri).astype(np.uint32))
    assert np.array_equal(got["t"], t)
    assert np.array_equal(got["u"], u.astype(np.float32)) and np.array_equal(got["v"], v.astype(np.float32))
    # and 2 000 of them against brute force over all 1 Mi triangles (no BVH on either side of the comparison)
    tb, ttb, _, _ = oracle.closest_hit(tris, o[:2000], d[:2000])
    assert np.array_equal(tb, tri[:2000]) and np.array_equal(ttb, t[:2000])
    # (2) whole frame through the ray-buffer path: same records; any-hit == (closest != miss)
    st = torch.cuda.current_stream().cuda_stream
    rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
    out2 = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    scene.cast_dev(rays.data_ptr(), n, out2.data_ptr(), stream=st)
    occ = torch.zeros(n, dtype=torch.uint8, device="cuda")
    scene.occluded_dev(rays.data_ptr(), n, occ.data_ptr(), stream=st)
    torch.cuda.synchronize()
    assert torch.equal(out2, d_hits)
    assert np.array_equal(occ.cpu().numpy().astype(bool), hit)
    # (3) closest-ness: with tmax = t nothing is hit (no triangle has 0 < t' < t); with tmax just above t the
    #     same record comes back
    r = rays.cpu().numpy().view(rt.RAY_DTYPE).copy()
    sel = np.nonzero(hit)[0][::3]
    r1 = r[sel]; r1["tmax"] = hits["t"][sel]
    assert (scene.Cast(r1)["tri"] == rt.MISS).all()
    assert (scene.Occluded(r1) == 0).all()
    r1["tmax"] = np.nextafter(hits["t"][sel], np.inf)
    assert scene.Cast(r1).tobytes() == hits[sel].tobytes()
    # (4) tile shards of two 'ranks' (different tile shape) put together == the frame
    a = frame(rt, scene, cam, tile=(128, 32), tile_first=0, tile_stride=2)
    b = frame(rt, scene, cam, tile=(128, 32), tile_first=1, tile_stride=2)
    assert torch.equal(torch.maximum(a, b), d_hits) and int((torch.minimum(a, b) != 0).sum()) == 0
    # (5) work counters are sane
    counts = torch.zeros(6, dtype=torch.int64, device="cuda")
    scene.primary_work_dev(cam, counts.data_ptr(), stream=st)
    torch.cuda.synchronize()
    visits, tests, vslots, tslots, wleaf, widle = (int(x) for x in counts.cpu())
    assert n < visits < 400 * n and hit.sum() <= tests < 100 * n


def test_c3_triangle_order_does_not_matter(rt, soup):
    """Permuting the input triangles permutes the ids and changes nothing else (a different BVH is built)."""
    from raytracer_amd import workloads
    tris, scene, cam = soup
    small = rt.Camera.from_cam14(workloads.SOUP_CAM14, (480, 270))
    h0 = frame(rt, scene, small, tile=(0, 0)).cpu().numpy().view(rt.HIT_DTYPE)
    perm = np.random.default_rng(2).permutation(len(tris))
    s2 = rt.Scene(tris[perm])
    h1 = frame(rt, s2, small, tile=(0, 0)).cpu().numpy().view(rt.HIT_DTYPE)
    m = h0["tri"] != rt.MISS
    assert np.array_equal(m, h1["tri"] != rt.MISS)
    assert np.array_equal(perm[h1["tri"][m]], h0["tri"][m])
    assert np.array_equal(h0["t"], h1["t"]) and np.array_equal(h0["u"], h1["u"]) and np.array_equal(h0["v"], h1["v"])


def test_c5_solver_known_answer_64k(rt):
    """64 K patches, 16 GiB F with every row summing to 0.9: E = (1 − 0.9ρ)·c has the fixed point B ≡ c."""
    import torch
    from raytracer_amd import workloads
    n = 65536
    F = workloads.synthetic_F(n, seed=1)
    # rows of the matrix against numpy on a few rows
    rows = [0, 1, 4099, n - 1]
    Fr = F[rows].cpu().numpy().astype(np.float64)
    assert np.allclose(Fr.sum(1), 0.9, atol=1e-5) and all(Fr[k, r] == 0 for k, r in enumerate(rows))
    rho = np.array([0.7, 0.5, 0.3])
    c = np.array([100.0, 20.0, 3.0])
    E = torch.from_numpy(np.tile(((1 - 0.9 * rho) * c).astype(np.float32), (n, 1))).cuda()
    st = torch.cuda.current_stream().cuda_stream
    # one sweep from a random B against numpy on the sampled rows
    B0 = torch.rand((n, 3), device="cuda") * 10
    B1 = torch.zeros_like(B0)
    norms = torch.zeros(3, dtype=torch.float64, device="cuda")
    rt.jacobi_sweep_dev(F.data_ptr(), n, n, 0, n, rho, E.data_ptr(), B0.data_ptr(), B1.data_ptr(), norms.data_ptr(), st)
    torch.cuda.synchronize()
    want = E[rows].cpu().numpy().astype(np.float64) + rho[None, :] * (Fr @ B0.cpu().numpy().astype(np.float64))
    assert np.max(np.abs(B1[rows].cpu().numpy() - want) / want) < 2e-7
    # the whole solve
    B = torch.zeros_like(E)
    stats = rt.api.rtmi_solve_stats()
    rc = rt.lib.rtmi_radiosity_solve_dev(F.data_ptr(), n, n, 0, n, rho.ctypes.data, E.data_ptr(), B.data_ptr(), 1e-6, 200, 2,
                                         None, None, rt.api.C.byref(stats), st)
    assert rc == 0, rt.lib.rtmi_last_error()
    got = B.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(got - c[None, :]) / np.linalg.norm(np.tile(c, (n, 1))) < 1e-4     # the north star's bar
    assert np.max(np.abs(got / c[None, :] - 1)) < 2e-5
    assert 5 < stats.sweeps < 60


C5_BOX_ROWS = [0, 21000, 43210, 64743]     # first, two interior (different walls), last patch of the 64 744-patch box


def test_c5_form_factor_rows_of_the_64k_box_equal_the_oracle(rt, oracle):
    """The matrix bench.py builds (64 744-patch box with a ceiling opening, light above it): four of its rows, the light
    column and the light's own row against the oracle — bit-exact in fp32.  The oracle takes its closest hits through
    its own hierarchy here (rto_form_factors_bvh, pinned to the scan over all triangles on small scenes by
    tests/test_oracle_golden.py): one such row is 4·10^10 triangle tests by the scan."""
    from raytracer_amd import workloads
    verts, normals, light = workloads.closed_box(74, hole=0.3)
    n = len(verts)
    assert n == 64744
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, light)
    h = oracle.bvh_build(verts)
    for r in C5_BOX_ROWS:
        got = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=r, n_rows=1).cpu().numpy()[0]
        want = oracle.form_factors_bvh(verts, normals, 10, 1, row0=r, n_rows=1, bvh=h)[0].astype(np.float32)
        assert np.array_equal(got[:n].view(np.uint32), want.view(np.uint32)), "row %d" % r
        fl = oracle.light_terms(verts, normals, light, 30, 1, mirror_bug=1, row0=r, n_rows=1)
        assert got[n] == np.float32(fl[0])
        assert 0.5 < float(want.sum()) < 1.1 and got[r] == 0.0
    oracle.bvh_free(h)
    # the light's own row is zero; rows that straddle the end of the patches
    tail = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=n - 1, n_rows=2).cpu().numpy()
    assert np.all(tail[1] == 0.0) and tail[0, : n].sum() > 0.5
    # the blocks bench.py builds (4096 rows per launch) give the same rows as single-row launches
    blk = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=20480, n_rows=4096).cpu().numpy()
    one = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=21000, n_rows=1).cpu().numpy()[0]
    assert np.array_equal(blk[21000 - 20480], one)
    assert rt.lib.rtmi_debug_form_factor_kernel(rad.handle) == 1        # measured on that call's first rows: in a room the path walk wins
    # the CRCs bench.py prints for these rows of the matrix it times (formfactor.crc32_of_pinned_rows) are the oracle's,
    # committed by oracle/gen_golden_c5_rows.py: bench line and suite cannot drift apart
    import json, os, zlib
    from conftest import GOLDEN
    pin = json.load(open(os.path.join(GOLDEN, "c5_box_rows.json")))
    assert pin["patches"] == n and sorted(int(k) for k in pin["crc32"]) == C5_BOX_ROWS
    for r in C5_BOX_ROWS:
        got = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=r // 4096 * 4096, n_rows=min(4096, n + 1 - r // 4096 * 4096)).cpu().numpy()[r % 4096]
        assert "%08x" % zlib.crc32(got[: n + 1].tobytes()) == pin["crc32"][str(r)], "row %d" % r
    scene.check()


def test_c5_whole_64k_matrix_path_walk_equals_root_traversal(rt, monkeypatch):
    """Every one of the 64 744 x 64 745 entries bench.py times: the path walk (plane flags, certain-hit shortcut, per-pair
    beam) against the root-traversal kernel that tests every ray from the root — compared on the device block by block
    (4096 rows, 1 GB each), bit for bit.  ≈ 0.6 s + 4 s of GPU time."""
    import torch
    from raytracer_amd import workloads
    verts, normals, light = workloads.closed_box(74, hole=0.3)
    n = len(verts)
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, light)
    ld = (n + 1 + 3) // 4 * 4
    A = torch.empty((4096, ld), dtype=torch.float32, device="cuda")
    Bf = torch.empty((4096, ld), dtype=torch.float32, device="cuda")
    lc = np.ascontiguousarray(light, np.float64)
    st = torch.cuda.current_stream().cuda_stream
    differing, nonzero = 0, 0
    for first in range(0, n + 1, 4096):
        cnt = min(4096, n + 1 - first)
        for kernel, out in (("path", A), ("flat", Bf)):
            monkeypatch.setenv("RTMI_FF_KERNEL", kernel)
            rc = rt.lib.rtmi_form_factors_dev(rad.handle, lc.ctypes.data, 10, 30, 1, 0, first, cnt, out.data_ptr(), ld, st)
            assert rc == 0, rt.lib.rtmi_last_error().decode()
        torch.cuda.synchronize()
        differing += int((A[:cnt, : n + 1].view(torch.int32) != Bf[:cnt, : n + 1].view(torch.int32)).sum())
        nonzero += int((A[:cnt, :n] > 0).sum())
        assert differing == 0, "rows %d..%d" % (first, first + cnt)
    monkeypatch.delenv("RTMI_FF_KERNEL", raising=False)
    assert rt.lib.rtmi_debug_form_factor_planes(rad.handle) == 1
    assert 0.7 < nonzero / float(n) ** 2 < 0.85                  # pairs on different walls see each other (5/6 of all, less the opening)
    scene.check()


def test_frame_path_full_size_properties(rt, oracle, monkeypatch):
    """1920x1080 frames of the 147 K-triangle studio through rtmi_render_frame_dev (depth 3, 10 AO rays): properties that
    do not depend on the size — (a) a sample of pixels equals the oracle's recursion (same seeded AO stream), (b) the
    frame does not depend on how it is cut into chunks or tiles, nor on the order in which the compacted generations
    happened to be filled (two runs are identical), (c) emission is linear: with ambient 0 the frame lit by two lights is
    the sum of the frames lit by each alone where light 0 casts no shadow."""
    import torch
    from raytracer_amd import workloads as W
    verts, obj, mats, lights = W.studio(64, 0.1)
    nrm = W.face_normals(verts)
    scene = rt.Scene(verts, obj)
    cam = rt.Camera.from_cam14(W.STUDIO_CAM14, (1920, 1080))
    sh = rt.Shader(scene, nrm, mats, lights, max_tree_depth=3, ao_rays=10, seed=4)
    rgb, pix = sh.Rasterize(cam)
    # (a)
    rng = np.random.default_rng(3)
    pl = np.stack([rng.integers(0, 1080, 3000), rng.integers(0, 1920, 3000)], 1).astype(np.uint32)
    d = oracle.primary_dirs(oracle.raster_details(W.STUDIO_CAM14, 1920, 1080), pl)
    o = np.tile(W.STUDIO_CAM14[:3], (len(d), 1))
    bvh = oracle.bvh_build(verts)
    idx = pl[:, 0].astype(np.int64) * 1920 + pl[:, 1]
    want = np.concatenate([oracle.trace(verts, nrm, obj, mats, lights, o[i:i + 1], d[i:i + 1], max_depth=3, ao_rays=10, seed=4,
                                        first_index=int(idx[i]), bvh=bvh) for i in range(len(pl))])
    oracle.bvh_free(bvh)
    assert np.abs(rgb.reshape(-1, 3)[idx] - want).max() < 1e-11
    # (b)
    again, _ = sh.Rasterize(cam)
    assert np.array_equal(again, rgb)
    monkeypatch.setenv("RTMI_SHADE_WS_MB", "256")
    sh_small = rt.Shader(scene, nrm, mats, lights, max_tree_depth=3, ao_rays=10, seed=4)
    chunks, _ = sh_small.Rasterize(cam)
    monkeypatch.delenv("RTMI_SHADE_WS_MB")
    assert np.array_equal(chunks, rgb)
    parts = [sh.Rasterize(cam, tile=(64, 64), tile_first=r, tile_stride=4)[0] for r in range(4)]
    assert np.array_equal(sum(parts), rgb) and all(((p != 0).any(2)).sum() > 400000 for p in parts)
    # (c)
    _, _, mats0, _ = W.studio(64, 0.0)
    far = lights.copy()
    far[0, :3] = [0.0, 0.0, 1.0e6]            # light 0 (the shadow caster) straight above, nothing above the scene's top faces
    f01, _ = rt.Shader(scene, nrm, mats0, far, max_tree_depth=1, ao_rays=0).Rasterize(cam)
    l0 = far.copy(); l0[1, 3:] = 0.0
    l1 = far.copy(); l1[0, 3:] = 0.0
    f0, _ = rt.Shader(scene, nrm, mats0, l0, max_tree_depth=1, ao_rays=0).Rasterize(cam)
    f1, _ = rt.Shader(scene, nrm, mats0, l1, max_tree_depth=1, ao_rays=0).Rasterize(cam)
    hits = torch.zeros(1920 * 1080 * 24, dtype=torch.uint8, device="cuda")
    scene.primary_dev(cam, hits.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ob = hits.cpu().numpy().view(rt.HIT_DTYPE)["object"].reshape(1080, 1920)
    lin = (ob != rt.MISS) & (ob != 3)          # object 3 = the microfacet box
    assert lin.sum() > 1000000 and np.abs((f0 + f1) - f01)[lin].max() < 1e-12


def test_c4_room_fused_secondary_rays_equal_the_stored_ray_paths(rt):
    """C4 at full size (99 804-triangle room, 1920x1080, 16 AO rays per pixel = 33 M rays; hard shadows; soft shadows on a
    strip of the frame): the launches that build their rays inside the query kernel write exactly what the
    generate -> store -> query paths write, and the occlusion statistics are those of a closed room."""
    import torch
    from raytracer_amd import workloads
    room, normals, light = workloads.closed_box(91)
    scene = rt.Scene(room)
    cam = rt.Camera.from_cam14(np.array([1.0, 1.0, 5.0, 6.0, 6.0, 1.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.5625, 0.5625]), (1920, 1080))
    n, spp = cam.x_res * cam.y_res, 16
    st = torch.cuda.current_stream().cuda_stream
    rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda"); hits = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    nrm = torch.from_numpy(np.ascontiguousarray(normals)).cuda()
    rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
    scene.cast_dev(rays.data_ptr(), n, hits.data_ptr(), stream=st)
    # ambient occlusion
    ao = torch.zeros(n * spp * 64, dtype=torch.uint8, device="cuda")
    occ = torch.zeros(n * spp, dtype=torch.uint8, device="cuda"); occ2 = torch.full_like(occ, 3)
    scene.ao_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, ao.data_ptr(), st)
    scene.occluded_dev(ao.data_ptr(), n * spp, occ.data_ptr(), st)
    scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, occ2.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(occ, occ2)
    frac = float(occ.float().mean())
    assert 0.02 < frac < 0.3                                   # corners and edges of the room occlude, open floor does not
    del ao
    # hard shadows: the light hangs inside the closed room, every surface point that faces it sees it
    sh = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    socc = torch.zeros(n, dtype=torch.uint8, device="cuda"); socc2 = torch.full_like(socc, 3)
    scene.shadow_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, light, sh.data_ptr(), st)
    scene.occluded_dev(sh.data_ptr(), n, socc.data_ptr(), st)
    scene.shadow_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, light, socc2.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(socc, socc2)
    # soft shadows on the first 64 rows
    m, NP = 64 * cam.x_res, 100
    srays = torch.zeros(m * NP * 64, dtype=torch.uint8, device="cuda"); shits = torch.zeros(m * NP * 24, dtype=torch.uint8, device="cuda")
    fac = torch.zeros(m, dtype=torch.float64, device="cuda"); fac2 = torch.full_like(fac, -1.0)
    contrib = torch.zeros(m * NP, dtype=torch.uint8, device="cuda")
    scene.soft_shadow_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), m, light, NP, srays.data_ptr(), st)
    scene.cast_dev(srays.data_ptr(), m * NP, shits.data_ptr(), stream=st)
    rt.soft_shadow_factor_dev(hits.data_ptr(), shits.data_ptr(), m, NP, fac.data_ptr(), st)
    scene.soft_shadow_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), m, light, NP, contrib.data_ptr(), fac2.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(fac, fac2)
