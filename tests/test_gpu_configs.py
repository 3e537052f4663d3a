"""The BASELINE.json configurations that are parity cases rather than bench lines:
C2 (closed 'Cornell' box, ~4 K patches: form factors + solve) and C4 (room, incoherent AO + soft-shadow any-hit
rays, 16 spp, tiles sharded over ranks)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    import raytracer_amd
    return raytracer_amd


def test_c2_closed_box_4k_patches(rt, oracle):
    from raytracer_amd import workloads
    verts, normals, light = workloads.closed_box(18, hole=0.34)     # 3 888 − 72 ceiling patches: light above the opening
    n = len(verts)
    assert n == 3816
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, light)
    F = rad.findFormFactor(samples=10, light_samples=30, seed=1).cpu().numpy()[:, : n + 1]
    # rows against the oracle (brute force over all patches, the reference's formula in double): bit-exact in fp32
    for r in (0, 1234, n - 1):
        want = oracle.form_factors(verts, normals, 10, 1, row0=r, n_rows=1)[0]
        fl = oracle.light_terms(verts, normals, light, 30, 1, mirror_bug=1, row0=r, n_rows=1)[0]
        assert np.array_equal(F[r, :n], want.astype(np.float32)) and F[r, n] == np.float32(fl)
    # a closed environment: every patch sees patches in all directions, row sums ≈ 1 (Monte-Carlo, 10 samples)
    rs = F[:n, :n].sum(1)
    assert 0.93 < rs.mean() < 1.0 and rs.min() > 0.5 and rs.max() < 1.6   # what leaves through the opening is missing
    assert (F[:n, n] > 0).sum() > 100                                     # patches that see the light through it
    assert np.all(F[n] == 0) and np.all(np.diag(F) == 0)
    # solve: against the exact fixed point of (I − ρF)B = E on the same F (LAPACK, double)
    rad.progressiveRefinement(tol_rel=1e-7, check_every=2)
    B = rad.setRadiosityForAllElements()
    K = np.eye(n + 1) - 0.70 * F.astype(np.float64)
    K[n] = 0; K[n, n] = 1
    E = np.zeros(n + 1); E[n] = 68500.0
    Bd = np.linalg.solve(K, E)[:n]
    assert np.linalg.norm(B - Bd) / np.linalg.norm(Bd) < 1e-4          # north star
    assert np.linalg.norm(B - Bd) / np.linalg.norm(Bd) < 5e-6          # what fp32 storage actually gives
    assert rad.stats.sweeps < 200


def test_c2_small_box_vs_reference_algorithm(rt, oracle):
    """768-patch box: the GPU solve against the reference's own progressive-refinement algorithm (restated in the
    oracle, pinned bit-exactly to the compiled reference on the golden scenes) run on the same F."""
    from raytracer_amd import workloads
    verts, normals, light = workloads.closed_box(8, hole=0.4)
    n = len(verts)
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, light)
    F = rad.findFormFactor(samples=10, light_samples=30, seed=3).cpu().numpy()[:, : n + 1].astype(np.float64)
    rad.progressiveRefinement(tol_rel=1e-7, check_every=2)
    B = rad.setRadiosityForAllElements()
    Bref, shots = oracle.progressive_refinement(F, oracle.area(verts), max_shots=2000000)
    assert shots > 0 and np.linalg.norm(Bref[:n]) > 0
    assert np.linalg.norm(B - Bref[:n]) / np.linalg.norm(Bref[:n]) < 1e-4


def _primary(rt, scene, cam):
    import torch
    n = cam.x_res * cam.y_res
    rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
    scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st)
    torch.cuda.synchronize()
    return rays.cpu().numpy().view(rt.RAY_DTYPE), out.cpu().numpy().view(rt.HIT_DTYPE)


def test_c4_room_scene_ao_and_soft_shadows_16spp(rt, oracle):
    """scene_room_bed_sample (152 triangles, the reference's own room): 16 AO + 16 soft-shadow any-hit rays per
    hit pixel of a strided sample, against brute force with the reference's triangle test."""
    from raytracer_amd import workloads
    g = load_golden("g2_scene_room_bed_sample.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    rays, hits = _primary(rt, scene, cam)
    sel = np.nonzero(hits["tri"] != rt.MISS)[0][::5]
    assert len(sel) > 5000
    P = workloads.hit_points(rays["o"][sel], rays["d"][sel], hits["t"][sel])
    N = g["tri_normal"][hits["tri"][sel]]
    N = np.where((N * rays["d"][sel]).sum(1)[:, None] > 0, -N, N)      # face the viewer
    o, d, tmax = workloads.ao_rays(P, N, 16)
    occ = scene.Occluded(rt.make_rays(o, d, tmax))
    assert np.array_equal(occ, oracle.any_hit(g["tris"], o, d, tmax))
    assert 0.02 < occ.mean() < 0.98
    o, d, tmax = workloads.soft_shadow_rays(P, g["light_center"], 16)
    sh = scene.Occluded(rt.make_rays(o, d, tmax))
    assert np.array_equal(sh, oracle.any_hit(g["tris"], o, d, tmax))
    assert 0.0 < sh.mean() < 1.0


def test_c4_tessellated_room_100k_tris_incoherent_any_hit(rt, oracle):
    """A ~100 K-triangle room (closed box, 91×91×2 per wall, plus an inner box): AO + shadow rays from the primary
    hits, any-hit on the GPU against the oracle's BVH closest hit (occluded ⇔ closest t < tmax); the pixel set is
    the union of two 'ranks'' tiles."""
    import torch
    from raytracer_amd import multi, workloads
    room, rn, light = workloads.closed_box(91)                          # 99 372 triangles
    inner, _, _ = workloads.closed_box(6, side=2.0)
    inner = inner + np.tile([4.0, 4.0, 0.0], 3)[None, :]
    tris = np.concatenate([room, inner])
    scene = rt.Scene(tris)
    cam14 = np.array([1.0, 1.0, 5.0, 6.0, 6.0, 1.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.5625, 0.5625])
    cam = rt.Camera.from_cam14(cam14, (960, 540))
    rays, hits = _primary(rt, scene, cam)
    assert (hits["tri"] != rt.MISS).all()                               # inside a closed room every ray hits
    # the two ranks' pixel sets are disjoint and complete (tiles 64×64 round-robin)
    assert sum(multi.pixels_of_rank(960, 540, 64, 64, r, 2) for r in range(2)) == 960 * 540
    sel = np.arange(0, len(hits), 7)
    P = workloads.hit_points(rays["o"][sel], rays["d"][sel], hits["t"][sel])
    e1 = tris[hits["tri"][sel], 3:6] - tris[hits["tri"][sel], 0:3]
    e2 = tris[hits["tri"][sel], 6:9] - tris[hits["tri"][sel], 0:3]
    N = np.cross(e1, e2); N /= np.linalg.norm(N, axis=1)[:, None]
    N = np.where((N * rays["d"][sel]).sum(1)[:, None] > 0, -N, N)
    o, d, tmax = workloads.ao_rays(P, N, 16)
    so, sd, stmax = workloads.soft_shadow_rays(P, light, 16)
    o = np.concatenate([o, so]); d = np.concatenate([d, sd]); tmax = np.concatenate([tmax, stmax])
    occ = scene.Occluded(rt.make_rays(o, d, tmax))
    h = oracle.bvh_build(tris)
    (tri, t, _, _), _ = oracle.bvh_closest_hit(h, o, d, tmax)
    oracle.bvh_free(h)
    assert np.array_equal(occ.astype(bool), tri >= 0)
    assert 0.01 < occ.mean() < 0.9
    # closest hit on the same incoherent rays, bit-exact
    got = scene.Cast(rt.make_rays(o[::4], d[::4], tmax[::4]))
    assert np.array_equal(got["tri"], np.where(tri[::4] < 0, rt.MISS, tri[::4]).astype(np.uint32))
    assert np.array_equal(got["t"], t[::4])
