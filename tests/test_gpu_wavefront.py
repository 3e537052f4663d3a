"""The device-resident wavefront: primary hits -> shadow / ambient-occlusion rays generated on the GPU -> any-hit,
against the oracle's restatement of Shader::generateNormalShadows / findOcclusionNew and the golden vectors of the
compiled reference."""
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


class Frame:
    """primary rays + hits of a golden scene, resident on the device"""

    def __init__(self, rt, g):
        import torch
        self.scene = rt.Scene(g["tris"], g["tri_obj"])
        self.cam = rt.Camera.from_cam14(g["cam"], g["res"])
        self.n = self.cam.x_res * self.cam.y_res
        self.st = torch.cuda.current_stream().cuda_stream
        self.rays = torch.zeros(self.n * 64, dtype=torch.uint8, device="cuda")
        self.hits = torch.zeros(self.n * 24, dtype=torch.uint8, device="cuda")
        self.normals = torch.from_numpy(np.ascontiguousarray(g["tri_normal"])).cuda()
        rt.primary_rays_dev(self.cam, self.rays.data_ptr(), stream=self.st)
        self.scene.cast_dev(self.rays.data_ptr(), self.n, self.hits.data_ptr(), stream=self.st)
        torch.cuda.synchronize()
        self.h_rays = self.rays.cpu().numpy().view(rt.RAY_DTYPE)
        self.h_hits = self.hits.cpu().numpy().view(rt.HIT_DTYPE)


def test_shadow_wavefront_test1(rt, oracle):
    import torch
    g = load_golden("g2_test1.npz")
    f = Frame(rt, g)
    out = torch.zeros(f.n * 64, dtype=torch.uint8, device="cuda")
    f.scene.shadow_rays_dev(f.normals.data_ptr(), f.rays.data_ptr(), f.hits.data_ptr(), f.n, g["light_center"], out.data_ptr(), f.st)
    occ = torch.zeros(f.n, dtype=torch.uint8, device="cuda")
    f.scene.occluded_dev(out.data_ptr(), f.n, occ.data_ptr(), f.st)
    torch.cuda.synchronize()
    sr = out.cpu().numpy().view(rt.RAY_DTYPE)
    hit = f.h_hits["tri"] != rt.MISS
    assert np.all(sr["reserved"][~hit] != 0) and np.all(sr["reserved"][hit] == 0)
    # the oracle's chain: hit point (object.cpp:23) -> shadow ray (shader.cpp:116-127): every double identical
    tri = f.h_hits["tri"][hit]
    P = oracle.hit_point(f.h_rays["o"][hit], f.h_rays["d"][hit], g["tris"][tri, 0:3], g["tri_normal"][tri])
    so, sd = oracle.shadow_rays(P, g["tri_normal"][tri], g["light_center"])
    assert np.array_equal(sr["o"][hit], so) and np.array_equal(sr["d"][hit], sd)
    # golden: the reference's own shadow rays and their occlusion, on the fixture's pixel subset
    k = (g["pix"][:, 0].astype(np.int64) * f.cam.x_res + g["pix"][:, 1])[g["shadow_src"]]
    assert np.max(np.abs(sr["o"][k] - g["shadow_o"])) < 1e-9 and np.max(np.abs(sr["d"][k] - g["shadow_d"])) < 1e-9
    o = occ.cpu().numpy()
    assert np.array_equal(o[k], g["shadow_occluded"])
    assert np.all(o[~hit] == 0)
    # the fused launch (shadow rays built inside the any-hit kernel) writes the same bytes
    occ2 = torch.full((f.n,), 7, dtype=torch.uint8, device="cuda")
    f.scene.shadow_occlusion_dev(f.normals.data_ptr(), f.rays.data_ptr(), f.hits.data_ptr(), f.n, g["light_center"], occ2.data_ptr(), f.st)
    torch.cuda.synchronize()
    assert torch.equal(occ, occ2)
    assert np.array_equal(o[hit], oracle.any_hit(g["tris"], so, sd))


def test_ao_wavefront_room(rt, oracle):
    import torch
    g = load_golden("g2_scene_room_bed_sample.npz")
    f = Frame(rt, g)
    spp, R, seed = 4, 2.0, 9
    out = torch.zeros(f.n * spp * 64, dtype=torch.uint8, device="cuda")
    f.scene.ao_rays_dev(f.normals.data_ptr(), f.rays.data_ptr(), f.hits.data_ptr(), f.n, spp, R, seed, out.data_ptr(), f.st)
    occ = torch.zeros(f.n * spp, dtype=torch.uint8, device="cuda")
    f.scene.occluded_dev(out.data_ptr(), f.n * spp, occ.data_ptr(), f.st)
    # the fused launch (rays built inside the any-hit kernel) writes the same bytes
    occ_fused = torch.full((f.n * spp,), 7, dtype=torch.uint8, device="cuda")
    f.scene.ao_occlusion_dev(f.normals.data_ptr(), f.rays.data_ptr(), f.hits.data_ptr(), f.n, spp, R, seed, occ_fused.data_ptr(), f.st)
    torch.cuda.synchronize()
    assert torch.equal(occ, occ_fused)
    frac = torch.zeros(f.n, dtype=torch.float64, device="cuda")
    rt.occlusion_fraction_dev(occ.data_ptr(), f.n, spp, frac.data_ptr(), f.st)
    torch.cuda.synchronize()
    ar = out.cpu().numpy().view(rt.RAY_DTYPE).reshape(f.n, spp)
    hit = f.h_hits["tri"] != rt.MISS
    assert np.all(ar["reserved"][~hit] != 0) and np.all(ar["reserved"][hit] == 0) and np.all(ar["tmax"][hit] == R)
    # oracle: same stream (seed, pixel index, sample); cos/sin are the only calls that may differ between the
    # device's and the host's libm, by an ulp: directions to 1e-14, occlusion bits equal on >= 99.99 % of rays
    tri = np.where(hit, f.h_hits["tri"], 0)
    P = oracle.hit_point(f.h_rays["o"], f.h_rays["d"], g["tris"][tri, 0:3], g["tri_normal"][tri])
    P[~hit] = 0
    oo, od = oracle.ao_rays(P, g["tri_normal"][tri], R, spp, seed)
    oo = oo.reshape(f.n, spp, 3); od = od.reshape(f.n, spp, 3)
    assert np.max(np.abs(ar["d"][hit] - od[hit])) < 1e-14 and np.max(np.abs(ar["o"][hit] - oo[hit])) < 1e-13
    assert (ar["d"][hit] == od[hit]).all(axis=-1).mean() > 0.9
    want = oracle.any_hit(g["tris"], oo[hit].reshape(-1, 3), od[hit].reshape(-1, 3), np.full(hit.sum() * spp, R)).reshape(-1, spp)
    got = occ.cpu().numpy().reshape(f.n, spp)
    assert (got[hit] == want).mean() >= 0.9999
    assert np.all(got[~hit] == 0)
    fr = frac.cpu().numpy()
    assert np.array_equal(fr, 1.0 - got.sum(1) / float(spp))
    assert 0.5 < fr[hit].mean() < 1.0


def test_ao_fraction_vs_the_references_findOcclusionNew(rt):
    """48 pixels of the room scene, 4000 AO rays each, against the compiled reference's own Monte-Carlo means."""
    import torch
    g = load_golden("g2_scene_room_bed_sample.npz")
    a = load_golden("g3_ao_room.npz")
    scene = rt.Scene(g["tris"], g["tri_obj"])
    hm = g["cast_hit"].astype(bool)
    sel = np.linspace(0, hm.sum() - 1, 48).astype(int)               # how oracle/gen_golden_ao.py chose its points
    pix = g["pix"][hm][sel]
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    O, dR, dU = cam.raster_details()
    d = O[None, :] + (pix[:, 1:2] + 0.5) * dR[None, :] - (pix[:, 0:1] + 0.5) * dU[None, :]
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = rt.make_rays(np.tile(g["cam"][:3], (48, 1)), d)
    hits = scene.Cast(rays)
    assert (hits["tri"] != rt.MISS).all()
    NR = int(a["nrays"])
    st = torch.cuda.current_stream().cuda_stream
    d_rays = torch.from_numpy(rays.view(np.uint8)).cuda(); d_hits = torch.from_numpy(hits.view(np.uint8)).cuda()
    nrm = torch.from_numpy(np.ascontiguousarray(g["tri_normal"])).cuda()
    out = torch.zeros(48 * NR * 64, dtype=torch.uint8, device="cuda")
    scene.ao_rays_dev(nrm.data_ptr(), d_rays.data_ptr(), d_hits.data_ptr(), 48, NR, float(a["radius"]), 3, out.data_ptr(), st)
    occ = torch.zeros(48 * NR, dtype=torch.uint8, device="cuda")
    scene.occluded_dev(out.data_ptr(), 48 * NR, occ.data_ptr(), st)
    frac = torch.zeros(48, dtype=torch.float64, device="cuda")
    rt.occlusion_fraction_dev(occ.data_ptr(), 48, NR, frac.data_ptr(), st)
    torch.cuda.synchronize()
    mine, ref = frac.cpu().numpy(), a["ref_ao"]
    se = np.sqrt(np.maximum(ref * (1 - ref), 1e-4) / NR) * np.sqrt(2.0)
    assert np.all(np.abs(mine - ref) <= 5 * se + 2e-3), np.abs(mine - ref).max()


def test_soft_shadow_wavefront_studio(rt, oracle):
    """primary hits -> 100 soft-shadow rays per hit generated on the device -> closest hit -> factor; against the oracle's
    restated rays (every double) and the compiled reference's own Shader::generateSoftShadows (golden)"""
    import torch
    g = load_golden("g6_shade_studio.npz")
    s = load_golden("g6_soft_shadow_studio.npz")
    n_tris = len(g["tris"])
    scene = rt.Scene(g["tris"], np.arange(n_tris, dtype=np.uint32))      # one object per triangle, as the golden scene
    cam = rt.Camera.from_cam14(g["cam14"], g["res"])
    n = cam.x_res * cam.y_res
    st = torch.cuda.current_stream().cuda_stream
    rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda"); hits = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    nrm = torch.from_numpy(np.ascontiguousarray(g["normals"])).cuda()
    rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
    scene.cast_dev(rays.data_ptr(), n, hits.data_ptr(), stream=st)
    NP = 100
    srays = torch.zeros(n * NP * 64, dtype=torch.uint8, device="cuda"); shits = torch.zeros(n * NP * 24, dtype=torch.uint8, device="cuda")
    fac = torch.zeros(n, dtype=torch.float64, device="cuda")
    scene.soft_shadow_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, g["lights"][0, :3], NP, srays.data_ptr(), st)
    scene.cast_dev(srays.data_ptr(), n * NP, shits.data_ptr(), stream=st)
    rt.soft_shadow_factor_dev(hits.data_ptr(), shits.data_ptr(), n, NP, fac.data_ptr(), st)
    torch.cuda.synchronize()
    k = s["ray_index"]
    h = hits.cpu().numpy().view(rt.HIT_DTYPE)
    assert np.array_equal(h["tri"][k].astype(np.int64), s["tri"].astype(np.int64))
    sr = srays.cpu().numpy().view(rt.RAY_DTYPE).reshape(n, NP)
    so, sd = oracle.soft_shadow_rays(s["P"], g["lights"][0, :3], NP)
    assert np.array_equal(sr["o"][k].reshape(-1, 3), so) and np.array_equal(sr["d"][k].reshape(-1, 3), sd)
    assert np.array_equal(fac.cpu().numpy()[k], s["factor"])
    miss = h["tri"] == rt.MISS
    assert np.all(sr["reserved"][miss] != 0) and np.all(fac.cpu().numpy()[miss] == 1.0)
    # the fused launch (rays built inside the closest-hit kernel): the same factors, and the per-ray verdicts
    contrib = torch.full((n * NP,), 9, dtype=torch.uint8, device="cuda")
    fac2 = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
    scene.soft_shadow_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, g["lights"][0, :3], NP, contrib.data_ptr(), fac2.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(fac, fac2)
    sh = shits.cpu().numpy().view(rt.HIT_DTYPE).reshape(n, NP)
    want = (sh["tri"] != rt.MISS) & (sh["object"] != h["object"][:, None]) & ~miss[:, None]
    assert np.array_equal(contrib.cpu().numpy().reshape(n, NP), want.astype(np.uint8))


def test_fused_secondary_ray_entries_argument_errors_and_empty_scene(rt):
    """rtmi_ao_occlusion_dev / rtmi_soft_shadow_dev: NULL buffers, a non-positive radius and zero points are refused with
    a message; zero rays is a no-op; an empty scene answers 'nothing hit' for every slot."""
    import torch
    st = torch.cuda.current_stream().cuda_stream
    scene = rt.Scene(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0.0]]))
    n, spp = 4, 3
    rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda"); hits = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    nrm = torch.zeros(3, dtype=torch.float64, device="cuda"); occ = torch.zeros(n * spp, dtype=torch.uint8, device="cuda")
    fac = torch.zeros(n, dtype=torch.float64, device="cuda")
    with pytest.raises(rt.RtmiError, match="radius"):
        scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 0.0, 1, occ.data_ptr(), st)
    with pytest.raises(rt.RtmiError, match="NULL"):
        scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), 0, n, spp, 2.0, 1, occ.data_ptr(), st)
    with pytest.raises(rt.RtmiError, match="NULL|n_points"):
        scene.soft_shadow_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, [0, 0, 1], 0, occ.data_ptr(), fac.data_ptr(), st)
    scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), 0, spp, 2.0, 1, occ.data_ptr(), st)   # n == 0
    # hit records that are all misses (tri = MISS): every slot 0, factor 1
    h = np.zeros(n, rt.HIT_DTYPE); h["tri"] = rt.MISS; h["object"] = rt.MISS
    hits.copy_(torch.from_numpy(h.view(np.uint8).copy()))
    occ.fill_(5); fac.fill_(-1.0)
    scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, occ.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(occ.sum()) == 0
    occ.fill_(5)
    scene.soft_shadow_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, [0, 0, 1], spp, occ.data_ptr(), fac.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(occ.sum()) == 0 and bool((fac == 1.0).all())
    empty = rt.Scene(np.zeros((0, 9)))
    occ.fill_(5); fac.fill_(-1.0)
    empty.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, occ.data_ptr(), st)
    torch.cuda.synchronize()
    assert int(occ.sum()) == 0
    empty.soft_shadow_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, [0, 0, 1], spp, occ.data_ptr(), fac.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool((fac == 1.0).all())
