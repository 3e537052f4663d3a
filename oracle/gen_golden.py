#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by RUNNING THE COMPILED REFERENCE.

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python oracle/gen_golden.py
The fixtures hold inputs and the reference's outputs (data only, SURVEY.md §8c G1–G5); the GPU
box never sees the reference.  The reference's BSP build and its form-factor sampler are
unseeded (scene.cpp:544-550, util.cpp:99-100), so anything that depends on them (the radiosity
element list = BSP leaves) is whatever this particular run produced; the fixture stores it.
"""
import json
import os
import sys
import tempfile
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_py import Oracle, Reference  # noqa: E402

REF_SCENES = "/root/reference/Render2/scenes"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def workdir():
    wd = tempfile.mkdtemp(prefix="rtref_")
    os.symlink(REF_SCENES, os.path.join(wd, "scenes"))
    return wd


def g1_mt(ref, rng):
    """G1: Möller–Trumbore known-answer set incl. edge / vertex / parallel / behind / det≈1e-8."""
    n = 4096
    v0 = rng.uniform(-5, 5, (n, 3)); v1 = v0 + rng.uniform(-2, 2, (n, 3)); v2 = v0 + rng.uniform(-2, 2, (n, 3))
    o = rng.uniform(-10, 10, (n, 3))
    # aim most rays at a point in/near the triangle
    a = rng.uniform(-0.2, 1.2, n); b = rng.uniform(-0.2, 1.2, n)
    tgt = v0 + a[:, None] * (v1 - v0) + b[:, None] * (v2 - v0)
    k = np.arange(n)
    edge = (k % 8 == 1); tgt[edge] = (v0 + a[:, None] * (v1 - v0))[edge]            # on edge v0-v1 (v = 0)
    edge2 = (k % 8 == 2); tgt[edge2] = (v0 + np.clip(a, 0, 1)[:, None] * (v2 - v0))[edge2]  # on edge v0-v2 (u = 0)
    vert = (k % 8 == 3); tgt[vert] = np.where((k % 3 == 0)[:, None], v0, np.where((k % 3 == 1)[:, None], v1, v2))[vert]
    hyp = (k % 8 == 4); s = np.clip(a, 0, 1); tgt[hyp] = (v0 + s[:, None] * (v1 - v0) + (1 - s)[:, None] * (v2 - v0))[hyp]
    d = tgt - o
    behind = (k % 8 == 5); d[behind] *= -1
    d /= np.sqrt((d * d).sum(1))[:, None]
    # parallel / near-parallel rays: direction in the triangle plane (+ tiny tilt so det ~ 1e-8)
    par = (k % 8 == 6)
    nrm = np.cross(v1 - v0, v2 - v0)
    inpl = (v1 - v0) * rng.uniform(-1, 1, (n, 1)) + (v2 - v0) * rng.uniform(-1, 1, (n, 1))
    inpl /= np.sqrt((inpl * inpl).sum(1))[:, None]
    tilt = 10.0 ** rng.uniform(-10, -7, n) * rng.choice([-1, 1], n) / np.sqrt((nrm * nrm).sum(1))
    dd = inpl + tilt[:, None] * nrm / np.sqrt((nrm * nrm).sum(1))[:, None]
    dd /= np.sqrt((dd * dd).sum(1))[:, None]
    d[par] = dd[par]
    exact_par = par & (k % 16 == 6); d[exact_par] = inpl[exact_par]
    t, u, v = ref.mt_batch(o, d, v0, v1, v2)
    print("G1: %d pairs, %d hits(t>0), %d zero" % (n, (t > 0).sum(), (t == 0).sum()))
    np.savez_compressed(os.path.join(OUT, "g1_mt.npz"), o=o, d=d, v0=v0, v1=v1, v2=v2, t=t, u=u, v=v)


def strided_pixels(xr, yr, target):
    total = xr * yr
    step = max(1, total // target)
    # a stride co-prime with the row length so columns are covered too
    while np.gcd(step, xr) != 1:
        step += 1
    idx = np.arange(0, total, step)
    return np.stack([idx // xr, idx % xr], 1).astype(np.uint32)


def g2_scene(ref, ora, stem, n_sample, rng, with_g3=False):
    """G2 (+G3): flattened original triangles + camera; strided primary rays through Scene::Cast and
    through brute force over the originals with the reference's triangle_intersection; CRC of the
    full-frame hit-id image."""
    wd = workdir()
    h = ref.scene_load(wd, stem)
    tris, obj, nrm = ref.triangles(h)
    cam, res = ref.camera(h)
    xr, yr = int(res[0]), int(res[1])
    rd = ref.raster_details(h)
    rc = ref.build_bsp(h, wd)
    out = dict(tris=tris, tri_obj=obj, tri_normal=nrm, cam=cam, res=res, raster=rd, bsp_rc=np.int32(rc))
    lc = ref.light_center(h)
    if lc is not None:
        out["light_center"] = lc
    pix = strided_pixels(xr, yr, n_sample)
    dirs = ref.primary_dirs(rd, pix)
    o = np.tile(cam[:3], (len(pix), 1))
    br = ref.brute(h, o, dirs)
    out.update(pix=pix, dirs_head=dirs[:512], tri=br["tri"], t=br["t"], u=br["u"], v=br["v"])
    if rc == 0:
        ca = ref.cast(h, o, dirs)
        hm = ca["hit"].astype(bool)
        print("  %s: Cast vs brute: hit mismatches %d, max rel dt %.3g" % (
            stem, int((hm != (br["tri"] >= 0)).sum()),
            float(np.max(np.abs(ca["dist"][hm] - br["t"][hm]) / br["t"][hm])) if hm.any() else 0.0))
        out.update(cast_hit=ca["hit"], cast_dist=ca["dist"], cast_obj=ca["obj"], cast_point_hits=ca["point"][hm],
                   cast_normal_hits=ca["normal"][hm], cast_bu_hits=ca["bu"][hm], cast_bv_hits=ca["bv"][hm])
    # full-frame hit-id image CRC (brute force with the reference's MT; the contract of SURVEY §0.5)
    ii, jj = np.meshgrid(np.arange(yr, dtype=np.uint32), np.arange(xr, dtype=np.uint32), indexing="ij")
    allpix = np.stack([ii.ravel(), jj.ravel()], 1)
    alld = ref.primary_dirs(rd, allpix)
    allo = np.tile(cam[:3], (len(allpix), 1))
    full = ref.brute(h, allo, alld)
    out["full_tri_crc32"] = np.uint32(zlib.crc32(full["tri"].astype("<i4").tobytes()))
    out["full_t_crc32"] = np.uint32(zlib.crc32(full["t"].astype("<f8").tobytes()))
    out["full_hits"] = np.int64((full["tri"] >= 0).sum())
    out["ref_brute_seconds_full"] = np.float64(full["seconds"])
    print("  %s: %d tris, %dx%d, full-frame hits %d, brute %.2fs" % (stem, len(tris), xr, yr, out["full_hits"], full["seconds"]))
    if with_g3 and rc == 0 and lc is not None:
        # G3 shadow rays built from the Cast hits exactly as generateNormalShadows does
        hm = ca["hit"].astype(bool)
        so, sd = ref.shadow_rays(ca["point"][hm], ca["normal"][hm], lc)
        sh = ref.cast(h, so, sd)
        out.update(shadow_src=np.nonzero(hm)[0].astype(np.int32), shadow_P=ca["point"][hm], shadow_N=ca["normal"][hm],
                   shadow_o=so, shadow_d=sd, shadow_occluded=sh["hit"])
        # AO-style rays: fixed seeded directions in the normal's hemisphere, origin P + 1e-4·dir
        # (shader.cpp:218-226); occluded ⇔ Cast ∧ distance < 2.0
        P = ca["point"][hm]; N = ca["normal"][hm]
        rdv = rng.normal(size=P.shape); rdv /= np.sqrt((rdv * rdv).sum(1))[:, None]
        flip = (rdv * N).sum(1) < 0
        rdv[flip] *= -1
        ao_o = P + rdv * 1e-4
        ao = ref.cast(h, ao_o, rdv)
        out.update(ao_o=ao_o, ao_d=rdv, ao_hit=ao["hit"], ao_dist=ao["dist"],
                   ao_occluded=(ao["hit"].astype(bool) & (ao["dist"] < 2.0)).astype(np.uint8))
        print("  %s: shadow rays %d (occluded %d), AO rays occluded %d" % (
            stem, len(so), int(sh["hit"].sum()), int(out["ao_occluded"].sum())))
    np.savez_compressed(os.path.join(OUT, "g2_%s.npz" % stem), **out)


def g4_radiosity(ref, ora, stem, rng, seed=1, S=10, SL=30):
    """G4: element list (= this run's BSP leaves), injected sample points from the oracle's seeded
    generator, the reference's per-sample visibility / light-blocked bits, F assembled by the
    restated formula on those bits' points, and the reference's progressiveRefinement on that F."""
    wd = workdir()
    h = ref.scene_load(wd, stem)
    rc = ref.build_bsp(h, wd)
    assert rc == 0, "reference BSP build threw on %s" % stem
    ne = ref.rad_setup(h)
    verts, nrm, obj = ref.rad_elements(h, ne)
    lc = ref.light_center(h)
    # every ordered pair i != j, S samples: points from rto_sample_point(seed, i, j, k, 0|1)
    ei, ej, xi, xj = [], [], [], []
    for i in range(ne):
        for j in range(ne):
            if i == j:
                continue
            for k in range(S):
                ei.append(i); ej.append(j)
                xi.append(ora.sample_point(verts[i], seed, i, j, k, 0))
                xj.append(ora.sample_point(verts[j], seed, i, j, k, 1))
    ei = np.array(ei, np.int32); ej = np.array(ej, np.int32); xi = np.array(xi); xj = np.array(xj)
    vis = ref.rad_visible(h, ei, ej, xi, xj)
    # light samples
    li, lx = [], []
    for i in range(ne):
        for k in range(SL):
            li.append(i); lx.append(ora.sample_point(verts[i], seed, i, 0xFFFFFFFF, k, 0))
    li = np.array(li, np.int32); lx = np.array(lx)
    lb = ref.rad_light_blocked(h, li, lx)
    # the reference's own random estimates, for a statistical check of the restated formulas
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, ne, 12), rng.integers(0, ne, 12)) if a != b]
    reps = 300
    mc_mean = np.array([np.mean([ref.rad_ff_term(h, a, b) for _ in range(reps)]) for a, b in pairs])
    lt_i = rng.integers(0, ne, 6)
    lt_mean = np.array([np.mean([ref.rad_ff_light_term(h, int(a)) for _ in range(reps)]) for a in lt_i])
    # F by the oracle's restated formula (same points) -> injected into the reference solver
    F = np.zeros((ne + 1, ne + 1))
    F[:ne, :ne] = ora.form_factors(verts, nrm, S, seed)
    F[:ne, ne] = ora.light_terms(verts, nrm, lc, SL, seed, mirror_bug=1)
    t_ff = ref.rad_find_form_factor(h)          # allocates FF/K (and times the reference's own build)
    F_ref_random = ref.rad_get_FF(h)
    ref.rad_set_FF(h, F)
    t_solve = ref.rad_solve(h)
    B = ref.rad_B(h)
    resid = ref.rad_residual(h)
    Bs = ref.rad_scatter_B(h, ne)
    assert np.array_equal(Bs, B[:ne])
    print("G4 %s: %d elements, vis %d/%d, light-blocked %d/%d, ref FF build %.2fs, solve %.4fs, residual %.3g" % (
        stem, ne, int(vis.sum()), len(vis), int(lb.sum()), len(lb), t_ff, t_solve, resid))
    np.savez_compressed(os.path.join(OUT, "g4_%s.npz" % stem), verts=verts, normals=nrm, obj=obj, light_center=lc,
                        seed=np.uint64(seed), S=np.int32(S), SL=np.int32(SL), ei=ei, ej=ej, xi=xi, xj=xj, vis=vis,
                        li=li, lx=lx, light_blocked=lb, mc_pairs=np.array(pairs, np.int32), mc_mean=mc_mean,
                        mc_reps=np.int32(reps), lt_i=lt_i.astype(np.int32), lt_mean=lt_mean, F=F,
                        F_ref_random_rowsum=F_ref_random.sum(1), B_ref=B, ref_residual=np.float64(resid),
                        ref_ff_seconds=np.float64(t_ff), ref_solve_seconds=np.float64(t_solve))


def g5_btxt():
    lines = [l for l in open("/root/reference/Render2/B.txt").read().splitlines() if l.strip()]
    last = float(lines[-1].split(":")[-1])
    json.dump({"source": "Render2/B.txt", "lines": len(lines), "last_value": last},
              open(os.path.join(OUT, "g5_btxt_facts.json"), "w"))
    print("G5: B.txt lines %d last %g" % (len(lines), last))


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261110)
    ref, ora = Reference(), Oracle()
    if sys.argv[1:] == ["g4"]:      # only the radiosity fixtures (they depend on the oracle's sample generator; round 3 changed it)
        rng = np.random.default_rng(20261111)
        g4_radiosity(ref, ora, "radiosity_test", rng)
        g4_radiosity(ref, ora, "practice_room_enclosed", rng)
        return
    g1_mt(ref, rng)
    print("G2/G3:")
    g2_scene(ref, ora, "test1", 12500, rng, with_g3=True)
    for stem in ("test2", "test5", "test6", "test7", "scene_room_bed_sample"):
        g2_scene(ref, ora, stem, 4000, rng)
    g4_radiosity(ref, ora, "radiosity_test", rng)
    g4_radiosity(ref, ora, "practice_room_enclosed", rng)
    g5_btxt()


if __name__ == "__main__":
    main()
