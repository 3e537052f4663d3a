"""Synthetic inputs for the configurations BASELINE.json names.  Host-side numpy only:
these produce the arrays a host driver would hand to the C ABI; nothing here computes a result.

    soup(n)            C3: n random small triangles in [-1,1]^3, camera looking down +z at 3840x2160
    closed_box(k)      C2/C5: 6 walls of k x k x 2 triangles each, normals inward, side 10, + light centre
    synthetic_F(...)   C5 (solver only): device-resident fp32 matrix, zero diagonal, rows scaled to sum 0.9
"""
import numpy as np

# C3 camera: eye, lookat, up, vpdist, x_win, y_win (the 14 numbers of Render2/ray_tracer.h:113-122)
SOUP_CAM14 = np.array([0, 0, -3, 0, 0, 0, 0, 1, 0, 2.75, -1.6, 1.6, -0.9, 0.9], np.float64)
SOUP_RES = (3840, 2160)


def soup(n_tris=1 << 20, seed=1, edge=0.01):
    """n_tris x 9 doubles: centres uniform in [-1,1]^3, the two edge vectors uniform in [-edge,edge]^3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.uniform(-1.0, 1.0, (n_tris, 3))
    e1 = rng.uniform(-edge, edge, (n_tris, 3))
    e2 = rng.uniform(-edge, edge, (n_tris, 3))
    return np.concatenate([c, c + e1, c + e2], axis=1)


def closed_box(k, side=10.0, hole=0.0):
    """Cornell-type box: returns (verts n x 9, normals n x 3 pointing inward, light centre).
    n = 12 k^2 when closed.  Wall order: z=0, z=side, y=0, y=side, x=0, x=side; rows of quads, 2 triangles per quad.
    hole > 0 removes the ceiling triangles whose centroid lies in the central square of width hole·side and
    puts the light above the opening (the reference's light test asks for NO hit at any distance,
    radiosity.cpp:290, so a light inside a closed box lights nothing)."""
    h = side / k
    g = np.arange(k) * h
    a, b = np.meshgrid(g, g, indexing="ij")
    a = a.ravel(); b = b.ravel()
    verts, normals = [], []
    for axis in range(3):
        for far in (0, 1):
            w = side if far else 0.0
            ua, ub = [(1, 2), (0, 2), (0, 1)][axis]

            def P(da, db):
                p = np.zeros((len(a), 3))
                p[:, axis] = w
                p[:, ua] = a + da
                p[:, ub] = b + db
                return p
            p00, p10, p11, p01 = P(0, 0), P(h, 0), P(h, h), P(0, h)
            t1 = np.concatenate([p00, p10, p11], 1)
            t2 = np.concatenate([p00, p11, p01], 1)
            t = np.stack([t1, t2], 1).reshape(-1, 9)
            nrm = np.zeros(3)
            nrm[axis] = -1.0 if far else 1.0
            verts.append(t)
            normals.append(np.tile(nrm, (len(t), 1)))
    verts = np.concatenate(verts)
    normals = np.concatenate(normals)
    light = np.array([0.5 * side, 0.5 * side, 0.9 * side])
    if hole > 0.0:
        c = (verts[:, 0:3] + verts[:, 3:6] + verts[:, 6:9]) / 3.0
        cut = (c[:, 2] == side) & (np.abs(c[:, 0] - 0.5 * side) < 0.5 * hole * side) & (np.abs(c[:, 1] - 0.5 * side) < 0.5 * hole * side)
        verts, normals = verts[~cut], normals[~cut]
        light = np.array([0.5 * side, 0.5 * side, 1.5 * side])
    return verts, normals, light


def _quad(p0, p1, p3, k):
    """k x k quads spanned by p0->p1 and p0->p3, two triangles each; normal = (p1-p0) x (p3-p0) direction."""
    p0, p1, p3 = (np.asarray(x, np.float64) for x in (p0, p1, p3))
    e1, e2 = (p1 - p0) / k, (p3 - p0) / k
    out = []
    for a in range(k):
        for b in range(k):
            q = p0 + a * e1 + b * e2
            out.append(np.concatenate([q, q + e1, q + e1 + e2]))
            out.append(np.concatenate([q, q + e1 + e2, q + e2]))
    return np.array(out)


def _box(lo, hi, k=1):
    """axis-aligned box, outward faces"""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    f = [((x0, y0, z0), (x0, y1, z0), (x1, y0, z0)),  # bottom (-z)
         ((x0, y0, z1), (x1, y0, z1), (x0, y1, z1)),  # top (+z)
         ((x0, y0, z0), (x1, y0, z0), (x0, y0, z1)),  # -y
         ((x0, y1, z0), (x0, y1, z1), (x1, y1, z0)),  # +y
         ((x0, y0, z0), (x0, y0, z1), (x0, y1, z0)),  # -x
         ((x1, y0, z0), (x1, y1, z0), (x1, y0, z1))]  # +x
    return np.concatenate([_quad(a, b, c, k) for a, b, c in f])


def face_normals(verts):
    """Unit((v2 - v1) x (v3 - v2)), the reference's normal for faces without vertex normals (params.cpp:199-208)"""
    v = np.asarray(verts, np.float64).reshape(-1, 3, 3)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 1])
    return n / np.sqrt((n * n).sum(1))[:, None]


# material rows: diffuse, specular, ambient, reflectivity, translucency (3 each), Phong_exp, ref_index, roughness
def material(diffuse=(0, 0, 0), specular=(0, 0, 0), ambient=(0, 0, 0), reflectivity=(0, 0, 0), translucency=(0, 0, 0),
             phong_exp=0.0, ref_index=0.0, roughness=0.0):
    return np.array(list(diffuse) + list(specular) + list(ambient) + list(reflectivity) + list(translucency) +
                    [phong_exp, ref_index, roughness], np.float64)


STUDIO_CAM14 = np.array([10.0, -9.0, 8.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 2.75, -1.25, 1.25, -1.0, 1.0])


def studio(k=2, ambient=0.0):
    """An open scene that exercises every branch of Shader::Shade (shader.cpp:395-645): a diffuse floor, a mirror-like
    back wall, a Phong box, a microfacet box and a refracting sheet, lit by two lights from above (the reference calls a
    point shadowed when ANYTHING lies along the ray to light 0, shader.cpp:131, so the scene has no ceiling).
    Returns (verts n x 9, object id per triangle, materials n_obj x 18, lights n_l x 6)."""
    parts = [
        (_quad((-8, -8, 0), (8, -8, 0), (-8, 8, 0), 2 * k), material(diffuse=(0.84, 0.37, 0.38), ambient=(ambient,) * 3,
                                                                    reflectivity=(0.1, 0.1, 0.1), phong_exp=2.0)),
        (_quad((-8, 8, 0), (8, 8, 0), (-8, 8, 6), k), material(diffuse=(0.1, 0.1, 0.12), specular=(0.3, 0.3, 0.3),
                                                              ambient=(ambient,) * 3, reflectivity=(0.6, 0.6, 0.55),
                                                              phong_exp=20.0)),
        (_box((-4, -1, 0), (-1.5, 1.5, 2.5), k), material(diffuse=(0.2, 0.5, 0.8), specular=(0.5, 0.5, 0.5),
                                                          ambient=(ambient,) * 3, phong_exp=8.0)),
        (_box((1.0, 1.0, 0), (3.5, 3.5, 1.5), k), material(diffuse=(0.6, 0.6, 0.2), specular=(0.4, 0.35, 0.3),
                                                          ambient=(ambient,) * 3, reflectivity=(0.05, 0.05, 0.05),
                                                          roughness=0.4)),
        (_quad((1.0, -4.0, 0.2), (4.5, -2.5, 0.2), (1.0, -4.0, 3.2), k), material(translucency=(0.2, 0.3, 0.4),
                                                                                 ref_index=1.5)),
        (_box((-1.0, -5.5, 0), (0.5, -4.0, 1.5), 1), material(diffuse=(0.3, 0.3, 0.3), translucency=(0.15, 0.1, 0.05),
                                                            ref_index=1.33)),
    ]
    verts = np.concatenate([p for p, _ in parts])
    obj = np.concatenate([np.full(len(p), i, np.uint32) for i, (p, _) in enumerate(parts)])
    mats = np.stack([m for _, m in parts])
    lights = np.array([[3.0, -2.0, 12.0, 1.0, 1.0, 1.0], [-6.0, -8.0, 9.0, 0.3, 0.2, 0.1]])
    return verts, obj, mats, lights


def synthetic_F(n, n_rows=None, row0=0, seed=1, row_sum=0.9, device="cuda", ld=None):
    """C5 solver-only matrix, generated on the device in row blocks: entries uniform(0,1), zero diagonal,
    every row rescaled to sum `row_sum`.  Rows [row0, row0+n_rows) of the n x n matrix (fp32, ld >= n)."""
    import torch
    n_rows = n - row0 if n_rows is None else n_rows
    ld = (n + 3) // 4 * 4 if ld is None else ld
    F = torch.empty((n_rows, ld), dtype=torch.float32, device=device)
    blk = 2048
    for r in range(0, n_rows, blk):
        m = min(blk, n_rows - r)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000003 + row0 + r)   # depends on the GLOBAL row only: shards agree with the whole
        x = torch.rand((m, n), generator=g, device=device, dtype=torch.float32)
        idx = torch.arange(m, device=device)
        x[idx, row0 + r + idx] = 0.0
        x *= (row_sum / x.sum(dim=1, keepdim=True, dtype=torch.float64)).float()
        F[r:r + m, :n] = x
        if ld > n:
            F[r:r + m, n:] = 0.0
    return F


def hit_points(o, d, t):
    """P = o + t·d for hit records (host side of the wavefront: where secondary rays start)."""
    return o + t[:, None] * d


def ao_rays(P, N, spp, radius=2.0, offset=1e-4, seed=1):
    """C4: spp cosine-weighted hemisphere rays per hit point (the reference's AO, shader.cpp:140-232,
    draws its directions from an unseeded RNG, so only the construction is mirrored: origin P + offset·dir,
    occluded ⇔ a hit closer than `radius`).  Returns (o, d, tmax) with spp consecutive rays per point."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(P)
    u1 = rng.random((n, spp)); u2 = rng.random((n, spp))
    r = np.sqrt(u1); phi = 2.0 * np.pi * u2
    x = r * np.cos(phi); y = r * np.sin(phi); z = np.sqrt(np.maximum(0.0, 1.0 - u1))
    # orthonormal frame around N
    a = np.where(np.abs(N[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    T = np.cross(N, a); T /= np.linalg.norm(T, axis=1)[:, None]
    B = np.cross(N, T)
    d = x[..., None] * T[:, None, :] + y[..., None] * B[:, None, :] + z[..., None] * N[:, None, :]
    d /= np.linalg.norm(d, axis=2)[..., None]
    o = P[:, None, :] + offset * d
    return o.reshape(-1, 3), d.reshape(-1, 3), np.full(n * spp, radius)


def soft_shadow_rays(P, light_center, spp, half=1.0, offset=0.05, seed=2):
    """C4: spp rays per hit point towards jittered points of a 2·half square through the light centre (the
    reference's 25×4 grid on a 2×2 square, shader.cpp:55-91: origin P + 0.05·dir); tmax = distance to the sample."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(P)
    j = rng.uniform(-half, half, (n, spp, 2))
    L = np.empty((n, spp, 3))
    L[..., 0] = light_center[0] + j[..., 0]
    L[..., 1] = light_center[1] + j[..., 1]
    L[..., 2] = light_center[2]
    v = L - P[:, None, :]
    dist = np.linalg.norm(v, axis=2)
    d = v / dist[..., None]
    o = P[:, None, :] + offset * d
    return o.reshape(-1, 3), d.reshape(-1, 3), (dist - offset).reshape(-1)
