"""CPU-side checks: the C-ABI library loads and exports exactly what include/rtmi.h declares, the host-only
entry points behave, the product never touches oracle/, and the rank-sharding logic (world_size 2, gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden


def header_symbols():
    h = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(rtmi_[a-z_0-9]+)\s*\(", h)) - {"rtmi_exchange_fn"})


def test_library_exports_every_declared_symbol():
    import raytracer_amd as rt
    syms = header_symbols()
    assert sorted(rt.SYMBOLS) == syms
    L = ctypes.CDLL(rt.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", rt.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r" T (rtmi_[a-z_0-9]+)", out)))
    assert exported == syms


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    import raytracer_amd as rt
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert rt.lib.rtmi_device_count() == 0
    with pytest.raises(rt.RtmiError):
        rt.Scene(np.zeros((1, 9)))


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "rt_oracle" not in txt, f


@pytest.mark.parametrize("stem", ["test1", "test2", "test5", "test6", "test7", "scene_room_bed_sample"])
def test_camera_setup_matches_reference_raster_details(stem):
    """rtmi_camera_setup runs on the host: RasterDetails (rasterizer.cpp:40-56) bit for bit."""
    import raytracer_amd as rt
    g = load_golden("g2_%s.npz" % stem)
    cam = rt.Camera.from_cam14(g["cam"], g["res"])
    O, dR, dU = cam.raster_details()
    assert np.array_equal(np.concatenate([O, dR, dU]), g["raster"][9:18])
    assert (cam.x_res, cam.y_res) == tuple(int(v) for v in g["res"])
    with pytest.raises(rt.RtmiError):
        rt.Camera(g["cam"][0:3], g["cam"][3:6], g["cam"][6:9], 1.0, [-1, 1], [-1, 1], 0, 10)


def test_row_and_tile_sharding_cover_everything_once():
    from raytracer_amd import multi
    for n, w in [(65536, 8), (49, 2), (11, 4), (5, 8), (3889, 3)]:
        seen = np.zeros(n, int)
        for r in range(w):
            r0, nr, blk = multi.rows_of_rank(n, r, w)
            seen[r0:r0 + nr] += 1
            assert nr <= blk and r0 == min(n, r * blk)
        assert (seen == 1).all()
    for (x, y, tw, th, w) in [(3840, 2160, 64, 64, 8), (500, 400, 48, 40, 3), (17, 9, 16, 16, 2)]:
        tot = sum(multi.pixels_of_rank(x, y, tw, th, r, w) for r in range(w))
        assert tot == x * y
        tiles = sorted(t for r in range(w) for t in multi.tiles_of_rank(x, y, tw, th, r, w))
        assert tiles == list(range(len(tiles)))


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import multi
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "g4_practice_room_enclosed.npz"))
F = g["F"].astype(np.float32); n = F.shape[0]
rho = np.array([0.7, 0.7, 0.7])
E = np.zeros((n, 3), np.float32); E[n - 1] = 68500.0
row0, n_rows, blk = multi.rows_of_rank(n, rank, world)
Fl = F[row0:row0 + n_rows].astype(np.float64)          # this rank's rows only
def sweep(B_old, B_new, norms):                       # CPU stand-in for the HIP sweep (tests only)
    bo = B_old.numpy()[:n].astype(np.float64)
    bn = (E[row0:row0 + n_rows].astype(np.float64) + rho[None, :] * (Fl @ bo)).astype(np.float32)
    B_new[row0:row0 + n_rows] = torch.from_numpy(bn)
    d = bn.astype(np.float64) - bo[row0:row0 + n_rows]
    norms[0] += float((d ** 2).sum()); norms[1] += float(((bn - E[row0:row0 + n_rows]) ** 2).sum())
    norms[2] = max(float(norms[2]), float(np.abs(d).max()) if len(d) else 0.0)
r = multi.solve_sharded(sweep, torch.from_numpy(E), n, tol_rel=1e-7, max_sweeps=500, check_every=3)
np.save(sys.argv[5] + ".%d.npy" % rank, r.B.numpy())
print(rank, r.sweeps, r.rel_change)
dist.destroy_process_group()
'''


def test_sharded_solve_world2_gloo(tmp_path, oracle):
    """Two ranks, gloo: F row-sharded, all-gather of B per sweep — every rank ends with the reference's B."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 2000)
    out = str(tmp_path / "B")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, out], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    g = load_golden("g4_practice_room_enclosed.npz")
    B0, B1 = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert np.array_equal(B0, B1)
    ne = len(g["verts"])
    Bd = oracle.direct_solve(g["F"])
    assert np.linalg.norm(B0[:ne, 0] - Bd[:ne]) / np.linalg.norm(Bd[:ne]) < 1e-5
    assert np.linalg.norm(B0[:ne, 0] - g["B_ref"][:ne]) / np.linalg.norm(g["B_ref"][:ne]) < 1e-4
    assert B0[ne, 0] == 68500.0


FRAME_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import multi
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
X, Y = 150, 70
full = (np.arange(Y * X * 3).reshape(Y, X, 3) * 7 % 251).astype(np.uint8)
def render(out, tile, first, stride):            # CPU stand-in for rtmi_render_frame_dev: writes this rank's tiles only
    tx = (X + tile[0] - 1) // tile[0]
    for t in multi.tiles_of_rank(X, Y, tile[0], tile[1], first, stride):
        x0, y0 = (t % tx) * tile[0], (t // tx) * tile[1]
        out[y0:y0 + tile[1], x0:x0 + tile[0]] = torch.from_numpy(full[y0:y0 + tile[1], x0:x0 + tile[0]])
img = multi.render_frame_sharded(render, X, Y, tile=(32, 24), device="cpu")
assert np.array_equal(img.numpy(), full), "assembled frame differs"
rgb = multi.render_frame_sharded(lambda o, t, f, s: render(o, t, f, s), X, Y, tile=(32, 24), dtype=torch.float64, device="cpu")
assert np.array_equal(rgb.numpy(), full.astype(np.float64))
print(rank, "ok")
dist.destroy_process_group()
'''


def test_sharded_frame_world2_gloo(tmp_path):
    """Two ranks, gloo: tiles dealt round-robin, one all-reduce assembles the frame on every rank."""
    script = tmp_path / "frame_worker.py"
    script.write_text(FRAME_WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs


SCENE_STEMS = ["test1", "test2", "test5", "test6", "test7", "scene_room_bed_sample", "radiosity_test", "practice_room_enclosed"]


@pytest.mark.parametrize("stem", SCENE_STEMS)
def test_scene_file_reader_matches_the_references_buildscene(stem):
    """rtmi_scene_file_load on the reference's own scene files (tests/golden/scenes, data copied by
    oracle/gen_golden_scenes.py) against what the compiled reference's Scene::BuildScene made of them: every double equal"""
    import raytracer_amd as rt
    g = np.load(os.path.join(GOLDEN, "g7_scene_files.npz"))
    s = rt.BuildScene(os.path.join(GOLDEN, "scenes"), stem)
    for key in ("tris", "tri_obj", "normals", "mats", "lights", "cam14", "res"):
        assert np.array_equal(s[key], g[stem + "/" + key]), key


def test_scene_file_reader_errors(tmp_path):
    import raytracer_amd as rt
    with pytest.raises(rt.RtmiError, match="could not open"):
        rt.BuildScene(str(tmp_path), "nothing_here")
    (tmp_path / "a.sdf").write_text("quadlight (0,0,0) (1,0,0) (1,1,0) (0,1,0)\n")
    with pytest.raises(rt.RtmiError, match="quadlight"):
        rt.BuildScene(str(tmp_path), "a")
    # an .obj whose first line does not name the object is ignored, as in the reference (scene.cpp:400-405)
    (tmp_path / "b.sdf").write_text("eye (0, 0, 5)\nobject thing\nend\n")
    (tmp_path / "thing.obj").write_text("o other\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    s = rt.BuildScene(str(tmp_path), "b")
    assert len(s["tris"]) == 0 and len(s["mats"]) == 1 and s["cam14"][2] == 5.0 and tuple(s["res"]) == (400, 400)
    (tmp_path / "thing.obj").write_text("o thing\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 4 3\nf 1 2 9\n")
    with pytest.raises(rt.RtmiError, match="does not exist"):
        rt.BuildScene(str(tmp_path), "b")


def test_lenient_reader_takes_what_the_references_reader_cannot(tmp_path):
    """the reference's reader stops at 512-byte lines (scene.cpp:211), wants `o <file name>` first
    (:400-405) and splits triangles and quads only (params.cpp:214).  tools/gen_obj.py writes a mesh that breaks all three
    (plus relative indices, a//c corners, groups); RTMI_SCENE_LENIENT reads it to exactly the triangles it stands for, the
    default mode treats it as the reference does: the object stays empty."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_obj
    import raytracer_amd as rt
    want = gen_obj.generate(str(tmp_path), "terrain", 12)
    strict = rt.BuildScene(str(tmp_path), "terrain")
    assert len(strict["tris"]) == 0 and len(strict["mats"]) == 1                      # name mismatch: the file is dropped
    got = rt.BuildScene(str(tmp_path), "terrain", lenient=True)
    assert np.array_equal(got["tris"], want) and len(want) > 12 * 12 * 2
    assert np.all(got["tri_obj"] == 0) and got["lights"].shape == (1, 6) and tuple(got["res"]) == (960, 540)
    n = got["normals"]
    assert np.allclose((n * n).sum(1), 1.0) and (n[:, 2] > 0).mean() > 0.95          # unit normals, the terrain faces up
    # a strict-conforming file reads the same in both modes
    (tmp_path / "plain.sdf").write_text("object plain\n")
    (tmp_path / "plain.obj").write_text("o plain\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\nf 1 2 3\n")
    a, b = rt.BuildScene(str(tmp_path), "plain"), rt.BuildScene(str(tmp_path), "plain", lenient=True)
    assert len(a["tris"]) == 3 and np.array_equal(a["tris"], b["tris"]) and np.array_equal(a["normals"], b["normals"])
    # five corners: the reference keeps the first triangle only, lenient mode fans on
    (tmp_path / "plain.obj").write_text("o plain\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0.5 1.5 0\nv 0 1 0\nf 1 2 3 4 5\n")
    assert len(rt.BuildScene(str(tmp_path), "plain")["tris"]) == 1 and len(rt.BuildScene(str(tmp_path), "plain", lenient=True)["tris"]) == 3
    with pytest.raises(rt.RtmiError, match="could not open"):
        (tmp_path / "gone.sdf").write_text("object not_there\n")
        rt.BuildScene(str(tmp_path), "gone", lenient=True)


def test_squared_visibility_threshold_is_exact():
    """rtmi_rad_device.hpp compares |p − xj|² with kEpsilonSquaredExact instead of sqrt(|p − xj|²) with 1e-5 (radiosity.cpp:249):
    the constant must be the least double whose correctly rounded root is >= 1e-5, then the two tests agree on every double."""
    import math
    src = open(os.path.join(ROOT, "raytracer_amd", "csrc", "rtmi_rad_device.hpp")).read()
    T = float.fromhex(re.search(r"kEpsilonSquaredExact = (0x[0-9a-fp.+-]+);", src).group(1))
    assert math.sqrt(T) >= 1e-5 and math.sqrt(math.nextafter(T, 0.0)) < 1e-5
    x = T                                              # and monotone around it, a few thousand neighbours each way
    for _ in range(4000):
        x = math.nextafter(x, math.inf)
        assert math.sqrt(x) >= 1e-5
    x = T
    for _ in range(4000):
        x = math.nextafter(x, 0.0)
        assert math.sqrt(x) < 1e-5


def test_vertex_values_equal_the_oracles_scan(oracle):
    """rtmi_radiosity_vertex_values (host code: corners sorted by position) against the oracle's O(n²) scan — equal to the
    bit — on the three fixture scenes the compiled reference pinned (g10) and on a closed box of 2 028 patches, where every
    vertex is shared by 4 to 8 triangles of up to three walls; plus the cases without neighbours and the argument errors."""
    import raytracer_amd as rt
    from raytracer_amd import workloads
    g = load_golden("g10_vertex_values.npz")
    for stem in ("radiosity_test", "practice_room_enclosed", "sheet"):
        v, B = g[stem + "/verts"], g[stem + "/B"]
        mine = rt.api.vertex_values(v, B)
        assert np.array_equal(mine, oracle.vertex_values(v, B)), stem
        assert np.allclose(mine, g[stem + "/vertex_values"], rtol=1e-15, atol=0)
    tris = workloads.closed_box(13)[0].reshape(-1, 9)
    B = np.random.default_rng(3).uniform(0.1, 5.0, len(tris))
    mine = rt.api.vertex_values(tris, B)
    assert np.array_equal(mine, oracle.vertex_values(tris, B))
    assert mine.min() >= B.min() - 1e-12 and mine.max() <= 2 * B.max()
    lonely = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [5, 5, 5, 6, 5, 5, 5, 6, 5.0]])
    assert np.array_equal(rt.api.vertex_values(lonely, [2.0, 3.0]), [[2, 2, 2], [3, 3, 3]])       # nobody shares a vertex: B_e
    assert rt.lib.rtmi_radiosity_vertex_values(None, 0, None, None) == 0
    assert rt.lib.rtmi_radiosity_vertex_values(None, 3, None, None) == -1 and b"NULL" in rt.lib.rtmi_last_error()
    bad = lonely.copy(); bad[1, 4] = np.nan
    with pytest.raises(rt.RtmiError, match="non-finite"):
        rt.api.vertex_values(bad, [2.0, 3.0])


def test_header_is_plain_c(tmp_path):
    """include/rtmi.h is the boundary a C, C++, cgo or ctypes host binds: it must compile as C99 and as C++17 with nothing
    else included, and a C program using its structs must link against librtmi.so."""
    src = tmp_path / "use.c"
    src.write_text('#include "rtmi.h"\n'
                   'int main(void) {\n'
                   '  rtmi_primary_sample s = {0.5, 0.5, 0, 0, 0};\n'
                   '  rtmi_trace_params p = {1, 10, 2.0, 1, RTMI_SHADE_RADIOSITY_BRANCH, 0};\n'
                   '  unsigned char id[RTMI_COMM_ID_BYTES];\n'
                   '  (void)s; (void)p; (void)id;\n'
                   '  return rtmi_scene_check(0) == -1 && rtmi_last_error()[0] ? 0 : 1;\n'
                   '}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "use.o")], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "use_cpp.o")], check=True)
    lib = os.path.join(ROOT, "raytracer_amd")
    subprocess.run(["gcc", str(tmp_path / "use.o"), "-L", lib, "-lrtmi", "-Wl,-rpath," + lib, "-o", str(tmp_path / "use")], check=True)
    assert subprocess.run([str(tmp_path / "use")]).returncode == 0       # no GPU needed: a NULL scene is refused with a message


def test_the_references_large_assets_load():
    """SURVEY.md §8f rank 3 names two of the reference's own meshes as the point of the row: scenes/full_room.obj (22 `o`
    groups, 14 676 faces — the reference's reader drops it, every `o` name differs from the object's) and
    scenes/Golden_Buddha_low.obj (v/vt/vn corners, `g`, `mtllib`, `usemtl`, `s` lines).  Both are data fixtures under
    tests/golden/scenes with a view of ours (real_*.sdf).  Triangles are checked against an independent parse of the files."""
    import raytracer_amd as rt
    scenes = os.path.join(GOLDEN, "scenes")

    def parse(path):          # v lines, f lines; a quad is (v1, v2, v3) + (v3, v4, v1), params.cpp:214-241
        v, tris = [], []
        for line in open(path):
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                v.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(c.split("/")[0]) - 1 for c in t[1:]]
                tris.append(v[idx[0]] + v[idx[1]] + v[idx[2]])
                for k in range(2, len(idx) - 1):
                    tris.append(v[idx[k]] + v[idx[k + 1]] + v[idx[0]])
        return np.array(tris)

    room = rt.BuildScene(scenes, "real_full_room", lenient=True)
    want = parse(os.path.join(scenes, "full_room.obj"))
    assert len(room["tris"]) == 14676 and np.array_equal(room["tris"], want)
    assert len(rt.BuildScene(scenes, "real_full_room")["tris"]) == 0                       # as the reference: nothing matches
    n2 = (room["normals"] ** 2).sum(1)                                                    # unit normals; the zero vector for the
    assert np.all((np.abs(n2 - 1.0) < 1e-9) | (n2 == 0)) and int((n2 == 0).sum()) == 78    # asset's 78 faces without a direction
    for lenient in (False, True):                                                         # the Buddha's `o` line matches: both modes
        b = rt.BuildScene(scenes, "real_buddha", lenient=lenient)
        assert len(b["tris"]) == 3367 and np.array_equal(b["tris"], parse(os.path.join(scenes, "Golden_Buddha_low.obj")))
    assert rt.check_hierarchy(room["tris"])[0] == 0                                       # its hierarchy keeps every invariant


@pytest.mark.parametrize("case", ["n1", "n2", "n3", "n5", "n64", "soup20k", "offset", "flat", "mixed_scales", "grid", "duplicates"])
def test_hierarchy_invariants_on_the_host(case):
    """rtmi_debug_check_hierarchy (host only): the binary + 4-wide hierarchies rtmi_scene_create builds keep every
    invariant the kernels rely on — child boxes decoded as the kernels decode them contain the children's fp32 bounds,
    every triangle is reachable exactly once, the LDS stack bound covers every path."""
    import raytracer_amd as rt
    from raytracer_amd import workloads
    rng = np.random.default_rng(3)
    if case.startswith("n"):
        tris = workloads.soup(int(case[1:]), seed=2, edge=0.3)
    elif case == "soup20k":
        tris = workloads.soup(20000, seed=4, edge=0.02)
    elif case == "offset":
        tris = workloads.soup(3000, seed=5, edge=0.1) + np.tile([1.0e5, -3.0e4, 7.0e3], 3)
    elif case == "flat":
        tris = workloads.soup(2000, seed=6, edge=0.2)
        tris[:, [1, 4, 7]] = -2.5
    elif case == "mixed_scales":
        tris = np.concatenate([workloads.soup(2000, seed=7, edge=1e-6), workloads.soup(50, seed=8, edge=200.0)])
    elif case == "grid":
        tris = workloads.closed_box(12)[0]
    else:
        t = workloads.soup(300, seed=9, edge=0.2)
        tris = np.concatenate([t, t, t])
    bad, wide_nodes, need, depth = rt.check_hierarchy(tris)
    n = len(tris)
    assert bad == 0
    assert (wide_nodes == 0) == (n == 1)
    if n > 1:
        assert (n - 1 + 2) // 3 <= wide_nodes <= n - 1           # between all-4-wide and all-binary
        assert need <= 3 * depth and need + 1 <= 64              # fits the dynamic LDS of a 256-lane workgroup


def test_hierarchy_checker_catches_corruptions(tmp_path):
    """tests/cpp/wide_mutations.cpp: the checker accepts the builder's output and flags a lowered hi byte, a raised lo
    byte, shifted child bases, a stack bound one too small and a lowered grid exponent."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wide_mutations")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(root, "raytracer_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "wide_mutations.cpp"), os.path.join(root, "raytracer_amd", "csrc", "bvh_build.cpp"),
                    "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_scripts_compile():
    """bench.py, __graft_entry__.py and every tool parse (they only run on the GPU box)."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, doraise=True)
