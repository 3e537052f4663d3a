"""TEST INFRASTRUCTURE.  Writes tests/golden/g10_vertex_values.npz: the radiosity the compiled reference's own
returnQuadTreeColor (radiosity_helper.h:156-268) attaches to the vertices of every element when QuadTreeNode::nextAdj holds
"the elements that share this vertex" — the reference's own build of nextAdj is commented out (radiosity.cpp:167-199), so this
script fills it (ref_rad_set_adjacency) with the adjacency rtmi_radiosity_vertex_values uses: the same three doubles.
Scenes: the element lists of radiosity_test and practice_room_enclosed (BSP leaves) and a generated 6 x 6-quad sheet with a fold.
Run in this container only:   python oracle/gen_golden_vertex.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from oracle_py import Oracle, Reference  # noqa: E402

ref, ora = Reference(), Oracle()
work = tempfile.mkdtemp()
shutil.copytree("/root/reference/Render2/scenes", os.path.join(work, "scenes"),
                ignore=lambda d, names: [n for n in names if os.path.isdir(os.path.join(d, n))])


def adjacency(verts):
    """(ei, ej, point) for every ordered pair of different elements that have a vertex with the same three doubles"""
    owners = {}
    for e, t in enumerate(verts.reshape(-1, 3, 3)):
        for p in t:
            owners.setdefault(tuple((p + 0.0).tolist()), set()).add(e)
    ei, ej, pts = [], [], []
    for p, es in owners.items():
        for a in sorted(es):
            for b in sorted(es):
                if a != b:
                    ei.append(a); ej.append(b); pts.append(p)
    return np.array(ei, np.int32), np.array(ej, np.int32), np.array(pts, np.float64).reshape(-1, 3)


# a sheet of quads with a fold, handed to the reference as triangles (ref_scene_from_triangles)
def sheet(k=6):
    P = np.array([[(0.37 * j - 1.0, 0.41 * i - 1.2, 0.3 * abs(j - 2.5) + 0.05 * i) for j in range(k + 1)] for i in range(k + 1)])
    t = []
    for i in range(k):
        for j in range(k):
            t.append([P[i, j], P[i, j + 1], P[i + 1, j + 1]])
            t.append([P[i, j], P[i + 1, j + 1], P[i + 1, j]])
    return np.array(t).reshape(-1, 9)


out = {}
for stem in ("radiosity_test", "practice_room_enclosed", "sheet"):
    for attempt in range(20):
        h = ref.scene_from_triangles(sheet(), light=[0.0, 0.0, 5.0]) if stem == "sheet" else ref.scene_load(work, stem)
        if ref.build_bsp(h, work) == 0:
            break
    else:
        raise SystemExit("the reference's BSP build kept throwing on " + stem)
    ne = ref.rad_setup(h)
    verts, nrm, obj = ref.rad_elements(h, ne)
    B = 0.05 + 4.0 * ((np.arange(ne) * 7919) % ne) / ne
    ref.rad_set_element_radiosity(h, B)
    flat = ref.rad_vertex_radiosity(h, ne)
    bad = ~np.isfinite(flat).all(1)   # zero-area fragments of the BSP build: the reference's area weights are 0/0 there
    print("   %d elements without an area (NaN weights in the reference), left out" % bad.sum())
    # as shipped: nextAdj empty, every vertex gets B_e (times the area weight of that vertex, which is 1 up to one rounding:
    # Area(v2, v1, v3) / Area(v1, v2, v3), radiosity_helper.h:235-238)
    assert np.allclose(flat[~bad], np.repeat(B[:, None], 3, 1)[~bad], rtol=1e-15, atol=0)
    ei, ej, pts = adjacency(verts)
    ref.rad_set_adjacency(h, ei, ej, pts)
    vr = ref.rad_vertex_radiosity(h, ne)
    mine = ora.vertex_values(verts, B)
    err = np.abs(mine - vr)[~bad].max() / np.abs(vr[~bad]).max()
    print("%-24s %3d elements, %4d adjacency records, %d of %d vertex values differ from flat; oracle vs reference max rel diff %.2e" % (
        stem, ne, len(ei), int((vr != flat).sum()), vr.size, err))
    assert err < 1e-12
    for k, v in dict(verts=verts, B=B, vertex_values=vr, n_adjacency=np.array(len(ei)), no_area=bad).items():
        out[stem + "/" + k] = v
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_vertex_values.npz"), **out)
shutil.rmtree(work, ignore_errors=True)
