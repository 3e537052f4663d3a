#!/usr/bin/env python3
"""Work counters of the form-factor kernel on the closed box (diagnostic build: make -C raytracer_amd/csrc count)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTMI_LIB"] = os.path.join(ROOT, "raytracer_amd", "librtmi_count.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 74
if len(sys.argv) > 3 and sys.argv[3] == "mixed":     # the furnished room of tools/quick_ff_mixed.py
    vb, nb, _ = workloads.closed_box(k)
    fv = np.concatenate([workloads._box((1.0, 1.0, 0.0), (3.5, 2.5, 1.2), 24), workloads._box((6.0, 0.2, 0.0), (9.5, 1.5, 6.0), 32),
                         workloads._box((0.3, 5.0, 0.0), (2.8, 9.5, 1.0), 28), workloads._box((7.0, 7.0, 2.0), (9.8, 9.8, 2.3), 20),
                         workloads._box((4.2, 4.2, 0.0), (5.6, 5.6, 3.0), 20)])
    verts, normals = np.concatenate([vb, fv]), np.concatenate([nb, workloads.face_normals(fv)])
elif len(sys.argv) > 3:    # a scene file of tests/golden/scenes instead of the box (lenient reader), e.g. real_full_room
    sc = rt.BuildScene(os.path.join(ROOT, "tests", "golden", "scenes"), sys.argv[3], lenient=True)
    verts, normals = sc["tris"], (-sc["normals"] if sys.argv[3] == "real_buddha" else sc["normals"])
else:
    verts, normals, light = workloads.closed_box(k)
box = rt.Scene(verts)
rad = rad = rt.Radiosity(box, normals, None)
out = (C.c_uint64 * 96)()
row0 = int(os.environ["ROW0"]) if "ROW0" in os.environ else (20000 if len(sys.argv) > 3 and sys.argv[3] == 'mixed' else 1000 if len(sys.argv) <= 3 else 0)
rad.findFormFactor(samples=10, seed=1, row0=row0, n_rows=rows)
torch.cuda.synchronize()
rt.lib.rtmi_debug_ff_counts(out)
c = list(out)
names = ["sample iterations (wave)", "um_i loop iterations", "um_j loop iterations", "own_j loop iterations", "lanes with >=1 sibling hit", "trace_from calls (wave)",
         "node steps (wave)", "leaf steps (wave)", "from_root lanes", "shoot lanes", "prepass own-level iterations", "node-step lanes", "leaf-step lanes",
         "sample iterations with a ray", "lanes with >=2", "lanes with >=3"]
it = max(1, c[13])
for n, v in zip(names, c):
    print("%-32s %14d   per ray-iteration %.3f" % (n, v, v / it))
print("per pair-row (wave): beam-walk visits %.2f, path-i levels %.2f, shared path-j levels %.2f, own path-j levels %.2f" % (
    c[80] * 10.0 / c[0], c[81] * 10.0 / c[0], c[82] * 10.0 / c[0], c[10] * 10.0 / c[0]))
print("sample iterations taken by the certain-hit path (no triangle test, no walk): %d of %d with a ray" % (c[79], c[13]))
print("sibling boxes hit by a ray, by level (lane count): path i", [int(x) for x in c[16:40]])
print("                                                   path j", [int(x) for x in c[48:72]])
print("trace_from calls with a grazing lane in the wave %d (grazing lanes entering %d, all lanes entering %d); node steps without / with a grazing lane %d / %d, leaf steps %d / %d" % (
    c[72], c[73], c[78], c[74], c[75], c[76], c[77]))
print("wave-rows %d; with a lane that keeps a sibling for its rays %d (%.3f); such lanes per wave-row with any %.1f; wave-rows with 1-6 / 7-12 / 13-25 / more such lanes: %d / %d / %d / %d" % (
    c[83], c[84], c[84] / max(1, c[83]), c[85] / max(1, c[84]), c[86], c[87], c[88], c[89]))
print("passes over dealt samples %d (%.2f per wave-row with such lanes)" % (c[90], c[90] / max(1, c[84])))
