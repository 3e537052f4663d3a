#!/usr/bin/env python3
"""Form-factor rate on a furnished room: the 64 K-patch box of the bench with finely meshed boxes standing in it (a table,
a wardrobe, a bed, a shelf) — path walk, root traversal, and the library's own choice; 2048 rows from the walls and 2048 from
the furniture."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads as W
torch.cuda.set_device(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 74
vb, nb, _ = W.closed_box(k)
furn = [W._box((1.0, 1.0, 0.0), (3.5, 2.5, 1.2), 24), W._box((6.0, 0.2, 0.0), (9.5, 1.5, 6.0), 32), W._box((0.3, 5.0, 0.0), (2.8, 9.5, 1.0), 28),
        W._box((7.0, 7.0, 2.0), (9.8, 9.8, 2.3), 20), W._box((4.2, 4.2, 0.0), (5.6, 5.6, 3.0), 20)]
fv = np.concatenate(furn)
verts = np.concatenate([vb, fv]); normals = np.concatenate([nb, W.face_normals(fv)])
scene = rt.Scene(verts)
rad = rt.Radiosity(scene, normals, None)
print("room %d wall patches + %d furniture patches" % (len(vb), len(fv)), flush=True)
for name, row0 in (("wall rows", 20000), ("furniture rows", len(vb) + 1000)):
    line = "%-15s" % name
    for mode in ("path", "flat", ""):
        if mode: os.environ["RTMI_FF_KERNEL"] = mode
        else: os.environ.pop("RTMI_FF_KERNEL", None)
        rad.findFormFactor(samples=10, seed=1, row0=row0, n_rows=1024); torch.cuda.synchronize()
        t0 = time.perf_counter()
        F = rad.findFormFactor(samples=10, seed=1, row0=row0, n_rows=2048); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line += "  %s %.1f ms = %.1f G pair-samples/s" % (mode or "library's choice (%d)" % rt.lib.rtmi_debug_form_factor_kernel(rad.handle), dt * 1e3, 10.0 * 2048 * rad.n / dt / 1e9)
    os.environ.pop("RTMI_FF_KERNEL", None)
    print(line, flush=True)
