#!/usr/bin/env python3
"""Form-factor build rate on the reference's real meshes (tests/golden/scenes): the path-walk kernel (RTMI_FF_KERNEL=path), the root-traversal
kernel (=flat) and what the library picks by itself, 2048 rows, 10 samples per pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import raytracer_amd as rt
torch.cuda.set_device(0)
for stem in ("real_full_room", "real_buddha"):
    s = rt.BuildScene(os.path.join(ROOT, "tests", "golden", "scenes"), stem, lenient=True)
    normals = -s["normals"] if stem == "real_buddha" else s["normals"]
    scene = rt.Scene(s["tris"])
    rad = rt.Radiosity(scene, normals, None)
    rows = min(2048, rad.n)
    line = "%-15s %6d patches, %d rows:" % (stem, rad.n, rows)
    for mode in ("path", "flat", ""):
        if mode: os.environ["RTMI_FF_KERNEL"] = mode
        else: os.environ.pop("RTMI_FF_KERNEL", None)
        rad.findFormFactor(samples=10, seed=1, row0=0, n_rows=min(1024, rows)); torch.cuda.synchronize()   # warm-up (and, unforced, the library's own timing of the two)
        t0 = time.perf_counter()
        F = rad.findFormFactor(samples=10, seed=1, row0=0, n_rows=rows); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line += "  %s %.1f ms = %.1f G pair-samples/s (nonzero %.3f)" % (mode or "library's choice (%d)" % rt.lib.rtmi_debug_form_factor_kernel(rad.handle), dt * 1e3, 10.0 * rows * rad.n / dt / 1e9, float((F > 0).float().mean()))
    os.environ.pop("RTMI_FF_KERNEL", None)
    print(line, flush=True)
