#!/usr/bin/env python3
"""C2 (Cornell-type box, 3 816 patches with a ceiling opening): form factors, solve, display — device and wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
verts, normals, light = workloads.closed_box(18, hole=0.3)
scene = rt.Scene(verts)
rad = rt.Radiosity(scene, normals, light)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rad.findFormFactor(samples=10, light_samples=30, seed=1); torch.cuda.synchronize(); t1 = time.perf_counter()
    rad.progressiveRefinement(tol_rel=1e-7); torch.cuda.synchronize(); t2 = time.perf_counter()
print("C2 %d patches: form factors %.2f ms; solve %d sweeps: wall %.3f ms, device %.3f ms (%.1f us per sweep)" % (
    len(verts), 1e3 * (t1 - t0), rad.stats.sweeps, 1e3 * (t2 - t1), 1e3 * rad.stats.seconds, 1e6 * rad.stats.seconds / rad.stats.sweeps))
