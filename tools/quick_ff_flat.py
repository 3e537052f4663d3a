#!/usr/bin/env python3
"""Form-factor kernel on a single flat wall of 65 522 patches: every pair is culled by its cosines, no visibility ray is
shot — the cost of sampling alone (two points per sample, direction, cosines)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
verts = workloads._quad((0, 0, 0), (10, 0, 0), (0, 10, 0), 181)
normals = np.tile([0.0, 0.0, 1.0], (len(verts), 1))
rad = rt.Radiosity(rt.Scene(verts), normals, None)
rows = 512
rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=1000, n_rows=8); torch.cuda.synchronize()
t0 = time.perf_counter()
F = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=1000, n_rows=rows); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("flat wall %d patches | %d rows %.4f s  %.1f M pair-samples/s (no rays)  F max %.3g" % (len(verts), rows, dt, 10.0 * rows * len(verts) / dt / 1e6, float(F.max())))
