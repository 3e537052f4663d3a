#!/usr/bin/env python3
"""Form-factor kernel time against the number of samples per pair (fixed cost per pair + cost per sample), 4096 rows of the 65 712-patch box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
verts, normals, light = workloads.closed_box(74)
box = rt.Scene(verts)
rad = rt.Radiosity(box, normals, light)
rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=1000, n_rows=8)
for s in (1, 2, 5, 10, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rad.findFormFactor(samples=s, light_samples=30, seed=1, row0=1000, n_rows=4096)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("samples %2d: %.2f ms per 4096 rows = %.3f ns per pair = %.3f ns per pair-sample" % (s, dt * 1e3, dt * 1e9 / 4096 / rad.n, dt * 1e9 / 4096 / rad.n / s), flush=True)
