#!/usr/bin/env python3
"""Iteration aid: form-factor build rate on the 65 712-patch closed box (a sample of rows) + digest of the rows."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
k = int(sys.argv[2]) if len(sys.argv) > 2 else 74
row0 = int(os.environ.get("ROW0", "1000"))
verts, normals, light = workloads.closed_box(k)
box = rt.Scene(verts)
rad = rt.Radiosity(box, normals, light)
rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=row0, n_rows=1024)   # warm-up; the library times its two kernels here
torch.cuda.synchronize()
t0 = time.perf_counter()
F = rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=row0, n_rows=rows)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
dig = hashlib.sha1(np.ascontiguousarray(F.cpu().numpy()[:, : rad.n]).tobytes()).hexdigest()[:12]   # the matrix's columns, not the pitch's padding
if os.environ.get("DUMP"):
    np.save(os.environ["DUMP"], F.cpu().numpy())
i = box.info()
print("box %d patches depth %d | %d rows %.4f s  %.1f M pair-samples/s  full build %.1f s  row sum %.4f digest %s %s" % (
    rad.numOfElements, i.max_depth, rows, dt, 10.0 * rows * (rad.n - 2) / dt / 1e6, dt * rad.n / rows, float(F[0].sum()), dig,
    os.environ.get("TAG", "")), flush=True)
