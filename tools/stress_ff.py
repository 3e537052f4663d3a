#!/usr/bin/env python3
"""Fuzz of the form-factor kernels: random rooms with random furniture, soups and terrains, random samples per pair, seeds and
row ranges — the path walk (with its dealing of samples, certain hits, plane flags) against the root traversal, bit for bit.
usage: stress_ff.py [seconds=120] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_scenes = n_bad = n_entries = 0
while time.time() < t_end:
    kind = rng.integers(0, 4)
    if kind == 0:      # a room with boxes in it
        k = int(rng.integers(3, 14))
        v, nrm, _ = workloads.closed_box(k, hole=float(rng.choice([0.0, 0.3])))
        parts, normals = [v], [nrm]
        for _ in range(int(rng.integers(0, 5))):
            lo = rng.uniform(0.2, 7.0, 3); lo[2] = float(rng.choice([0.0, rng.uniform(0.0, 5.0)]))
            hi = np.minimum(lo + rng.uniform(0.3, 3.0, 3), 9.8)
            b = workloads._box(tuple(lo), tuple(hi), int(rng.integers(1, 5)))
            parts.append(b); normals.append(workloads.face_normals(b))
        verts, normals = np.concatenate(parts), np.concatenate(normals)
    elif kind == 1:    # a soup
        verts = workloads.soup(int(rng.integers(64, 1500)), seed=int(rng.integers(1, 1 << 30)), edge=float(rng.uniform(0.05, 0.6)))
        normals = workloads.face_normals(verts)
    elif kind == 2:    # a room scaled and moved (the plane flags' thresholds follow the scene's size)
        v, nrm, _ = workloads.closed_box(int(rng.integers(3, 10)))
        s = float(10.0 ** rng.uniform(-2.5, 2.0)); off = rng.uniform(-1.0, 1.0, 3) * float(10.0 ** rng.uniform(-1, 3.5))
        verts, normals = (v.reshape(-1, 3, 3) * s + off).reshape(-1, 9), nrm
    else:              # two rooms side by side sharing a wall with a doorway's worth of gap
        a, na, _ = workloads.closed_box(int(rng.integers(3, 9)), hole=0.4)
        b = (a.reshape(-1, 3, 3) + np.array([10.0 + float(rng.choice([0.0, 0.5])), 0.0, 0.0])).reshape(-1, 9)
        verts, normals = np.concatenate([a, b]), np.concatenate([na, na])
    n = len(verts)
    samples = int(rng.choice([1, 2, 3, 5, 7, 10, 10, 10, 13, 21, 33, 64, 65]))
    seed = int(rng.integers(0, 1 << 40))
    row0 = int(rng.integers(0, max(1, n - 1)))
    n_rows = int(min(n - row0, rng.integers(1, 400)))
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals, None)
    os.environ["RTMI_FF_KERNEL"] = "path"
    Fp = rad.findFormFactor(samples=samples, seed=seed, row0=row0, n_rows=n_rows)[:, :n].clone()
    os.environ["RTMI_FF_KERNEL"] = "flat"
    Ff = rad.findFormFactor(samples=samples, seed=seed, row0=row0, n_rows=n_rows)[:, :n]
    bad = int((Fp.view(torch.int32) != Ff.view(torch.int32)).sum())
    scene.check()
    n_scenes += 1; n_entries += n_rows * n
    if bad:
        n_bad += 1
        print("MISMATCH: kind %d n %d samples %d seed %d rows %d+%d: %d entries differ" % (kind, n, samples, seed, row0, n_rows, bad), flush=True)
    rad.close(); scene.close()
    if n_scenes % 50 == 0:
        print("%d scenes, %d entries, %d scenes with a mismatch" % (n_scenes, n_entries, n_bad), flush=True)
print("done: %d scenes, %d entries compared, %d scenes with a mismatch" % (n_scenes, n_entries, n_bad))
sys.exit(1 if n_bad else 0)
