#!/usr/bin/env python3
"""What a host-pointer call costs for small batches (the reference's per-ray Scene::Cast left un-batched goes through it):
rtmi_closest_hit / rtmi_any_hit on 1, 16, 256, 4096 rays of the test1 scene, microseconds per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
torch.cuda.set_device(0)
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g2_test1.npz"))
scene = rt.Scene(g["tris"], g["tri_obj"])
rng = np.random.default_rng(1)
for n in (1, 16, 256, 4096, 65536):
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    rays = rt.make_rays(np.tile(g["cam"][:3], (n, 1)), d)
    scene.Cast(rays); scene.Occluded(rays)
    reps = 300 if n <= 4096 else 30
    t0 = time.perf_counter()
    for _ in range(reps): scene.Cast(rays)
    t1 = time.perf_counter()
    for _ in range(reps): scene.Occluded(rays)
    t2 = time.perf_counter()
    print("n = %5d: rtmi_closest_hit %.1f us per call, rtmi_any_hit %.1f us per call %s" % (n, (t1 - t0) / reps * 1e6, (t2 - t1) / reps * 1e6, os.environ.get("TAG", "")), flush=True)
