#!/usr/bin/env python3
"""Closest-hit / any-hit throughput of the C3 frame against the triangle count (soup of n triangles, 3840x2160)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, workloads.SOUP_RES)
n = cam.x_res * cam.y_res
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
occ = torch.zeros(n, dtype=torch.uint8, device="cuda")
rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(f, reps=5):
    f(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for logn in [int(a) for a in sys.argv[1:]] or [16, 18, 20, 22, 23]:
    nt = 1 << logn
    tris = workloads.soup(nt, seed=1, edge=0.01 * (2 ** ((20 - logn) / 3.0)))   # keep the covered fraction comparable
    t0 = time.perf_counter(); scene = rt.Scene(tris); tb = time.perf_counter() - t0
    i = scene.info()
    a = timed(lambda: scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st))
    b = timed(lambda: scene.cast_dev(rays.data_ptr(), n, out.data_ptr(), stream=st))
    c = timed(lambda: scene.occluded_dev(rays.data_ptr(), n, occ.data_ptr(), stream=st))
    hit = float(occ.float().mean())
    print("2^%d = %8d tris | build+upload %.2f s depth %d, %6.1f MB | primary closest %7.1f Mrays/s | buffered closest %7.1f | buffered any-hit %7.1f | hit fraction %.3f" % (
        logn, nt, tb, i.max_depth, i.device_bytes / 1e6, n / a / 1e3, n / b / 1e3, n / c / 1e3, hit), flush=True)
    scene.close()
