#!/usr/bin/env python3
"""Stress of the traversal queue's item hand-out (shared range per workgroup, claims from one counter): many launches of
random sizes and tile dealings on two streams, every result compared with the records of ONE reference launch — an item handed
out twice or not at all would show as a wrong or missing record."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scene = rt.Scene(workloads.soup(1 << 18, seed=2, edge=0.02))
cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, (1920, 1080))
n = cam.x_res * cam.y_res
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
ref = torch.zeros(n, 24, dtype=torch.uint8, device="cuda")
scene.primary_dev(cam, ref.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
rays = torch.zeros(n, 64, dtype=torch.uint8, device="cuda")
rt.primary_rays_dev(cam, rays.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
rng = np.random.default_rng(1)
bad = 0
t0 = time.perf_counter()
for it in range(reps):
    W = int(rng.integers(1, 9)); first = int(rng.integers(0, W)); tw = int(rng.choice([16, 32, 64, 100])); th = int(rng.choice([16, 24, 64]))
    out = torch.full((n, 24), 0xAB, dtype=torch.uint8, device="cuda")
    m = int(rng.integers(1, n)); off = int(rng.integers(0, n - m + 1))
    out2 = torch.zeros(m, 24, dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(s0):
        scene.primary_dev(cam, out.data_ptr(), tile=(tw, th), tile_first=first, tile_stride=W, stream=s0.cuda_stream)
    with torch.cuda.stream(s1):
        scene.cast_dev(rays[off:].data_ptr(), m, out2.data_ptr(), stream=s1.cuda_stream)
    torch.cuda.synchronize()
    written = (out != 0xAB).any(1)
    tiles_x = (cam.x_res + tw - 1) // tw
    px = torch.arange(n, device="cuda")
    tile_of = (px // cam.x_res // th) * tiles_x + (px % cam.x_res) // tw
    mine = (tile_of >= first) & ((tile_of - first) % W == 0)
    ok = bool(torch.equal(written, mine)) and bool(torch.equal(out[mine], ref[mine])) and bool(torch.equal(out2, ref[off:off + m]))
    if not ok:
        bad += 1
        print("iteration %d: mismatch (W %d first %d tile %dx%d; buffer of %d rays at %d)" % (it, W, first, tw, th, m, off), flush=True)
    if it % 50 == 49:
        print("%d launches pairs done, %d bad, %.1f s" % (it + 1, bad, time.perf_counter() - t0), flush=True)
scene.check()
print("stress: %d pairs of concurrent launches, %d mismatches" % (reps, bad), flush=True)
sys.exit(1 if bad else 0)
