#!/usr/bin/env python3
"""Host and device cost of one (nearly empty) launch of the queue kernels: what a launch costs besides its rays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
scene = rt.Scene(workloads.soup(1 << 20, seed=1))
cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, workloads.SOUP_RES)
n = cam.x_res * cam.y_res
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for stride in (2040, 64, 8):       # 1 tile, 32 tiles, 255 tiles of the 2040
    f = lambda: scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), tile_first=0, tile_stride=stride, stream=st)
    f(); torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): f()
    t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
    print("1/%d of the tiles (%d rays): host %.1f us per call, device %.1f us per launch" % (
        stride, n // stride, 1e6 * (t1 - t0) / reps, 1e3 * e0.elapsed_time(e1) / reps), flush=True)
