#!/usr/bin/env python3
"""One rank's share of a dealt frame (every W-th tile of the C3 frame, W = 2, 4, 8) against RTMI_MIN_RAYS_PER_LANE: does a
smaller persistent grid (fewer, longer-lived waves) shorten a launch that has only a few rays per resident lane?"""
import os, sys
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
def timed(f, reps=20):
    f(); f(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
full = timed(lambda: scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st))
print("full frame %.3f ms" % full, flush=True)
for W in (2, 4, 8):
    line = "1/%d of the tiles (ideal %.3f ms):" % (W, full / W)
    for m in ((0,) if "short" in sys.argv else (0, 2, 3, 4, 6, 8, 12, 16)):
        os.environ["RTMI_MIN_RAYS_PER_LANE"] = str(m)
        t = timed(lambda: scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), tile_first=1, tile_stride=W, stream=st))
        line += "  min %d: %.3f" % (m, t)
    print(line, flush=True)
os.environ["RTMI_MIN_RAYS_PER_LANE"] = "0"
