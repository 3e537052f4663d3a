"""What splitting a 4K frame of the C3 soup into per-rank tile launches costs on one GPU (the launches of an N-GPU run, one after
the other), on one stream and spread over up to 4 streams."""
import os, sys
sys.path.insert(0, "/root/repo")
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
def timed(f, reps=10):
    f(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
full = timed(lambda: scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st))
for W in (2, 4, 8):
    def f():
        for s in range(W):
            scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), tile_first=s, tile_stride=W, stream=st)
    t = timed(f)
    streams = [torch.cuda.Stream() for _ in range(min(4, W))]
    def g():
        for s in range(W):
            scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), tile_first=s, tile_stride=W, stream=streams[s % len(streams)].cuda_stream)
    torch.cuda.synchronize()
    g(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(10): g()
    torch.cuda.synchronize()
    t4 = (time.perf_counter() - t0) / 10 * 1e3
    print("full frame %.3f ms; as %d launches of 1/%d of the tiles: %.3f ms on one stream (%.1f %% more), %.3f ms on %d streams (%.1f %% more)" % (
        full, W, W, t, 100 * (t / full - 1), t4, len(streams), 100 * (t4 / full - 1)))
