#!/usr/bin/env python3
"""Where the end of a launch goes (diagnostic build: tools/build_variant.sh tail "-DRTMI_TAIL_DIAG", RTMI_LIB=.../librtmi_tail.so):
every wave of ONE k_primary_closest launch prints when it started, when it found its queue dry and when it ended; this script
launches a share of the C3 frame (every W-th tile; W = 1: all of it) and prints the distribution."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.cuda.set_device(0)
scene = rt.Scene(workloads.soup(1 << 20, seed=1))
cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, workloads.SOUP_RES)
out = torch.zeros(cam.x_res * cam.y_res * 24, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), tile_first=1 if W > 1 else 0, tile_stride=W, stream=st)
    torch.cuda.synchronize()
import ctypes, numpy as np
lib = ctypes.CDLL(os.environ["RTMI_LIB"])
buf = np.zeros(5 * 8192, dtype=np.uint64)
assert lib.rtmi_debug_tail_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
a = buf.reshape(8192, 5).astype(np.int64)
a = a[a[:, 0] > 0]
t0, dry, end = a[:, 0], a[:, 1], a[:, 2]
start = t0.min()
us = lambda t: (t - start) / 100.0
print("share 1/%d: waves %d; launch spans %.1f us (first wave start to last wave end)" % (W, len(a), us(end.max())))
print("wave start: median %.1f max %.1f us" % (np.median(us(t0)), us(t0.max())))
d = dry[dry > 0]
print("queue found dry (%d waves): min %.1f median %.1f max %.1f us" % (len(d), us(d.min()), np.median(us(d)), us(d.max())))
print("wave end: p1 %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(us(end), [1, 10, 50, 90, 99, 100])))
after = (end - np.where(dry > 0, dry, end)) / 100.0
print("time from dry to end per wave: median %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(after, [50, 90, 99, 100])))
r, b = a[:, 3], a[:, 4]
print("rounds after dry: median %d p90 %d max %d; mean busy lanes in those rounds %.1f; us per round after dry %.2f" % (np.median(r), np.percentile(r, 90), r.max(), b.sum() / max(1, r.sum()), after.sum() / max(1, r.sum())))
for back in (400, 300, 250, 200, 150, 100, 50, 20):
    t = end.max() - back * 100
    print("  %3d us before the end: %4d waves still running, %d have found the queue dry" % (back, int((end > t).sum()), int(((dry > 0) & (dry < t)).sum())))
