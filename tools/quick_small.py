"""Small device-pointer launches on the C3 soup (latency-bound: a ray is ~45 dependent node steps): ms per launch for
2^12 .. 2^20 buffered rays.  usage (through gpurun): python tools/quick_small.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.set_device(0)
import raytracer_amd as rt
from raytracer_amd import workloads

scene = rt.Scene(workloads.soup(1 << 20))
rng = np.random.default_rng(1)
nmax = 1 << 20
o = np.tile(np.array([0.5, 0.5, -1.2]), (nmax, 1))
d = rng.random((nmax, 3)) - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = rt.make_rays(o, d)
d_rays = torch.from_numpy(rays.view(np.uint8).reshape(nmax, 64)).cuda()
d_out = torch.zeros(nmax * 24, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
line = []
for lg in (12, 14, 16, 17, 18, 19, 20):
    n = 1 << lg
    for _ in range(3):
        scene.cast_dev(d_rays.data_ptr(), n, d_out.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        scene.cast_dev(d_rays.data_ptr(), n, d_out.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    line.append("2^%d %.3f" % (lg, e0.elapsed_time(e1) / 20))
print("ms per launch:", "  ".join(line))
scene.check()
