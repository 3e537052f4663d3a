"""The host-pointer entry rtmi_closest_hit (rays and records in host memory, staged over PCIe inside the call) on the C3
soup: PCIe-inclusive Mrays/s for several batch sizes, beside the device-resident rate.
usage (through gpurun): python tools/quick_host_path.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.set_device(0)
import raytracer_amd as rt
from raytracer_amd import workloads

verts = workloads.soup(1 << 20)
scene = rt.Scene(verts)
rng = np.random.default_rng(1)
for n in (1 << 16, 1 << 20, 8294400):
    o = np.tile(np.array([0.5, 0.5, -1.2]), (n, 1))
    t = rng.random((n, 3))
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = rt.make_rays(o, d)
    out = scene.Cast(rays)          # warm
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        out = scene.Cast(rays)
        best = min(best, time.time() - t0)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, 64)).cuda()
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    scene.cast_dev(d_rays.data_ptr(), n, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t0 = time.time()
    scene.cast_dev(d_rays.data_ptr(), n, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("%8d rays: host pointers %.2f ms = %.1f Mrays/s (%.1f GB/s over the link), device pointers %.2f ms = %.1f Mrays/s"
          % (n, best * 1e3, n / best / 1e6, n * 88 / best / 1e9, dt * 1e3, n / dt / 1e6), flush=True)
