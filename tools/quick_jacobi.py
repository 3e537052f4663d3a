#!/usr/bin/env python3
"""Iteration aid: Jacobi sweep bandwidth at 64 K patches (synthetic F) + a checksum of B after 4 sweeps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = int(sys.argv[2]) if len(sys.argv) > 2 else n          # rows of this "rank" (8192 = the 8-GPU shard at 64 K)
F = workloads.synthetic_F(n, n_rows=rows, seed=1)
rho = np.array([0.7, 0.7, 0.7])
E = torch.rand((n, 3), device="cuda"); Ba = E.clone(); Bb = torch.zeros_like(Ba)
norms = torch.zeros(3, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def sw(a, b): rt.jacobi_sweep_dev(F.data_ptr(), n, n, 0, rows, rho, E.data_ptr(), a.data_ptr(), b.data_ptr(), norms.data_ptr(), st)
for _ in range(2): sw(Ba, Bb); sw(Bb, Ba)
chk = float(Ba.double().sum())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): sw(Ba, Bb); sw(Bb, Ba)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("n=%d rows=%d sweep %.4f ms  %.1f GB/s  (%.1f %% of 8 TB/s)  checksum %.9e %s" % (n, rows, ms, 4.0 * rows * n / ms / 1e6, 4.0 * rows * n / ms / 1e6 / 80, chk, os.environ.get("TAG", "")), flush=True)
