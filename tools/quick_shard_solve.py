#!/usr/bin/env python3
"""What the exchange costs per sweep at the shard size of an 8-GPU solve, measured on ONE GPU: 8 192 rows of a 65 536-patch
system (2 GiB of F), (a) the sweep kernel alone, launches back to back; (b) rtmi_radiosity_solve_sharded_dev with a
communicator of one rank — sweep + in-place RCCL all-gather enqueued per sweep, two all-reduces + one host wait per check;
(c) the Python loop multi.solve_sharded over the same sweep (the round-1 path).  The one-rank all-gather moves no data over
xGMI; what shows here is the launch / enqueue / synchronisation cost of the exchange path, which is what binds at 0.34 ms
per sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import multi, workloads
torch.cuda.set_device(0)
n, rows = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ld = n
F = workloads.synthetic_F(n, n_rows=rows, row0=0, seed=1, ld=ld)
rho = np.array([0.7, 0.7, 0.7])
E = torch.zeros((n, 3), dtype=torch.float32, device="cuda"); E[::97] = 68500.0 / 97.0
st = torch.cuda.current_stream().cuda_stream
Ba = E.clone(); Bb = torch.zeros_like(Ba); norms = torch.zeros(3, dtype=torch.float64, device="cuda")
def sweep(a, b, nm):
    rt.jacobi_sweep_dev(F.data_ptr(), ld, n, 0, rows, rho, E.data_ptr(), a.data_ptr(), b.data_ptr(), nm.data_ptr(), st)
for _ in range(4): sweep(Ba, Bb, norms)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): sweep(Ba, Bb, norms); sweep(Bb, Ba, norms)
e1.record(); torch.cuda.synchronize()
k_ms = e0.elapsed_time(e1) / 100
out = {"rows": rows, "kernel_ms_per_sweep": round(k_ms, 4), "GBs": round(4.0 * rows * n / k_ms / 1e6, 1)}
for ce in (4, 16):
    B, s = multi.solve_sharded_library(F.data_ptr(), ld, n, 0, rows, rho, E, tol_rel=0.0, max_sweeps=8, check_every=ce)      # warm-up, creates the communicator
    t0 = time.perf_counter()
    B, s = multi.solve_sharded_library(F.data_ptr(), ld, n, 0, rows, rho, E, tol_rel=0.0, max_sweeps=96, check_every=ce)
    torch.cuda.synchronize()
    out["library_wall_ms_per_sweep_check_every_%d" % ce] = round((time.perf_counter() - t0) * 1e3 / s.sweeps, 4)
t0 = time.perf_counter()
r = multi.solve_sharded(sweep, E, n, tol_rel=0.0, max_sweeps=96, check_every=4)
torch.cuda.synchronize()
out["python_loop_wall_ms_per_sweep_world1"] = round((time.perf_counter() - t0) * 1e3 / r.sweeps, 4)
print(out, flush=True)
