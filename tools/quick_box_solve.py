#!/usr/bin/env python3
"""Where the solve on the 64 K box's own matrix spends its time: the wrapper (emission, upload), the library call, the sweeps."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
from raytracer_amd.api import rtmi_solve_stats
torch.cuda.set_device(0)
verts, normals, light = workloads.closed_box(int(sys.argv[1]) if len(sys.argv) > 1 else 74, hole=0.3)   # the bench's box (a light inside a closed box lights nothing)
box = rt.Scene(verts)
rad = rt.Radiosity(box, normals, light)
rad.findFormFactor(samples=10, light_samples=30, seed=1)
torch.cuda.synchronize()
n, ld = rad.n, rad.ld
print("n %d ld %d" % (n, ld))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rad.progressiveRefinement(tol_rel=1e-4, max_sweeps=1000, check_every=4)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("wrapper: %.2f ms, %d sweeps" % (dt * 1e3, rad.stats.sweeps))
rho = np.array([rad.REFLECTIVITY_INDEX] * 3, np.float64)
E = rad.emission()
t0 = time.perf_counter(); E = rad.emission(); print("emission(): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); dE = torch.from_numpy(E).cuda(); torch.cuda.synchronize(); print("upload: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
dB = torch.zeros_like(dE)
st = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    stats = rtmi_solve_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = rt.lib.rtmi_radiosity_solve_dev(rad.FF.data_ptr(), ld, n, 0, n, rho.ctypes.data, dE.data_ptr(), dB.data_ptr(), 1e-4, 1000, 4, None, None, C.byref(stats), st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("library call: %.2f ms, %d sweeps, rc %d" % (dt * 1e3, stats.sweeps, rc))
B0, B1 = dE.clone(), torch.empty_like(dE)
norms = torch.zeros(3 * 64, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    e0.record()
    for k in range(10):
        rt.jacobi_sweep_dev(rad.FF.data_ptr(), ld, n, 0, n, rho, dE.data_ptr(), B0.data_ptr(), B1.data_ptr(), norms.data_ptr(), st)
        B0, B1 = B1, B0
    e1.record(); torch.cuda.synchronize()
    print("sweep kernel: %.3f ms each = %.0f GB/s" % (e0.elapsed_time(e1) / 10, 4.0 * n * n / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9))
