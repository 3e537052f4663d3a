#!/usr/bin/env python3
"""Sweeps to tolerance of the solve loop: Jacobi against Chebyshev steps for several assumed lower ends of the spectrum
(RTMI_CHEB_LOW_PERCENT), on a synthetic matrix (row sums 0.9) and on the real form-factor matrix of the C2 box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
def run(name, dF, ld, n, rho, dE, tol, ce=4):
    line = "%-28s tol %.0e:" % (name, tol)
    ref = None
    for mode in ("jacobi", "-100", "-50", "-25", "-10", "0"):
        if mode == "jacobi": os.environ["RTMI_SOLVER"] = "jacobi"
        else:
            os.environ.pop("RTMI_SOLVER", None); os.environ["RTMI_CHEB_LOW_PERCENT"] = mode
        dB = torch.zeros_like(dE)
        s = rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), dB.data_ptr(), tol, 2000, ce, st)
        if ref is None:
            tight = torch.zeros_like(dE)
            rt.api.radiosity_solve_dev(dF.data_ptr(), ld, n, rho, dE.data_ptr(), tight.data_ptr(), 1e-9, 5000, ce, st)
            ref = tight.double()
        err = float((dB.double() - ref).norm() / ref.norm())
        line += "  %s: %d sweeps (%d acc, b %.3f) err %.1e |" % (mode, s.sweeps, s.accelerated_sweeps, s.spectral_radius, err)
    print(line, flush=True)
n = 8192
F = workloads.synthetic_F(n, seed=1); ld = F.shape[1]
E = torch.zeros((n, 3), device="cuda"); E[::97] = 68500.0 / 97; E[n - 1] = 68500.0
for tol in (1e-4, 1e-6):
    run("synthetic %d" % n, F, ld, n, np.array([0.7, 0.7, 0.7]), E, tol)
for k, hole in ((18, 0.3), (40, 0.3)):
    verts, normals, light = workloads.closed_box(k, hole=hole)
    rad = rt.Radiosity(rt.Scene(verts), normals, light)
    Fb = rad.findFormFactor(samples=10, light_samples=30, seed=1)
    nb = rad.n
    Eb = torch.from_numpy(rad.emission()).cuda()
    for tol in (1e-4, 1e-7):
        run("box k=%d (%d patches)" % (k, nb), Fb, rad.ld, nb, np.array([0.7, 0.7, 0.7]), Eb, tol)
