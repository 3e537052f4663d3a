#!/usr/bin/env python3
"""k_jacobi_sweep against the row pitch of F: n columns, ld = n rounded up to a multiple of 4, 32, 64, 256, 1024 floats (and n = 65536)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
rho = np.array([0.5] * 3, np.float64)
for n in (int(sys.argv[1]) if len(sys.argv) > 1 else 64745, 65536):
    for mult in (4, 32, 64, 256, 1024):
        ld = (n + mult - 1) // mult * mult
        F = torch.rand((n, ld), dtype=torch.float32, device="cuda") * (1.8 / n)
        E = torch.rand((n, 3), dtype=torch.float32, device="cuda")
        B0, B1 = E.clone(), torch.empty_like(E)
        norms = torch.zeros(3 * 64, dtype=torch.float64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            e0.record()
            for k in range(10):
                rt.jacobi_sweep_dev(F.data_ptr(), ld, n, 0, n, rho, E.data_ptr(), B0.data_ptr(), B1.data_ptr(), norms.data_ptr(), st)
                B0, B1 = B1, B0
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        print("n %d ld %d (multiple of %4d): %.3f ms per sweep = %.0f GB/s of algorithmic bytes" % (n, ld, mult, best, 4.0 * n * n / (best * 1e-3) / 1e9), flush=True)
        del F
