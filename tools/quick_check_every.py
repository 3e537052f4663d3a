import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
verts, normals, light = workloads.closed_box(74, hole=0.3)   # the bench's box: 64 744 patches, the light above the ceiling's opening
box = rt.Scene(verts)
rad = rt.Radiosity(box, normals, light)
rad.findFormFactor(samples=10, light_samples=30, seed=1)
torch.cuda.synchronize()
ref = None
for ce in (4, 4, 3, 2, 1, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rad.progressiveRefinement(tol_rel=1e-4, max_sweeps=1000, check_every=ce)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    B = rad.B[: rad.numOfElements, 0].double().cpu().numpy()
    if ref is None:
        rad.progressiveRefinement(tol_rel=1e-9, max_sweeps=1000, check_every=4)
        ref = rad.B[: rad.numOfElements, 0].double().cpu().numpy()
    print("check_every %d: %.2f ms, %d sweeps, rel_change %.2e, rel L2 error vs 1e-9 solve %.2e" % (ce, dt * 1e3, rad.stats.sweeps, rad.stats.rel_change, np.linalg.norm(B - ref) / np.linalg.norm(ref)), flush=True)
