import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["RTMI_VERBOSE"] = "1"
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
verts, normals, light = workloads.closed_box(73)
t0 = time.perf_counter(); box = rt.Scene(verts); torch.cuda.synchronize(); print("scene %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); rad = rt.Radiosity(box, normals, light); torch.cuda.synchronize(); print("patches %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=0, n_rows=1024); torch.cuda.synchronize(); print("first call %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); rad.findFormFactor(samples=10, light_samples=30, seed=1, row0=0, n_rows=1024); torch.cuda.synchronize(); print("second call %.1f ms" % ((time.perf_counter() - t0) * 1e3))
