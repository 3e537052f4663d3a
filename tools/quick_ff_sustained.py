"""Is the form-factor kernel's rate sustained?  The same 4096 rows of the 64 K box built 16 times back to back (event
time of each launch), then 4096-row slices across the matrix.  usage (through gpurun): python tools/quick_ff_sustained.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.set_device(0)
import raytracer_amd as rt
from raytracer_amd import workloads

verts, normals, light = workloads.closed_box(74, hole=0.2)
scene = rt.Scene(verts)
rad = rt.Radiosity(scene, normals, light)
n = rad.n
ld = (n + 3) // 4 * 4
F = torch.zeros((4096, ld), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def build(row0, rows):
    rt.api._check(rt.lib.rtmi_form_factors_dev(rad.handle, rt.api._ptr(rad.light_center), 10, 30, 1, 0, row0, rows, F.data_ptr(), ld, st))


build(0, 1024)          # tables + the library's choice of kernel
torch.cuda.synchronize()
for label, starts in (("the same rows 16 times", [1000] * 16), ("slices across the matrix", list(range(0, n - 4096, 4096)))):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(starts) + 1)]
    ev[0].record()
    for k, r0 in enumerate(starts):
        build(r0, 4096)
        ev[k + 1].record()
    torch.cuda.synchronize()
    print(label + ": ms per launch", " ".join("%.1f" % ev[k].elapsed_time(ev[k + 1]) for k in range(len(starts))), flush=True)
