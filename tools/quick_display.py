#!/usr/bin/env python3
"""Radiosity display pass (rtmi_radiosity_raster_dev: the reference's O(elements) trace_ray per pixel) on the C2 / C5
boxes: ms per frame and element tests per second."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
for k, res in ((18, (1920, 1080)), (74, (1920, 1080))):
    verts, normals, light = workloads.closed_box(k)
    scene = rt.Scene(verts)
    rad = rt.Radiosity(scene, normals)
    n_el = len(verts)
    vr = torch.rand(n_el, 3, dtype=torch.float64, device="cuda")
    cam = rt.Camera.from_cam14(np.array([3.0, 4.0, 6.0, 0.0, 2.0, 3.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.5625, 0.5625]), res)
    n = res[0] * res[1]
    rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda"); pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
    import ctypes as C
    def go():
        rt.api._check(rt.lib.rtmi_radiosity_raster_dev(rad.handle, vr.data_ptr(), C.byref(cam.c), 0.5, 0.5, 0, 0, 0, 1, rgb.data_ptr(), pix.data_ptr(), st))
    for mode in ("1", "2"):
        os.environ["RTMI_DISPLAY_QUERY"] = mode
        go(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); go(); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        print("display pass (%s): %6d elements, %dx%d: %.2f ms; mean grey %.1f" % (
            "BVH of the elements" if mode == "2" else "loop over all elements = %.0f G element tests/s" % (n_el * n / ms / 1e6),
            n_el, res[0], res[1], ms, float(pix.float().mean())), flush=True)
