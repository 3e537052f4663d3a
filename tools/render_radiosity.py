#!/usr/bin/env python3
"""The reference's radiosity pipeline end to end on the MI355X — what `scene.radiosity = new Radiosity(&scene, &camera)`
(Render2/radiosity.cpp:18-60, commented out in the shipped main.cpp:31) followed by Rasterizer::Radiosity_Raster does:
BuildScene -> elements = the scene's triangles -> findFormFactor -> progressiveRefinement -> setRadiosityForAllElements
-> Radiosity_Raster -> PPM (flat shading per element: the shared-vertex averages of returnQuadTreeColor need the
reference's adjacency pass and come in through the adapter only).
usage: render_radiosity.py <scenes_dir> <stem> <out.ppm>    or    render_radiosity.py --box K <out.ppm>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import workloads

if sys.argv[1] == "--box":
    k, out = int(sys.argv[2]), sys.argv[3]
    verts, normals, light = workloads.closed_box(k, hole=0.3)
    cam = rt.Camera.from_cam14(np.array([3.0, 4.0, 6.0, 0.0, 2.0, 3.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.75, 0.75]), (800, 600))
else:
    s = rt.BuildScene(sys.argv[1], sys.argv[2])
    out = sys.argv[3]
    verts, normals = s["tris"], s["normals"]
    light = s["lights"][0, :3] if len(s["lights"]) else None
    cam = rt.Camera.from_cam14(s["cam14"], s["res"])
t0 = time.time()
scene = rt.Scene(verts)
rad = rt.Radiosity(scene, normals, light)
rad.findFormFactor(samples=10, light_samples=30, seed=1)
t1 = time.time()
rad.progressiveRefinement(tol_rel=1e-6)
B = rad.setRadiosityForAllElements()
t2 = time.time()
_, pix = rad.Radiosity_Raster(cam)
t3 = time.time()
rt.write_ppm(out, pix)
print("%d elements: form factors %.3f s, solve %.3f s (%d sweeps), display %.3f s; radiosity min %.3g mean %.3g max %.3g -> %s" % (
    len(verts), t1 - t0, t2 - t1, rad.stats.sweeps, t3 - t2, B.min(), B.mean(), B.max(), out))
