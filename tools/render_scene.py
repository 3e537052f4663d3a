#!/usr/bin/env python3
"""main.cpp of the reference (Render2/main.cpp:6-51) on the MI355X: BuildScene -> BuildBSP -> Rasterize -> PPM.
usage: render_scene.py <scenes_dir> <stem> <out.ppm> [--mode normal|aa|dof] [--depth N] [--ao N]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

ap = argparse.ArgumentParser()
ap.add_argument("scenes_dir"); ap.add_argument("stem"); ap.add_argument("out")
ap.add_argument("--mode", default="normal", choices=["normal", "aa", "dof"])
ap.add_argument("--depth", type=int, default=1)    # default_max_tree_depth, base.h:70
ap.add_argument("--ao", type=int, default=10)      # shader.cpp:481
a = ap.parse_args()
t0 = time.time()
s = rt.BuildScene(a.scenes_dir, a.stem)
scene = rt.Scene(s["tris"], s["tri_obj"])
sh = rt.Shader(scene, s["normals"], s["mats"], s["lights"], max_tree_depth=a.depth, ao_rays=a.ao)
cam = rt.Camera.from_cam14(s["cam14"], s["res"])
mode = {"normal": rt.RASTER_NORMAL, "aa": rt.RASTER_ANTI_ALIASING, "dof": rt.RASTER_DEPTH_OF_FIELD}[a.mode]
t1 = time.time()
_, pix = sh.Rasterize(cam, mode=mode)
t2 = time.time()
rt.write_ppm(a.out, pix)
print("%s: %d triangles, %d objects, %d lights, %dx%d -> %s (read+build %.2f s, frame %.3f s incl. first-launch costs)" % (
    a.stem, len(s["tris"]), len(s["mats"]), len(s["lights"]), cam.x_res, cam.y_res, a.out, t1 - t0, t2 - t1))
