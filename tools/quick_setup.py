"""Host-side set-up times of the radiosity path on the 64 K-patch box: scene build, patch set, path tables (first
form-factor call), shader.  usage (through gpurun): python tools/quick_setup.py [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.set_device(0)
import raytracer_amd as rt
from raytracer_amd import workloads

k = int(sys.argv[1]) if len(sys.argv) > 1 else 74
verts, normals, light = workloads.closed_box(k, hole=0.2)
torch.cuda.synchronize()


def timed(what, fn):
    t = time.time()
    out = fn()
    torch.cuda.synchronize()
    print("%-44s %.3f s" % (what, time.time() - t), flush=True)
    return out


print(len(verts), "patches")
scene = timed("rt.Scene (rtmi_scene_create)", lambda: rt.Scene(verts))
rad = timed("rt.Radiosity (rtmi_patches_create)", lambda: rt.Radiosity(scene, normals, light))
timed("first form-factor call, 8 rows (path tables)", lambda: rad.findFormFactor(samples=10, seed=1, row0=0, n_rows=8))
timed("second form-factor call, 8 rows", lambda: rad.findFormFactor(samples=10, seed=1, row0=0, n_rows=8))
timed("scene.check()", lambda: scene.check())
