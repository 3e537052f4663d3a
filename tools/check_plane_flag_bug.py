#!/usr/bin/env python3
"""Does a library build show round 2's plane-flag bug (flags of the group's first patch applied to every lane)?  Builds the
multi-plane scenes of tests/test_gpu_parity.py and counts entries where the path walk and the root traversal differ.
    RTMI_LIB=raytracer_amd/variants/librtmi_r02.so python tools/check_plane_flag_bug.py     (round 2's kernel: differences)
    python tools/check_plane_flag_bug.py                                                     (this tree: none)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import raytracer_amd as rt
from test_gpu_parity import _multi_plane_scenes
torch.cuda.set_device(0)
for name, verts, normals in _multi_plane_scenes():
    n = len(verts)
    scene = rt.Scene(verts)
    F = {}
    for kernel in ("path", "flat"):
        os.environ["RTMI_FF_KERNEL"] = kernel
        rad = rt.Radiosity(scene, normals, None)
        F[kernel] = rad.findFormFactor(samples=10, seed=31).cpu().numpy()[:, :n]
        rad.close()
    bad = np.argwhere(F["path"].view(np.uint32) != F["flat"].view(np.uint32))
    vis_wrong = int(((F["path"] > 0) & (F["flat"] == 0)).sum())
    print("%-8s %5d patches: %d of %d entries differ (path walk sees %d pairs the root traversal finds hidden) %s  [%s]" % (
        name, n, len(bad), n * n, vis_wrong, bad[:4].tolist(), os.environ.get("RTMI_LIB", "librtmi.so")), flush=True)
