#!/usr/bin/env python3
"""Whole frames through rtmi_render_frame_dev (Scene::Trace + Shader::Shade on the device): the shipped test1 frame
beside the compiled reference's own time for it (tests/golden/g6_shade_test1.npz: trace_seconds, this container's CPU),
and a ~150 K-triangle studio at 1920x1080 (depth 1 and 3, 10 AO rays).  Prints ms per frame, Mpixels/s and the largest
colour difference to the CPU oracle on a pixel sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads as W
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream


def time_frames(sh, cam, reps=5):
    n = cam.x_res * cam.y_res
    rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda"); pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
    sh.render_frame_dev(cam, rgb.data_ptr(), pix.data_ptr(), stream=st); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        sh.render_frame_dev(cam, rgb.data_ptr(), pix.data_ptr(), stream=st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, rgb.cpu().numpy()


g = np.load(os.path.join(ROOT, "tests", "golden", "g6_shade_test1.npz"))
scene = rt.Scene(g["tris"], g["tri_obj"])
sh = rt.Shader(scene, g["normals"], g["mats"], g["lights"], max_tree_depth=1, ao_rays=10)
cam = rt.Camera.from_cam14(g["cam14"], g["res"])
ms, _ = time_frames(sh, cam, 20)
print("C1 test1 500x400, 24 tris, AO 10, depth 1: %.3f ms/frame = %.0f Mpixels/s; compiled reference Scene::Trace loop: %.0f ms (1 core) -> x%.0f" % (
    ms, 0.2 / ms * 1e3, 1e3 * float(g["trace_seconds"]), 1e3 * float(g["trace_seconds"]) / ms), flush=True)

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
verts, obj, mats, lights = W.studio(k, 0.1)
nrm = W.face_normals(verts)
t0 = time.time(); scene = rt.Scene(verts, obj); tb = time.time() - t0
cam = rt.Camera.from_cam14(W.STUDIO_CAM14, (1920, 1080))
from oracle_py import Oracle
ora = Oracle()
bvh = ora.bvh_build(verts)
rd = ora.raster_details(W.STUDIO_CAM14, 1920, 1080)
rng = np.random.default_rng(1)
pixs = np.stack([rng.integers(0, 1080, 20000), rng.integers(0, 1920, 20000)], 1).astype(np.uint32)
d = ora.primary_dirs(rd, pixs); o = np.tile(W.STUDIO_CAM14[:3], (len(d), 1))
idx = pixs[:, 0].astype(np.int64) * 1920 + pixs[:, 1]
for depth, ao in ((1, 0), (1, 10), (3, 10), (5, 10)):
    sh = rt.Shader(scene, nrm, mats, lights, max_tree_depth=depth, ao_rays=ao, seed=1)
    ms, rgb = time_frames(sh, cam)
    t0 = time.time()
    want = np.concatenate([ora.trace(verts, nrm, obj, mats, lights, o[i:i + 1], d[i:i + 1], max_depth=depth, ao_rays=ao, seed=1,
                                     first_index=int(idx[i]), bvh=bvh) for i in range(0, 2000)]) if depth <= 3 else None
    tc = time.time() - t0
    err = np.abs(rgb[idx[:2000]] - want).max() if want is not None else float("nan")
    print("studio(%d) %d tris (build %.2f s), 1920x1080, depth %d, AO %d: %.2f ms/frame = %.0f Mpixels/s; max |colour - oracle| on 2000 pixels %.2e (oracle %.1f us/pixel)" % (
        k, len(verts), tb, depth, ao, ms, 1920 * 1080 / ms / 1e3, err, tc / 2000 * 1e6), flush=True)
