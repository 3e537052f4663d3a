#!/usr/bin/env python3
"""For rocprofv3 --kernel-trace --stats: a few 1920x1080 frames of the studio scene through rtmi_render_frame_dev.
argv: k depth ao_rays frames"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads as W
k, depth, ao, frames = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 64), (2, 1), (3, 10), (4, 5)))
torch.cuda.set_device(0)
verts, obj, mats, lights = W.studio(k, 0.1)
scene = rt.Scene(verts, obj)
sh = rt.Shader(scene, W.face_normals(verts), mats, lights, max_tree_depth=depth, ao_rays=ao)
cam = rt.Camera.from_cam14(W.STUDIO_CAM14, (1920, 1080))
n = 1920 * 1080
rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda"); pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
for _ in range(frames):
    sh.render_frame_dev(cam, rgb.data_ptr(), pix.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("mean colour", rgb.mean(0).tolist())
