#!/usr/bin/env python3
"""Iteration aid: C3 closest-hit throughput + a digest of the frame (equal digests = identical hit records)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tris = workloads.soup(1 << 20, seed=1)
scene = rt.Scene(tris)
i = scene.info()
cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, workloads.SOUP_RES)
n = cam.x_res * cam.y_res
out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
dig = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
print("leaf<=%d depth %d nodes %d build %.2fs | %.3f ms  %.1f Mrays/s  digest %s %s" % (
    i.max_leaf_size, i.max_depth, i.n_nodes, i.build_seconds, ms, n / ms / 1e3, dig, os.environ.get("TAG", "")), flush=True)
cnt = torch.zeros(6, dtype=torch.int64, device="cuda")
scene.primary_work_dev(cam, cnt.data_ptr(), stream=st); torch.cuda.synchronize()
a, b, c, d, wl, wi = (int(x) for x in cnt.cpu())
print("  per ray: %.1f visits %.1f tests; lane occupancy node loop %.3f tri loop %.3f; wave-iters/ray node %.2f tri %.2f; node-step lanes: %.3f busy, %.3f hold a leaf, %.3f without a ray" % (
    a / n, b / n, a / max(c, 1), b / max(d, 1), c / 64 / n, d / 64 / n, a / max(c, 1), wl / max(c, 1), wi / max(c, 1)), flush=True)
# incoherent rays: the same rays in a random order through the ray-buffer path
rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda")
rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
perm = torch.randperm(n, device="cuda")
rays = rays.view(n, 64)[perm].contiguous()
o2 = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
scene.cast_dev(rays.data_ptr(), n, o2.data_ptr(), stream=st)
torch.cuda.synchronize()
e0.record()
for _ in range(3):
    scene.cast_dev(rays.data_ptr(), n, o2.data_ptr(), stream=st)
e1.record(); torch.cuda.synchronize()
ms2 = e0.elapsed_time(e1) / 3
ok = torch.equal(o2.view(n, 24), out.view(n, 24)[perm])
print("  shuffled rays via buffer: %.3f ms  %.1f Mrays/s  same records: %s" % (ms2, n / ms2 / 1e3, ok), flush=True)
