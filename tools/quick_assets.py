#!/usr/bin/env python3
"""The reference's own large meshes (tests/golden/scenes: full_room.obj, Golden_Buddha_low.obj) through the lenient reader,
scene build and the frame path at 1920x1080: load / build seconds, ms per frame at depth 1 and 3 (10 AO rays), primary Mrays/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import raytracer_amd as rt
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream
for stem in ("real_full_room", "real_buddha"):
    t0 = time.perf_counter()
    s = rt.BuildScene(os.path.join(ROOT, "tests", "golden", "scenes"), stem, lenient=True)
    t1 = time.perf_counter()
    scene = rt.Scene(s["tris"], s["tri_obj"])
    t2 = time.perf_counter()
    cam = rt.Camera.from_cam14(s["cam14"], (1920, 1080))
    n = cam.x_res * cam.y_res
    out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    rgb = torch.zeros(n, 3, dtype=torch.float64, device="cuda"); pix = torch.zeros(n, 3, dtype=torch.uint8, device="cuda")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def timed(f, reps=10):
        f(); torch.cuda.synchronize(); a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    pm = timed(lambda: scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), stream=st))
    line = "%-15s %6d triangles: read %.3f s, scene %.3f s; 1920x1080 primary %.3f ms = %.0f Mrays/s" % (stem, len(s["tris"]), t1 - t0, t2 - t1, pm, n / pm / 1e3)
    for depth in (1, 3):
        sh = rt.Shader(scene, s["normals"], s["mats"], s["lights"], max_tree_depth=depth, ao_rays=10)
        ms = timed(lambda: sh.render_frame_dev(cam, rgb.data_ptr(), pix.data_ptr(), stream=st), 5)
        line += "; frame depth %d %.2f ms" % (depth, ms)
    print(line, flush=True)
