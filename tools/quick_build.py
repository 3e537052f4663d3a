"""Seconds per phase of rtmi_scene_create on the C3 soup (RTMI_VERBOSE=1 prints them) and with fewer build threads.
usage (through gpurun): python tools/quick_build.py [n_tris]"""
import os
import sys
import time

os.environ["RTMI_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.set_device(0)
import raytracer_amd as rt
from raytracer_amd import workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
verts = workloads.soup(n)
for threads in (16, 8, 4):
    os.environ["RTMI_BUILD_THREADS"] = str(threads)
    t = time.time()
    scene = rt.Scene(verts)
    print("threads %2d: Scene() %.2f s, library's build_seconds %.2f" % (threads, time.time() - t, scene.info().build_seconds), flush=True)
    scene.close()
print("os.cpu_count()", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
