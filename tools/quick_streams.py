"""Which streams the 8 tile launches of one rank are dealt over, and what that does to their wall time (C3 frame)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
scene = rt.Scene(workloads.soup(1 << 20, seed=1))
cam = rt.Camera.from_cam14(workloads.SOUP_CAM14, workloads.SOUP_RES)
out = torch.zeros(cam.x_res * cam.y_res * 24, dtype=torch.uint8, device="cuda")
pool = [torch.cuda.Stream() for _ in range(16)]
def wall(handles, reps=10):
    def g():
        for s in range(8):
            scene.primary_dev(cam, out.data_ptr(), tile=(64, 64), tile_first=s, tile_stride=8, stream=handles[s % len(handles)])
    g(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, idx in (("0-3", [0, 1, 2, 3]), ("4-7", [4, 5, 6, 7]), ("6-9", [6, 7, 8, 9]), ("0-1", [0, 1]), ("0-2", [0, 1, 2]), ("0-7", list(range(8))),
                  ("0-3 again", [0, 1, 2, 3]), ("1-4", [1, 2, 3, 4]), ("0,2,4,6", [0, 2, 4, 6])):
    print("streams %-10s %.3f ms" % (name, wall([pool[i].cuda_stream for i in idx])), flush=True)
for i in range(10):
    print("streams only %d    %.3f ms" % (i, wall([pool[i].cuda_stream])), flush=True)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
own = []
for _ in range(4):
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0
    own.append(h.value)
print("streams own 4 (hipStreamNonBlocking) %.3f ms" % wall(own), flush=True)
print("streams own 2 %.3f ms" % wall(own[:2]), flush=True)
print("streams pairs (0,1) %.3f (2,3) %.3f (4,5) %.3f (1,3) %.3f (0,2) %.3f" % tuple(wall([pool[a].cuda_stream, pool[b].cuda_stream]) for a, b in ((0, 1), (2, 3), (4, 5), (1, 3), (0, 2))), flush=True)
