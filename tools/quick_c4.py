#!/usr/bin/env python3
"""C4 on one GPU: ~100 K-triangle room, 1920x1080, primary closest hit -> 16 AO rays (radius 2) + 1 shadow ray per
pixel generated on the device -> any-hit.  Prints stage times and Mrays/s (device-resident, no host round trip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import workloads
torch.cuda.set_device(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
room, rn, light = workloads.closed_box(91)
inner, inn, _ = workloads.closed_box(6, side=2.0)
inner = inner + np.tile([4.0, 4.0, 0.0], 3)[None, :]
tris = np.concatenate([room, inner]); normals = np.concatenate([rn, -inn])
scene = rt.Scene(tris)
cam = rt.Camera.from_cam14(np.array([1.0, 1.0, 5.0, 6.0, 6.0, 1.0, 0, 0, 1, 1.0, -1.0, 1.0, -0.5625, 0.5625]), (1920, 1080))
n = cam.x_res * cam.y_res
st = torch.cuda.current_stream().cuda_stream
rays = torch.zeros(n * 64, dtype=torch.uint8, device="cuda"); hits = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
nrm = torch.from_numpy(np.ascontiguousarray(normals)).cuda()
ao = torch.zeros(n * spp * 64, dtype=torch.uint8, device="cuda"); occ = torch.zeros(n * spp, dtype=torch.uint8, device="cuda")
sh = torch.zeros(n * 64, dtype=torch.uint8, device="cuda"); socc = torch.zeros(n, dtype=torch.uint8, device="cuda")
frac = torch.zeros(n, dtype=torch.float64, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
def frame(record):
    if record: ev[0].record()
    rt.primary_rays_dev(cam, rays.data_ptr(), stream=st)
    scene.cast_dev(rays.data_ptr(), n, hits.data_ptr(), stream=st)
    if record: ev[1].record()
    scene.ao_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, ao.data_ptr(), st)
    if record: ev[2].record()
    scene.occluded_dev(ao.data_ptr(), n * spp, occ.data_ptr(), st)
    if record: ev[3].record()
    scene.shadow_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, light, sh.data_ptr(), st)
    scene.occluded_dev(sh.data_ptr(), n, socc.data_ptr(), st)
    if record: ev[4].record()
    rt.occlusion_fraction_dev(occ.data_ptr(), n, spp, frac.data_ptr(), st)
    if record: ev[5].record()
frame(False); torch.cuda.synchronize()
frame(True); torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
print("C4 room %d tris, %dx%d, %d AO spp: primary %.2f ms (%.0f Mrays/s) | AO gen %.2f ms | AO any-hit %.2f ms (%.0f Mrays/s) | shadow gen+any-hit %.2f ms | reduce %.2f ms | frame %.2f ms = %.0f Mrays/s overall; AO mean %.3f shadowed %.3f" % (
    len(tris), cam.x_res, cam.y_res, spp, t[0], n / t[0] / 1e3, t[1], t[2], n * spp / t[2] / 1e3, t[3], t[4], sum(t),
    n * (spp + 2) / sum(t) / 1e3, float(frac.mean()), float(socc.float().mean())), flush=True)
# the same AO bytes from the fused launch (rays built in the any-hit kernel's refill)
occ2 = torch.zeros_like(occ)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, occ2.data_ptr(), st); torch.cuda.synchronize()
e0.record()
scene.ao_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, spp, 2.0, 1, occ2.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
tf = e0.elapsed_time(e1)
print("   fused AO (rtmi_ao_occlusion_dev): %.2f ms against %.2f ms for generate + any-hit (%.0f Mrays/s); same bytes: %s; frame %.2f ms" % (
    tf, t[1] + t[2], n * spp / tf / 1e3, bool(torch.equal(occ, occ2)), sum(t) - t[1] - t[2] + tf), flush=True)
socc2 = torch.zeros_like(socc)
def sh2(): scene.shadow_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, light, sh.data_ptr(), st); scene.occluded_dev(sh.data_ptr(), n, socc.data_ptr(), st)
def sh1(): scene.shadow_occlusion_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), n, light, socc2.data_ptr(), st)
for f in (sh2, sh1):
    f(); torch.cuda.synchronize(); e0.record(); f(); e1.record(); torch.cuda.synchronize()
    print("   hard shadows, %d rays, %s: %.3f ms" % (n, "generate + any-hit" if f is sh2 else "fused (rtmi_shadow_occlusion_dev)", e0.elapsed_time(e1)), flush=True)
print("   same bytes: %s" % bool(torch.equal(socc, socc2)), flush=True)
# soft shadows (BASELINE config 4 names them; the shipped Shade has the call commented out): 100 rays per hit, 3 steps vs fused
NP = 100
m = n // 4                                         # a quarter of the frame: the 3-step path needs 88 B per ray
srays = torch.zeros(m * NP * 64, dtype=torch.uint8, device="cuda"); shits = torch.zeros(m * NP * 24, dtype=torch.uint8, device="cuda")
fac = torch.zeros(m, dtype=torch.float64, device="cuda"); fac2 = torch.zeros_like(fac); contrib = torch.zeros(m * NP, dtype=torch.uint8, device="cuda")
def three():
    scene.soft_shadow_rays_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), m, light, NP, srays.data_ptr(), st)
    scene.cast_dev(srays.data_ptr(), m * NP, shits.data_ptr(), stream=st)
    rt.soft_shadow_factor_dev(hits.data_ptr(), shits.data_ptr(), m, NP, fac.data_ptr(), st)
def fused():
    scene.soft_shadow_dev(nrm.data_ptr(), rays.data_ptr(), hits.data_ptr(), m, light, NP, contrib.data_ptr(), fac2.data_ptr(), st)
for f in (three, fused):
    f(); torch.cuda.synchronize(); e0.record(); f(); e1.record(); torch.cuda.synchronize()
    print("   soft shadows, %d hits x %d rays, %s: %.2f ms (%.0f Mrays/s)" % (m, NP, f.__name__, e0.elapsed_time(e1), m * NP / e0.elapsed_time(e1) / 1e3), flush=True)
print("   same factors: %s; mean factor %.3f" % (bool(torch.equal(fac, fac2)), float(fac.mean())), flush=True)
