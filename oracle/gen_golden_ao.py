"""TEST INFRASTRUCTURE.  Writes tests/golden/g3_ao_room.npz: the compiled reference's own
Shader::findOcclusionNew (shader.cpp:140-232) at 48 hit points of scene_room_bed_sample, 4000 rays each (its RNG is unseeded, so
the fixture is a Monte-Carlo mean; the tests compare within 5 standard errors).  Run in this container only:
    python oracle/gen_golden_ao.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_py import Oracle, Reference  # noqa: E402

ROOT = os.path.dirname(HERE)
g = np.load(os.path.join(ROOT, "tests", "golden", "g2_scene_room_bed_sample.npz"))
ref, ora = Reference(), Oracle()
work = tempfile.mkdtemp()
os.makedirs(os.path.join(work, "scenes"), exist_ok=True)
for attempt in range(20):
    h = ref.scene_from_triangles(g["tris"], g["tri_obj"], g["light_center"])
    if ref.build_bsp(h, work) == 0:
        break
else:
    raise SystemExit("the reference's BSP build kept throwing")
sel = np.linspace(0, len(g["cast_point_hits"]) - 1, 48).astype(int)
P = g["cast_point_hits"][sel]
N = g["cast_normal_hits"][sel]
NR = 4000
ao = ref.ao_occlusion(h, P, N, 2.0, NR)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g3_ao_room.npz"), P=P, N=N, radius=2.0, nrays=NR, ref_ao=ao)
print("reference AO: mean %.4f min %.4f max %.4f" % (ao.mean(), ao.min(), ao.max()))
# what the restatement gives on seeded square points
o, d = ora.ao_rays(P, N, 2.0, NR, 7)
occ = ora.any_hit(g["tris"], o, d, np.full(len(o), 2.0)).reshape(len(P), NR).mean(1)
print("oracle    AO: mean %.4f ; max |diff| %.4f" % ((1 - occ).mean(), np.abs((1 - occ) - ao).max()))
shutil.rmtree(work, ignore_errors=True)
