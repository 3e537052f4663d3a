"""TEST INFRASTRUCTURE.  For rtmi_scene_file_load (the restated Scene::BuildScene, Render2/scene.cpp:209-460):
copies the reference's small scene DATA files (.sdf / .obj of eight scenes; data, not source) to tests/golden/scenes/
and writes tests/golden/g7_scene_files.npz = what the compiled reference's own BuildScene makes of each of them:
triangles, object ids, triNormals, materials, lights (Center(GetBox), emission) and camera.
Run in this container only:   python oracle/gen_golden_scenes.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from oracle_py import Reference  # noqa: E402

SRC = "/root/reference/Render2/scenes"
DST = os.path.join(ROOT, "tests", "golden", "scenes")
STEMS = ["test1", "test2", "test5", "test6", "test7", "scene_room_bed_sample", "radiosity_test", "practice_room_enclosed"]
ref = Reference()
os.makedirs(DST, exist_ok=True)
out = {}
work = tempfile.mkdtemp()
shutil.copytree(SRC, os.path.join(work, "scenes"), ignore=lambda d, names: [n for n in names if os.path.isdir(os.path.join(d, n))])
for stem in STEMS:
    h = ref.scene_load(work, stem)
    tris, obj, nrm = ref.triangles(h)
    cam14, res = ref.camera(h)
    out[stem + "/tris"] = tris; out[stem + "/tri_obj"] = obj; out[stem + "/normals"] = nrm
    out[stem + "/mats"] = ref.materials(h); out[stem + "/lights"] = ref.lights(h)
    out[stem + "/cam14"] = cam14; out[stem + "/res"] = res
    shutil.copy(os.path.join(SRC, stem + ".sdf"), DST)
    with open(os.path.join(SRC, stem + ".sdf")) as f:
        for line in f:
            w = line.split()
            if len(w) >= 2 and w[0] == "object":
                shutil.copy(os.path.join(SRC, w[1] + ".obj"), DST)
    print("%-24s %5d triangles, %d objects, %d lights" % (stem, len(tris), len(out[stem + "/mats"]), len(out[stem + "/lights"])))
for f in os.listdir(DST):
    os.chmod(os.path.join(DST, f), 0o644)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g7_scene_files.npz"), **out)
shutil.rmtree(work, ignore_errors=True)
