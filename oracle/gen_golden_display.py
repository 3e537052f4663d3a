"""TEST INFRASTRUCTURE.  Writes tests/golden/g8_radiosity_display.npz: the compiled reference's own
Radiosity_Helper::trace_ray (radiosity_helper.cpp:318-365, incl. returnQuadTreeColor) and ToneMapRadiosity
(rasterizer.cpp:90-109) on the element lists of radiosity_test and practice_room_enclosed — the body of
Rasterizer::Radiosity_Raster's pixel loop — for every 4th pixel of every 4th row, with a known radiosity per element.
Also the per-vertex radiosities the reference's returnQuadTreeColor yields (evaluated at the vertices).
Run in this container only:   python oracle/gen_golden_display.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from oracle_py import Oracle, Reference  # noqa: E402

ref, ora = Reference(), Oracle()
work = tempfile.mkdtemp()
shutil.copytree("/root/reference/Render2/scenes", os.path.join(work, "scenes"),
                ignore=lambda d, names: [n for n in names if os.path.isdir(os.path.join(d, n))])
out = {}
for stem in ("radiosity_test", "practice_room_enclosed"):
    for attempt in range(20):
        h = ref.scene_load(work, stem)
        if ref.build_bsp(h, work) == 0:
            break
    else:
        raise SystemExit("the reference's BSP build kept throwing")
    ne = ref.rad_setup(h)
    verts, nrm, obj = ref.rad_elements(h, ne)
    B = 0.05 + 4.0 * ((np.arange(ne) * 7919) % ne) / ne          # distinct, spread over the tone curve
    ref.rad_set_element_radiosity(h, B)
    vr = ref.rad_vertex_radiosity(h, ne)
    cam14, res = ref.camera(h)
    rd = ref.raster_details(h)
    pix = np.array([(i, j) for i in range(0, res[1], 4) for j in range(0, res[0], 4)], np.uint32)
    d = ref.primary_dirs(rd, pix)
    o = np.tile(cam14[:3], (len(d), 1))
    rgb = ref.rad_trace_ray(h, o, d)
    mine = ora.radiosity_trace(verts, nrm, vr, o, d)
    print("%-24s %3d elements, %5d rays: oracle differs on %d; green %d, black %d" % (
        stem, ne, len(d), (mine != rgb).any(1).sum(), (rgb == [0, 1, 0]).all(1).sum(), (rgb == 0).all(1).sum()))
    for k, v in dict(verts=verts, normals=nrm, B=B, vertex_rad=vr, cam14=cam14, res=res, pix=pix, rgb=rgb,
                     pixels=ref.tone_map_radiosity(rgb)).items():
        out[stem + "/" + k] = v
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g8_radiosity_display.npz"), **out)
shutil.rmtree(work, ignore_errors=True)
