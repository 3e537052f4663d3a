import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C restatement (oracle/liboracle.so) — the checker, never the product."""
    from oracle_py import Oracle, ORACLE_SO
    if not os.path.exists(ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return Oracle()


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))


def frame_rays(oracle, cam14, res):
    """origins, directions, pixel list of a whole frame in the reference's pixel order (rasterizer.cpp:219-233)"""
    import numpy as np
    rd = oracle.raster_details(cam14, int(res[0]), int(res[1]))
    pix = np.array([(i, j) for i in range(int(res[1])) for j in range(int(res[0]))], np.uint32)
    d = oracle.primary_dirs(rd, pix)
    o = np.tile(np.asarray(cam14[:3], np.float64), (len(d), 1))
    return o, d, pix


def ao_steps(ref_rgb, occ1_rgb, tris, tri_obj, mats, lights, oracle, o, d):
    """The reference draws its ambient-occlusion rays from an unseeded generator (shader.cpp:148-150), so a colour
    with ambient != 0 is  c(occ = 1) − (k/N)·0.85·ambient·Σ emission  for the random count k of occluded rays
    (shader.cpp:231, :494).  Returns k·(10/N)... as a float per ray and channel (NaN where the ambient term is 0 or the
    pixel is shadowed / a miss, i.e. where the two colours must be equal)."""
    import numpy as np
    tri, _, _, _ = oracle.closest_hit(tris, o, d)
    amb = np.where(tri[:, None] >= 0, mats[tri_obj[np.maximum(tri, 0)], 6:9], 0.0)
    B = 0.85 * amb * lights[:, 3:6].sum(0)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(B != 0, (occ1_rgb - ref_rgb) / B, np.nan)
    return k
