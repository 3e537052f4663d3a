#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — writes tests/golden/c5_box_rows.json: CRC-32 of four rows (all patch columns + the light column, fp32)
of the form-factor matrix of the 64 744-patch box that bench.py builds and tests/test_gpu_full_size.py checks, computed by
the CPU oracle (oracle/rt_oracle.c through its own hierarchy, which tests/test_oracle_golden.py pins to its scan over all
triangles).  bench.py prints the CRCs of the rows it has just timed; the GPU test asserts them against this file, so the
bench line and the suite cannot drift apart.      python oracle/gen_golden_c5_rows.py      (about a minute on 8 cores)"""
import json, os, sys, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from oracle_py import Oracle
from raytracer_amd import workloads

ROWS = [0, 21000, 43210, 64743]
verts, normals, light = workloads.closed_box(74, hole=0.3)
n = len(verts)
ora = Oracle()
h = ora.bvh_build(verts)
out = {"scene": "workloads.closed_box(74, hole=0.3)", "patches": n, "samples": 10, "light_samples": 30, "seed": 1, "crc32": {}, "row_sum": {}}
for r in ROWS:
    row = np.zeros(n + 1, np.float32)
    row[:n] = ora.form_factors_bvh(verts, normals, 10, 1, row0=r, n_rows=1, bvh=h)[0].astype(np.float32)
    row[n] = np.float32(ora.light_terms(verts, normals, light, 30, 1, mirror_bug=1, row0=r, n_rows=1)[0])
    out["crc32"][str(r)] = "%08x" % zlib.crc32(row.tobytes())
    out["row_sum"][str(r)] = float(row[:n].astype(np.float64).sum())
    print(r, out["crc32"][str(r)], out["row_sum"][str(r)], flush=True)
ora.bvh_free(h)
json.dump(out, open(os.path.join(HERE, "..", "tests", "golden", "c5_box_rows.json"), "w"), indent=1)
