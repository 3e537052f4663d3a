#!/usr/bin/env python3
"""Writes a scene (<dir>/<stem>.sdf + <dir>/<stem>_mesh.obj) of the kind the reference's own reader cannot take
(Render2/scene.cpp:211, :400-405, Render2/params.cpp:214 — SURVEY.md §8f rank 3) and rtmi_scene_file_load_ex(...,
RTMI_SCENE_LENIENT) can: a k x k terrain whose faces are a mix of triangles, quads and pentagons, some written with vertex
normals (a//c), some with relative (negative) indices, in groups (`g`), under an `o` line that is NOT the file's name, with a
comment of more than 512 bytes and one 96-cornered face (a line of about 700 bytes).

    python tools/gen_obj.py <dir> <stem> [k=320]      k = 320: 122 881 faces (102 400 cells + the disc) = 225 374 triangles

generate(dir, stem, k) also returns the triangles the file stands for (n x 9, the reader's order and its quad / fan
split), so tests can compare the reader's output with them exactly."""
import os
import sys

import numpy as np


def height(x, y):
    return 0.6 * np.sin(0.9 * x) * np.cos(0.7 * y) + 0.25 * np.sin(2.3 * x + 1.0) + 0.15 * np.cos(3.1 * y)


def generate(out_dir, stem, k=320, size=16.0):
    os.makedirs(out_dir, exist_ok=True)
    g = np.linspace(-size / 2, size / 2, k + 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    Z = height(X, Y)
    # a centre vertex per cell for the pentagon cells
    xc = 0.5 * (g[:-1] + g[1:])
    XC, YC = np.meshgrid(xc, xc, indexing="ij")
    ZC = height(XC, YC) + 0.02
    verts = []          # as written (1-based order)
    lines = ["o some_other_name", "# " + "x" * 600, "mtllib nothing.mtl"]

    def vid(i, j):
        return i * (k + 1) + j + 1
    for i in range(k + 1):
        for j in range(k + 1):
            verts.append((X[i, j], Y[i, j], Z[i, j]))
            lines.append("v %r %r %r" % (float(X[i, j]), float(Y[i, j]), float(Z[i, j])))
    n_grid = len(verts)
    lines.append("vn 0.0 0.0 1.0")
    tris = []
    P = lambda idx: verts[idx - 1]

    def emit(a, b, c):
        tris.append(P(a) + P(b) + P(c))

    def face(idx, style):
        """idx: 1-based vertex ids; style 0 plain, 1 with the normal (a//1), 2 relative indices"""
        n_now = len(verts)
        if style == 1:
            lines.append("f " + " ".join("%d//1" % a for a in idx))
        elif style == 2:
            lines.append("f " + " ".join("%d" % (a - n_now - 1) for a in idx))
        else:
            lines.append("f " + " ".join("%d" % a for a in idx))
        emit(idx[0], idx[1], idx[2])                       # the reader's split: (v1,v2,v3), then (v_{m-1}, v_m, v1)
        for m in range(3, len(idx)):
            emit(idx[m - 1], idx[m], idx[0])
    for i in range(k):
        if i % 64 == 0:
            lines.append("g band%d" % (i // 64))
        for j in range(k):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            kind = (i * 7 + j * 3) % 5
            style = (i + j) % 3
            if kind == 0:                                    # two triangles
                face([a, b, c], style)
                face([a, c, d], style)
            elif kind == 1:                                  # a pentagon: the cell with an extra vertex on one edge
                verts.append((float(XC[i, j]), float(Y[i, j]), float(height(XC[i, j], Y[i, j]))))
                lines.append("v %r %r %r" % verts[-1])
                face([a, len(verts), b, c, d], style)
            else:                                            # a quad
                face([a, b, c, d], style)
    # one face with 96 corners: a disc above the terrain (its line is longer than the reference's 512-byte buffer)
    first = len(verts) + 1
    for t in range(96):
        ang = 2 * np.pi * t / 96
        verts.append((float(1.5 * np.cos(ang)), float(1.5 * np.sin(ang)), 2.0))
        lines.append("v %r %r %r" % verts[-1])
    face(list(range(first, first + 96)), 0)
    with open(os.path.join(out_dir, stem + "_mesh.obj"), "w") as f:
        f.write("\n".join(lines) + "\n")
    sdf = """# written by tools/gen_obj.py
eye (9.0, -11.0, 7.5)
lookat (0.0, 0.0, 0.0)
up (0.0, 0.0, 1.0)
vpdist 2.75
x_win (-1.6, 1.6)
y_win (-0.9, 0.9)
x_res 960
y_res 540
emission [1.0, 1.0, 1.0]
spherelight (3.0, -2.0, 12.0) 0.5
emission [0.0, 0.0, 0.0]
diffuse [0.55, 0.6, 0.4]
specular [0.2, 0.2, 0.2]
ambient [0.1, 0.1, 0.1]
Phong_exp 6.0
object %s_mesh
end
""" % stem
    with open(os.path.join(out_dir, stem + ".sdf"), "w") as f:
        f.write(sdf)
    return np.array(tris, np.float64)


if __name__ == "__main__":
    t = generate(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 320)
    print("%d triangles" % len(t))
