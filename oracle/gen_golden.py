#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (dev container only).

Runs oracle/_ref/ref_harness (built by `make -C oracle ref` from the reference
sources where they lie under /root/reference) and packs its raw dumps into
compressed .npz fixtures.  Fixtures are data only: inputs and the reference's
outputs.  Also writes the small procedural OBJ/MTL scene the `obj` fixture is
rendered from (tests/golden/mesh_scene.obj/.mtl).

usage: python oracle/gen_golden.py
"""
import math
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")


def run(mode_args, tmp):
    out = subprocess.run([HARNESS, tmp] + [str(a) for a in mode_args], check=True, capture_output=True, text=True).stdout
    arrays, info = {}, []
    for line in out.splitlines():
        if line.startswith("FIXTURE "):
            _, name, dtype, count = line.split()
            arrays[name] = np.fromfile(os.path.join(tmp, name + ".bin"), dtype="<" + dtype, count=int(count))
        elif line.startswith("INFO "):
            info.append(line)
    return arrays, info


def icosphere(level):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [tuple(c / math.sqrt(1 + t * t) for c in p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = [(v[a][k] + v[b][k]) / 2 for k in range(3)]
                l = math.sqrt(sum(c * c for c in m))
                v.append(tuple(c / l for c in m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return v, f


def write_obj_scene(path_obj, path_mtl):
    """Small procedural mesh scene: smooth icosphere (vn+vt), flat icosahedron (no vn: exercises the loader's
    face-normal fallback, loader.hpp:104-110), glass tetrahedron (Tf/Ni), floor, emissive quad (Ke)."""
    with open(path_mtl, "w") as m:
        m.write("newmtl floor\nKd 0.70 0.70 0.65\n\nnewmtl ball\nKd 0.80 0.30 0.20\n\n")
        m.write("newmtl shiny\nKd 0.50 0.50 0.55\nKs 0.80 0.80 0.80\nNs 50.0\n\n")
        m.write("newmtl glass\nKd 0.0 0.0 0.0\nTf 0.9 0.9 0.9\nNi 1.5\n\n")
        m.write("newmtl lamp\nKd 0.0 0.0 0.0\nKe 8.0 8.0 8.0\n")
    lines = ["mtllib " + os.path.basename(path_mtl)]
    nv = nvt = nvn = 0
    # smooth ball
    v, f = icosphere(2)
    lines += ["o ball", "usemtl ball"]
    for p in v:
        lines.append("v %.6f %.6f %.6f" % (0.7 * p[0] - 0.6, 0.7 * p[1] + 0.7, 0.7 * p[2]))
        lines.append("vn %.6f %.6f %.6f" % p)
        lines.append("vt %.6f %.6f" % (0.5 + math.atan2(p[2], p[0]) / (2 * math.pi), 0.5 + math.asin(max(-1, min(1, p[1]))) / math.pi))
    for a, b, c in f:
        lines.append("f %d/%d/%d %d/%d/%d %d/%d/%d" % (a + 1, a + 1, a + 1, b + 1, b + 1, b + 1, c + 1, c + 1, c + 1))
    nv += len(v); nvt += len(v); nvn += len(v)
    # flat icosahedron, no normals / uvs
    v2, f2 = icosphere(0)
    lines += ["o gem", "usemtl shiny"]
    for p in v2:
        lines.append("v %.6f %.6f %.6f" % (0.45 * p[0] + 0.9, 0.45 * p[1] + 0.45, 0.45 * p[2] + 0.6))
    for a, b, c in f2:
        lines.append("f %d %d %d" % (nv + a + 1, nv + b + 1, nv + c + 1))
    nv += len(v2)
    # glass tetrahedron
    tet = [(0.2, 0.0, 1.2), (0.8, 0.0, 1.6), (0.1, 0.0, 1.9), (0.4, 0.7, 1.55)]
    lines += ["o tet", "usemtl glass"]
    for p in tet:
        lines.append("v %.6f %.6f %.6f" % p)
    for a, b, c in [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]:
        lines.append("f %d %d %d" % (nv + a + 1, nv + b + 1, nv + c + 1))
    nv += 4
    # floor (a quad: tinyobj triangulates it) with normals
    lines += ["o floor", "usemtl floor"]
    for p in [(-3, 0, -3), (-3, 0, 3), (3, 0, 3), (3, 0, -3)]:
        lines.append("v %.6f %.6f %.6f" % p)
    lines.append("vn 0 1 0")
    nvn += 1
    lines.append("f %d//%d %d//%d %d//%d %d//%d" % (nv + 1, nvn, nv + 2, nvn, nv + 3, nvn, nv + 4, nvn))
    nv += 4
    # lamp: two triangles facing down
    lines += ["o lamp", "usemtl lamp"]
    for p in [(-0.8, 3.0, -0.8), (0.8, 3.0, -0.8), (0.8, 3.0, 0.8), (-0.8, 3.0, 0.8)]:
        lines.append("v %.6f %.6f %.6f" % p)
    lines.append("vn 0 -1 0")
    nvn += 1
    lines.append("f %d//%d %d//%d %d//%d" % (nv + 1, nvn, nv + 2, nvn, nv + 3, nvn))
    lines.append("f %d//%d %d//%d %d//%d" % (nv + 1, nvn, nv + 3, nvn, nv + 4, nvn))
    with open(path_obj, "w") as o:
        o.write("\n".join(lines) + "\n")


def write_ngon_scene(path_obj, path_mtl):
    """Polygons with more than four corners, which tinyobjloader ear-clips (tiny_obj_loader.hpp:1715-1936, the built-in method: the
    reference does not define TINYOBJLOADER_USE_MAPBOX_EARCUT): a concave pentagon and a concave octagon in axis-aligned planes, a
    concave hexagon (chevron) in a tilted plane, a convex hexagon with normals and uvs, a 7-gon whose first corner is reflex (the
    clipper has to skip it), over a floor quad and under an emissive pentagon (five corners: lights come out of the clipper too)."""
    with open(path_mtl, "w") as m:
        m.write("newmtl floor\nKd 0.70 0.70 0.65\n\nnewmtl red\nKd 0.80 0.25 0.20\n\nnewmtl blue\nKd 0.20 0.30 0.80\nKs 0.5 0.5 0.5\nNs 30.0\n\n")
        m.write("newmtl green\nKd 0.25 0.70 0.30\n\nnewmtl lamp\nKd 0.0 0.0 0.0\nKe 9.0 9.0 9.0\n")
    lines = ["mtllib " + os.path.basename(path_mtl)]
    nv = nvt = nvn = 0

    def poly(name, mtl, pts, normal=None, uvs=None):
        nonlocal nv, nvt, nvn
        lines.extend(["o " + name, "usemtl " + mtl])
        for q in pts:
            lines.append("v %.6f %.6f %.6f" % tuple(q))
        if uvs:
            for q in uvs:
                lines.append("vt %.6f %.6f" % tuple(q))
        if normal:
            lines.append("vn %.6f %.6f %.6f" % tuple(normal))
            nvn += 1
        toks = []
        for k in range(len(pts)):
            v = nv + k + 1
            if uvs and normal:
                toks.append("%d/%d/%d" % (v, nvt + k + 1, nvn))
            elif normal:
                toks.append("%d//%d" % (v, nvn))
            else:
                toks.append("%d" % v)
        lines.append("f " + " ".join(toks))
        nv += len(pts)
        if uvs:
            nvt += len(uvs)

    # convex hexagon with a normal and uvs, plane y = 0.6.  First on purpose: the reference loader's scratch vertices are malloc'ed and
    # uninitialised (loader.hpp:68-110), so a FIRST face without normals takes its fallback normal from garbage (undefined behaviour)
    hexa = [(math.cos(k * math.pi / 3), math.sin(k * math.pi / 3)) for k in range(6)]
    poly("hex", "green", [(1.6 + 0.5 * x, 0.6, 0.4 + 0.5 * z) for x, z in hexa], normal=(0, 1, 0), uvs=[(0.5 + 0.5 * x, 0.5 + 0.5 * z) for x, z in hexa])
    # concave pentagon (a house with its roof pushed in), plane z = -1.2, no normals
    poly("house", "red", [(-1.6, 0.1, -1.2), (-0.4, 0.1, -1.2), (-0.4, 1.3, -1.2), (-1.0, 0.7, -1.2), (-1.6, 1.3, -1.2)])
    # concave hexagon (chevron) in a tilted plane: points (u, v) -> origin + u * a + v * b
    a, b, o = (0.8, 0.1, 0.6), (-0.1, 0.98, 0.15), (0.2, 0.2, -0.9)
    chev = [(0.0, 0.0), (0.6, 0.35), (1.2, 0.0), (1.2, 0.9), (0.6, 1.25), (0.0, 0.9)]
    poly("chevron", "blue", [tuple(o[i] + u * a[i] + v * b[i] for i in range(3)) for u, v in chev])
    # concave octagon (an L), plane x = -2.2, with a normal only
    ell = [(0.0, -1.0), (0.0, 0.6), (0.5, 0.6), (0.5, -0.4), (1.4, -0.4), (1.4, -1.0), (0.9, -1.0), (0.4, -1.0)]
    poly("ell", "red", [(-2.2, 0.1 + y, z) for y, z in ell], normal=(1, 0, 0))
    # 7-gon whose FIRST corner is reflex, plane z = 1.0 facing the camera
    seven = [(0.0, 0.5), (-0.7, 0.1), (-0.7, 1.2), (0.0, 1.5), (0.7, 1.2), (0.7, 0.1), (0.3, 0.1)]
    poly("seven", "blue", [(-0.3 + x, y, 1.0) for x, y in seven])
    # floor quad
    poly("floor", "floor", [(-3, 0, -3), (-3, 0, 3), (3, 0, 3), (3, 0, -3)], normal=(0, 1, 0))
    # emissive pentagon facing down
    pent = [(math.cos(2 * math.pi * k / 5), math.sin(2 * math.pi * k / 5)) for k in range(5)]
    poly("lamp", "lamp", [(0.9 * x, 3.0, 0.9 * z) for x, z in pent], normal=(0, -1, 0))
    with open(path_obj, "w") as f:
        f.write("\n".join(lines) + "\n")


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build the reference harness first: make -C oracle ref")
    os.makedirs(GOLD, exist_ok=True)
    obj = os.path.join(GOLD, "mesh_scene.obj")
    write_obj_scene(obj, os.path.join(GOLD, "mesh_scene.mtl"))
    ngon = os.path.join(GOLD, "ngon_scene.obj")
    write_ngon_scene(ngon, os.path.join(GOLD, "ngon_scene.mtl"))
    jobs = {
        "rand": (["rand"], {}),
        "units": (["units"], {}),
        # BASELINE config-0 scene at two glibc seeds; small images so the fixture stays small
        "c0_seed1": (["c0", 1, 72, 48, 8, 50], dict(seed=1, width=72, height=48, spp=8, depth=50)),
        "c0_seed7": (["c0", 7, 96, 64, 4, 50], dict(seed=7, width=96, height=64, spp=4, depth=50)),
        "mixed": (["mixed", 3, 64, 48, 16, 20], dict(seed=3, width=64, height=48, spp=16, depth=20)),
        "obj": (["obj", obj, 5, 64, 48, 8, 12], dict(seed=5, width=64, height=48, spp=8, depth=12)),
        # polygons with more than four corners through tinyobj's ear clipping (arrays keep the harness's "obj_" prefix)
        "ngon": (["obj", ngon, 6, 64, 48, 8, 12], dict(seed=6, width=64, height=48, spp=8, depth=12)),
        # the reference's own main() scene (raytracing.cpp:77-90: Cornell box, depth 10) at a small size
        "cornell": (["cornell", 1, 56, 56, 16, 10], dict(seed=1, width=56, height=56, spp=16, depth=10)),
        # volume / fog: sphere-bounded media of all_feature_test() (demo.hpp:117-134) over a ground rect
        "smoke": (["smoke", 2, 64, 48, 16, 12], dict(seed=2, width=64, height=48, spp=16, depth=12)),
        # all_feature_test() (demo.hpp:94-143): 400 boxes, media, a failed image texture, translate(rotatey(bvh(1000 spheres)))
        "features": (["features", 4, 48, 32, 4, 10], dict(seed=4, width=48, height=32, spp=4, depth=10)),
    }
    only = sys.argv[1:]          # optional: names of the fixtures to (re)generate
    for name, (args, meta) in jobs.items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            arrays, info = run(args, tmp)
        for k, v in meta.items():
            arrays["meta_" + k] = np.array(v)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrays)
        print(name, {k: v.shape for k, v in arrays.items() if not k.startswith("meta_")})
        for line in info:
            print("   ", line)


if __name__ == "__main__":
    main()
