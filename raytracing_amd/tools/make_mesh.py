#!/usr/bin/env python3
"""Procedural OBJ/MTL workload for BASELINE configs[2] (no mesh assets ship with the reference, and none can be fetched).

A displaced icosphere (level 8 -> 1 310 720 triangles, smooth normals + uvs) standing on a floor quad under an emissive
quad, with diffuse / specular / transmissive MTL materials, written as plain OBJ text and then read back through the
host loader like any other asset.   python -m raytracing_amd.tools.make_mesh out.obj [--level 8]
"""
import argparse
import os

import numpy as np


def icosphere(level):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                  (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.sort(e, axis=1)
        uniq, inv = np.unique(key[:, 0] * (len(v) + 1) + key[:, 1], return_inverse=True)
        a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid = v[a] + v[b]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = inv.reshape(3, -1).T + len(v)      # per face: midpoints of edges 01, 12, 20
        v = np.concatenate([v, mid])
        f = np.concatenate([np.stack([f[:, 0], m[:, 0], m[:, 2]], 1), np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
                            np.stack([f[:, 2], m[:, 2], m[:, 1]], 1), np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)])
    return v, f


def write_scene(path, level=8, seed=12345):
    rng = np.random.default_rng(seed)
    v, f = icosphere(level)
    # smooth low-frequency displacement so the silhouette is not a sphere
    k = rng.normal(size=(6, 3)) * 2.5
    ph = rng.uniform(0, 6.28, size=6)
    disp = 1.0 + 0.08 * sum(np.sin(v @ k[i] + ph[i]) for i in range(6)) / 6 ** 0.5
    p = v * disp[:, None] * 0.9 + np.array([0.0, 1.05, 0.0])
    # vertex normals from face normals (area weighted)
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    n = np.zeros_like(p)
    for c in range(3):
        np.add.at(n, f[:, c], fn)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    uv = np.stack([0.5 + np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi), 0.5 + np.arcsin(np.clip(v[:, 1], -1, 1)) / np.pi], 1)
    mtl = os.path.splitext(path)[0] + ".mtl"
    with open(mtl, "w") as m:
        m.write("newmtl clay\nKd 0.75 0.45 0.30\n\nnewmtl glaze\nKd 0.30 0.45 0.70\nKs 0.60 0.60 0.60\nNs 80.0\n\n"
                "newmtl floor\nKd 0.70 0.70 0.65\n\nnewmtl lamp\nKd 0.0 0.0 0.0\nKe 12.0 12.0 12.0\n")
    nv = len(p)
    # faces with +y centroid in one band get the glossy material: split into two usemtl groups
    band = np.sin(6.0 * p[f].mean(axis=1)[:, 1]) > 0.6
    with open(path, "w") as o:
        o.write("mtllib %s\no blob\n" % os.path.basename(mtl))
        np.savetxt(o, p, fmt="v %.6f %.6f %.6f")
        np.savetxt(o, n, fmt="vn %.6f %.6f %.6f")
        np.savetxt(o, uv, fmt="vt %.6f %.6f")
        for name, sel in (("clay", ~band), ("glaze", band)):
            o.write("usemtl %s\n" % name)
            g = f[sel] + 1
            np.savetxt(o, np.stack([g[:, 0], g[:, 0], g[:, 0], g[:, 1], g[:, 1], g[:, 1], g[:, 2], g[:, 2], g[:, 2]], 1),
                       fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
        o.write("o floor\nusemtl floor\n")
        for q in [(-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4)]:
            o.write("v %.6f %.6f %.6f\n" % q)
        o.write("vn 0 1 0\n")
        o.write("f %d//%d %d//%d %d//%d %d//%d\n" % (nv + 1, nv + 1, nv + 2, nv + 1, nv + 3, nv + 1, nv + 4, nv + 1))
        o.write("o lamp\nusemtl lamp\n")
        for q in [(-1.0, 3.6, -1.0), (1.0, 3.6, -1.0), (1.0, 3.6, 1.0), (-1.0, 3.6, 1.0)]:
            o.write("v %.6f %.6f %.6f\n" % q)
        o.write("vn 0 -1 0\n")
        o.write("f %d//%d %d//%d %d//%d\nf %d//%d %d//%d %d//%d\n" % (nv + 5, nv + 2, nv + 6, nv + 2, nv + 7, nv + 2, nv + 5, nv + 2, nv + 7, nv + 2, nv + 8, nv + 2))
    return len(f) + 4


def _write_png(path, rgb):
    """8-bit RGB PNG (zlib + CRC only: no imaging library in the image)."""
    import struct
    import zlib
    h, w, _ = rgb.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgb.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


def write_soup(path, n_tri=10_000_000, n_mats=64, tex_size=1024, extent=50.0, edge=0.2, seed=12345):
    """BASELINE configs[3]: a triangle soup (uniform random centres in [-extent, extent]^3, edge ~`edge`) with `n_mats` MTL
    materials, each with its own procedural `tex_size`^2 PNG map_Kd, per-vertex uvs, under one emissive quad.  Incoherent,
    far larger than the caches: the node / primitive / texel fetch stress case."""
    rng = np.random.default_rng(seed)
    d = os.path.dirname(os.path.abspath(path))
    base = os.path.splitext(os.path.basename(path))[0]
    yy, xx = np.mgrid[0:tex_size, 0:tex_size]
    with open(os.path.splitext(path)[0] + ".mtl", "w") as m:
        for k in range(n_mats):
            img = np.stack([(xx * (k + 3) // 7 + yy) & 255, (yy * (k + 5) // 9 + 3 * k) & 255, ((xx ^ yy) + 11 * k) & 255], axis=2).astype(np.uint8)
            _write_png(os.path.join(d, f"{base}_t{k}.png"), img)
            m.write(f"newmtl m{k}\nKd 0.7 0.7 0.7\nmap_Kd {base}_t{k}.png\n\n")
        m.write("newmtl lamp\nKd 0.0 0.0 0.0\nKe 40.0 40.0 40.0\n")
    c = rng.uniform(-extent, extent, size=(n_tri, 3))
    p = c[:, None, :] + rng.normal(scale=edge * 0.5, size=(n_tri, 3, 3))
    uv = rng.uniform(0, 1, size=(n_tri * 3, 2))
    per = n_tri // n_mats
    with open(path, "w") as o:
        o.write(f"mtllib {base}.mtl\no soup\n")
        np.savetxt(o, p.reshape(-1, 3), fmt="v %.5f %.5f %.5f")
        np.savetxt(o, uv, fmt="vt %.5f %.5f")
        idx = np.arange(1, n_tri * 3 + 1).reshape(-1, 3)
        for k in range(n_mats):
            g = idx[k * per:(k + 1) * per if k < n_mats - 1 else n_tri]
            o.write(f"usemtl m{k}\n")
            np.savetxt(o, np.stack([g[:, 0], g[:, 0], g[:, 1], g[:, 1], g[:, 2], g[:, 2]], 1), fmt="f %d/%d %d/%d %d/%d")
        nv = n_tri * 3
        e = extent * 1.2
        o.write("o lamp\nusemtl lamp\n")
        for q in [(-e, extent * 1.3, -e), (e, extent * 1.3, -e), (e, extent * 1.3, e), (-e, extent * 1.3, e)]:
            o.write("v %.5f %.5f %.5f\n" % q)
        o.write("vn 0 -1 0\n")
        o.write("f %d//1 %d//1 %d//1\nf %d//1 %d//1 %d//1\n" % (nv + 1, nv + 2, nv + 3, nv + 1, nv + 3, nv + 4))
    return n_tri + 2


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--level", type=int, default=8)
    a = ap.parse_args()
    print(write_scene(a.out, a.level), "triangles ->", a.out)
