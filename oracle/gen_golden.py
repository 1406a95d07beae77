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


def test_pictures():
    """Deterministic pictures for the image-file fixture: a smooth colour field (what a texture mostly is), one with hard edges and
    saturated noise (coefficients that drive the inverse DCT into its clamps), and the small sizes whose planes are narrower
    than one block."""
    rng = np.random.default_rng(20260211)

    def field(w, h):
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        r = 127 + 120 * np.sin(x / 5.0 + y / 11.0)
        g = 127 + 120 * np.cos(x / 3.0) * np.sin(y / 4.0)
        b = (x * 7 + y * 13) % 256
        return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)

    def harsh(w, h):
        img = rng.integers(0, 2, (h, w, 3)) * 255
        img[h // 3: 2 * h // 3, w // 4: 3 * w // 4] = rng.integers(0, 256, (2 * h // 3 - h // 3, 3 * w // 4 - w // 4, 3))
        return img.astype(np.uint8)
    return field, harsh


def write_png_by_hand(path, arr, ctype, depth, interlaced):
    """PNG kinds Pillow does not write (Adam7 interlace, 16-bit RGB / RGBA): filter 0 rows through zlib."""
    import struct
    import zlib
    h, w, ch = arr.shape
    if depth == 16:                                   # low byte: something that must not show in the 8-bit result
        wide = np.stack([arr, (arr.astype(np.int32) * 37 % 256).astype(np.uint8)], -1).reshape(h, w, ch * 2)
    else:
        wide = arr
    raw = b""
    if interlaced:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = wide[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                for row in sub:
                    raw += b"\0" + row.tobytes()
    else:
        for row in wide:
            raw += b"\0" + row.tobytes()

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlaced else 0))
                + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def image_files(tmp):
    """(name, path) of the files of tests/golden/images.npz, written with Pillow (the fixture keeps the file bytes, so the
    tests do not depend on Pillow or on its version)."""
    from PIL import Image
    field, harsh = test_pictures()
    out = []

    def save(name, arr, mode=None, **kw):
        path = os.path.join(tmp, name)
        im = Image.fromarray(arr) if mode is None else Image.fromarray(arr).convert(mode)
        im.save(path, **kw)
        out.append((name, path))
        return path
    a, b = field(37, 29), harsh(50, 33)
    save("rgb444_q90.jpg", a, quality=90, subsampling=0)
    save("rgb422_q75.jpg", a, quality=75, subsampling=1)
    save("rgb420_q75.jpg", a, quality=75, subsampling=2)
    save("rgb420_q30_harsh.jpg", b, quality=30, subsampling=2)
    save("rgb444_q100_harsh.jpg", b, quality=100, subsampling=0)
    save("rgb420_q5_harsh.jpg", b, quality=5, subsampling=2)
    for sub in ("4:1:1", "4:4:0"):
        try:
            save("rgb%s_q80.jpg" % sub.replace(":", ""), a, quality=80, subsampling=sub)
        except Exception as e:                     # an older Pillow: the fixture simply lacks the case
            print("   (no %s: %s)" % (sub, e))
    save("rgb420_1x1.jpg", field(1, 1), quality=80, subsampling=2)
    save("rgb420_1x9.jpg", field(1, 9), quality=80, subsampling=2)
    save("rgb422_2x2.jpg", field(2, 2), quality=80, subsampling=1)
    save("rgb420_17x16.jpg", field(17, 16), quality=80, subsampling=2)
    save("rgb420_200x120.jpg", field(200, 120), quality=85, subsampling=2)
    save("grey_q85.jpg", a, mode="L", quality=85)
    save("grey_q20_harsh.jpg", b, mode="L", quality=20)
    save("prog444.jpg", a, quality=85, subsampling=0, progressive=True)
    save("prog420.jpg", field(61, 43), quality=70, subsampling=2, progressive=True)
    save("prog420_harsh.jpg", b, quality=40, subsampling=2, progressive=True)
    save("prog_grey.jpg", a, mode="L", quality=80, progressive=True)
    save("optimised420.jpg", a, quality=75, subsampling=2, optimize=True)
    try:
        save("restart420.jpg", field(61, 43), quality=75, subsampling=2, restart_marker_blocks=3)
        save("restart_rows444.jpg", a, quality=75, subsampling=0, restart_marker_rows=1)
        save("restart_prog420.jpg", field(61, 43), quality=75, subsampling=2, progressive=True, restart_marker_blocks=2)
    except Exception as e:
        print("   (no restart markers: %s)" % e)
    save("cmyk.jpg", a, mode="CMYK", quality=85)
    # components named 'R','G','B' (stb then skips the colour matrix, stb_image.hpp:3253,3825): patch the ids of a 4:4:4 file
    p = save("rgb_ids.jpg", a, quality=90, subsampling=0)
    raw = bytearray(open(p, "rb").read())
    sof = raw.index(b"\xff\xc0")
    sos = raw.index(b"\xff\xda")
    for k, ch in enumerate(b"RGB"):
        assert raw[sof + 10 + 3 * k] == k + 1 and raw[sos + 5 + 2 * k] == k + 1
        raw[sof + 10 + 3 * k] = ch
        raw[sos + 5 + 2 * k] = ch
    open(p, "wb").write(bytes(raw))
    # chroma at half height only (4:4:0), which Pillow does not write: a 48x16 4:2:2 file holds 3x2 MCUs of luma-luma-Cb-Cr
    # blocks; with the luma factors swapped to 1x2 and the size set to 21x27 the same entropy data is 3x2 MCUs of a 4:4:0 frame
    # (a scrambled picture, which is as good a test of the vertical chroma filter as any)
    p = save("rgb440_patched.jpg", field(48, 16), quality=80, subsampling=1)
    raw = bytearray(open(p, "rb").read())
    sof = raw.index(b"\xff\xc0")
    assert raw[sof + 11] == 0x21 and raw[sof + 5:sof + 9] == bytes([0, 16, 0, 48])
    raw[sof + 11] = 0x12
    raw[sof + 5:sof + 9] = bytes([0, 27, 0, 21])
    open(p, "wb").write(bytes(raw))
    # four components read as YCCK (Adobe transform 2) and, with the Adobe segment cut out, as YCbCr plus an ignored plane
    cmyk = bytearray(open(os.path.join(tmp, "cmyk.jpg"), "rb").read())
    app14 = cmyk.index(b"\xff\xeeAdobe".replace(b"Adobe", b"\x00\x0eAdobe"))
    assert cmyk[app14 + 15] == 0
    ycck = bytearray(cmyk)
    ycck[app14 + 15] = 2
    open(os.path.join(tmp, "ycck_patched.jpg"), "wb").write(bytes(ycck))
    out.append(("ycck_patched.jpg", os.path.join(tmp, "ycck_patched.jpg")))
    open(os.path.join(tmp, "four_planes_no_adobe.jpg"), "wb").write(bytes(cmyk[:app14] + cmyk[app14 + 16:]))
    out.append(("four_planes_no_adobe.jpg", os.path.join(tmp, "four_planes_no_adobe.jpg")))
    # PNG kinds the loader meets
    save("rgb.png", a)
    save("rgba.png", np.concatenate([a, a[:, :, :1]], -1))
    save("grey.png", a, mode="L")
    save("grey_alpha.png", np.concatenate([a[:, :, :1], a[:, :, 1:2]], -1))
    save("palette.png", a, mode="P")
    save("palette16.png", a, mode="P", bits=4)
    Image.fromarray(a).convert("P").save(os.path.join(tmp, "palette_trns.png"), transparency=3)
    out.append(("palette_trns.png", os.path.join(tmp, "palette_trns.png")))
    save("bilevel.png", (a[:, :, 0] > 127).astype(np.uint8) * 255, mode="1")
    save("rgb_trns.png", a, transparency=(int(a[0, 0, 0]), int(a[0, 0, 1]), int(a[0, 0, 2])))
    save("grey_trns.png", a, mode="L", transparency=int(np.array(Image.fromarray(a).convert("L"))[2, 2]))
    Image.fromarray((a[:, :, 0].astype(np.uint16) * 257 + 31)).save(os.path.join(tmp, "grey16.png"))
    out.append(("grey16.png", os.path.join(tmp, "grey16.png")))
    for name, arr, ctype, depth in (("interlaced_rgb.png", a, 2, 8), ("interlaced_grey.png", a[:, :, :1], 0, 8), ("rgb16.png", a, 2, 16), ("interlaced_rgba16.png", np.concatenate([a, a[:, :, :1]], -1), 6, 16)):
        path = os.path.join(tmp, name)
        write_png_by_hand(path, arr, ctype, depth, interlaced=name.startswith("interlaced"))
        out.append((name, path))
    # BMP, TGA, PGM / PPM: the other stb formats an MTL file may name
    import struct
    pal_img = Image.fromarray(a).convert("P")
    pal = np.zeros((256, 3), np.uint8)
    flat = np.array(pal_img.getpalette()[:768], np.uint8)
    pal.reshape(-1)[:flat.size] = flat
    idx = np.array(pal_img)
    grey = np.array(Image.fromarray(a).convert("L"))
    rgba = np.concatenate([a, a[:, :, 1:2]], -1)
    save("rgb.bmp", a)
    save("rgba.bmp", rgba)
    save("palette.bmp", a, mode="P")
    save("grey.bmp", a, mode="L")
    save("bilevel.bmp", (a[:, :, 0] > 127).astype(np.uint8) * 255, mode="1")

    def raw_file(name, data):
        path = os.path.join(tmp, name)
        open(path, "wb").write(data)
        out.append((name, path))

    def bmp(name, hsz, bpp, rows, palette=b"", compress=0, masks=b"", top_down=False):
        hh, ww = len(rows), None
        body = b"".join(r + b"\0" * ((-len(r)) & 3) for r in (rows if top_down else rows[::-1]))
        ww = a.shape[1]
        if hsz == 12:
            info = struct.pack("<IHHHH", 12, ww, hh, 1, bpp)
        else:
            info = struct.pack("<IiiHHIIiiII", hsz, ww, -hh if top_down else hh, 1, bpp, compress, len(body), 2835, 2835, 0, 0)
            if hsz >= 108:
                info += masks + b"\0" * (hsz - 40 - len(masks))
                masks = b""
        off = 14 + len(info) + len(masks) + len(palette)
        raw_file(name, b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + info + masks + palette + body)
    bgr_rows = [row[:, ::-1].tobytes() for row in a]
    bmp("topdown24.bmp", 40, 24, bgr_rows, top_down=True)
    bmp("os2_24.bmp", 12, 24, bgr_rows)
    bmp("os2_pal8.bmp", 12, 8, [row.tobytes() for row in idx], palette=pal[:, ::-1].tobytes())
    idx16 = (idx % 16).astype(np.uint8)
    packed4 = [bytes(((int(r[i]) << 4) | (int(r[i + 1]) if i + 1 < len(r) else 0)) for i in range(0, len(r), 2)) for r in idx16]
    bmp("pal4.bmp", 40, 4, packed4, palette=b"".join(bytes([int(c[2]), int(c[1]), int(c[0]), 0]) for c in pal[:16]))
    v555 = ((a[:, :, 0].astype(np.uint16) >> 3) << 10) | ((a[:, :, 1].astype(np.uint16) >> 3) << 5) | (a[:, :, 2].astype(np.uint16) >> 3)
    bmp("rgb555.bmp", 40, 16, [row.astype("<u2").tobytes() for row in v555])
    v565 = ((a[:, :, 0].astype(np.uint16) >> 3) << 11) | ((a[:, :, 1].astype(np.uint16) >> 2) << 5) | (a[:, :, 2].astype(np.uint16) >> 3)
    bmp("rgb565_bitfields.bmp", 40, 16, [row.astype("<u2").tobytes() for row in v565], compress=3, masks=struct.pack("<III", 0xf800, 0x07e0, 0x001f))
    x32 = (a[:, :, 0].astype(np.uint32) << 16) | (a[:, :, 1].astype(np.uint32) << 8) | a[:, :, 2].astype(np.uint32)
    bmp("xrgb32_alpha_all_zero.bmp", 40, 32, [row.astype("<u4").tobytes() for row in x32])
    odd = ((a[:, :, 0].astype(np.uint32) >> 2) << 20) | ((a[:, :, 1].astype(np.uint32) >> 1) << 9) | ((a[:, :, 2].astype(np.uint32) >> 5) << 2) | ((a[:, :, 1].astype(np.uint32) >> 4) << 28)
    bmp("odd_masks_v4.bmp", 108, 32, [row.astype("<u4").tobytes() for row in odd], compress=3, masks=struct.pack("<IIII", 63 << 20, 127 << 9, 7 << 2, 15 << 28))
    save("rgb.tga", a)
    save("rgba_rle.tga", rgba, compression="tga_rle")
    save("grey.tga", a, mode="L")
    save("grey_rle.tga", (a[:, :, 0] // 64 * 64), compression="tga_rle")
    save("palette.tga", a, mode="P")
    save("palette_rle.tga", (a // 96 * 96), mode="P", compression="tga_rle")
    save("rgb_topdown.tga", a, orientation=1)

    def tga(name, kind, bits, body, descriptor=0, cmap=b"", cmap_len=0, cmap_bits=0, ident=b""):
        raw_file(name, struct.pack("<BBBHHBHHHHBB", len(ident), 1 if cmap else 0, kind, 0, cmap_len, cmap_bits, 0, 0, a.shape[1], a.shape[0], bits, descriptor) + ident + cmap + body)
    tga("rgb555.tga", 2, 16, (v555 | 0x8000).astype("<u2").tobytes(), ident=b"five bits a channel")
    tga("grey_alpha16.tga", 3, 16, np.stack([grey, a[:, :, 2]], -1).tobytes(), descriptor=0x28)
    tga("palette_of_555.tga", 1, 8, (idx % 32).astype(np.uint8).tobytes(), cmap=v555[0, :32].astype("<u2").tobytes(), cmap_len=32, cmap_bits=16)
    save("palette.gif", a)
    save("interlaced.gif", a, interlace=True)
    save("sequential.gif", a, interlace=False)
    save("grey.gif", a, mode="L")
    Image.fromarray(a).convert("P").save(os.path.join(tmp, "transparent.gif"), transparency=5)
    out.append(("transparent.gif", os.path.join(tmp, "transparent.gif")))
    save("big_lzw.gif", field(200, 120))
    # a frame smaller than the screen, with a background index: the uncovered border takes the background entry
    small = Image.fromarray(a).convert("P")
    p = os.path.join(tmp, "frame_inside_screen.gif")
    small.save(p)
    raw = bytearray(open(p, "rb").read())
    assert raw[6:10] == struct.pack("<HH", a.shape[1], a.shape[0])
    raw[6:10] = struct.pack("<HH", a.shape[1] + 6, a.shape[0] + 5)
    raw[11] = 7                                                        # background colour index
    at = raw.index(b"\x2c" + struct.pack("<HHHH", 0, 0, a.shape[1], a.shape[0]))
    raw[at + 1:at + 5] = struct.pack("<HH", 4, 2)
    open(p, "wb").write(bytes(raw))
    out.append(("frame_inside_screen.gif", p))
    # Radiance .hdr, Photoshop .psd, Softimage .pic: written by hand (Pillow writes none of them)
    hh, ww = a.shape[:2]
    expo = (120 + (np.arange(hh)[:, None] * 3 + np.arange(ww)[None, :]) % 14).astype(np.uint8)      # 2^-16 .. 2^-3 times the mantissa: dark to beyond white
    expo[3, 5] = 0                                                                                    # the black pixel
    rgbe = np.concatenate([np.maximum(a, 1), expo[:, :, None]], -1)
    hdr_head = b"#?RADIANCE\nEXPOSURE=1.0\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (hh, ww)

    def hdr_rle_rows(px):
        out_rows = b""
        for row in px:
            out_rows += bytes([2, 2, ww >> 8, ww & 255])
            for k in range(4):
                col, i = row[:, k], 0
                while i < ww:
                    run = 1
                    while i + run < ww and run < 127 and col[i + run] == col[i]:
                        run += 1
                    if run >= 3:
                        out_rows += bytes([128 + run, int(col[i])]); i += run
                    else:
                        n = min(ww - i, 5)
                        out_rows += bytes([n]) + col[i:i + n].tobytes(); i += n
        return out_rows
    raw_file("rle.hdr", hdr_head + hdr_rle_rows(rgbe))
    raw_file("flat_rows.hdr", hdr_head.replace(b"#?RADIANCE", b"#?RGBE") + rgbe.tobytes())            # rows that do not start 2 2: stb's flat fallback
    raw_file("narrow.hdr", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X 5\n" % hh + rgbe[:, :5].tobytes())

    def psd(name, planes, depth, compression):
        head = b"8BPS" + struct.pack(">H6xHIIHH", 1, len(planes), hh, ww, depth, 3) + struct.pack(">III", 0, 0, 0) + struct.pack(">H", compression)
        if not compression:
            body = b"".join(pl.astype(">u2").tobytes() if depth == 16 else pl.astype(np.uint8).tobytes() for pl in planes)
        else:
            rows, counts = [], []
            for pl in planes:
                for row in pl.astype(np.uint8):
                    enc, i = b"", 0
                    while i < ww:
                        run = 1
                        while i + run < ww and run < 128 and row[i + run] == row[i]:
                            run += 1
                        if run >= 2:
                            enc += bytes([257 - run, int(row[i])]); i += run
                        else:
                            n = min(ww - i, 7)
                            enc += bytes([n - 1]) + row[i:i + n].tobytes(); i += n
                    enc += b"\x80"                                              # a no-op byte between rows
                    rows.append(enc); counts.append(len(enc))
            body = b"".join(struct.pack(">H", c) for c in counts) + b"".join(rows)
        raw_file(name, head + body)
    alpha = ((np.arange(hh)[:, None] * 37 + np.arange(ww)[None, :] * 11) % 256).astype(np.uint8)
    alpha[0, :4] = (0, 255, 1, 254)
    psd("rgb8_raw.psd", [a[:, :, 0], a[:, :, 1], a[:, :, 2]], 8, 0)
    psd("rgba8_rle.psd", [a[:, :, 0] // 32 * 32, a[:, :, 1], a[:, :, 2], alpha], 8, 1)
    psd("rgba16_raw.psd", [a[:, :, 0].astype(np.uint16) * 257, a[:, :, 1].astype(np.uint16) * 256 + 7, a[:, :, 2].astype(np.uint16) * 256 + 255, alpha.astype(np.uint16) * 257], 16, 0)

    def pic(name, packets):
        head = bytes([0x53, 0x80, 0xf6, 0x34]) + struct.pack(">f", 3.7) + b"made by hand".ljust(80, b"\0") + b"PICT" + struct.pack(">HHfHH", ww, hh, 1.0, 3, 0)
        table = b"".join(bytes([1 if k + 1 < len(packets) else 0, 8, kind, chan]) for k, (kind, chan, _) in enumerate(packets))
        body = b""
        for y in range(hh):
            for kind, chan, data in packets:
                row = data[y]                                     # (ww, n channels of this packet)
                if kind == 0:
                    body += row.tobytes()
                elif kind == 1:
                    i = 0
                    while i < ww:
                        run = 1
                        while i + run < ww and run < 255 and (row[i + run] == row[i]).all():
                            run += 1
                        body += bytes([run]) + row[i].tobytes(); i += run
                else:
                    i = 0
                    while i < ww:
                        run = 1
                        while i + run < ww and run < 300 and (row[i + run] == row[i]).all():
                            run += 1
                        if run >= 2:
                            body += (bytes([127 + run]) if run <= 128 else bytes([128]) + struct.pack(">H", run)) + row[i].tobytes(); i += run
                        else:
                            n = min(ww - i, 4)
                            body += bytes([n - 1]) + row[i:i + n].tobytes(); i += n
        raw_file(name, head + table + body)
    coarse = a // 64 * 64
    pic("rgb_raw.pic", [(0, 0xe0, a)])
    pic("rgb_pure_rle.pic", [(1, 0xe0, coarse)])
    pic("rgba_mixed_rle.pic", [(2, 0xe0, coarse), (2, 0x10, (alpha // 128 * 200)[:, :, None].astype(np.uint8))])
    pic("red_and_blue_only.pic", [(0, 0x80, a[:, :, :1]), (1, 0x20, coarse[:, :, 2:3])])
    raw_file("grey.pgm", b"P5\n# a comment\n%d %d\n255\n" % (a.shape[1], a.shape[0]) + grey.tobytes())
    raw_file("rgb.ppm", b"P6 %d %d 255\n" % (a.shape[1], a.shape[0]) + a.tobytes())
    raw_file("rgb_max100.ppm", b"P6\n%d\n%d\n100\n" % (a.shape[1], a.shape[0]) + (a // 3).astype(np.uint8).tobytes())
    raw_file("rgb16.ppm", b"P6 %d %d 65535 " % (a.shape[1], a.shape[0]) + np.stack([a, a[:, :, ::-1]], -1).tobytes())
    return out


def make_images_fixture():
    with tempfile.TemporaryDirectory() as tmp:
        files = image_files(tmp)
        arrays, _ = run(["images"] + [p for _, p in files], tmp)
        packed = {"names": np.array([n for n, _ in files])}
        for k, (name, path) in enumerate(files):
            packed["file%d" % k] = np.fromfile(path, dtype=np.uint8)
            packed["dims%d" % k] = arrays["img%d_dims" % k]
            packed["pixels%d" % k] = arrays["img%d_pixels" % k]
            print("   ", name, packed["dims%d" % k].tolist(), packed["file%d" % k].size, "bytes")
    np.savez_compressed(os.path.join(GOLD, "images.npz"), **packed)


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
        # what the object model composes and the demos do not: nested wrappers, volumes around boxes, groups as objects, instance in instance
        "nested": (["nested", 3, 60, 40, 8, 12], dict(seed=3, width=60, height=40, spp=8, depth=12)),
    }
    only = sys.argv[1:]          # optional: names of the fixtures to (re)generate
    if not only or "images" in only:
        # image files as the reference's image texture loads them (texture.hpp:87): file bytes in, texels out
        print("images")
        make_images_fixture()
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
