"""CPU tests: the C++ host mirror (raytracing_amd/host: scene/bvh/loader/flatten, the product's scene path) against the
golden fixtures of the REAL reference and against the oracle's independent flatten.  All comparisons are bit-exact."""
import os

import numpy as np
import pytest

import oracle_lib as O
from raytracing_amd.host_scenes import HostScene

ROW = 56


def desc_to_rows(d):
    """Re-create ref_harness.cpp's pre-order tree dump from a flattened scene."""
    rows, first_seen = [], {}

    def tex_row(ti, row):
        k = int(d["tex_kind"][ti]); t = d["tex_data"][ti]
        row[0] = k
        if k == 1:
            row[1:4] = t[0:3]
        elif k == 2:
            row[1:4] = d["tex_data"][int(t[0])][0:3]; row[4:7] = d["tex_data"][int(t[1])][0:3]
        elif k == 3:
            row[1:5] = t[0:4]

    def mat_row(mi, row):
        k = int(d["mat_kind"][mi]); m = d["mat_data"][mi]; t = int(d["mat_tex"][mi])
        row[0] = k
        if k in (2, 3):
            row[1] = m[0]
        if k != 3 and t >= 0:
            tex_row(t, row[2:])          # (fog, kind 6: base_color, like lambertian)
        if k == 5:
            row[9:21] = m[1:13]; row[21] = m[13]; row[22] = m[14]

    def leaf_row(kind, flags, mat, data, r):
        r[0] = kind; r[1] = flags & 1
        ng = {1: 4, 2: 9, 6: 24, 7: 11, 8: 5}.get(kind, 5)
        r[4:4 + ng] = data[:ng]
        if kind == 7:
            r[3] = flags & 6      # rotatey / translate wrappers around the box
        mat_row(mat, r[30:])

    node_seen = {}
    has_xf = len(np.asarray(d.get("xform_data", [])).ravel()) > 0

    def walk(code, parent_xf=-1):
        idx = len(rows)
        r = np.zeros(ROW)
        rows.append(r)
        if code >= 0:
            node_seen.setdefault(code, idx)       # a span-1 node names the same child twice: the harness dumps that sub-tree twice
            r[0] = 0; r[2] = node_seen[code]
            r[4:10] = d["node_bbox"][code]
            xf = int(d["node_xform"][code]) if has_xf else -1
            if xf >= 0 and parent_xf < 0:         # the root of an instanced sub-tree: the harness shows the wrappers on this entry
                x = d["xform_data"][xf]
                r[3] = x[5]; r[12:17] = x[0:5]
            l = walk(int(d["node_child"][code][0]), xf)
            rr = walk(int(d["node_child"][code][1]), xf)
            r[10] = l; r[11] = rr
        else:
            p = ~code
            first_seen.setdefault(p, idx)
            r[2] = first_seen[p]
            leaf_row(int(d["prim_kind"][p]), int(d["prim_flags"][p]), int(d["prim_mat"][p]), d["prim_data"][p], r)
        return idx

    for c in d["roots"]:
        walk(int(c))
    lights = np.zeros((len(d["light_kind"]), ROW))
    for i in range(len(lights)):
        leaf_row(int(d["light_kind"][i]), int(d["light_flags"][i]), int(d["light_mat"][i]), d["light_data"][i], lights[i])
        lights[i][2] = i
    return np.array(rows), lights


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).ravel().view(np.uint64)


CASES = {
    "c0_seed1": ("c0", lambda gd, m: HostScene.random_spheres(m["width"], m["height"], m["seed"]), lambda gd, m: O.Scene.c0(m["width"], m["height"], m["seed"])),
    "c0_seed7": ("c0", lambda gd, m: HostScene.random_spheres(m["width"], m["height"], m["seed"]), lambda gd, m: O.Scene.c0(m["width"], m["height"], m["seed"])),
    "mixed": ("mixed", lambda gd, m: HostScene.mixed(m["width"], m["height"], m["seed"]), lambda gd, m: O.Scene.mixed(m["width"], m["height"], m["seed"])),
    "cornell": ("cornell", lambda gd, m: HostScene.cornell(m["width"], m["height"], m["seed"]), lambda gd, m: O.Scene.cornell(m["width"], m["height"], m["seed"])),
    "features": ("features", lambda gd, m: HostScene.features(m["width"], m["height"], m["seed"]), lambda gd, m: O.Scene.features(m["width"], m["height"], m["seed"])),
    "smoke": ("smoke", lambda gd, m: HostScene.smoke(m["width"], m["height"], m["seed"]), lambda gd, m: O.Scene.smoke(m["width"], m["height"], m["seed"])),
    "obj": ("obj", lambda gd, m: HostScene.obj(os.path.join(gd, "mesh_scene.obj"), m["width"], m["height"], m["seed"]),
            lambda gd, m: O.obj_scene(os.path.join(gd, "mesh_scene.obj"), m["width"], m["height"], m["seed"])),
    # polygons with 5..8 corners, concave ones included: tinyobj's ear clipping, restated in host/loader.hpp and in the oracle's loader
    "ngon": ("obj", lambda gd, m: HostScene.obj(os.path.join(gd, "ngon_scene.obj"), m["width"], m["height"], m["seed"]),
             lambda gd, m: O.obj_scene(os.path.join(gd, "ngon_scene.obj"), m["width"], m["height"], m["seed"])),
}


@pytest.mark.parametrize("name", list(CASES))
def test_host_scene_and_bvh_match_reference_fixture(golden_dir, name):
    tag, make_host, make_oracle = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    m = {k: int(g["meta_" + k]) for k in ("width", "height", "seed")}
    hs = make_host(golden_dir, m)
    d = hs.arrays()
    rows, lights = desc_to_rows(d)
    # scene generation, loader, BVH builder and flatten vs the real reference's tree dump
    want = g[tag + "_tree"].reshape(-1, ROW).copy()
    want[want[:, 0] == 7, 15:21] = 0      # the harness also dumps a wrapped box's own bounding box; the flattened scene carries it in its parent node
    assert np.array_equal(bits(rows), bits(want)), "pre-order tree rows"
    assert np.array_equal(bits(lights), bits(g[tag + "_lights"])), "lights"
    assert np.array_equal(bits(d["camera"]), bits(g[tag + "_camera"])), "camera fields"
    # and against the oracle's own flatten: same primitive ids, same geometry, same tree links
    od = make_oracle(golden_dir, m).export()
    for k in ("node_bbox", "prim_data", "light_data"):
        assert np.array_equal(bits(d[k]), bits(od[k])), k
    for k in ("node_child", "roots", "prim_kind", "prim_flags", "light_kind", "light_flags"):
        assert np.array_equal(d[k], od[k]), k
    if "xform_data" in od:                     # instances: same transforms, same membership
        assert np.array_equal(bits(d["xform_data"]), bits(od["xform_data"]))
        assert np.array_equal(d["node_xform"], od["node_xform"]) and np.array_equal(d["prim_xform"], od["prim_xform"])
    else:
        assert d["xform_data"].size == 0
    hs.free()


def test_nested_wrappers_groups_and_volume_boxes_flatten_like_the_oracle(golden_dir):
    """nested_test() (host/demo.hpp) through scene::build_bvh and the flattener against the oracle's independent flatten of the same scene
    (orc_build_nested, itself pinned bit for bit to the fixture of the real reference: tests/test_oracle_golden.py `nested`): same tree,
    same primitive ids and records, same chains of transform records (translate(translate(.)), rotatey(translate(.)), an instance inside
    an instance), the innermost record folded into a box, outer flips, group links, volumes around boxes."""
    g = np.load(os.path.join(golden_dir, "nested.npz"))
    m = {k: int(g["meta_" + k]) for k in ("width", "height", "seed")}
    hs = HostScene.nested(m["width"], m["height"], m["seed"])
    d = hs.arrays()
    od = O.Scene.nested(m["width"], m["height"], m["seed"]).export()
    for k in ("node_bbox", "prim_data", "light_data", "xform_data", "camera"):
        assert np.array_equal(bits(d[k]), bits(od[k])), k
    for k in ("node_child", "roots", "prim_kind", "prim_flags", "light_kind", "light_flags", "node_xform", "prim_xform"):
        assert np.array_equal(d[k], od[k]), k
    assert np.array_equal(bits(d["camera"]), bits(g["nested_camera"]))
    fl, xf = np.asarray(d["prim_flags"]), np.asarray(d["xform_data"])
    assert (fl & 16).any() and (fl & 8).any() and (xf[:, 5].astype(int) & 8).any() and np.isinf(d["node_bbox"]).any()
    hs.free()


def test_one_object_under_two_wrapper_sets_is_refused(tmp_path):
    """The reference lets one object pointer appear twice, once moved (each use intersects it through its own wrappers); a flattened
    primitive has one record, one set of flags and one transform chain, so the flattener refuses that scene instead of rendering one of the
    two uses wrongly."""
    import subprocess
    src = tmp_path / "dup.cpp"
    src.write_text('#include "rt_host.hpp"\n#include <cstdio>\nint main(){scene scn(32,24);sphere* s=new sphere(vec3d(0,1,0),1,new lambertian(vec3d(0.5,0.5,0.5)));'
                   'scn.add_object(s);scn.add_object(new translate(s,vec3d(3,0,0)));scn.add_object(new sphere(vec3d(0,-100,0),100,new lambertian(vec3d(0.5,0.5,0.5))));'
                   'scn.add_light(new flip(new planexz(-1,1,-1,1,5,new emissive(vec3d(4,4,4)))));scn.set_camera(camera(vec3d(0,2,8),vec3d(0,1,0),vec3d(0,1,0),40,32.0/24,0,1,0,1));'
                   'scn.build_bvh();flat_scene f;try{scn.flatten(f);std::puts("flattened");}catch(const std::exception& e){std::printf("refused: %s\\n",e.what());}return 0;}\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "dup"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "raytracing_amd", "host"), str(src), "-o", str(exe),
                    "-L", os.path.join(root, "raytracing_amd"), "-lrtcore", "-lz", "-Wl,-rpath," + os.path.join(root, "raytracing_amd")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    assert out.startswith("refused: flatten: one object is used under two different sets of wrappers"), out


def test_png_roundtrip(tmp_path):
    """image_io.hpp: write a PNG with the host writer, read it back with the host reader through an image texture."""
    import subprocess
    src = tmp_path / "t.cpp"
    src.write_text('#include "rt_host.hpp"\n#include <cstdio>\nint main(int c,char**v){unsigned char px[4*3*3];for(int i=0;i<36;i++)px[i]=(unsigned char)(i*7);'
                   'if(!rt_write_png(v[1],4,3,3,px,true))return 1;std::vector<unsigned char> out;int w,h,ch;if(!rt_load_image(v[1],out,w,h,ch))return 2;'
                   'if(w!=4||h!=3||ch!=3)return 3;for(int y=0;y<3;y++)for(int x=0;x<12;x++)if(out[y*12+x]!=px[(2-y)*12+x])return 4;std::puts("ok");return 0;}\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "raytracing_amd", "host"), str(src), "-o", str(exe),
                    "-L", os.path.join(root, "raytracing_amd"), "-lrtcore", "-lz", "-Wl,-rpath," + os.path.join(root, "raytracing_amd")], check=True)
    assert subprocess.run([str(exe), str(tmp_path / "a.png")], capture_output=True, text=True).stdout.strip() == "ok"


def _image_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "images.npz"))
    return g, [str(n) for n in g["names"]]


def test_image_files_decode_to_the_reference_loaders_texels(golden_dir, tmp_path):
    """image_io.hpp / jpeg_decode.hpp against tests/golden/images.npz: the bytes of 86 image files and, for each, what the
    reference's image texture holds after stbi_load(path, &w, &h, &ch, 0) (texture.hpp:87; made by oracle/gen_golden.py with the
    reference's loader compiled here).  JPEG: baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 / 4:4:0 chroma, quality 5..100
    (the inverse DCT's clamps), restart intervals, optimised Huffman tables, grey, Adobe CMYK / YCCK, components named R,G,B, pictures
    narrower than one block.  PNG: every colour type, 1 / 4 / 8 / 16 bits, Adam7, palette and colour-key transparency.  BMP: 1 / 4 / 8-bit
    palettes, 16 / 24 / 32 bits with default and BI_BITFIELDS masks, OS/2 and V4 headers, top-down rows, an all-zero alpha plane.
    TGA: true-colour, grey, colour-mapped, 15 / 16-bit colours, run-length packed, both row orders.  GIF: sequential and interlaced
    frames, a transparent index, a frame smaller than its screen over a background index.  PGM / PPM incl. two-byte samples.  Radiance
    HDR (run-length rows, flat rows, stb's 8-bit tone curve), PSD (raw, PackBits, 16-bit, the un-matting of partial alpha), Softimage PIC
    (raw, pure and mixed run-length packets, separate channel packets) -- every format stb_image loads.
    Bar: size, channel count and every texel equal."""
    from raytracing_amd.host_scenes import load_image
    g, names = _image_fixture(golden_dir)
    assert len(names) >= 86 and {os.path.splitext(n)[1] for n in names} == {".jpg", ".png", ".bmp", ".gif", ".tga", ".pgm", ".ppm", ".hdr", ".psd", ".pic"}
    for k, name in enumerate(names):
        path = str(tmp_path / name)
        g["file%d" % k].tofile(path)
        w, h, ch, ok = [int(v) for v in g["dims%d" % k]]
        assert ok == 1, name
        got = load_image(path)
        assert got.shape == (h, w, ch), (name, got.shape, (h, w, ch))
        assert np.array_equal(got, g["pixels%d" % k].reshape(h, w, ch)), name


def test_damaged_image_files_are_refused_not_crashed_on(golden_dir, tmp_path):
    """A file cut short or with flipped bytes either decodes to a picture of its stated size or raises; a texture that fails to load
    renders cyan, as in the reference (texture.hpp:97-99)."""
    from raytracing_amd.host_scenes import load_image
    from raytracing_amd.rtcore import RtError
    g, names = _image_fixture(golden_dir)
    rng = np.random.default_rng(5)
    for name in ("rgb420_q75.jpg", "prog420.jpg", "restart420.jpg", "interlaced_rgb.png", "palette_trns.png", "pal4.bmp", "odd_masks_v4.bmp",
                 "rgba_rle.tga", "palette_of_555.tga", "rgb16.ppm", "interlaced.gif", "frame_inside_screen.gif", "rle.hdr", "rgba8_rle.psd", "rgba_mixed_rle.pic"):
        data = g["file%d" % names.index(name)]
        for trial in range(40):
            bad = data.copy()
            if trial % 2:
                bad = bad[:int(rng.integers(0, bad.size))]
            else:
                bad[rng.integers(0, bad.size, 3)] = rng.integers(0, 256, 3)
            path = str(tmp_path / ("bad_" + name))
            bad.tofile(path)
            try:
                a = load_image(path)
                assert a.ndim == 3 and a.size > 0
            except RtError:
                pass
    with pytest.raises(RtError):
        load_image(str(tmp_path / "no_such_file.jpg"))


def test_jpeg_map_kd_reaches_the_scene_through_the_obj_loader(golden_dir, tmp_path):
    """OBJ text -> MTL `map_Kd *.jpg` -> loader -> image texture -> flatten (loader.hpp:136-171): the flattened texels are the
    reference loader's, for a colour JPEG (3 channels) and a grey one (1 channel: the reference indexes it with its own channel
    count and reads three neighbouring bytes, texture.hpp:115-122 -- the scene carries the count, the kernels do the same)."""
    g, names = _image_fixture(golden_dir)
    files = ("prog420.jpg", "grey_q85.jpg")
    for name in files:
        g["file%d" % names.index(name)].tofile(str(tmp_path / name))
    (tmp_path / "m.mtl").write_text("".join(f"newmtl m{i}\nKd 0.7 0.7 0.7\nmap_Kd {name}\n\n" for i, name in enumerate(files)) + "newmtl l\nKe 5 5 5\n")
    (tmp_path / "s.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nv 0 0 1\nv 1 0 1\nvt 0 0\nvt 1 0\nvt 0 1\n"
                                    "usemtl m0\nf 1/1 2/2 3/3\nusemtl m1\nf 2/1 4/2 3/3\nusemtl l\nf 1/1 5/2 6/3\n")
    hs = HostScene.obj(str(tmp_path / "s.obj"), 16, 16, 1)
    d = hs.arrays()
    assert int((d["tex_kind"] == 3).sum()) == 2
    for i, name in enumerate(files):
        k = names.index(name)
        w, h, ch, _ = [int(v) for v in g["dims%d" % k]]
        ti = int(d["mat_tex"][int(d["prim_mat"][i])])
        tw, th, tch, color_based, off = [int(x) for x in d["tex_data"][ti][:5]]
        assert (tw, th, tch, color_based) == (w, h, ch, 1), name
        assert np.array_equal(d["tex_pixels"][off:off + w * h * ch], g["pixels%d" % k]), name
    hs.free()


def test_mtl_texture_options_and_material_library_lists(golden_dir, tmp_path):
    """What tinyobjloader accepts and the reference therefore loads (tiny_obj_loader.hpp:1247-1331, 2843-2893): `map_Kd` with options in
    front of the file name (each skipped with the number of arguments tinyobj consumes), a file name with a blank in it, CR-LF line
    ends, and `mtllib` with several names of which the first that opens is used (a backslash keeps a blank inside a name)."""
    g, names = _image_fixture(golden_dir)
    k = names.index("rgb420_q75.jpg")
    g["file%d" % k].tofile(str(tmp_path / "my tex.jpg"))
    (tmp_path / "my mats.mtl").write_bytes(b"newmtl a\r\nKd 0.5 0.5 0.5\r\nmap_Kd -s 2 2 1 -bm 0.3 -clamp on -mm 0 1 my tex.jpg\r\n\r\nnewmtl l\r\nKe 4 4 4\r\n")
    (tmp_path / "s.obj").write_bytes(b"mtllib missing.mtl my\\ mats.mtl unused.mtl\r\nv 0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nv 0 0 1\r\nvt 0 0\r\nvt 1 0\r\nvt 0 1\r\n"
                                     b"usemtl a\r\nf 1/1 2/2 3/3\r\nusemtl l\r\nf 1/1 3/2 4/3\r\n")
    hs = HostScene.obj(str(tmp_path / "s.obj"), 16, 16, 1)
    d = hs.arrays()
    assert int(hs.desc.contents.n_prims) == 2 and int(hs.desc.contents.n_lights) == 1
    ti = int(d["mat_tex"][int(d["prim_mat"][0])])
    w, h, ch, _ = [int(v) for v in g["dims%d" % k]]
    assert ti >= 0 and [int(x) for x in d["tex_data"][ti][:3]] == [w, h, ch]
    assert np.array_equal(d["tex_pixels"][int(d["tex_data"][ti][4]):][:w * h * ch], g["pixels%d" % k])
    hs.free()


def test_obj_loader_drops_faces_with_indices_outside_the_arrays(tmp_path):
    """A malformed or truncated OBJ (index 0, an index past the arrays, a relative index before their start) must not read outside the
    vertex arrays: such faces are dropped (tinyobj fails the parse on them), the good ones load, and a file with no usable face is
    an error through rth_last_error rather than a crash."""
    mtl = tmp_path / "m.mtl"
    mtl.write_text("newmtl a\nKd 0.5 0.5 0.5\nnewmtl l\nKe 3 3 3\n")
    good = "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvn 0 0 1\nusemtl a\nf 1 2 3\nusemtl l\nf 2/1/1 4/1/1 3/1/1\nusemtl a\n"
    bad_faces = ["f 0 0 0", "f 1 2 99", "f -9 1 2", "f 1/7 2/1 3/1", "f 1//5 2//1 3//1", "f 1 2 3 4 77"]
    p = tmp_path / "bad.obj"
    p.write_text(good + "\n".join(bad_faces) + "\n")
    hs = HostScene.obj(str(p), 16, 16, 1)
    assert int(hs.desc.contents.n_prims) == 2 and int(hs.desc.contents.n_lights) == 1
    hs.free()
    q = tmp_path / "allbad.obj"
    q.write_text("mtllib m.mtl\nv 0 0 0\nusemtl a\n" + "\n".join(bad_faces) + "\n")
    with pytest.raises(Exception, match="no triangles"):
        HostScene.obj(str(q), 16, 16, 1)
