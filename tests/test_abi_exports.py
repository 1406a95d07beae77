"""CPU test: the C-ABI library loads and exports every symbol include/rtcore.h declares (no compute without a GPU)."""
import os
import re

import raytracing_amd
from raytracing_amd.rtcore import EXPORTS, SceneDesc, RenderParams, RenderStats
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    header = open(os.path.join(ROOT, "include", "rtcore.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(rt_\w+)\s*\(", header, flags=re.M))
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)
    lib = raytracing_amd.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rt_version()


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors must have the sizes the C compiler gives include/rtcore.h (also proves the header is plain C)."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "rtcore.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(rt_scene_desc), '
                   'sizeof(rt_render_params), sizeof(rt_render_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(SceneDesc), C.sizeof(RenderParams), C.sizeof(RenderStats)]


def test_scene_description_is_validated_before_any_gpu_call():
    """rt_scene_create checks the whole description first (RT_EINVAL) and only then touches the device (RT_EHIP here: no GPU in this
    container, and no CPU fallback either): malformed wrapper records of the round-3 object model are refused with a message, not
    crashed on -- a chain that runs off the end of xform_data, box-boundary / rotatey / translate flags on primitives that cannot
    carry them, unknown flag bits, indices out of range."""
    import numpy as np
    import oracle_lib as O
    import raytracing_amd as R
    R.load_library()
    d = O.Scene.nested(60, 40, 3).export()

    def error_of(desc):
        with __import__("pytest").raises(R.RtError) as e:
            R.Scene(desc)
        return str(e.value)

    import torch
    if not torch.cuda.is_available():
        assert "error -2" in error_of(d)                              # a well-formed scene gets as far as the device
    cases = []
    b = dict(d); b["xform_data"] = d["xform_data"].copy(); b["xform_data"][-1, 5] = 4 | 8; cases.append((b, "cannot continue"))
    b = dict(d); b["xform_data"] = d["xform_data"].copy(); b["xform_data"][0, 5] = 8; cases.append((b, "ROTATEY and / or TRANSLATE"))
    for flag, what in ((8, "RT_PRIM_VOLUME"), (2, "prim_xform"), (32, "unknown primitive flag")):
        b = dict(d); b["prim_flags"] = d["prim_flags"].copy(); b["prim_flags"][2] |= flag; cases.append((b, what))     # primitive 2: a sphere
    b = dict(d); b["prim_flags"] = d["prim_flags"].copy(); b["prim_flags"][8] |= 4; cases.append((b, "boundary BOX"))   # 8: a sphere-bounded volume
    b = dict(d); b["prim_xform"] = d["prim_xform"].copy(); b["prim_xform"][2] = len(d["xform_data"]); cases.append((b, "prim_xform out of range"))
    b = dict(d); b["node_child"] = d["node_child"].copy(); b["node_child"][3, 1] = 10 ** 6; cases.append((b, "child code out of range"))
    for desc, what in cases:
        msg = error_of(desc)
        assert "error -1" in msg and what in msg, (what, msg)
