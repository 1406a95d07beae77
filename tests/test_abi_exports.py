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
