#!/usr/bin/env python3
"""Instruction counts of the marked regions (RT_ASM_MARK: node visit, leaf step, parts of the shade pass) of one render_wave_kernel variant in a
device assembly file made with  hipcc ... '-DRT_ASM_MARK(text)=asm volatile(text)' --cuda-device-only -S rt_kernels_fp64.hip -o k.s
usage: tools/isa_regions.py k.s [mangled-name prefix, default the fp64 OBJ variant]"""
import re
import sys

txt = open(sys.argv[1]).read()
prefix = sys.argv[2] if len(sys.argv) > 2 else "_ZN2rt18render_wave_kernelIdLb0ELb1ELi3E"
start = re.search(r"^" + re.escape(prefix) + r"\w*:", txt, re.M)
body = txt[start.end():]
body = body[:body.index("s_endpgm")]
lines = body.split("\n")
INST = re.compile(r"\s+(v|s|ds|global|buffer|scratch|flat)_\w+")


def region(a, b):
    out, on = [], False
    for l in lines:
        if a in l:
            on = True
        if on:
            out.append(l)
        if on and b in l:
            break
    return out


def stats(name, ls):
    n = sum(1 for l in ls if INST.match(l))
    valu = sum(1 for l in ls if re.match(r"\s+v_", l))
    f64 = sum(1 for l in ls if re.match(r"\s+v_\w+_f64", l))
    scr = sum(1 for l in ls if re.match(r"\s+scratch_", l))
    gl = sum(1 for l in ls if "global_load" in l)
    ds = sum(1 for l in ls if re.match(r"\s+ds_", l))
    print("  %-26s insts %5d  VALU %5d (f64 %4d)  scratch %4d  global_load %3d  ds %3d" % (name, n, valu, f64, scr, gl, ds))


stats("whole kernel", lines)
stats("node visit", region("wide node_step begin", "wide node_step end"))
stats("leaf step", region("leaf step begin", "leaf step end"))
stats("shade: record", region("shade: record", "shade: scatter"))
stats("shade: scatter", region("shade: scatter", "shade: pdf"))
stats("shade: pdf", region("shade: pdf", "shade: tail"))
stats("shade: tail", region("shade: tail", "shade: end"))
