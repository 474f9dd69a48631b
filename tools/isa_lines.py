#!/usr/bin/env python3
"""Static vector-instruction count of one kernel of csrc/device/pt_kernels.hip PER SOURCE LINE (no GPU needed): compiles with
-gline-tables-only (same code, .loc directives added), attributes every instruction of the kernel to the source location the
compiler names for it, and prints the totals per file and the heaviest lines.  Complements tools/isa_census.py (per function
totals) when the question is which lines of an inlined loop body the instructions of a node step belong to (LABNOTES.md section 4).
usage: python tools/isa_lines.py <substring of the mangled kernel name> [top N]   e.g.  tools/isa_lines.py k_trace_closestILb1ELb1ELb0E 40"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVICE = os.path.join(ROOT, "vk_gltf_renderer_amd", "csrc", "device")


def compile_asm(extra=()):
    out = os.path.join(tempfile.gettempdir(), "pt_kernels_lines.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + DEVICE,
           "-Wno-unused-function", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-freciprocal-math", "-fapprox-func",  # (csrc/Makefile: PT_KERNELS_FP)
           "--cuda-device-only", "-gline-tables-only", "-S", "-o", out, os.path.join(DEVICE, "pt_kernels.hip"), *extra]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def per_line(asm_path, kernel):
    files, counts, cur, inside = {}, collections.Counter(), None, False
    for line in open(asm_path):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", line)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = kernel in m.group(1)
            continue
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            inside = False
        if t.startswith("v_") and cur:
            counts[cur] += 1
    return counts


if __name__ == "__main__":
    kernel = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    c = per_line(compile_asm(), kernel)
    print("total vector instructions", sum(c.values()))
    byfile = collections.Counter()
    for (f, l), n in c.items():
        byfile[f] += n
    print(dict(byfile))
    for (f, l), n in sorted(c.items(), key=lambda kv: (kv[0][0], kv[0][1])):
        if n >= top:
            print(f"{n:5d}  {f}:{l}")
