"""Census of the compiled device code (no GPU needed): compiles csrc/device/pt_kernels.hip to gfx950 assembly and counts, per kernel
and per non-inlined device function, what the counters do not show -- flat loads / stores (a generic pointer: counted on both memory
counters, so every use drains everything in flight), full drains (s_waitcnt vmcnt(0) lgkmcnt(0)), scratch traffic, vector loads
whose address is an SGPR pair plus an offset (a uniform address read lane by lane when the index register is a constant), scalar
loads, LDS operations and vector ALU instructions.  LABNOTES.md section 4, "What the compiled code showed".

usage: python tools/isa_census.py [extra hipcc flags ...]      e.g.  python tools/isa_census.py -DSHADE_SIMPLE_WAVES=4
as a module: census(flags=()) -> {demangled function name: {counter: value}}"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVICE = os.path.join(ROOT, "vk_gltf_renderer_amd", "csrc", "device")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KEYS = ("flat_load", "flat_store", "drain", "scratch", "global_load", "sgpr_base_load", "global_store", "s_load", "lds", "valu", "vgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes")


def census(flags=(), source="pt_kernels.hip"):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "out.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + DEVICE,
               "-Wno-unused-function", "--cuda-device-only", "-S", "-o", asm, os.path.join(DEVICE, source), *flags]
        if source == "pt_kernels.hip":
            cmd += ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-freciprocal-math", "-fapprox-func"]  # (csrc/Makefile: PT_KERNELS_FP)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        text = open(asm).read()
    stats, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z[A-Za-z0-9_]*):", line)
        if m:
            cur = m.group(1)
            stats[cur] = dict.fromkeys(KEYS, 0)
            continue
        if cur is None:
            continue
        t = line.strip()
        s = stats[cur]
        if t.startswith("flat_load"):
            s["flat_load"] += 1
        elif t.startswith(("flat_store", "flat_atomic")):
            s["flat_store"] += 1
        elif t.startswith("global_load"):
            s["global_load"] += 1
            if re.search(r", s\[\d+:\d+\]", t):
                s["sgpr_base_load"] += 1
        elif t.startswith(("global_store", "global_atomic")):
            s["global_store"] += 1
        elif t.startswith("scratch_"):
            s["scratch"] += 1
        elif t.startswith("s_load"):
            s["s_load"] += 1
        elif t.startswith("ds_"):
            s["lds"] += 1
        elif t.startswith("v_"):
            s["valu"] += 1
        if t.startswith("s_waitcnt vmcnt(0) lgkmcnt(0)"):
            s["drain"] += 1
    # register / scratch figures of the kernels (the metadata block at the end of the file)
    for blk in re.split(r"\n  - \.agpr_count", text)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or name.group(1) not in stats:
            continue
        s = stats[name.group(1)]
        for key, pat in (("vgpr", r"\.vgpr_count:\s+(\d+)"), ("vgpr_spill", r"\.vgpr_spill_count:\s+(\d+)"), ("sgpr_spill", r"\.sgpr_spill_count:\s+(\d+)"),
                         ("scratch_bytes", r"\.private_segment_fixed_size:\s+(\d+)")):
            m = re.search(pat, blk)
            if m:
                s[key] = int(m.group(1))
    names = list(stats)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for mangled, d in zip(names, dem):
        d = re.sub(r"pt::\(anonymous namespace\)::", "", d)
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*", "", d)
        out[d] = stats[mangled]
    return out


if __name__ == "__main__":
    res = census(tuple(sys.argv[1:]))
    print(f"{'function':44s} " + " ".join(f"{k[:9]:>9s}" for k in KEYS))
    for name, s in res.items():
        if s["valu"] >= 50:
            print(f"{name[:44]:44s} " + " ".join(f"{s[k]:9d}" for k in KEYS))
