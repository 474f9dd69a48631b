#!/usr/bin/env python3
"""Static mix of vector-ALU instruction classes per kernel of csrc/device/pt_kernels.hip (no GPU needed), for the issue model of
bench.py's `issue_frac` (tools/make_pmc_latest.py, LABNOTES.md section 4).  On the SIMD-32 of gfx950 a wave64 instruction of the
FULL-rate class -- f32 fma / mul / add / sub, and / or / xor, 32-bit integer add / sub, mov -- issues over 2 cycles (MI355X_MICROARCH.md;
tools/microbench_valu.hip measures that class at 1.9-2.1 cycles per instruction and SIMD against 2.9-3.1 for everything else: min / max /
med3, conversions, shifts, bit-field ops, selects, compares, 24-bit and 32-bit multiplies, lane permutes in DPP form, packed f32 ops,
and 5.6 for a reciprocal).  The model charges 2 / 4 / 4 / 8 cycles to full / half / packed / transcendental instructions -- the upper
end of what was measured -- and a kernel's average follows from the static shares of its compiled code (the dynamic mix of its hot loop
may differ: a model, stated as such).
usage: python tools/valu_mix.py > profiles/r06_valu_mix.json"""
import collections
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_lines  # noqa: E402  (compile_asm)

FULL = {"v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_mad_f32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_mov_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
CYCLES = {"full": 2.0, "half": 4.0, "packed": 4.0, "transcendental": 8.0}


def short(name):  # as tools/summarize_pmc.py names kernels
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    asm = isa_lines.compile_asm()
    counts, cur = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if "k_" in m.group(1) else None
            if cur:
                counts[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            cur = None
        elif t.startswith("v_"):
            op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", t.split()[0])
            counts[cur]["full" if op in FULL else ("packed" if op.startswith("v_pk_") else ("transcendental" if op in TRANS else "half"))] += 1
    names = subprocess.run(["c++filt"] + list(counts), capture_output=True, text=True).stdout.split("\n")
    out = {"model_cycles_per_wave64_instruction": CYCLES, "source": "static instruction mix of the compiled kernels (tools/valu_mix.py), classes timed by tools/microbench_valu.hip",
           "kernels": {}}
    for mangled, name in zip(counts, names):
        c = counts[mangled]
        tot = sum(c.values())
        if tot == 0:
            continue
        out["kernels"][short(name)] = {"valu_static": tot, **{k: round(c[k] / tot, 3) for k in CYCLES},
                                       "avg_cycles": round(sum(CYCLES[k] * c[k] for k in CYCLES) / tot, 3)}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
