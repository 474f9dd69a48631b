#!/usr/bin/env python3
"""profiles/pmc_latest_<workload>.json from a tools/summarize_pmc.py summary: per-kernel HBM bytes per launch from the separate
FETCH_SIZE / WRITE_SIZE passes, corrected with the factors MEASURED on this renderer's access patterns (tools/calib_fetch.hip ->
profiles/r03_fetch_calibration.json; MI355X_MICROARCH.md: the counter reports half of a wide stream's bytes, other widths must be
calibrated), and what bench.py quotes as `roofline.traffic`.  Without a calibration file FETCH_SIZE is doubled (the guide's figure
for wide streams) like in round 2.
usage: tools/make_pmc_latest.py <summary.json> <workload> <frames in flight> [round] [calibration.json] [W H] > profiles/pmc_latest_<workload>.json"""
import json, os, sys

summary, workload, frames = json.load(open(sys.argv[1]))["kernels"], sys.argv[2], int(sys.argv[3])
try:
    MIX = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_valu_mix.json")))["kernels"]
except Exception:
    MIX = {}
calib_path = sys.argv[5] if len(sys.argv) > 5 else None
calib = json.load(open(calib_path))["factors"] if calib_path else None
res = [int(sys.argv[6]), int(sys.argv[7])] if len(sys.argv) > 7 else [1920, 1080]


def mean(*v):
    v = [x for x in v if x]
    return sum(v) / len(v) if v else None


# Counter / true bytes per kernel.  What the calibration shows (profiles/r03_fetch_calibration.json): the counter tallies REQUESTS at 64 B
# each -- a gather that stays inside one 64-B line (16-B texel footprints, 48-B vertex / triangle records) is counted exactly (1.00),
# a wide request (consecutive 16-B words of a wave: queue entries; an 80-B node record that straddles two lines) is a 128-B request
# counted as 64 (0.50 / 0.52).  A kernel's ratio is the harmonic mix of the two by the share s of its ALGORITHMIC read bytes that
# come as wide requests (LABNOTES.md section 4: bench.py's byte model):  r = 1 / (s / r_wide + (1 - s) / r_line).
if calib:
    r_wide, r_line, r_node = calib["fetch_stream16"], mean(calib["fetch_gather16"], calib["fetch_gather48"]), calib["fetch_gather80"]

    def mix(s_wide, wide=r_wide):
        return 1.0 / (s_wide / wide + (1.0 - s_wide) / r_line)

    r_fetch = {"shade_first": mix(68.0 / 980.0),    # per hit: queue entry 52 B + the path's misc record 16 B (round 4: in the entry) as streams; shade record /
                                                     # attributes 192 + records 480 + ~5 taps x 48 as gathers
               "shade": mix(100.0 / 1012.0),          # entry 52 B + misc / throughput / radiance 48 B as streams (round 4: the state travels in the queue entry)
               "trace_closest": mix(0.8, r_node),     # ~80 % of the bytes are 80-B node records (20 visits x 80 B against 8 triangles x 48 B)
               "trace_shadow": mix(0.8, r_node),
               "trace_primary": r_line,               # nodes and triangles through the scalar cache: 64-B lines
               "shadow_resolve": r_wide, "finish_sample": mix(0.5), "generate": r_wide}
    # WRITE_SIZE: a wide stream is counted exactly; a lone 16-B store is tallied as 32 B (the memory's write granule) -- taken as reported
    r_write = {k: calib["write_stream16"] for k in r_fetch}
else:
    r_fetch, r_write = {}, {}
# the timed (non-counting) template instances: k_shade<COUNT, SIMPLE, FIRST>, k_trace_closest<WIDE, HAS_ALPHA, COUNT>,
# k_trace_primary<HAS_ALPHA, COUNT, INTERVAL>, k_trace_shadow<WIDE, MODE, COUNT>; of several matches the one with the most time
pick = {"shade_first": lambda n: n.startswith("k_shade<false") and n.endswith("true>"),
        "shade": lambda n: n.startswith("k_shade<false") and n.endswith("false>"),
        "trace_closest": lambda n: n.startswith("k_trace_closest<") and n.endswith("false>"),
        "trace_primary": lambda n: n.startswith("k_trace_primary<") and n.split(", ")[1].startswith("false"),  # <HAS_ALPHA, COUNT, INTERVAL>
        "trace_shadow": lambda n: n.startswith("k_trace_shadow<") and n.endswith("false>"),
        "shadow_resolve": lambda n: n.startswith("k_shadow_resolve"),
        "finish_sample": lambda n: n.startswith("k_finish_sample"),
        "generate": lambda n: n.startswith("k_generate")}
out = {"round": int(sys.argv[4]) if len(sys.argv) > 4 else 2, "workload": workload, "frames_in_flight": frames, "resolution": res,
       "fetch_size_calibration": (os.path.join("profiles", os.path.basename(calib_path)) + " (tools/calib_fetch.hip: counter / true bytes measured for 16-B streams and 16 / 48 / 80-B gathers)") if calib else "the guide (wide streaming reads): FETCH_SIZE x 2",
       "fetch_size_factor": "per kernel, see kernels[*].fetch_counter_over_true" if calib else 2.0,
       "command": f"rocprofv3 --pmc <one counter group per pass: FETCH_SIZE | WRITE_SIZE | SQ_* | TCC_*> -- python bench.py --workload {workload}" + (f" --width {res[0]} --height {res[1]}" if (workload == "helmet" and res[0] == 3840) else "") + (" --denoise" if os.environ.get("PMC_DENOISE") else "") + " --steps 3 --warmup 1 --no-cpu-baseline --also none",
       "note": "hbm_bytes_per_launch = FETCH_SIZE / (counter-over-true ratio of the kernel's access class) + WRITE_SIZE / (ratio of its writes); the raw counters and the bounds [raw, 2 x raw] are kept next to it", "kernels": {}}
for key, match in pick.items():
    cands = sorted((n for n, k in summary.items() if match(n) and "FETCH_SIZE" in k and "WRITE_SIZE" in k),
                   key=lambda n: summary[n].get("avg_us", 0.0) * max(1, summary[n].get("dispatches_pmc", 1)))
    for name, k in ((n, summary[n]) for n in cands[-1:]):
        if True:
            d = max(1, k.get("dispatches_pmc", 1))
            fetch, write = k["FETCH_SIZE"] * 1024 / d, k["WRITE_SIZE"] * 1024 / d
            rf, rw = r_fetch.get(key) or 0.5, r_write.get(key) or 1.0
            out["kernels"][key] = {"kernel": name, "dispatches": d, "fetch_size_bytes_per_launch_raw": round(fetch), "write_size_bytes_per_launch": round(write),
                                   "fetch_counter_over_true": round(rf, 4), "write_counter_over_true": round(rw, 4),
                                   "hbm_bytes_per_launch": round(fetch / rf + write / rw), "hbm_bytes_bounds": [round(fetch + write), round(2 * fetch + write)],
                                   "avg_us": round(k.get("avg_us", 0.0), 1)}
            # from the SQ / TCC passes of the same command (tools/profile.sh): how much of the launch the 1024 SIMDs spend issuing vector
            # instructions -- instructions x (cycles per wave64 instruction) / (SIMDs x duration x 2.4 GHz).  The cycles come from the kernel's
            # static class mix (profiles/r06_valu_mix.json, tools/valu_mix.py): 2 for the full-rate class (f32 fma / mul / add, and / or / xor,
            # integer add, mov), 4 for the rest (min / max, conversions, shifts, selects, compares, packed f32), 8 for transcendentals -- classes
            # timed by tools/microbench_valu.hip.  A model (the dynamic mix of the hot loop may differ); the bounds at 2 and at 4 cycles for every
            # instruction stand beside it -- also the lanes active per vector instruction, and the L2 hit rate
            if k.get("SQ_INSTS_VALU") and k.get("avg_us"):
                per2 = k["SQ_INSTS_VALU"] / d * 2.0 / (1024.0 * k["avg_us"] * 1e-6 * 2.4e9)
                mixk = MIX.get(name)
                if mixk:
                    out["kernels"][key]["issue_frac"] = round(min(1.0, per2 * mixk["avg_cycles"] / 2.0), 3)
                    out["kernels"][key]["issue_cycles_per_instruction"] = mixk["avg_cycles"]
                out["kernels"][key]["issue_frac_bounds"] = [round(per2, 3), round(min(1.0, 2.0 * per2), 3)]
                out["kernels"][key]["valu_insts_per_launch"] = round(k["SQ_INSTS_VALU"] / d)
            if k.get("SQ_THREAD_CYCLES_VALU") and k.get("SQ_INSTS_VALU"):
                out["kernels"][key]["active_lanes"] = round(min(64.0, k["SQ_THREAD_CYCLES_VALU"] / k["SQ_INSTS_VALU"]), 1)  # (the counter runs a few per cent high: a dense kernel reads 64-67)
            if k.get("l2_hit_rate") is not None:
                out["kernels"][key]["l2_hit_rate"] = round(k["l2_hit_rate"], 3)
            if k.get("frac_wait_any") is not None:
                out["kernels"][key]["wave_time_waiting"] = round(k["frac_wait_any"], 3)
K = out["kernels"]


def group(*keys):
    n = sum(K[k]["dispatches"] for k in keys if k in K)
    return round(sum(K[k]["hbm_bytes_per_launch"] * K[k]["dispatches"] for k in keys if k in K) / max(1, n))


# keyed like bench.py's kernel table (one entry per kernel of the step)
out["bench_kernel_traffic"] = {k: K[k]["hbm_bytes_per_launch"] for k in ("trace_primary", "shade_first", "trace_closest", "shade", "trace_shadow", "finish_sample") if k in K}
print(json.dumps(out, indent=1))
