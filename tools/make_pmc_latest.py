#!/usr/bin/env python3
"""profiles/pmc_latest.json from a tools/summarize_pmc.py summary: per-kernel HBM bytes per launch (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE) and the per-group averages bench.py quotes as `roofline.traffic`.
usage: tools/make_pmc_latest.py gpurun_out/prof_r02_helmet/summary.json helmet 32 [round] > profiles/pmc_latest.json"""
import json, sys

summary, workload, frames = json.load(open(sys.argv[1]))["kernels"], sys.argv[2], int(sys.argv[3])
# the timed (non-counting) template instances: k_shade<COUNT, SIMPLE, FIRST>, k_trace_closest<WIDE, HAS_ALPHA, COUNT>,
# k_trace_primary<HAS_ALPHA, COUNT>, k_trace_shadow<WIDE, MODE, COUNT>
pick = {"shade_first": lambda n: n.startswith("k_shade<false") and n.endswith("true>"),
        "shade": lambda n: n.startswith("k_shade<false") and n.endswith("false>"),
        "trace_closest": lambda n: n.startswith("k_trace_closest<") and n.endswith("false>"),
        "trace_primary": lambda n: n.startswith("k_trace_primary<") and n.endswith("false>"),
        "trace_shadow": lambda n: n.startswith("k_trace_shadow<") and n.endswith("false>"),
        "shadow_resolve": lambda n: n.startswith("k_shadow_resolve"),
        "finish_sample": lambda n: n.startswith("k_finish_sample"),
        "generate": lambda n: n.startswith("k_generate")}
out = {"round": int(sys.argv[4]) if len(sys.argv) > 4 else 2, "workload": workload, "frames_in_flight": frames, "resolution": [1920, 1080],
       "command": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --workload {workload} --steps 3 --warmup 1 --no-cpu-baseline",
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported (KB).", "kernels": {}}
for key, match in pick.items():
    for name, k in summary.items():
        if match(name) and "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            d = max(1, k.get("dispatches_pmc", 1))
            fetch, write = k["FETCH_SIZE"] * 1024 / d, k["WRITE_SIZE"] * 1024 / d
            out["kernels"][key] = {"kernel": name, "dispatches": d, "fetch_size_bytes_per_launch_raw": round(fetch), "write_size_bytes_per_launch": round(write),
                                   "hbm_bytes_per_launch": round(2 * fetch + write), "avg_us": round(k.get("avg_us", 0.0), 1)}
K = out["kernels"]


def group(*keys):
    n = sum(K[k]["dispatches"] for k in keys if k in K)
    return round(sum(K[k]["hbm_bytes_per_launch"] * K[k]["dispatches"] for k in keys if k in K) / max(1, n))


# keyed like bench.py's kernel table (one entry per kernel of the step)
out["bench_kernel_traffic"] = {k: K[k]["hbm_bytes_per_launch"] for k in ("trace_primary", "shade_first", "trace_closest", "shade", "trace_shadow") if k in K}
print(json.dumps(out, indent=1))
