"""bench.py's reporting contract, checked without a GPU: the roofline table cannot print a fraction above 1 (round-3 review: the algorithmic
byte model printed 1.05 / 1.10 / 1.27), every configuration of the default run finds its committed counter passes, the rank launcher
refuses what it cannot run, and the committed round-4 line obeys all of it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _timing(ms=10.0):
    return {"tracePrimaryMs": ms, "tracePrimaryLaunches": 1, "shadeFirstMs": ms, "shadeFirstLaunches": 1, "traceClosestMs": 3 * ms, "traceClosestLaunches": 3,
            "shadeMs": 3 * ms, "shadeLaunches": 3, "traceShadowMs": 2 * ms, "traceShadowLaunches": 2, "accumulateMs": 1.0, "totalMs": 9 * ms}


def _counters(scale=1.0):
    return {"cameraPaths": 2.0e6 * scale, "segments": 7.0e6 * scale, "surfaceHits": 6.5e6 * scale, "shadowRays": 3.5e6 * scale, "nodesPrimary": 9.0e5 * scale,
            "trisPrimary": 6.0e5 * scale, "nodesClosest": 1.0e8 * scale, "trisClosest": 4.0e7 * scale, "nodesShadow": 4.5e7 * scale, "trisShadow": 2.0e7 * scale,
            "textureTaps": 7.0e6 * scale}


def test_every_default_configuration_finds_its_counter_passes():
    """profiles/pmc_latest_<workload>.json must match the frames in flight and the resolution bench.py actually runs: a changed default
    without new counter passes would silently print traffic: null."""
    lines = [(n, bench.ALSO_LINES[n]) for n in bench.ALSO_DEFAULT.split(",")] + [("headline", (bench.NORTH_STAR["workload"], 0, 0, False, True))]
    for name, (wl, w, h, _den, _par) in lines:
        cfg = bench.WORKLOADS[wl]
        W, H = w or cfg["width"], h or cfg["height"]
        F = bench.frames_in_flight(cfg.get("in_flight", bench.IN_FLIGHT_DEFAULT), W, H)
        pmc = bench.load_pmc(wl, F, W, H)
        if wl in bench.PMC_PENDING:
            continue
        assert pmc is not None, (name, wl, F, W, H)
        for k in ("trace_primary", "shade_first", "trace_closest", "shade", "trace_shadow"):
            e = pmc["kernels"][k]
            assert e["hbm_bytes_per_launch"] > 0 and 0.0 < e["issue_frac"] <= 1.0 and e["issue_frac_bounds"][0] <= e["issue_frac"] <= e["issue_frac_bounds"][1] + 1e-9, (name, k, e)
            assert 1.0 <= e["active_lanes"] <= 64.0


def test_kernel_table_fractions_are_fractions():
    """hbm rows: frac = counter traffic / launch time / 8 TB/s (None without counters), the algorithmic figure stands beside it under its own
    name; valu rows: useful lane-operations against the vector peak, with the per-kernel instruction counts."""
    per_frame, first = _counters(), {k: v * (0.3 if k != "cameraPaths" else 1.0) for k, v in _counters().items()}
    F = bench.IN_FLIGHT_DEFAULT
    pmc = bench.load_pmc("atrium", F, 1920, 1080)
    assert pmc is not None
    rows = bench.kernel_table(per_frame, first, _timing(), frames=F, in_flight=F, pmc=pmc)
    for name, r in rows.items():
        assert r["frac"] is None or 0.0 <= r["frac"], name
        if r["bound"] == "hbm" and r["traffic"]:
            assert r["frac"] == pytest.approx(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, rel=1e-3)
            assert "algorithmic_frac" in r and "algorithmic_GBps" in r
        if r["bound"] == "valu":
            assert r["peak"] == pytest.approx(78.64, abs=0.01) and 0.0 < r["issue_frac"] <= 1.0
    # the packet walk is charged its interval test (50 instructions of all 64 lanes per node), not the per-ray test's count
    tp = rows["trace_primary"]
    assert tp["useful_laneops_per_launch"] == round(64 * (per_frame["nodesPrimary"] * bench.VALU_PER_PACKET_NODE + per_frame["trisPrimary"] * bench.VALU_PER_TRI) * F)
    # no counters (another resolution / batch size): no invented fraction for the memory-bound kernels
    rows = bench.kernel_table(per_frame, first, _timing(), frames=128, in_flight=96, pmc=None)
    assert rows["shade"]["frac"] is None and rows["shade"]["traffic"] is None and rows["shade"]["algorithmic_frac"] > 0
    assert rows["trace_closest"]["frac"] > 0  # (useful work needs no counters)


def test_committed_round4_line_obeys_the_contract():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")).read().strip().splitlines()[-1])
    assert line["unit"] == "Msamples/s" and line["n_gpus"] == 1 and line["config"]["workload"].startswith("configs[2]") and line["north_star"]["value"] == line["value"]
    lines = {"headline": line, **line["also"]}
    assert set(line["also"]) == {"helmet", "helmet_4k", "street", "glass_denoise"}  # (the default set of that round)
    for name, ln in lines.items():
        assert ln["roofline"]["traffic"] is not None, name  # no line without its counter passes
        for k, r in ln["kernels"].items():
            assert r["frac"] is not None and 0.0 < r["frac"] <= 1.0, (name, k, r["frac"])
            if "issue_frac" in r:
                assert 0.0 < r["issue_frac"] <= 1.0, (name, k)
        if name != "helmet_4k":  # (the same scene as "helmet": one parity leg)
            p = ln["parity"]
            assert p["spp"] == bench.WORKLOADS[bench.ALSO_LINES[name][0] if name in bench.ALSO_LINES else "atrium"]["spp"]
            assert p["rel_l2"] <= 1e-3 and p["within_tolerance"], (name, p["rel_l2"])
            assert ln["cpu_baseline"]["kind"] == "port" and ln["cpu_baseline"]["cores"] >= 1


def _committed_full_records():
    prof = os.path.join(ROOT, "profiles")
    return sorted(f for f in os.listdir(prof) if f.endswith("_bench_default.json") or f.endswith("_bench_full.json"))


@pytest.mark.parametrize("record", _committed_full_records())
def test_final_line_is_compact(record):
    """Round 4's default run printed ONE 33 KB line and the driver's record of it could not be parsed.  The last stdout line is now built by
    compact_line() from the full record (which goes to bench_full.json and stderr): under 6 KB -- in practice under 4 -- with every field the
    driver and the review read, numbers only in `roofline`, and one short record per other configuration."""
    full = json.loads(open(os.path.join(ROOT, "profiles", record)).read().strip().splitlines()[-1])
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.FINAL_LINE_LIMIT and len(text) < 4096, len(text)
    assert "\n" not in text
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "north_star", "also"):
        assert k in line, k
    for k in ("workload", "resolution", "spp_per_step", "frames_in_flight", "max_depth", "library"):
        assert k in line["config"], k
    roof = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "hbm_frac", "avg_launch_ms", "launches", "traffic"):
        assert k in roof, k
    assert all(not isinstance(v, str) or k in ("bound", "kernel", "unit") for k, v in roof.items())  # numbers only
    assert 0.0 < roof["frac"] <= 1.0 and 0.0 < roof["hbm_frac"] <= 1.0
    assert roof["hbm_frac"] == pytest.approx(roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, rel=1e-3)
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["value"] > 0
    assert line["parity"]["within_tolerance"] and line["parity"]["rel_l2"] <= 1e-3
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    for name, e in line["also"].items():
        assert e["value"] == full["also"][name]["value"] and set(e["roofline"]) >= {"kernel", "bound", "frac"}, name


def test_emit_prints_the_compact_line_last(tmp_path, capfd):
    full = json.loads(open(os.path.join(ROOT, "profiles", _committed_full_records()[-1])).read().strip().splitlines()[-1])
    os.environ["BENCH_FULL"] = str(tmp_path / "full.json")
    try:
        bench.emit(full)
    finally:
        del os.environ["BENCH_FULL"]
    out, err = capfd.readouterr()
    assert out.count("\n") == 1 and json.loads(out) == bench.compact_line(full)
    assert json.loads(open(tmp_path / "full.json").read()) == full and json.loads(err.strip().splitlines()[-1]) == full


def test_committed_round5_record_obeys_the_contract():
    """The full record of the round's last default run (what bench_full.json held): every kernel fraction inside (0, 1], every configuration with its counter
    passes, parity within the tolerance at the configuration's own sample count.  Schema and consistency only: a committed artifact cannot regress, so no
    performance threshold is asserted on it (the live numbers are the driver's BENCH_rNN.json)."""
    line = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().strip().splitlines()[-1])
    assert line["unit"] == "Msamples/s" and line["n_gpus"] == 1 and line["config"]["workload"].startswith("configs[2]") and line["north_star"]["value"] == line["value"]
    assert set(line["also"]) == {"helmet", "helmet_4k", "street", "glass_denoise"}  # (the default set of that round)
    for name, ln in {"headline": line, **line["also"]}.items():
        assert ln["roofline"]["traffic"] is not None, name
        for k, r in ln["kernels"].items():
            assert r["frac"] is not None and 0.0 < r["frac"] <= 1.0, (name, k, r["frac"])
        if name != "helmet_4k":
            assert ln["parity"]["rel_l2"] <= 1e-3 and ln["parity"]["within_tolerance"], name
            assert ln["cpu_baseline"]["kind"] == "port" and "-march=native" in ln["cpu_baseline"]["flags"]


def test_committed_round6_record_obeys_the_contract():
    """Schema and consistency of the round's last default run (profiles/r06_bench_default.json + the compact line the driver parses): strong scaling at N = 1, the
    uncut-geometry value beside the headline, every default `also` line present with its counter passes, fractions inside (0, 1], parity legs reported with their tolerance flag."""
    full = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_default.json")).read().strip().splitlines()[-1])
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read().strip().splitlines()[-1])
    assert line == bench.compact_line(full)
    assert full["scaling"] == "strong" and full["n_gpus"] == 1 and full["config"]["frames_in_flight"] == bench.IN_FLIGHT_DEFAULT and full["config"]["spp_per_step"] == 256
    assert full["value_uncut_geometry"]["value"] > 0 and full["value_uncut_geometry"]["scene_triangles"] < full["config"]["scene_triangles"]
    assert set(full["also"]) == set(bench.ALSO_DEFAULT.split(","))
    for name, ln in {"headline": full, **full["also"]}.items():
        assert "error" not in ln, name
        assert ln["roofline"]["traffic"] is not None, name
        for k, r in ln["kernels"].items():
            assert r["frac"] is not None and 0.0 < r["frac"] <= 1.0, (name, k, r["frac"])
        if "parity" in ln:
            assert ln["parity"]["within_tolerance"] == (ln["parity"]["rel_l2"] <= 1e-3) and ln["parity"]["spp"] == bench.WORKLOADS[bench.ALSO_LINES[name][0] if name != "headline" else "atrium"]["spp"]
    assert full["also"]["glass_denoise"]["config"]["scene_triangles"] > 1000000 and full["also"]["glass_denoise"]["per_frame"]["textureTaps"] > 0  # the dragon + textured slabs


def test_more_ranks_than_devices_is_refused_before_anything_runs():
    """`python bench.py --gpus N` starts its own ranks -- and on a box with fewer devices (here: none) it must exit non-zero without a JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0", "--workload", "box"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in (r.stdout + r.stderr)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_step_shape_of_strong_and_weak_scaling():
    """Frame bookkeeping of `bench.py --gpus N`: strong = the configuration's own frames per step at every world size, all of them in flight (a GPU holds
    1/N of their path slots); weak = frames per step and frames in flight grow with N.  The two agree at N = 1, and both respect the path-slot budget."""
    W, H = 1920, 1080
    assert bench.step_shape("strong", 1, 128, 256, W, H) == bench.step_shape("weak", 1, 128, 256, W, H) == (128, 256)
    for n in (2, 4, 8):
        F, step = bench.step_shape("strong", n, 128, 256, W, H)
        assert step == 256 and F == 256                    # the step never grows; its 256 frames are in flight together ...
        assert F / n == 256 / n                            # ... which is 128 / 64 / 32 frames' worth of path slots per GPU
        Fw, stepw = bench.step_shape("weak", n, 128, 256, W, H)
        assert stepw == 256 * n and Fw == min(1024, 128 * n) and Fw / n == 128
    # 4K: the slot budget caps a single GPU at 64 frames in flight; eight GPUs hold a whole 256-frame step between them
    assert bench.step_shape("strong", 1, 128, 256, 3840, 2160) == (64, 256)
    assert bench.step_shape("strong", 8, 128, 256, 3840, 2160) == (256, 256)
    for n in (1, 2, 4, 8):
        for mode in ("strong", "weak"):
            F, _ = bench.step_shape(mode, n, 128, 256, 3840, 2160)
            assert F * 3840 * 2160 / n <= bench.SLOT_BUDGET * 1.02 and (F % 64 == 0 or F < 64)
