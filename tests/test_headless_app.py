"""Headless entry + BenchmarkController: log format compatible with the reference's own parser
(utils/benchmark/benchmark_results.py) and, on a GPU, an end-to-end run of the reference's benchmark command line."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from vk_gltf_renderer_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(capi.LIB_DIR, "mi_gltf_renderer")
SUMMARY_KEYS = {"type", "frames", "maxFrames", "ptSamples", "effective_spp", "measured_effective_spp", "resolution_w", "resolution_h", "wall_ms",
                "ms_per_frame", "total_wall_ms", "total_ms_per_frame", "warmup_frames", "measured_frames", "throughput_MSps", "spp_per_sec", "schema"}


def _records(log):
    return [json.loads(line.split("BENCHMARK_JSON ", 1)[1]) for line in log.splitlines() if "BENCHMARK_JSON " in line]


def _run(args):
    if not os.path.exists(EXE):
        pytest.skip("mi_gltf_renderer not built (run __graft_entry__.build())")
    env = dict(os.environ, LD_LIBRARY_PATH=capi.LIB_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    return subprocess.run([EXE] + args, capture_output=True, text=True, env=env, timeout=300)


def test_benchmark_log_schema():
    """Keys and arithmetic of src/benchmarking.cpp:248-304 (warm-up frame excluded, effective spp = frames x ptSamples)."""
    r = _run(["--benchmarkSelftest"])
    assert r.returncode == 0, r.stderr
    recs = _records(r.stdout)
    assert [x["type"] for x in recs] == ["headless_start", "headless_progress", "headless_progress", "headless_summary"]
    s = recs[-1]
    assert set(s) == SUMMARY_KEYS and all(x["schema"] == 1 for x in recs)
    assert (s["frames"], s["ptSamples"], s["effective_spp"], s["warmup_frames"], s["measured_frames"], s["measured_effective_spp"]) == (3, 2, 6, 1, 2, 4)
    legacy = re.search(r"HEADLESS_SUMMARY frames=3 maxFrames=3 ptSamples=2 effective_spp=6 measured_effective_spp=4 resolution=64x32 wall_ms=", r.stdout)
    assert legacy
    ref_parser = "/root/reference/utils/benchmark"
    if os.path.isdir(ref_parser):  # only in the authoring container; the GPU box has no /root/reference
        sys.path.insert(0, ref_parser)
        import benchmark_results
        parsed = benchmark_results.parse_headless_summary(r.stdout)
        assert parsed is not None and parsed["frames"] == "3" and parsed["measured_frames"] == "2"
        assert float(parsed["throughput_MSps"]) == pytest.approx(s["throughput_MSps"])


def test_cli_rejects_unknown_and_non_headless():
    assert _run(["--noSuchFlag", "1"]).returncode == 2
    assert _run(["--scenefile", "x.glb"]).returncode == 2  # windowed mode does not exist here


@pytest.mark.gpu
def test_headless_benchmark_command_line(tmp_path, assets):
    """The reference's recommended benchmark invocation (docs/benchmarking.md:16-23) runs unchanged."""
    out = tmp_path / "box.hdr"
    r = _run(["--headless", "--size", "256", "256", "--scenefile", os.path.join(assets, "Box.glb"), "--hdrfile", os.path.join(assets, "std_env.hdr"),
              "--frames", "8", "--maxFrames", "8", "--ptSamples", "2", "--ptAdaptiveSampling", "0", "--renderSystem", "0", "--envSystem", "1",
              "--ptMaxDepth", "4", "--output", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    s = _records(r.stdout)[-1]
    assert s["type"] == "headless_summary" and s["effective_spp"] == 16 and s["measured_frames"] == 7 and s["throughput_MSps"] > 0
    assert out.exists() and out.read_bytes().startswith(b"#?RADIANCE")
    # the saved HDR is the 16-spp accumulation the library holds: compare against the C-ABI path with the same parameters
    import parity_util as pu
    from vk_gltf_renderer_amd import pathtracer as ptmod
    setup = pu.Setup(os.path.join(assets, "Box.glb"), 256, 256, hdr_path=os.path.join(assets, "std_env.hdr"), max_depth=4, spp_per_frame=2)
    img = pu.render_gpu(setup, 8, collect_counters=False)["accum"]
    hdr = ptmod.HdrEnvironment(path=str(out))
    e = hdr.env.contents
    saved = np.ctypeslib.as_array(e.rgba, shape=(e.height, e.width, 4))[..., :3]
    assert np.abs(saved - img[..., :3]).max() <= img[..., :3].max() / 128 + 1e-3  # RGBE has an 8-bit mantissa


@pytest.mark.gpu
def test_headless_frames_in_flight_same_image(tmp_path, assets):
    """--framesInFlight batches the app frames of a headless run into shared wavefront launches (default 32): the saved accumulation
    is the one of the frame-by-frame run (--framesInFlight 1, the reference's loop) bit for bit, and the summary counts the same frames."""
    outs, recs = [], []
    for n in (1, 5, 32):
        out = tmp_path / f"f{n}.hdr"
        r = _run(["--headless", "--size", "200", "120", "--scenefile", os.path.join(assets, "shader_ball.gltf"), "--hdrfile", os.path.join(assets, "std_env.hdr"),
                  "--frames", "23", "--maxFrames", "23", "--ptSamples", "1", "--ptAdaptiveSampling", "0", "--envSystem", "1", "--framesInFlight", str(n),
                  "--output", str(out)])
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out.read_bytes())
        recs.append(_records(r.stdout)[-1])
    assert outs[0] == outs[1] == outs[2]
    assert all((s["frames"], s["effective_spp"], s["warmup_frames"], s["measured_frames"]) == (23, 23, 1, 22) for s in recs)


@pytest.mark.gpu
def test_adaptive_sampling_controller(assets):
    """Auto SPP (reference: PathTracer::updateAdaptiveSampling, src/renderer_pathtracer.cpp:1326-1374): with the slowest target
    (10 frames per second) a 64x64 frame leaves headroom every frame, so the samples per frame climb by one per frame from frame 5 on;
    an explicit --ptSamples switches the controller off."""
    base = ["--headless", "--size", "64", "64", "--scenefile", os.path.join(assets, "Box.glb"), "--hdrfile", os.path.join(assets, "std_env.hdr"), "--envSystem", "1",
            "--frames", "20", "--maxFrames", "20", "--output", "/tmp/adaptive_test.hdr"]
    r = _run(base + ["--ptAdaptiveSampling", "1", "--ptPerformanceTarget", "3"])
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"ADAPTIVE_SAMPLING samples_per_frame_at_end=(\d+) total_samples=(\d+)", r.stdout)
    # frames 0..4 run 1 spp; frames 5..19 see headroom and add one sample each: 2, 3, ... 16
    assert m and int(m.group(1)) == 16 and int(m.group(2)) == 5 + sum(range(2, 17)), r.stdout[-800:]
    r = _run(base + ["--ptSamples", "2"])
    assert r.returncode == 0 and "ADAPTIVE_SAMPLING" not in r.stdout
    assert _records(r.stdout)[-1]["effective_spp"] == 40


def _read_png_rgba8(path):
    """Minimal PNG reader for what GltfRenderer::savePng writes (8-bit RGBA, filter 0 on every row)."""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(b):
        n, tag = struct.unpack(">I4s", b[pos:pos + 8])
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", b[pos + 8:pos + 18])
            assert (depth, ctype) == (8, 6)
        if tag == b"IDAT":
            idat += b[pos + 8:pos + 8 + n]
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
def test_tonemapped_output_and_denoiser_cadence(tmp_path, assets):
    """--output x.png saves eImgTonemapped: the device tonemapper with the reference's tm* switches (src/renderer.cpp:173-179), fed
    by the denoised image when the denoiser is on (src/renderer.cpp:1006-1016).  The denoiser keeps the OptiX adapter's switches:
    auto-denoise every optixAutoDenoiseInterval frames (src/optix_denoiser.cpp:773-800), plus the final frame of a headless run."""
    from vk_gltf_renderer_amd import pathtracer as ptmod
    import parity_util as pu
    out = tmp_path / "box.png"
    base = ["--headless", "--size", "96", "64", "--scenefile", os.path.join(assets, "Box.glb"), "--hdrfile", os.path.join(assets, "std_env.hdr"), "--envSystem", "1",
            "--frames", "12", "--maxFrames", "12", "--ptSamples", "1", "--ptMaxDepth", "3", "--output", str(out)]
    r = _run(base + ["--tmMethod", "3", "--tmExposure", "1.5", "--tmAutoExposure", "0", "--tmSaturation", "0.9"])
    assert r.returncode == 0 and "DENOISER" not in r.stdout, r.stdout + r.stderr
    png = _read_png_rgba8(out)
    assert png.shape == (64, 96, 4) and (png[..., 3] == 255).all()
    # the same frames through the C-ABI, tonemapped by the library with the same parameters
    setup = pu.Setup(os.path.join(assets, "Box.glb"), 96, 64, hdr_path=os.path.join(assets, "std_env.hdr"), max_depth=3)
    t = ptmod.PathTracer(setup.scene)
    try:
        t.set_environment(setup.hdr)
        t.resize(96, 64)
        t.set_frame_info(setup.frame_info)
        t.set_sky(setup.sky)
        for f in range(12):
            t.render_frame(setup.frame_params(f, f))
        want = t.tonemap(method="aces", exposure=1.5, saturation=0.9)
    finally:
        t.close()
    assert np.abs(png[..., :3].astype(int) - want[..., :3].astype(int)).max() <= 1
    r = _run(base + ["--optixEnable", "1", "--optixAutoDenoiseInterval", "5"])
    assert r.returncode == 0, r.stdout + r.stderr
    # frames 5 and 10 by the cadence, the final (12th) frame by the headless save
    assert "DENOISER passes=3 final_image=denoised" in r.stdout, r.stdout[-600:]
    den = _read_png_rgba8(out)
    assert den.shape == (64, 96, 4) and np.abs(den.astype(int) - png.astype(int)).max() > 0


QUICK_CFG = os.path.join(ROOT, "tests", "golden", "quick_sequences.cfg")  # the reference's utils/benchmark/quick.cfg, shortened


def test_sequencer_log_parses_with_the_reference_tooling():
    """tests/golden/sequencer_log_mi355x.txt is the output of `mi_gltf_renderer --benchmark 1 --sequencefile quick_sequences.cfg` on an
    MI355X (test_scripted_sequencer runs the same command on the GPU box): the reference's own parser must read it."""
    log = open(os.path.join(ROOT, "tests", "golden", "sequencer_log_mi355x.txt")).read()
    ref_parser = "/root/reference/utils/benchmark"
    if not os.path.isdir(ref_parser):
        pytest.skip("the reference's benchmark scripts are only present in the authoring container")
    sys.path.insert(0, ref_parser)
    import benchmark_results
    parsed = benchmark_results.parse_benchmark(log, "shader_ball")
    assert [(b["id"], b["name"]) for b in parsed] == [(0, "Warmup"), (1, "PT 1spp"), (2, "Rasterizer")]
    assert parsed[1]["timers"]["PathTracer::onRender"]["VK"] == pytest.approx(1.331) and parsed[1]["timers"]["PathTracer::onRender"]["CPU"] > 0
    assert parsed[1]["memory"]["Scene"]["Device Used"] == 28643577 and parsed[1]["memory"]["PathTracer"]["Device Allocated"] == 9172120
    assert parsed[2]["timers"] == {}
    assert benchmark_results.primary_timer_ms(parsed[1], "VK", ["GltfRenderer::onRender", "PathTracer::onRender"]) == pytest.approx(1.331)


@pytest.mark.gpu
def test_scripted_sequencer(tmp_path, assets):
    """`--benchmark 1 --sequencefile quick.cfg` (docs/benchmarking.md "Scripted sequencer"): per sequence a ParameterSequence timer block
    and the memory snapshot (BENCHMARK_ADV + BENCHMARK_JSON sequence_memory) in the shape utils/benchmark/benchmark_results.py parses."""
    r = _run(["--benchmark", "1", "--sequencefile", QUICK_CFG, "--size", "160", "120", "--scenefile", os.path.join(assets, "shader_ball.gltf"),
              "--hdrfile", os.path.join(assets, "std_env.hdr")])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    # the consumer's own regular expressions (benchmark_results.py parse_benchmark / _parse_legacy_memory_records)
    seqs = re.split(r'ParameterSequence\s+(\d+)\s+"([^"]+)"\s*=', r.stdout)[1:]
    assert [(int(seqs[i]), seqs[i + 1]) for i in range(0, len(seqs), 3)] == [(0, "Warmup"), (1, "PT 1spp"), (2, "Rasterizer")]
    timer = re.compile(r'Timer\s+"([^"]+)"\s*;\s*GPU;\s*avg\s+(\d+);.*?CPU;\s*avg\s+(\d+);')
    t = [timer.findall(seqs[i + 2].split("BENCHMARK_ADV")[0]) for i in range(0, len(seqs), 3)]
    assert [x[0][0] for x in t[:2]] == ["PathTracer::onRender"] * 2 and t[2] == []  # no rasterizer on this path: no timers
    assert all(int(x[0][1]) > 0 and int(x[0][2]) > 0 for x in t[:2])
    legacy = re.findall(r"Memory (\w+); Host used\s+(\d+); Device Used\s+(\d+); Device Allocated\s+(\d+);", r.stdout)
    assert [m[0] for m in legacy] == ["Scene", "PathTracer"] * 3 and all(int(m[2]) > 0 for m in legacy[:4])
    mem = [x for x in _records(r.stdout) if x["type"] == "sequence_memory"]
    assert [x["id"] for x in mem] == [0, 1, 2] and all(x["schema"] == 1 for x in mem)
    assert [s["category"] for s in mem[1]["memory"]] == ["Scene", "PathTracer"]
    assert int(legacy[2][2]) == mem[1]["memory"][0]["device_used"] > 100000  # the shader ball's geometry + BVH
    ref_parser = "/root/reference/utils/benchmark"
    if os.path.isdir(ref_parser):  # authoring container only
        sys.path.insert(0, ref_parser)
        import benchmark_results
        parsed = benchmark_results.parse_benchmark(r.stdout, "shader_ball")
        assert [b["name"] for b in parsed] == ["Warmup", "PT 1spp", "Rasterizer"]
        assert parsed[1]["timers"]["PathTracer::onRender"]["VK"] > 0 and parsed[1]["memory"]["Scene"]["Device Used"] > 100000


def test_sequencer_argument_errors():
    assert _run(["--benchmark", "1", "--scenefile", "x.glb"]).returncode == 2  # no script
    assert _run(["--benchmark", "1", "--sequencestring", 'SEQUENCE "a" --sequenceframes 1']).returncode == 2  # no scene


def test_png_and_jpeg_writers(tmp_path):
    """The headless run saves eImgTonemapped as .png or, like the reference by default, .jpg (src/renderer.cpp:557-573): both writers on a
    synthetic image (no GPU), read back with Pillow -- the PNG exactly, the baseline JPEG (quality 90, 4:4:4) within coding error -- and
    the JPEG also with this repo's own decoder."""
    PIL_Image = pytest.importorskip("PIL.Image")
    prefix = str(tmp_path / "img")
    r = _run(["--saveSelftest", prefix])
    assert r.returncode == 0, r.stdout + r.stderr
    yy, xx = np.mgrid[0:61, 0:83]
    want = np.stack([(127 + 120 * np.sin(xx / 9.0)).astype(np.uint8), (127 + 120 * np.cos(yy / 7.0 + xx / 23.0)).astype(np.uint8), ((xx * 3 + yy * 2) % 256).astype(np.uint8)], -1)
    png = np.asarray(PIL_Image.open(prefix + ".png").convert("RGBA"))
    assert png.shape == (61, 83, 4) and (png[..., :3] == want).all() and (png[..., 3] == 255).all()
    jim = PIL_Image.open(prefix + ".jpg")
    assert jim.format == "JPEG" and jim.size == (83, 61) and jim.mode == "RGB"
    jpg = np.asarray(jim).astype(int)
    err = np.abs(jpg - want.astype(int))
    assert err.mean() < 3.0 and np.percentile(err, 99) < 24, (err.mean(), np.percentile(err, 99))  # the sawtooth blue channel has hard edges
    # and through the glTF front end's own JPEG decoder
    from vk_gltf_renderer_amd import pathtracer as ptmod
    from vk_gltf_renderer_amd import scenegen
    b = scenegen.GlbBuilder()
    t = b.texture(b.image_bytes(open(prefix + ".jpg", "rb").read(), "image/jpeg"))
    b.material({"pbrMetallicRoughness": {"metallicRoughnessTexture": {"index": t}}})
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    b.node(mesh=b.mesh([b.primitive(pos, np.array([0, 1, 2]), material=0)]))
    sc = ptmod.Scene(b.save(str(tmp_path / "j.glb")))
    tex = sc.desc.contents.textures[0]
    own = np.ctypeslib.as_array(tex.levels[0], shape=(61, 83, 4)).astype(int)
    assert np.abs(own[..., :3] - jpg).max() <= 4  # stb-style reconstruction vs libjpeg: rounding only


@pytest.mark.gpu
def test_animation_switches(tmp_path, assets):
    """--animTime poses the clip once and accumulates the still scene; --animStep advances the clip every app frame and restarts
    the accumulation like GltfRenderer::onRender does after updateAnimation (reference: src/renderer.cpp:657-662, :2065-2170).
    Both are checked against the C-ABI path on the scene posed through mi_scene_update_animation."""
    import parity_util as pu
    from vk_gltf_renderer_amd import pathtracer as ptmod
    from vk_gltf_renderer_amd import scenegen
    glb = scenegen.scene_animated(str(tmp_path / "sculpture.glb"))
    hdr = os.path.join(assets, "std_env.hdr")
    common = ["--headless", "--size", "192", "128", "--scenefile", glb, "--hdrfile", hdr, "--ptSamples", "1", "--ptAdaptiveSampling", "0", "--envSystem", "1",
              "--ptMaxDepth", "4"]

    def saved(path):
        h = ptmod.HdrEnvironment(path=str(path))  # owns the pixels: keep it alive until they are copied
        e = h.env.contents
        return np.ctypeslib.as_array(e.rgba, shape=(e.height, e.width, 4))[..., :3].copy()

    def expected(time, frames):
        st = pu.Setup(glb, 192, 128, hdr_path=hdr, max_depth=4)
        assert st.scene.update_animation(0, time)
        return pu.render_gpu(st, frames, collect_counters=False)["accum"][..., :3]

    out = tmp_path / "scrub.hdr"
    r = _run(common + ["--frames", "6", "--maxFrames", "6", "--animTime", "0.83", "--output", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    want = expected(0.83, 6)
    assert np.abs(saved(out) - want).max() <= want.max() / 128 + 1e-3
    still = tmp_path / "still.hdr"
    assert _run(common + ["--frames", "6", "--maxFrames", "6", "--output", str(still)]).returncode == 0
    assert np.abs(saved(still) - want).max() > 0.05  # the rest pose is another image

    out = tmp_path / "play.hdr"
    r = _run(common + ["--frames", "5", "--maxFrames", "100", "--animStep", "0.25", "--output", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    assert _records(r.stdout)[-1]["frames"] == 5
    want = expected(1.25, 1)  # five steps of 0.25 s from the clip's start; every step restarts the accumulation
    assert np.abs(saved(out) - want).max() <= want.max() / 128 + 1e-3
