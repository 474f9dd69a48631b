"""Guards what round 3 found in the compiled shade path (LABNOTES.md section 4, "What the compiled code showed"): the descriptors of the
non-inlined helpers and of the shade kernels come through the scalar cache, the tables are read as global memory, helper results
travel in registers.  None of it changes a pixel, so no parity test would notice it coming back -- the generated code does.
CPU-only: hipcc cross-compiles gfx950 without a GPU (tools/isa_census.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_census  # noqa: E402


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(isa_census.HIPCC):
        pytest.skip("no hipcc")
    return isa_census.census()


def _one(table, prefix):
    hits = [v for k, v in table.items() if k.startswith(prefix)]
    assert hits, (prefix, sorted(table)[:20])
    return hits


def test_helpers_read_descriptors_through_the_scalar_cache(table):
    # (pt::sampleLightsCall: inlined at its call sites since round 6 -- its descriptor reads are then the shade kernels' own, checked below: no vector
    #  load of a uniform address, and well over a hundred scalar loads in the kernels that carry its body)
    for name, min_s_loads in (("pt::evalPhysicalSky", 20), ("missEnvironmentCall", 10), ("primaryMissBackplateCall", 4), ("pt::getShadowTransmissionCall", 4)):
        for s in _one(table, name):
            assert s["flat_load"] == 0 and s["flat_store"] == 0 and s["drain"] == 0, (name, s)
            assert s["s_load"] >= min_s_loads, (name, s)


def test_texture_fetch_reads_global_memory(table):
    for s in _one(table, "pt::getTextureRef"):
        assert s["flat_load"] == 0 and s["drain"] == 0 and s["scratch"] == 0, s


def test_shade_kernels_have_no_generic_loads_and_no_uniform_vector_loads(table):
    kernels = {k: v for k, v in table.items() if k.startswith("k_shade<")}
    assert len(kernels) == 8
    for name, s in kernels.items():
        assert s["flat_load"] <= 1 and s["flat_store"] == 0 and s["drain"] == 0, (name, s)  # (one: the sRGB table's copy into LDS)
        # sc. / fc. after a store or a call: scalar, not a vector load of a uniform address.  (The later-bounce kernels have two loads in SGPR-base form that are
        # real gathers: finishMisses reads the listed entries' slot / ray through a uniform queue base + the per-lane position it took from LDS.)
        assert s["sgpr_base_load"] <= (2 if name.endswith(", false>") else 0), (name, s)
        assert s["s_load"] >= 90, (name, s)         # ... and sampleLights' light / sky tables arrive through the scalar cache (its body is inlined here)
    for name in ("k_shade<false, true, true>", "k_shade<true, true, true>"):  # the bounce-0 launch of the common flavour: nothing spilled
        assert kernels[name]["vgpr_spill"] == 0 and kernels[name]["scratch"] == 0, (name, kernels[name])


def test_finish_and_resolve_kernels(table):
    for s in _one(table, "k_finish_sample"):
        assert s["flat_load"] == 0 and s["drain"] == 0 and s["vgpr_spill"] == 0, s
    for s in _one(table, "k_shadow_resolve<"):
        assert s["flat_load"] == 0 and s["drain"] == 0, s


def test_walk_kernels_keep_their_wave_budget(table):
    """Four waves per SIMD: 128 VGPRs.  The spills they have belong to the ray feed, not to the node loop (checked by reading
    the assembly in round 3); a jump in their number means the inner loop started spilling."""
    for name, s in table.items():
        if name.startswith("k_trace_closest<true") or name.startswith("k_trace_shadow<true, 0") or name.startswith("k_trace_shadow<true, 1"):
            assert s["vgpr"] <= 128 and s["vgpr_spill"] <= 24 and s["flat_load"] == 0, (name, s)
