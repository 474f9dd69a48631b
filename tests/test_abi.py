"""The C-ABI shared libraries load and export every symbol include/*.h declares (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from vk_gltf_renderer_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"MI_PT_API[^;(]*?\b(mi_[a-z0-9_]+)\s*\(", text)))


def _exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_host_library_exports_header(built):
    names = _declared("mi_host.h")
    assert len(names) >= 14
    exported = _exported(os.path.join(capi.LIB_DIR, "libmi_host.so"))
    assert not [n for n in names if n not in exported]
    assert sorted(capi.HOST_SYMBOLS) == names  # the ctypes table is the header, nothing more, nothing less
    capi.host_lib()


def test_pt_library_exports_header():
    lib = os.path.join(capi.LIB_DIR, "libmi_pt.so")
    if not os.path.exists(lib):
        pytest.skip("libmi_pt.so not built yet (run __graft_entry__.build())")
    names = _declared("mi_pt.h")
    assert len(names) >= 20
    exported = _exported(lib)
    assert not [n for n in names if n not in exported]
    assert sorted(capi.PT_SYMBOLS) == names
    # nothing from the oracle may be linked into the product
    assert not [s for s in exported if s.startswith("oracle_")]
    needed = subprocess.run(["readelf", "-d", lib], check=True, capture_output=True, text=True).stdout
    assert "oracle" not in needed


def test_pt_library_is_built_from_this_tree():
    """Stale-binary guard: libmi_pt.so is git-ignored and travels prebuilt to the GPU box; its baked-in source id
    (csrc/Makefile: sha1 over csrc/device/* + include/mi_pt*.h) must be the one of the sources in this tree."""
    lib = os.path.join(capi.LIB_DIR, "libmi_pt.so")
    if not os.path.exists(lib):
        pytest.skip("libmi_pt.so not built yet (run __graft_entry__.build())")
    version = capi.pt_lib().mi_pt_version().decode()
    assert f"src={capi.device_source_id()}" in version, (version, capi.device_source_id())


@pytest.mark.gpu
def test_pt_library_is_built_from_this_tree_on_the_gpu_box():
    test_pt_library_is_built_from_this_tree()


def test_product_fails_loudly_without_gpu(built, assets):
    """No CPU fallback: without a HIP device mi_pt_create must return MI_PT_ERR_NO_DEVICE, never render on the host."""
    lib = os.path.join(capi.LIB_DIR, "libmi_pt.so")
    if not os.path.exists(lib):
        pytest.skip("libmi_pt.so not built yet")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible")
    except ImportError:
        pass
    from vk_gltf_renderer_amd import pathtracer as ptmod
    scene = ptmod.Scene(os.path.join(assets, "Box.glb"))
    with pytest.raises(ptmod.MiError) as e:
        ptmod.PathTracer(scene)
    assert "rc=-2" in str(e.value) or "no HIP device" in str(e.value)


def test_struct_layouts_match_reference_contract():
    """Sizes/offsets the reference asserts (src/gltf_material_cache.cpp:46-56) and SURVEY Appendix C."""
    M = capi.MiGltfShadeMaterial
    assert (M.pbrBaseColorFactor.offset, M.pbrRoughnessFactor.offset, M.alphaMode.offset, M.occlusionStrength.offset, M.doubleSided.offset) == (0, 32, 40, 48, 52)
    assert C.sizeof(M) == 288 and C.sizeof(M) % 8 == 0
    assert M.pbrBaseColorTexture.offset == 232 and M.retroreflectionTexture.offset == 274
    assert C.sizeof(capi.MiGltfRenderNode) == 136 and C.sizeof(capi.MiGltfLight) == 64 and C.sizeof(capi.MiGltfTextureInfo) == 32
    assert C.sizeof(capi.MiSceneFrameInfo) == 396
