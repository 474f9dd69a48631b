"""ctypes binding of oracle/liboracle_pt.so — the CPU oracle (test infrastructure; never imported by the product)."""
import ctypes as C
import os
import subprocess

from vk_gltf_renderer_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None

P, VP, f32, i32, u32 = C.POINTER, C.c_void_p, C.c_float, C.c_int, C.c_uint32


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


FLAGS = "-O3"  # of the library lib() loaded ("-O3 -march=native" after use_native())


def use_native():
    """Before the first lib(): build oracle/_native/liboracle_pt.so (-O3 -march=native) on THIS host and load that one -- for the timed
    CPU baseline of bench.py.  Falls back to the portable build when it cannot be built here.  Returns the flags in use."""
    global _native, FLAGS
    r = subprocess.run(["make", "-C", ORACLE_DIR, "-s", "native"], capture_output=True, text=True)
    _native = r.returncode == 0 and os.path.exists(os.path.join(ORACLE_DIR, "_native", "liboracle_pt.so"))
    FLAGS = "-O3 -march=native" if _native else "-O3"
    return FLAGS


_native = False


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "_native", "liboracle_pt.so") if _native else os.path.join(ORACLE_DIR, "liboracle_pt.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        sig = {
            "oracle_pt_create": (i32, [P(capi.MiPtSceneDesc), P(VP)]),
            "oracle_pt_destroy": (None, [VP]),
            "oracle_pt_set_environment": (i32, [VP, P(capi.MiPtEnvironment)]),
            "oracle_pt_resize": (i32, [VP, i32, i32]),
            "oracle_pt_set_frame_info": (i32, [VP, P(capi.MiSceneFrameInfo)]),
            "oracle_pt_set_sky": (i32, [VP, P(capi.MiSkyPhysicalParameters)]),
            "oracle_pt_set_tile_partition": (i32, [VP, i32, i32, i32]),
            "oracle_pt_render_frame": (i32, [VP, P(capi.MiPathtraceParams), i32]),
            "oracle_pt_accum": (P(f32), [VP]),
            "oracle_pt_depth": (P(f32), [VP]),
            "oracle_pt_selection": (P(u32), [VP]),
            "oracle_pt_albedo": (P(f32), [VP]),
            "oracle_pt_normal": (P(f32), [VP]),
            "oracle_pt_get_stats": (i32, [VP, P(capi.MiPtStats)]),
            "oracle_xxhash32": (u32, [u32, u32, u32]),
            "oracle_rand": (f32, [P(u32)]),
            "oracle_sky_eval": (None, [P(capi.MiSkyPhysicalParameters), P(f32), P(f32)]),
            "oracle_sky_pdf": (f32, [P(capi.MiSkyPhysicalParameters), P(f32)]),
            "oracle_sky_sample": (None, [P(capi.MiSkyPhysicalParameters), f32, f32, P(f32)]),
            "oracle_bsdf_eval": (None, [P(f32), P(f32), P(f32), P(f32), P(f32)]),
            "oracle_bsdf_sample": (None, [P(f32), P(f32), P(f32), P(f32)]),
            "oracle_round_to_half": (f32, [f32]),
            "oracle_fresnel_dielectric_unpolarized": (f32, [f32, f32]),
            "oracle_fresnel_schlick": (f32, [f32, f32]),
            "oracle_fresnel_conductor": (None, [f32, f32, f32, f32, P(f32)]),
            "oracle_thin_film": (None, [f32, f32, f32, f32, f32, P(f32)]),
            "oracle_ggx_ndf": (f32, [f32, f32, P(f32)]),
            "oracle_ggx_g1": (f32, [f32, f32, P(f32)]),
            "oracle_ggx_sample_vndf": (None, [f32, f32, P(f32), f32, f32, P(f32)]),
            "oracle_hg_pdf": (f32, [f32, f32]),
            "oracle_hg_sample": (None, [f32, f32, f32, P(f32), P(f32)]),
            "oracle_light_contribution": (None, [P(capi.MiGltfLight), P(f32), P(f32), P(f32)]),
            "oracle_sheen_ndf": (f32, [f32, f32]),
            "oracle_sheen_sample": (None, [f32, f32, f32, P(f32)]),
            "oracle_vcavities_g": (f32, [f32, f32, f32, f32, f32]),
            "oracle_lobe_weights": (None, [P(f32), f32, P(f32)]),
            "oracle_lobe_index": (i32, [C.c_char_p]),
            "oracle_env_sample": (None, [P(capi.MiPtEnvironment), P(f32), P(f32)]),
            "oracle_point_offset": (None, [P(f32), P(f32), P(f32), P(f32), P(f32)]),
            "oracle_ray_cone_footprint": (f32, [f32, f32, f32, P(f32), P(f32)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
