"""ctypes declarations for the C-ABI in include/mi_pt.h, include/mi_pt_shaderio.h and include/mi_host.h.

Pure plumbing: every struct mirrors the C header field for field (sizes are asserted at import time) and every
function prototype is declared once here.  No arithmetic of the hot path lives in Python.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

f32 = C.c_float
i32 = C.c_int
u32 = C.c_uint32


class MiGltfRenderNode(C.Structure):
    _fields_ = [("objectToWorld", f32 * 16), ("worldToObject", f32 * 16), ("materialID", i32), ("renderPrimID", i32)]


class MiGltfLight(C.Structure):
    _fields_ = [("direction", f32 * 3), ("type", i32), ("position", f32 * 3), ("radius", f32), ("color", f32 * 3),
                ("intensity", f32), ("angularSizeOrInvRange", f32), ("innerAngle", f32), ("outerAngle", f32), ("_pad", i32)]


class MiGltfTextureInfo(C.Structure):
    _fields_ = [("uvTransform", f32 * 6), ("index", i32), ("texCoord", i32)]


_MAT_TEXTURE_SLOTS = [
    "pbrBaseColorTexture", "normalTexture", "pbrMetallicRoughnessTexture", "emissiveTexture", "occlusionTexture",
    "transmissionTexture", "thicknessTexture", "clearcoatTexture", "clearcoatRoughnessTexture", "clearcoatNormalTexture",
    "specularTexture", "specularColorTexture", "iridescenceTexture", "iridescenceThicknessTexture", "anisotropyTexture",
    "sheenColorTexture", "sheenRoughnessTexture", "pbrDiffuseTexture", "pbrSpecularGlossinessTexture",
    "diffuseTransmissionTexture", "diffuseTransmissionColorTexture", "retroreflectionTexture"]


class MiGltfShadeMaterial(C.Structure):
    _fields_ = [
        ("pbrBaseColorFactor", f32 * 4), ("emissiveFactor", f32 * 3), ("normalTextureScale", f32),
        ("pbrRoughnessFactor", f32), ("pbrMetallicFactor", f32), ("alphaMode", i32), ("alphaCutoff", f32),
        ("occlusionStrength", f32), ("doubleSided", i32), ("attenuationColor", f32 * 3), ("ior", f32),
        ("transmissionFactor", f32), ("thicknessFactor", f32), ("attenuationDistance", f32), ("clearcoatFactor", f32),
        ("specularColorFactor", f32 * 3), ("clearcoatRoughness", f32), ("specularFactor", f32), ("unlit", i32),
        ("iridescenceFactor", f32), ("iridescenceThicknessMinimum", f32), ("iridescenceThicknessMaximum", f32),
        ("iridescenceIor", f32), ("anisotropyRotation", f32 * 2), ("sheenColorFactor", f32 * 3),
        ("anisotropyStrength", f32), ("sheenRoughnessFactor", f32), ("dispersion", f32), ("pbrModel", i32),
        ("pbrDiffuseFactor", f32 * 4), ("pbrSpecularFactor", f32 * 3), ("pbrGlossinessFactor", f32),
        ("diffuseTransmissionColor", f32 * 3), ("diffuseTransmissionFactor", f32), ("retroreflectionFactor", f32),
        ("multiscatterColorFactor", f32 * 3), ("scatterAnisotropy", f32),
    ] + [(n, C.c_uint16) for n in _MAT_TEXTURE_SLOTS] + [("_pad16", C.c_uint16 * 2), ("_pad", C.c_uint64)]


class MiSceneFrameInfo(C.Structure):
    _fields_ = [
        ("viewMatrix", f32 * 16), ("projInv", f32 * 16), ("viewInv", f32 * 16), ("viewProjMatrix", f32 * 16),
        ("prevMVP", f32 * 16), ("jitter", f32 * 2), ("imageSize", f32 * 2), ("flags", i32), ("envRotation", f32),
        ("envBlur", f32), ("envIntensity", f32), ("backgroundColor", f32 * 3), ("visualization", i32),
        ("infinitePlaneDistance", f32), ("infinitePlaneBaseColor", f32 * 3), ("infinitePlaneMetallic", f32),
        ("infinitePlaneRoughness", f32), ("shadowCatcherDarkenAmount", f32)]


class MiPathtraceParams(C.Structure):
    _fields_ = [("maxDepth", i32), ("frameCount", i32), ("fireflyClampThreshold", f32), ("texGradScale", f32),
                ("numSamples", i32), ("totalSamples", i32), ("focalDistance", f32), ("aperture", f32), ("flags", i32),
                ("pixelAngle", f32), ("mouseCoord", f32 * 2)]


class MiSkyPhysicalParameters(C.Structure):
    _fields_ = [("rgbUnitConversion", f32 * 3), ("multiplier", f32), ("haze", f32), ("redblueshift", f32),
                ("saturation", f32), ("horizonHeight", f32), ("groundColor", f32 * 3), ("horizonBlur", f32),
                ("nightColor", f32 * 3), ("sunDiskIntensity", f32), ("sunDirection", f32 * 3), ("sunDiskScale", f32),
                ("sunGlowIntensity", f32), ("yIsUp", i32)]


class MiEnvAccel(C.Structure):
    _fields_ = [("alias", u32), ("q", f32)]


class MiTonemapperData(C.Structure):
    _fields_ = [("method", i32), ("isActive", i32), ("exposure", f32), ("brightness", f32), ("contrast", f32), ("saturation", f32), ("vignette", f32),
                ("autoExposure", i32), ("autoExposureSpeed", f32), ("evMinValue", f32), ("evMaxValue", f32), ("enableCenterMetering", i32)]


TONEMAP_METHODS = ("filmic", "uncharted", "clip", "aces", "agx", "khronos_pbr")  # MiTonemapMethod 0..5 (src/renderer.cpp:173)


class MiPtRenderPrimitive(C.Structure):
    _fields_ = [("indices", C.POINTER(u32)), ("triangleCount", u32), ("vertexCount", u32), ("positions", C.POINTER(f32)),
                ("normals", C.POINTER(f32)), ("colors", C.POINTER(u32)), ("tangents", C.POINTER(f32)),
                ("texCoords0", C.POINTER(f32)), ("texCoords1", C.POINTER(f32)), ("opaqueTriangleCount", u32), ("reserved", u32)]


class MiPtTexture(C.Structure):
    _fields_ = [("levels", C.POINTER(C.POINTER(C.c_uint8))), ("width", i32), ("height", i32), ("numLevels", i32),
                ("srgb", i32), ("magFilter", i32), ("minFilter", i32), ("mipmapMode", i32), ("wrapS", i32), ("wrapT", i32)]


class MiPtSceneDesc(C.Structure):
    _fields_ = [("materials", C.POINTER(MiGltfShadeMaterial)), ("numMaterials", i32),
                ("textureInfos", C.POINTER(MiGltfTextureInfo)), ("numTextureInfos", i32),
                ("renderNodes", C.POINTER(MiGltfRenderNode)), ("numRenderNodes", i32),
                ("renderNodeVisible", C.POINTER(C.c_uint8)),
                ("renderPrimitives", C.POINTER(MiPtRenderPrimitive)), ("numRenderPrimitives", i32),
                ("lights", C.POINTER(MiGltfLight)), ("numLights", i32),
                ("textures", C.POINTER(MiPtTexture)), ("numTextures", i32)]


class MiPtEnvironment(C.Structure):
    _fields_ = [("rgba", C.POINTER(f32)), ("accel", C.POINTER(MiEnvAccel)), ("width", i32), ("height", i32), ("integral", f32)]


class MiPtCreateOptions(C.Structure):
    _fields_ = [("device", i32), ("collectCounters", i32), ("bvhBuilder", i32), ("reserved", i32 * 5)]


class MiPtStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "cameraPaths", "segments", "shadowRays", "nodesClosest", "trisClosest", "nodesShadow", "trisShadow",
        "textureTaps", "bvhNodeCount", "bvhTriangleCount", "bvhNodeBytes", "bvhTriangleBytes", "surfaceHits", "nodesPrimary", "trisPrimary")]


class MiPtMemory(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sceneBytes", "rendererBytes", "deviceUsedBytes", "deviceTotalBytes", "pathStateBytes", "pathSlots")]


class MiPtFrameTiming(C.Structure):
    _fields_ = [("totalMs", f32), ("generateMs", f32), ("traceClosestMs", f32), ("sortMs", f32), ("shadeMs", f32),
                ("traceShadowMs", f32), ("accumulateMs", f32), ("traceClosestLaunches", i32), ("shadeLaunches", i32),
                ("traceShadowLaunches", i32), ("bounceIterations", i32), ("tracePrimaryMs", f32), ("shadeFirstMs", f32),
                ("tracePrimaryLaunches", i32), ("shadeFirstLaunches", i32)]


class MiCamera(C.Structure):
    _fields_ = [("eye", f32 * 3), ("center", f32 * 3), ("up", f32 * 3), ("fovDegrees", f32), ("znear", f32),
                ("zfar", f32), ("orthographic", i32), ("xmag", f32), ("ymag", f32)]


assert C.sizeof(MiGltfRenderNode) == 136
assert C.sizeof(MiGltfLight) == 64
assert C.sizeof(MiGltfTextureInfo) == 32
assert C.sizeof(MiGltfShadeMaterial) == 288
assert C.sizeof(MiSceneFrameInfo) == 396

MI_PT_ABI_VERSION = 6  # include/mi_pt.h
MI_PT_USE_DLSS, MI_PT_USE_OPTIX_DENOISER, MI_PT_FIRST_FRAME = 1, 2, 4
MI_SCENE_IS_ORTHOGRAPHIC, MI_SCENE_USE_SOLID_BACKGROUND, MI_SCENE_USE_HDR_ENVIRONMENT = 1, 2, 4
MI_SCENE_USE_INFINITE_PLANE, MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER = 8, 16
MI_LIGHT_NONE, MI_LIGHT_DIRECTIONAL, MI_LIGHT_SPOT, MI_LIGHT_POINT = 0, 1, 2, 3  # shaders/gltf_scene_io.h.slang:72-78

P = C.POINTER
VP = C.c_void_p

HOST_SYMBOLS = {
    "mi_scene_load": (i32, [C.c_char_p, P(VP)]),
    "mi_scene_destroy": (None, [VP]),
    "mi_scene_desc": (P(MiPtSceneDesc), [VP]),
    "mi_scene_num_cameras": (i32, [VP]),
    "mi_scene_camera": (i32, [VP, i32, P(MiCamera)]),
    "mi_scene_bounds": (None, [VP, P(f32), P(f32)]),
    "mi_scene_num_triangles": (C.c_uint64, [VP]),
    "mi_hdr_load": (i32, [C.c_char_p, P(VP)]),
    "mi_scene_recompute_tangents": (i32, [VP, i32, i32]),
    "mi_mikktspace": (i32, [P(f32), P(f32), P(f32), u32, P(u32), u32, P(f32)]),
    "mi_scene_cut_alpha": (C.c_int64, [VP, i32]),
    "mi_scene_num_animations": (i32, [VP]),
    "mi_scene_animation_info": (i32, [VP, i32, P(f32), P(f32), C.c_char_p, i32]),
    "mi_scene_update_animation": (i32, [VP, i32, f32]),
    "mi_hdr_from_pixels": (i32, [i32, i32, P(f32), P(VP)]),
    "mi_hdr_destroy": (None, [VP]),
    "mi_hdr_env": (P(MiPtEnvironment), [VP]),
    "mi_default_sky": (None, [P(MiSkyPhysicalParameters)]),
    "mi_default_params": (None, [P(MiPathtraceParams)]),
    "mi_camera_frame_info": (None, [P(MiCamera), i32, i32, P(MiSceneFrameInfo), P(f32), P(f32)]),
    "mi_host_last_error": (C.c_char_p, []),
}

PT_SYMBOLS = {
    "mi_pt_create": (i32, [P(MiPtSceneDesc), P(MiPtCreateOptions), P(VP)]),
    "mi_pt_destroy": (i32, [VP]),
    "mi_pt_set_environment": (i32, [VP, P(MiPtEnvironment)]),
    "mi_pt_resize": (i32, [VP, i32, i32]),
    "mi_pt_set_frame_info": (i32, [VP, P(MiSceneFrameInfo)]),
    "mi_pt_set_sky": (i32, [VP, P(MiSkyPhysicalParameters)]),
    "mi_pt_set_tile_partition": (i32, [VP, i32, i32, i32]),
    "mi_pt_bind_accum": (i32, [VP, VP]),
    "mi_pt_bind_guides": (i32, [VP, VP, VP, VP]),
    "mi_pt_render_frame": (i32, [VP, P(MiPathtraceParams), VP]),
    "mi_pt_render_frames": (i32, [VP, P(MiPathtraceParams), i32, VP]),
    "mi_pt_synchronize": (i32, [VP]),
    "mi_pt_read_accum": (i32, [VP, P(f32)]),
    "mi_pt_write_accum": (i32, [VP, P(f32)]),
    "mi_pt_read_guides": (i32, [VP, P(f32), P(f32)]),
    "mi_pt_read_selection": (i32, [VP, P(u32)]),
    "mi_pt_read_depth": (i32, [VP, P(f32)]),
    "mi_pt_set_frame_queue": (i32, [VP, i32]),
    "mi_pt_accum_device_ptr": (VP, [VP]),
    "mi_pt_denoise": (i32, [VP, i32, f32, f32, f32, P(f32), VP]),
    "mi_pt_denoise_svgf": (i32, [VP, i32, f32, f32, f32, P(f32), VP]),
    "mi_pt_denoised_device_ptr": (VP, [VP]),
    "mi_pt_tonemap": (i32, [VP, P(MiTonemapperData), i32, f32, P(C.c_uint8), VP]),
    "mi_pt_tonemapped_device_ptr": (VP, [VP]),
    "mi_pt_default_tonemapper": (None, [P(MiTonemapperData), i32]),
    "mi_pt_get_memory": (i32, [VP, P(MiPtMemory)]),
    "mi_pt_get_stats": (i32, [VP, P(MiPtStats)]),
    "mi_pt_reset_stats": (i32, [VP]),
    "mi_pt_enable_timing": (i32, [VP, i32]),
    "mi_pt_get_frame_timing": (i32, [VP, P(MiPtFrameTiming)]),
    "mi_pt_last_error": (C.c_char_p, []),
    "mi_pt_version": (C.c_char_p, []),
    "mi_pt_abi_version": (i32, []),
    "mi_pt_update_render_nodes": (i32, [VP, P(MiGltfRenderNode), i32, P(C.c_uint8)]),
    "mi_pt_update_lights": (i32, [VP, P(MiGltfLight), i32]),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return lib


_host = None
_pt = None


def host_lib():
    """libmi_host.so — the scene front end (CPU)."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "libmi_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (make -C vk_gltf_renderer_amd/csrc)")
        _host = _bind(C.CDLL(path), HOST_SYMBOLS)
    return _host


def pt_lib():
    """libmi_pt.so — the HIP path tracer.  There is no CPU fallback: a missing library is a hard error."""
    global _pt
    if _pt is None:
        # MI_PT_LIB selects another build of the same library (e.g. the -DTRACE_PROFILE diagnostics build, tools/profile_lanes.sh)
        path = os.environ.get("MI_PT_LIB") or os.path.join(LIB_DIR, "libmi_pt.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension was not built; the product path has no fallback")
        _pt = _bind(C.CDLL(path), PT_SYMBOLS)
        if _pt.mi_pt_abi_version() != MI_PT_ABI_VERSION:  # the structs of this module mirror ONE layout version of include/mi_pt.h
            raise RuntimeError(f"{path}: ABI version {_pt.mi_pt_abi_version()}, this binding was written for {MI_PT_ABI_VERSION}")
    return _pt


def device_source_id():
    """sha1 (first 16 hex digits) over the device sources + public headers, exactly as csrc/Makefile bakes it into libmi_pt.so:
    mi_pt_version() of a library built from this tree contains "src=<this>"."""
    import glob
    import hashlib
    root = os.path.dirname(_HERE)
    dev = os.path.join(_HERE, "csrc", "device")
    files = sorted(glob.glob(os.path.join(dev, "*.hip")) + glob.glob(os.path.join(dev, "*.h")), key=os.path.basename)
    files += [os.path.join(root, "include", "mi_pt.h"), os.path.join(root, "include", "mi_pt_shaderio.h")]
    h = hashlib.sha1()
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
