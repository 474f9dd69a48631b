/*
 * mi_pt_shaderio.h — host<->device data contract of the path-trace hot path, restated as plain C PODs.
 *
 * Every struct here is byte-compatible with the struct of the same name (minus the `Mi` prefix) that the
 * reference shares between its C++ host and its Slang device code under *scalar block layout*
 * (reference: shaders/gltf_scene_io.h.slang:41-322, shaders/shaderio.h:148-196, layout asserts
 * src/gltf_material_cache.cpp:46-56).  A maintainer can memcpy the reference's std::vector<shaderio::X>
 * straight into these.
 *
 * Matrix convention (reference: glm on the host, Slang `mul(v, M)` on the device): 16 floats, column-major,
 * m[4*c + r]; translation lives in m[12..14].
 *
 * All MAT_EXT_* gates of the reference (shaders/gltf_material_config.h) are ON here, which the reference
 * documents as its default layout.
 */
#ifndef MI_PT_SHADERIO_H
#define MI_PT_SHADERIO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: shaders/gltf_scene_io.h.slang:41-47 (136 B) */
typedef struct MiGltfRenderNode
{
  float objectToWorld[16];
  float worldToObject[16];
  int   materialID;
  int   renderPrimID;
} MiGltfRenderNode;

/* reference: shaders/gltf_scene_io.h.slang:72-78 */
enum MiLightType
{
  MI_LIGHT_NONE        = 0,
  MI_LIGHT_DIRECTIONAL = 1,
  MI_LIGHT_SPOT        = 2,
  MI_LIGHT_POINT       = 3
};

/* reference: shaders/gltf_scene_io.h.slang:85-100 (64 B) */
typedef struct MiGltfLight
{
  float direction[3];
  int   type;
  float position[3];
  float radius;
  float color[3];
  float intensity;
  float angularSizeOrInvRange;
  float innerAngle;
  float outerAngle;
  int   _pad;
} MiGltfLight;

/* reference: shaders/gltf_scene_io.h.slang:103-108 */
enum MiAlphaMode
{
  MI_ALPHA_OPAQUE = 0,
  MI_ALPHA_MASK   = 1,
  MI_ALPHA_BLEND  = 2
};

/* reference: shaders/gltf_scene_io.h.slang:112-116 */
enum MiPbrModel
{
  MI_PBR_METALLIC_ROUGHNESS  = 0,
  MI_PBR_SPECULAR_GLOSSINESS = 1
};

/* reference: shaders/gltf_scene_io.h.slang:121-128 (32 B). uvTransform is the Slang float3x2: three rows of two
 * floats; uv' = (u,v,1) * uvTransform. Host packing: src/gltf_material_cache.cpp:81-84. */
typedef struct MiGltfTextureInfo
{
  float uvTransform[6];
  int   index;    /* glTF texture index, -1 = none */
  int   texCoord; /* 0 or 1 */
} MiGltfTextureInfo;

/* reference: shaders/gltf_scene_io.h.slang:147-310 (288 B with every MAT_EXT_* on) */
typedef struct MiGltfShadeMaterial
{
  float pbrBaseColorFactor[4]; /*   0 */
  float emissiveFactor[3];     /*  16 */
  float normalTextureScale;    /*  28 */
  float pbrRoughnessFactor;    /*  32 */
  float pbrMetallicFactor;     /*  36 */
  int   alphaMode;             /*  40 */
  float alphaCutoff;           /*  44 */
  float occlusionStrength;     /*  48 */
  int   doubleSided;           /*  52 */
  float attenuationColor[3];   /*  56  KHR_materials_volume */
  float ior;                   /*  68  KHR_materials_ior */
  float transmissionFactor;    /*  72  KHR_materials_transmission */
  float thicknessFactor;       /*  76  KHR_materials_volume */
  float attenuationDistance;   /*  80 */
  float clearcoatFactor;       /*  84  KHR_materials_clearcoat */
  float specularColorFactor[3];/*  88  KHR_materials_specular */
  float clearcoatRoughness;    /* 100 */
  float specularFactor;        /* 104 */
  int   unlit;                 /* 108  KHR_materials_unlit */
  float iridescenceFactor;     /* 112  KHR_materials_iridescence */
  float iridescenceThicknessMinimum;
  float iridescenceThicknessMaximum;
  float iridescenceIor;
  float anisotropyRotation[2]; /* 128  (sin, cos) */
  float sheenColorFactor[3];   /* 136  KHR_materials_sheen */
  float anisotropyStrength;    /* 148 */
  float sheenRoughnessFactor;  /* 152 */
  float dispersion;            /* 156  KHR_materials_dispersion */
  int   pbrModel;              /* 160  KHR_materials_pbrSpecularGlossiness */
  float pbrDiffuseFactor[4];   /* 164 */
  float pbrSpecularFactor[3];  /* 180 */
  float pbrGlossinessFactor;   /* 192 */
  float diffuseTransmissionColor[3]; /* 196 KHR_materials_diffuse_transmission */
  float diffuseTransmissionFactor;   /* 208 */
  float retroreflectionFactor;       /* 212 KHR_materials_retroreflection */
  float multiscatterColorFactor[3];  /* 216 KHR_materials_volume_scatter */
  float scatterAnisotropy;           /* 228 */
  /* 22 texture-info slots; 0 = "no texture" sentinel (textureInfos[0] is reserved) */
  uint16_t pbrBaseColorTexture; /* 232 */
  uint16_t normalTexture;
  uint16_t pbrMetallicRoughnessTexture;
  uint16_t emissiveTexture;
  uint16_t occlusionTexture;
  uint16_t transmissionTexture;
  uint16_t thicknessTexture;
  uint16_t clearcoatTexture;
  uint16_t clearcoatRoughnessTexture;
  uint16_t clearcoatNormalTexture;
  uint16_t specularTexture;
  uint16_t specularColorTexture;
  uint16_t iridescenceTexture;
  uint16_t iridescenceThicknessTexture;
  uint16_t anisotropyTexture;
  uint16_t sheenColorTexture;
  uint16_t sheenRoughnessTexture;
  uint16_t pbrDiffuseTexture;
  uint16_t pbrSpecularGlossinessTexture;
  uint16_t diffuseTransmissionTexture;
  uint16_t diffuseTransmissionColorTexture;
  uint16_t retroreflectionTexture; /* 274 */
  uint16_t _pad16[2];              /* 276: natural padding up to the 8-aligned trailing pad */
  uint64_t _pad;                   /* 280 */
} MiGltfShadeMaterial;

/* reference: shaders/shaderio.h:138-145 */
enum MiSceneFrameInfoFlags
{
  MI_SCENE_IS_ORTHOGRAPHIC               = 1 << 0,
  MI_SCENE_USE_SOLID_BACKGROUND          = 1 << 1,
  MI_SCENE_USE_HDR_ENVIRONMENT           = 1 << 2,
  MI_SCENE_USE_INFINITE_PLANE            = 1 << 3,
  MI_SCENE_INFINITE_PLANE_SHADOW_CATCHER = 1 << 4
};

/* reference: shaders/shaderio.h:148-168 (396 B) */
typedef struct MiSceneFrameInfo
{
  float viewMatrix[16];
  float projInv[16];
  float viewInv[16];
  float viewProjMatrix[16];
  float prevMVP[16];
  float jitter[2];
  float imageSize[2];
  int   flags;
  float envRotation;
  float envBlur;
  float envIntensity;
  float backgroundColor[3];
  int   visualization;
  float infinitePlaneDistance;
  float infinitePlaneBaseColor[3];
  float infinitePlaneMetallic;
  float infinitePlaneRoughness;
  float shadowCatcherDarkenAmount;
} MiSceneFrameInfo;

/* reference: shaders/shaderio.h:170-175 */
enum MiPathtracerFlags
{
  MI_PT_USE_DLSS           = 1 << 0, /* accepted and ignored: DLSS is out of scope */
  MI_PT_USE_OPTIX_DENOISER = 1 << 1, /* repurposed: capture albedo/normal guides for the a-trous pass */
  MI_PT_FIRST_FRAME        = 1 << 2
};

/* The scalar head of PathtracePushConstant (reference: shaders/shaderio.h:179-196). The four device pointers of the
 * reference (frameInfo, skyParams, gltfScene, prevRenderNodeObjectToWorld) are owned by the library and set through
 * mi_pt_set_frame_info / mi_pt_set_sky / mi_pt_create, so they do not appear here. */
typedef struct MiPathtraceParams
{
  int   maxDepth;              /* default 5 */
  int   frameCount;            /* seed input: xxhash32(x, y, frameCount) */
  float fireflyClampThreshold; /* default 10 */
  float texGradScale;          /* default 1 */
  int   numSamples;            /* spp per frame, default 1 */
  int   totalSamples;          /* samples accumulated before this frame */
  float focalDistance;
  float aperture;
  int   flags; /* MiPathtracerFlags */
  float pixelAngle;
  float mouseCoord[2]; /* debug only; ignored */
} MiPathtraceParams;

/* Physical sun & sky parameters (reference: nvshaders/sky_io.h.slang `SkyPhysicalParameters`, external to the
 * reference tree; consumed at shaders/pathtrace_functions.h.slang:422-429,470-471; host default src/renderer.cpp:1328,
 * yIsUp src/renderer.cpp:707). Field list restated from the public nvpro_core2 header; see DESIGN.md §oracle. */
typedef struct MiSkyPhysicalParameters
{
  float rgbUnitConversion[3];
  float multiplier;
  float haze;
  float redblueshift;
  float saturation;
  float horizonHeight;
  float groundColor[3];
  float horizonBlur;
  float nightColor[3];
  float sunDiskIntensity;
  float sunDirection[3];
  float sunDiskScale;
  float sunGlowIntensity;
  int   yIsUp;
} MiSkyPhysicalParameters;

/* Alias-table entry for HDR importance sampling (reference: nvshaders/hdr_io.h.slang `EnvAccel`, external;
 * consumed at shaders/gltf_pathtrace.slang:69 and pathtrace_functions.h.slang:437). */
typedef struct MiEnvAccel
{
  uint32_t alias;
  float    q;
} MiEnvAccel;

/* Tonemapper parameters (reference: nvshaders/tonemap_io.h.slang `TonemapperData`, external to the reference tree; held in
 * Resources::tonemapperData with `.autoExposure = 1`, src/resources.hpp:212; set from the command line by tmMethod / tmExposure /
 * tmGamma (-> brightness) / tmContrast / tmSaturation / tmWhitePoint (-> vignette), src/renderer.cpp:173-179; consumed by
 * GltfRenderer::tonemap, src/renderer.cpp:1041-1050).  Field list restated from the public nvpro_core2 header. */
enum MiTonemapMethod
{
  MI_TONEMAP_FILMIC      = 0,
  MI_TONEMAP_UNCHARTED   = 1,
  MI_TONEMAP_CLIP        = 2,
  MI_TONEMAP_ACES        = 3,
  MI_TONEMAP_AGX         = 4,
  MI_TONEMAP_KHRONOS_PBR = 5
};

typedef struct MiTonemapperData
{
  int   method;     /* MiTonemapMethod */
  int   isActive;   /* 0: the input is passed through (clamped to 8 bits), src/renderer.cpp:1042-1047 */
  float exposure;   /* linear multiplier applied before the curve */
  float brightness; /* display gamma applied after the curve: c^(1/brightness) */
  float contrast;   /* about mid grey 0.5 */
  float saturation; /* about Rec.601 luma */
  float vignette;   /* radial darkening, 0 = off */
  int   autoExposure;         /* 1: exposure is further scaled by key 0.18 / geometric-mean luminance of the image */
  float autoExposureSpeed;    /* adaptation rate, 1/s (temporal easing 1 - exp(-dt * speed)) */
  float evMinValue;           /* log2-luminance range of the metering histogram */
  float evMaxValue;
  int   enableCenterMetering; /* 1: the central disc of the image weighs four times */
} MiTonemapperData;

#ifdef __cplusplus
}
static_assert(sizeof(MiGltfRenderNode) == 136, "GltfRenderNode layout");
static_assert(sizeof(MiGltfLight) == 64, "GltfLight layout");
static_assert(sizeof(MiGltfTextureInfo) == 32, "GltfTextureInfo layout");
static_assert(sizeof(MiGltfShadeMaterial) == 288, "GltfShadeMaterial layout");
static_assert(sizeof(MiSceneFrameInfo) == 396, "SceneFrameInfo layout");
static_assert(sizeof(MiTonemapperData) == 48, "TonemapperData layout");
#endif

#endif /* MI_PT_SHADERIO_H */
