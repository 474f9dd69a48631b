/*
 * oracle_pt.h — C interface of the CPU oracle (liboracle_pt.so).  TEST INFRASTRUCTURE ONLY: nothing under
 * vk_gltf_renderer_amd/ may include, link or call this.  It consumes the same MiPtSceneDesc / MiPtEnvironment /
 * MiSceneFrameInfo / MiPathtraceParams tables as the product's C-ABI (include/mi_pt.h) so a test feeds both sides
 * identical bytes.
 */
#ifndef ORACLE_PT_H
#define ORACLE_PT_H
#include "../include/mi_pt.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct OraclePt OraclePt;
int  oracle_pt_create(const MiPtSceneDesc* scene, OraclePt** out);
void oracle_pt_destroy(OraclePt* o);
int  oracle_pt_set_environment(OraclePt* o, const MiPtEnvironment* env);
int  oracle_pt_resize(OraclePt* o, int w, int h);
int  oracle_pt_set_frame_info(OraclePt* o, const MiSceneFrameInfo* f);
int  oracle_pt_set_sky(OraclePt* o, const MiSkyPhysicalParameters* s);
int  oracle_pt_set_tile_partition(OraclePt* o, int rank, int world, int tileSize);
int  oracle_pt_render_frame(OraclePt* o, const MiPathtraceParams* params, int threads);
const float*    oracle_pt_accum(OraclePt* o);
const float*    oracle_pt_depth(OraclePt* o);
const uint32_t* oracle_pt_selection(OraclePt* o);
const float*    oracle_pt_albedo(OraclePt* o);
const float*    oracle_pt_normal(OraclePt* o);
int             oracle_pt_get_stats(OraclePt* o, MiPtStats* s);
/* known-answer hooks */
uint32_t oracle_xxhash32(uint32_t x, uint32_t y, uint32_t z);
float    oracle_rand(uint32_t* seed);
void     oracle_sky_eval(const MiSkyPhysicalParameters* s, const float* dir, float* rgb);
float    oracle_sky_pdf(const MiSkyPhysicalParameters* s, const float* dir);
void     oracle_sky_sample(const MiSkyPhysicalParameters* s, float u, float v, float* dirPdfRgb);
/* material array layout (29 floats): baseColor[3] roughness[2] metallic ior1 ior2 specular specularColor[3]
 * transmission thickness clearcoat clearcoatRoughness sheenColor[3] sheenRoughness iridescence iridescenceIor
 * iridescenceThickness diffuseTransmissionFactor diffuseTransmissionColor[3] dispersion retroreflection;
 * shading frame is T=(1,0,0) B=(0,1,0) N=Ng=(0,0,1). */
void  oracle_bsdf_eval(const float* m, const float* k1, const float* k2, const float* xi, float* out7);
void  oracle_bsdf_sample(const float* m, const float* k1, const float* xi, float* out8);
float oracle_round_to_half(float f);
/* closed-form pins (tests/test_oracle_pins.py) */
float oracle_fresnel_dielectric_unpolarized(float eta, float cosTheta);
float oracle_fresnel_schlick(float ior, float cosTheta);
void  oracle_fresnel_conductor(float n_a, float n_b, float k_b, float cosTheta, float* rs_rp);
void  oracle_thin_film(float thickness, float coatingIor, float baseIor, float incomingIor, float cosTheta, float* rgb);
float oracle_ggx_ndf(float ax, float ay, const float* h);
float oracle_ggx_g1(float ax, float ay, const float* k);
void  oracle_ggx_sample_vndf(float ax, float ay, const float* k, float u, float v, float* h);
float oracle_hg_pdf(float cosTheta, float g);
void  oracle_hg_sample(float u, float v, float g, const float* wi, float* wo);
/* out8 = incidentVector[3] distance intensity[3] pdf of singleLightContribution(light, pos, xi) */
void  oracle_light_contribution(const MiGltfLight* light, const float* pos, const float* xi, float* out8);
#ifdef __cplusplus
}
#endif
#endif
