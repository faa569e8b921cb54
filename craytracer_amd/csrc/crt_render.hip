/*
 * crt_render.hip -- wavefront path tracing on top of the ray-query ABI (restricted form).
 *
 * The reference traces one path per call: pathTrace (trace.h:399-532) loops over bounces and calls
 * intersectTest once per bounce and, through uniformSampleOneLight / estimateDirect (trace.h:359-397,
 * :221-357), shadowIntersectTest once per bounce.  Here every live path advances one bounce per
 * kernel: the rays of all paths go through crt_intersect_batch_dev, a shading kernel rebuilds the
 * ShadeRec of each hit (the tail of the reference's intersectors: triangle.cpp:154-168,:185-204,
 * rect.cpp:34-46, sphere.cpp:3-31), samples one light and the BSDF, and appends a shadow ray and the
 * next path ray to device queues; the shadow rays go through crt_shadow_batch_dev and their
 * unoccluded contributions are added.  The sample loop of threadFunc (main.cpp:88-149) is the outer
 * loop.  This file is a client of include/crt_gpu.h like any other caller; it sees no scene internals.
 *
 * Arithmetic follows the reference operation by operation (float32, separate multiply / add; double
 * where the reference's expressions are double because PI / INV_PI are, util/vec.h:6-10).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <chrono>
#include <new>
#include <vector>

#include "../../include/crt_gpu.h"
#include "crt_internal.h"
#include "crt_camera.h"
#include "crt_libm.h"
#include "crt_shaderec.h"
#include "crt_texture.h"

namespace {

#define RD_PI      3.14159265358979323846264338327950288         /* util/vec.h:6 */
#define RD_INV_PI  (1.0 / RD_PI)                                  /* util/vec.h:10 */

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 v3p(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* util/vec.h:154 */
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float length(V3 a) { return (float)sqrt((double)dot(a, a)); }             /* util/vec.h:240 */
__device__ __forceinline__ V3 normalize(V3 a)                                                          /* util/vec.h:206-215 */
{
    const float len = (float)sqrt((double)dot(a, a));
    if (len == 0) return v3(0.f, 0.f, 0.f);
    return scale(a, 1.0f / len);
}
__device__ __forceinline__ bool is_black(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }  /* vec3_equal(a, BLACK) */
/* orthoNormalTransform, util/math.cpp:10-18 */
__device__ __forceinline__ V3 ortho(V3 u, V3 v, V3 w, V3 a) { return add(add(scale(u, a.x), scale(v, a.y)), scale(w, a.z)); }
/* getVec3InLocalBasis, util/math.cpp:31-39; JITTERED_UP, util/constants.h:27 */
__device__ __forceinline__ V3 jittered_up() { return v3(0.0072f, 1.0f, 0.0034f); }
__device__ __forceinline__ V3 in_local_basis(V3 a, V3 normal)
{
    const V3 w = normal;
    const V3 v = normalize(cross(w, jittered_up()));
    const V3 u = cross(v, w);
    return ortho(u, v, w, a);
}
/* mapSampleToDisk / mapSampleToHemisphere, sampling.cpp:606-618 */
__device__ __forceinline__ void sample_disk(float* out, float s0, float s1)
{
    const float phi = (float)(2.0f * M_PI * (double)s0);
    const float radius = sqrtf(s1);
    out[0] = crt_libm_cosf(phi) * radius;                          /* glibc's values to the bit: crt_libm.h */
    out[1] = crt_libm_sinf(phi) * radius;
}
__device__ __forceinline__ V3 sample_hemisphere(float s0, float s1)
{
    float d[2];
    sample_disk(d, s0, s1);
    const float zz = 1.0f - d[0] * d[0] - d[1] * d[1];
    return v3(d[0], d[1], sqrtf((0.0f > zz) ? 0.0f : zz));            /* max(0, .), util/util.h:14 */
}

/* OrenNayar_f, reflection.cpp:191-214; sinTheta / cosPhi / sinPhi, util/math.h:13-36.  w: the vectors the
 * reference hands it -- tangent space from BSDF_sample_f, WORLD space from BSDF_f (reflection.cpp:731 passes
 * wi, wo, not wi_local, wo_local) */
__device__ __forceinline__ float on_sin_theta(V3 w) { const float s2 = 1.0f - w.z * w.z; return sqrtf((0.0f > s2) ? 0.0f : s2); }
__device__ __forceinline__ float on_clamp(float x) { return (x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x)); }
__device__ __forceinline__ float oren_nayar_scale(V3 wi, V3 wo, float a, float b)
{
    const float sin_theta_i = on_sin_theta(wi), sin_theta_o = on_sin_theta(wo);
    float max_cos = 0.0f;
    if ((double)sin_theta_i > 1e-4 && (double)sin_theta_o > 1e-4) {
        const float sin_phi_i = (sin_theta_i == 0) ? 0 : on_clamp(wi.y / sin_theta_i), cos_phi_i = (sin_theta_i == 0) ? 1 : on_clamp(wi.x / sin_theta_i);
        const float sin_phi_o = (sin_theta_o == 0) ? 0 : on_clamp(wo.y / sin_theta_o), cos_phi_o = (sin_theta_o == 0) ? 1 : on_clamp(wo.x / sin_theta_o);
        const float d_cos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
        max_cos = (0.0f > d_cos) ? 0.0f : d_cos;
    }
    float sin_alpha, tan_beta;
    if (fabsf(wi.z) > fabsf(wo.z)) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / fabsf(wi.z); }
    else                           { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / fabsf(wo.z); }
    return (float)((double)(a + b * max_cos * sin_alpha * tan_beta) * RD_INV_PI);
}

/* calcFresnelDielectric, reflection.cpp:52-76, for the tangent-space normal (0, 0, +-1) */
__device__ __forceinline__ float fresnel_dielectric(float nz, float woz, float etaT, float etaI)
{
    float cos_theta_i = nz * woz;                                 /* vec3_dot with n = (0, 0, nz): 0*x + 0*y + nz*z */
    if (cos_theta_i < 0.0f) { const float tmp = etaI; etaI = etaT; etaT = tmp; cos_theta_i = fabsf(cos_theta_i); }
    const float s2 = 1.0f - cos_theta_i * cos_theta_i;
    const float sin_theta_i = sqrtf((0.0f > s2) ? 0.0f : s2);
    const float sin_theta_t = etaI / etaT * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    const float c2 = 1 - sin_theta_t * sin_theta_t;
    const float cos_theta_t = sqrtf((0.0f > c2) ? 0.0f : c2);
    const float r_parl = ((etaT * cos_theta_i) - (etaI * cos_theta_t)) / ((etaT * cos_theta_i) + (etaI * cos_theta_t));
    const float r_perp = ((etaI * cos_theta_i) - (etaT * cos_theta_t)) / ((etaI * cos_theta_i) + (etaT * cos_theta_t));
    return (r_parl * r_parl + r_perp * r_perp) / 2;
}

/* calcFresnelDielectric for any normal: only vec3_dot(n, wo) enters the result */
__device__ __forceinline__ float fresnel_dielectric_cos(float cos_theta_i, float etaT, float etaI)
{
    return fresnel_dielectric(1.0f, cos_theta_i, etaT, etaI);     /* 1 * c is c */
}

/* ---- PLASTIC / GLASS / METAL: microfacet.cpp and the glossy lobes of reflection.cpp -------------------------------------
 * exp / log / tan / atan: the double-precision function rounded once, as in the oracle (glibc's float routines are not
 * correctly rounded and no device libm returns their bits; against the reference that is a last-ulp difference in about
 * 1 % of the lobe evaluations, tests/test_trace_pieces.py).  cosf / sinf are glibc's (crt_libm.h). */
__device__ __forceinline__ float r_expf(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float r_logf(float x) { return (float)log((double)x); }
__device__ __forceinline__ float r_tanf(float x) { return (float)tan((double)x); }
__device__ __forceinline__ float r_atanf(float x) { return (float)atan((double)x); }
__device__ __forceinline__ float maxr(float a, float b) { return (a > b) ? a : b; }                  /* util/util.h:14 */
__device__ __forceinline__ float mf_cos_phi(V3 w) { const float st = on_sin_theta(w); return (st == 0) ? 1.0f : on_clamp(w.x / st); }
__device__ __forceinline__ float mf_sin_phi(V3 w) { const float st = on_sin_theta(w); return (st == 0) ? 0.0f : on_clamp(w.y / st); }
__device__ __forceinline__ bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }              /* util/math.h:49-52 */
__device__ __forceinline__ float pow5(float v) { return (v * v) * (v * v) * v; }                      /* reflection.cpp:460-463 */

/* MicrofacetDistribution_D, BECKMANN, microfacet.cpp:4-21 */
__device__ float beckmann_D(V3 wh, float ax, float ay)
{
    const float c2 = wh.z * wh.z;
    const float tan2 = maxr(0.0f, 1.0f - c2) / c2;
    if (isinf(tan2)) return 0.0f;
    const float cos4 = c2 * c2;
    const float cp = mf_cos_phi(wh), sp = mf_sin_phi(wh);
    const float e = r_expf(-tan2 * (cp * cp / (ax * ax) + sp * sp / (ay * ay)));
    return (float)((double)e / (RD_PI * (double)ax * (double)ay * (double)cos4));
}
/* MicrofacetDistribution_Lambda / _G, microfacet.cpp:33-55,:72-75 */
__device__ float beckmann_lambda(V3 w, float ax, float ay)
{
    const float abs_tan = fabsf(on_sin_theta(w) / w.z);
    if (isinf(abs_tan)) return 0.0f;
    const float cp = mf_cos_phi(w), sp = mf_sin_phi(w);
    const float alpha = sqrtf(cp * cp * ax * ax + sp * sp * ay * ay);
    const float a = 1.0f / (alpha * abs_tan);
    if (a >= 1.6f) return 0.0f;
    return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}
__device__ __forceinline__ float beckmann_G(V3 wo, V3 wi, float ax, float ay)
{
    return 1.0f / (1.0f + beckmann_lambda(wo, ax, ay) + beckmann_lambda(wi, ax, ay));
}
/* MicrofacetDistribution_sample_wh, microfacet.cpp:77-135 */
__device__ V3 beckmann_sample_wh(V3 wo, float s0, float s1, float ax, float ay)
{
    float tan2, phi;
    float log_sample = r_logf(s0);
    if (isinf(log_sample)) log_sample = 0.0f;
    if (ax == ay) {
        tan2 = -ax * ax * log_sample;
        phi = (float)((double)(s1 * 2.0f) * RD_PI);
    } else {
        phi = r_atanf(ay / ax * r_tanf((float)(2.0f * RD_PI * (double)s1 + 0.5f * RD_PI)));
        if (s1 > 0.5f) phi = (float)((double)phi + RD_PI);
        const float sin_phi = crt_libm_sinf(phi), cos_phi = crt_libm_cosf(phi);
        const float ax2 = ax * ax, ay2 = ay * ay;
        tan2 = -log_sample / (cos_phi * cos_phi / ax2 + sin_phi * sin_phi / ay2);
    }
    const float cos_theta = 1.0f / sqrtf(1.0f + tan2);
    const float sin_theta = sqrtf(maxr(0.0f, 1.0f - cos_theta * cos_theta));
    V3 wh = v3(sin_theta * crt_libm_cosf(phi), sin_theta * crt_libm_sinf(phi), cos_theta);            /* sphericalDirection, util/math.h:91-93 */
    if (!same_hemisphere(wo, wh)) wh = neg(wh);
    return wh;
}
/* calcFresnelConductor, reflection.cpp:78-157, one colour channel */
__device__ float fresnel_conductor(float cos_theta_i, float etaT, float etaI, float k)
{
    const float c2 = cos_theta_i * cos_theta_i, s2 = 1.0f - c2;
    const float eta = etaT / etaI, etaK = k / etaI;
    const float eta2 = eta * eta, etaK2 = etaK * etaK;
    float t0 = eta2 - etaK2;
    t0 = t0 - s2;
    float e22 = eta2 * etaK2;
    e22 = e22 * 4.0f;
    const float a2plusb2 = sqrtf(t0 * t0 + e22);
    const float t1 = a2plusb2 + c2;
    float a = a2plusb2 + t0;
    a = a * 0.5f;
    a = sqrtf(a);
    const float t2 = a * (2.0f * cos_theta_i);
    const float rs = (t1 - t2) / (t1 + t2);
    float t3 = a2plusb2 * c2;
    t3 = t3 + s2 * s2;
    const float t4 = t2 * s2;
    float rp = (t3 - t4) / (t3 + t4);
    rp = rp * rs;
    return (rp + rs) * 0.5f;
}
/* calcReflectRayDir(wi, wh, -wo), util/ray.cpp:11-17 */
__device__ __forceinline__ V3 reflect_about(V3 wo, V3 wh)
{
    const V3 inc = neg(wo);
    const float magnitude = -2.0f * dot(inc, wh);
    return add(inc, scale(wh, magnitude));
}
/* MicrofacetReflection_f, reflection.cpp:297-327: *f is written only when the lobe has a value (the callers start from black).
 * m: the device copy of the material (alpha in uroughness / vroughness).  metal: conductor Fresnel with etaT = color, etaI = 1. */
__device__ void mf_reflection_f(V3* f, V3 wi, V3 wo, const crt_material& m, bool metal)
{
    const float cos_o = fabsf(wo.z), cos_i = fabsf(wi.z);
    V3 wh = add(wi, wo);
    if (cos_i == 0.0f || cos_o == 0.0f) return;
    if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return;
    wh = normalize(wh);
    V3 fr;
    if (!metal) {
        const float kr = 1.0f - fresnel_dielectric_cos(dot(wh, wi), m.ior_in, m.ior_out);
        fr = v3(kr, kr, kr);
    } else {
        const float c = dot(wi, wh);
        fr = v3(fresnel_conductor(c, m.color[0], 1.0f, m.k[0]), fresnel_conductor(c, m.color[1], 1.0f, m.k[1]),
                fresnel_conductor(c, m.color[2], 1.0f, m.k[2]));
    }
    const float sf = beckmann_D(wh, m.uroughness, m.vroughness) * beckmann_G(wo, wi, m.uroughness, m.vroughness) / (4.0f * cos_i * cos_o);
    *f = mul(scale(v3(1.0f, 1.0f, 1.0f), sf), fr);                 /* color = white for METAL and GLASS, materials.cpp:175,:182 */
}
/* MicrofacetFresnel_f / _pdf, reflection.cpp:356-387,:447-458 */
__device__ void mf_fresnel_f(V3* f, V3 wi, V3 wo, const crt_material& m)
{
    if (same_hemisphere(wo, wi)) return;
    const float cos_o = wo.z, cos_i = wi.z;
    if (cos_i == 0.0f || cos_o == 0.0f) { *f = v3(0.f, 0.f, 0.f); return; }
    const float eta = cos_o > 0.0f ? (m.ior_in / m.ior_out) : (m.ior_out / m.ior_in);
    V3 wh = normalize(add(wo, scale(wi, eta)));
    if (wh.z < 0.0f) wh = neg(wh);
    const float F = fresnel_dielectric_cos(dot(wh, wo), m.ior_in, m.ior_out);
    const float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
    const float denom = cos_i * cos_o * sqrt_denom * sqrt_denom;
    const float factor = 1.0f / eta;
    const float numerator = beckmann_D(wh, m.uroughness, m.vroughness) * beckmann_G(wo, wi, m.uroughness, m.vroughness) * eta * eta *
        fabsf(dot(wi, wh)) * fabsf(dot(wo, wh)) * factor * factor;
    *f = scale(scale(v3(1.0f, 1.0f, 1.0f), 1.0f - F), fabsf(numerator / denom));
}
__device__ float mf_fresnel_pdf(V3 wi, V3 wo, const crt_material& m)
{
    if (same_hemisphere(wo, wi)) return 0.0f;
    const float eta = wo.z > 0.0f ? (m.ior_in / m.ior_out) : (m.ior_out / m.ior_in);
    const V3 wh = normalize(add(wo, scale(wi, eta)));
    const float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
    const float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
    return beckmann_D(wh, m.uroughness, m.vroughness) * fabsf(wh.z) * dwh_dwi;
}
/* FresnelBlendDiffuse_f, reflection.cpp:479-490 (kd = color) */
__device__ V3 fb_diffuse_f(V3 wi, V3 wo, const crt_material& m)
{
    V3 d = scale(v3p(m.color), (float)(28.0f / (23.0f * RD_PI)));
    d = mul(d, sub(v3(1.0f, 1.0f, 1.0f), v3p(m.ks)));
    d = scale(d, 1.0f - pow5(1.0f - 0.5f * fabsf(wi.z)));
    d = scale(d, 1.0f - pow5(1.0f - 0.5f * fabsf(wo.z)));
    return d;
}
/* FresnelBlendSpecular_f / _pdf, reflection.cpp:507-522,:537-547 */
__device__ void fb_specular_f(V3* f, V3 wi, V3 wo, const crt_material& m)
{
    V3 wh = add(wi, wo);
    if (is_black(wh)) return;
    wh = normalize(wh);
    const float c = dot(wi, wh);
    const V3 ks = v3p(m.ks);
    const V3 fresnel = add(scale(sub(v3(1.0f, 1.0f, 1.0f), ks), pow5(1.0f - c)), ks);                 /* schlickFresnel, :465-477 */
    *f = scale(fresnel, beckmann_D(wh, m.uroughness, m.vroughness) / (4.0f * fabsf(c) * maxr(fabsf(wi.z), fabsf(wo.z))));
}
__device__ float fb_specular_pdf(V3 wi, V3 wo, const crt_material& m)
{
    if (!same_hemisphere(wo, wi)) return 0.0f;
    const V3 wh = normalize(add(wo, wi));
    return beckmann_D(wh, m.uroughness, m.vroughness) / (2.0f * dot(wo, wh));
}
__device__ __forceinline__ float cos_hemisphere_pdf(V3 wi, V3 wo)                                     /* reflection.cpp:6-15 */
{
    return same_hemisphere(wi, wo) ? (float)((double)fabsf(wi.z) * RD_INV_PI) : 0.0f;
}
/* BSDF_sample_f (reflection.cpp:750-811) for the BSDFs of PLASTIC (two lobes), GLASS and METAL (materials.cpp:142-186), in
 * tangent space.  Returns the pdf (0: the path ends); *glossy = the sampled lobe's flags hold GLOSSY.
 * u_rand stands in for rand() in MicrofacetFresnel_sample_f (reflection.cpp:419). */
__device__ __noinline__ float glossy_sample_f(const crt_material& m, V3 wo, float s0, float s1, float u_rand, V3* f_out, V3* wi_out, bool* glossy)
{
    V3 f = v3(0.f, 0.f, 0.f), wi = v3(0.f, 0.f, 0.f);
    float pdf = 0.0f;
    const float ax = m.uroughness, ay = m.vroughness;
    if (m.type == CRT_MAT_METAL) {                                /* MicrofacetReflection_sample_f, :329-344 */
        *glossy = true;
        if (wo.z == 0.0f) return 0.0f;
        const V3 wh = beckmann_sample_wh(wo, s0, s1, ax, ay);
        wi = reflect_about(wo, wh);
        if (!same_hemisphere(wo, wi)) return 0.0f;
        mf_reflection_f(&f, wi, wo, m, true);
        pdf = beckmann_D(wh, ax, ay) * fabsf(wh.z) / (4.0f * dot(wo, wh));
    } else if (m.type == CRT_MAT_GLASS) {                         /* MicrofacetFresnel_sample_f, :389-445 */
        *glossy = true;
        if (wo.z == 0.0f) return 0.0f;
        const V3 wh = beckmann_sample_wh(wo, s0, s1, ax, ay);
        const float kr = fresnel_dielectric_cos(dot(wh, wo), m.ior_in, m.ior_out);
        if (u_rand <= kr) {
            wi = reflect_about(wo, wh);
            if (!same_hemisphere(wo, wi)) return 0.0f;
            mf_reflection_f(&f, wi, wo, m, false);                /* the stack MicrofacetReflection's is_metal, unset there, is false here */
            pdf = beckmann_D(wh, ax, ay) * fabsf(wh.z) / (4.0f * dot(wo, wh));
        } else {
            const float eta = wo.z > 0.0f ? (m.ior_out / m.ior_in) : (m.ior_in / m.ior_out);
            /* refract, util/ray.cpp:42-59 */
            const float cos_i = dot(wh, wo);
            const float sin2_i = maxr(0.0f, 1.0f - cos_i * cos_i);
            const float sin2_t = eta * eta * sin2_i;
            if (sin2_t >= 1.0f) return 0.0f;
            const float cos_t = sqrtf(1.0f - sin2_t);
            wi = add(scale(neg(wo), eta), scale(wh, eta * cos_i - cos_t));
            mf_fresnel_f(&f, wi, wo, m);
            pdf = mf_fresnel_pdf(wi, wo, m);
        }
    } else {                                                      /* PLASTIC: [0] FresnelBlendDiffuse, [1] FresnelBlendSpecular */
        const int idx = (int)(s0 * 2);
        const float r0 = s0 * 2 - idx;
        if (idx == 0) {                                           /* FresnelBlendDiffuse_sample_f, :492-499 */
            *glossy = false;
            wi = sample_hemisphere(r0, s1);
            if (wo.z < 0.0f) wi.z = -wi.z;
            f = fb_diffuse_f(wi, wo, m);
            pdf = cos_hemisphere_pdf(wi, wo);
            if (pdf == 0.0f) return 0.0f;
            pdf += fb_specular_pdf(wi, wo, m);
            V3 g = v3(0.f, 0.f, 0.f);
            fb_specular_f(&g, wi, wo, m);
            f = add(f, g);
        } else {                                                  /* FresnelBlendSpecular_sample_f, :524-535 */
            *glossy = true;
            const V3 wh = beckmann_sample_wh(wo, r0, s1, ax, ay);
            wi = reflect_about(wo, wh);
            if (!same_hemisphere(wo, wi)) return 0.0f;
            fb_specular_f(&f, wi, wo, m);
            pdf = fb_specular_pdf(wi, wo, m);
            if (pdf == 0.0f) return 0.0f;
            pdf += cos_hemisphere_pdf(wi, wo);
            f = add(f, fb_diffuse_f(wi, wo, m));
        }
    }
    *f_out = f; *wi_out = wi;
    return pdf;
}

/* crt_rng: DESIGN.md "N1".  Three rounds of the murmur3 finaliser over (seed, pixel, sample, dim). */
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ float crt_rng(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    uint32_t h = fmix32(pixel ^ seed_lo);
    h = fmix32(h + sample * 0x9E3779B1u + seed_hi);
    h = fmix32(h ^ (dim * 0x85EBCA77u + 0x165667B1u));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
enum { DIM_IMAGE = 0, DIM_LENS = 2, DIM_BOUNCE0 = 4, DIMS_PER_BOUNCE = 8,
       DIM_LIGHT = 0, DIM_SCATTER = 2, DIM_BSDF = 4, DIM_PICK = 6, DIM_RR = 7 };

/* per-object shading record, indexed by insertion id (crt_hit.obj) */
struct ShadeObj {
    int32_t type, mat;
    float p[14];      /* TRIANGLE / FLAT: normal[3] tangent[3] | RECTANGLE: normal[3] dpdu[3] point[3] | SPHERE: center[3] radius |
                         SMOOTH: triangle index, normal_mat index (as int bits) | every other kind: index into shapes[] (as int bits) */
};
enum { LIGHT_ENV = 0, LIGHT_AREA = 1 };
struct LightRec {
    int32_t kind, obj_type;
    float color[3], intensity, power, pad;
    float p[12];      /* RECTANGLE: point width height normal | SPHERE: center radius | DISK: center normal radius */
};

struct RenderDev {
    const ShadeObj* objs; const crt_material* mats; const LightRec* lights;
    const float* vnormals; const float* vtangents; const float* normal_mats;      /* SMOOTH_TRIANGLE: crt_mesh_shading arrays */
    const crt_shape* shapes; int32_t num_shapes;                                  /* plane / disk / box / cylinder / torus / instanced / compound hits */
    int32_t num_lights, has_env, max_depth;
    float env_color[3], env_intensity, env_world_radius;
    /* textures: Texture records with device texel pointers, Matte.diffuse / Plastic.diffuse per material (-1 = none), texcoords of the
     * mesh triangles ([tri][6]), and the environment map (EnvLight type TEXTURE) with EnvLight.transform */
    const crt_texture* tex; const int32_t* mat_tex; const float* vuv;
    int32_t env_map;                      /* 1 + texture index, 0 = constant colour */
    float env_xf[9];
    uint32_t pix_offset;                  /* first pixel of the band this renderer renders (row_begin * width) */
    uint32_t seed_lo, seed_hi, npix;
    crt_camera cam; crt_film film;
    /* per path */
    float* beta; float* L; uint32_t* good;
    /* queues */
    crt_ray* rays[2]; uint32_t* qpath[2]; crt_hit* hits;
    crt_ray* srays; float* sdist; uint32_t* spath; float* scontrib; uint8_t* socc;
    /* queue sizes, all on the device: word 2b = rays queued for bounce b (word 0 = paths of the pass), word 2(b+1)+1 = shadow
     * rays queued by bounce b; bounce b's shading appends to both with one 64-bit atomic on the pair at word 2(b+1) */
    uint32_t* counters;
    float* color;                         /* [npix*3] running sum, main.cpp:139-141 */
};

/* Queue append for a 256-thread block: ONE atomic per block for both queues (the two counters are
 * the halves of one 64-bit word; neither half can carry, a queue holds < 2^31 entries).  Same-address
 * atomics serialise in the L2, and one per wave per queue was what the shading kernel spent most of
 * its time on.  All threads of the block must call; returns the slots through next_slot / shadow_slot. */
__device__ __forceinline__ void queue_push2(unsigned long long* counters, bool want_next, bool want_shadow,
                                            unsigned* next_slot, unsigned* shadow_slot)
{
    __shared__ unsigned s_cnt[2][4];
    __shared__ unsigned long long s_base;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mn = __ballot(want_next), ms = __ballot(want_shadow);
    if (lane == 0) { s_cnt[0][wave] = (unsigned)__popcll(mn); s_cnt[1][wave] = (unsigned)__popcll(ms); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tn = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
        const unsigned ts = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
        s_base = (tn | ts) ? atomicAdd(counters, (unsigned long long)tn | ((unsigned long long)ts << 32)) : 0ull;
    }
    __syncthreads();
    unsigned bn = (unsigned)s_base, bs = (unsigned)(s_base >> 32);
    for (unsigned w = 0; w < wave; w++) { bn += s_cnt[0][w]; bs += s_cnt[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    *next_slot = bn + (unsigned)__popcll(mn & below);
    *shadow_slot = bs + (unsigned)__popcll(ms & below);
}

/* getEnvLightIncRadiance (lights.cpp:233-248) for a direction already taken through EnvLight.transform */
__device__ __forceinline__ V3 env_radiance(const RenderDev& rd, const V3& dir)
{
    if (rd.env_map == 0) return scale(v3p(rd.env_color), rd.env_intensity);          /* CONSTANT */
    const crt_texture t = rd.tex[rd.env_map - 1];
    const float d[3] = { dir.x, dir.y, dir.z };
    float uv[2], c[3];
    crt_env_uv(uv, d);
    crt_tex_color(c, t.width, t.height, t.comp, t.is_float, t.data, uv);
    return scale(v3p(c), rd.env_intensity);
}
/* mat3_mult_vec3(out, env_light->transform, v), trace.h:300,:439 */
__device__ __forceinline__ V3 env_transform(const RenderDev& rd, const V3& v)
{
    return v3(dot(v3p(rd.env_xf), v), dot(v3p(rd.env_xf + 3), v), dot(v3p(rd.env_xf + 6), v));
}

/* threadFunc's per-pixel head, main.cpp:102-115: two Sampler_getSample values, calcImageCoord, calcCameraRay */
__global__ __launch_bounds__(256) void k_pt_generate(RenderDev rd, unsigned num_paths, unsigned first_sample)
{
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= num_paths) return;
    if (p == 0) rd.counters[0] = num_paths;                      /* size of bounce 0's queue */
    const unsigned pixel = rd.pix_offset + p % rd.npix, sample = first_sample + p / rd.npix;     /* pixel index in the whole frame */
    const float i0 = crt_rng(rd.seed_lo, rd.seed_hi, pixel, sample, DIM_IMAGE), i1 = crt_rng(rd.seed_lo, rd.seed_hi, pixel, sample, DIM_IMAGE + 1);
    const float l0 = crt_rng(rd.seed_lo, rd.seed_hi, pixel, sample, DIM_LENS), l1 = crt_rng(rd.seed_lo, rd.seed_hi, pixel, sample, DIM_LENS + 1);
    crt_ray ray;
    cam_primary_ray(&ray, rd.cam, rd.film, pixel, i0, i1, l0, l1);
    rd.rays[0][p] = ray;
    rd.qpath[0][p] = p;
    rd.beta[3 * (size_t)p] = 1.0f; rd.beta[3 * (size_t)p + 1] = 1.0f; rd.beta[3 * (size_t)p + 2] = 1.0f;   /* trace.h:405-406 */
    rd.L[3 * (size_t)p] = 0.0f; rd.L[3 * (size_t)p + 1] = 0.0f; rd.L[3 * (size_t)p + 2] = 0.0f;
    rd.good[p] = 0;
}

/* One iteration of pathTrace's loop body (trace.h:413-527) for every queued path, minus the two
 * scene queries: the closest hit arrives in hits[], the shadow query leaves through the shadow queue. */
__global__ __launch_bounds__(256) void k_pt_shade(RenderDev rd, unsigned n_max, int bounce, int cur, unsigned first_sample)
{
    unsigned n = rd.counters[2 * bounce];                        /* this bounce's queue size: never seen by the host */
    n = n < n_max ? n : n_max;
    if (blockIdx.x * blockDim.x >= n) return;                    /* the whole block: queue_push2 below needs all of a block's threads */
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    bool push_shadow = false, push_next = false;
    crt_ray sray, nray; float sdist = 0.f; V3 scontrib = v3(0.f, 0.f, 0.f);
    unsigned p = 0;
    if (active) {
        p = rd.qpath[cur][i];
        const crt_ray ray = rd.rays[cur][i];
        const crt_hit hit = rd.hits[i];
        const unsigned pixel = rd.pix_offset + p % rd.npix, samp = first_sample + p / rd.npix;
        V3 beta = v3p(rd.beta + 3 * (size_t)p);
        V3 L = v3p(rd.L + 3 * (size_t)p);
        unsigned good = rd.good[p] & 0x7FFFFFFFu;
        bool spec = (rd.good[p] >> 31) != 0;                      /* sampled_flags & (SPECULAR | GLOSSY) of the previous bounce */
        const bool is_hit = hit.obj >= 0;
        ShadeObj so; so.type = -1; so.mat = 0;
        crt_material mat; mat.type = 0;
        if (is_hit) { so = rd.objs[hit.obj]; mat = rd.mats[so.mat]; }
        /* emitted / environment light: on bounce 0, and after a specular lobe was sampled (trace.h:419) */
        if (bounce == 0 || spec) {
            if (is_hit) {
                if (mat.type == CRT_MAT_EMISSIVE) {
                    V3 inc = scale(v3p(mat.color), mat.intensity);
                    inc = mul(inc, beta);
                    L = add(L, inc);
                    good++;
                }
            } else {
                V3 env = v3(0.f, 0.f, 0.f);                       /* the reference leaves it uninitialised without an env light */
                if (rd.has_env) env = env_radiance(rd, env_transform(rd, v3p(ray.d)));   /* trace.h:436-441, lights.cpp:233-248 */
                env = mul(env, beta);
                L = add(L, env);
                good++;
            }
        }
        bool alive = is_hit && bounce < rd.max_depth && mat.type != CRT_MAT_EMISSIVE;    /* trace.h:459 */
        if (alive) {
            /* ---- ShadeRec ---- */
            const V3 ro = v3p(ray.o), rdir = v3p(ray.d);
            const V3 hp = add(ro, scale(rdir, hit.t));            /* getPointOnRay, util/ray.cpp:4-9 */
            const V3 wo = neg(rdir);
            V3 wo_sr = wo;                                        /* ShadeRec.wo: what estimateDirect hands BSDF_f (trace.h:328); differs from -d only behind a disk */
            V3 normal, dpdu;
            if (so.type == CRT_TRIANGLE) {                        /* triangle.cpp:159-166 */
                normal = v3p(so.p); dpdu = v3p(so.p + 3);
                if (dot(wo, normal) < 0.0f) normal = neg(normal);
            } else if (so.type == CRT_FLAT_TRIANGLE) {            /* triangle.cpp:199-202: not flipped */
                normal = v3p(so.p); dpdu = v3p(so.p + 3);
            } else if (so.type == CRT_SMOOTH_TRIANGLE) {          /* rayIntersectSmoothTriangle, triangle.cpp:226-251 */
                const size_t tri = (size_t)__float_as_int(so.p[0]);
                const float* nm = rd.normal_mats + 9 * (size_t)__float_as_int(so.p[1]);
                const float* vn = rd.vnormals + 9 * tri;
                const float* vt = rd.vtangents + 9 * tri;
                const float w0 = 1.0f - hit.beta - hit.gamma;
                /* interpTriangleVec3, triangle.cpp:206-216 */
                const V3 ni = normalize(add(add(scale(v3p(vn), w0), scale(v3p(vn + 3), hit.beta)), scale(v3p(vn + 6), hit.gamma)));
                normal = normalize(v3(dot(v3p(nm), ni), dot(v3p(nm + 3), ni), dot(v3p(nm + 6), ni)));        /* mat3_mult_vec3 */
                const V3 ti = normalize(add(add(scale(v3p(vt), w0), scale(v3p(vt + 3), hit.beta)), scale(v3p(vt + 6), hit.gamma)));
                dpdu = normalize(v3(dot(v3p(nm), ti), dot(v3p(nm + 3), ti), dot(v3p(nm + 6), ti)));
            } else if (so.type == CRT_RECTANGLE) {                /* rect.cpp:5-16,:34-46 */
                const V3 rn = v3p(so.p);
                const V3 disp = sub(v3p(so.p + 6), ro);
                normal = dot(disp, rn) < 0.0f ? neg(rn) : rn;
                dpdu = v3p(so.p + 3);
                if (dot(wo, normal) < 0.0f) { normal = neg(normal); dpdu = neg(dpdu); }
            } else if (so.type != CRT_SPHERE) {                   /* plane, disk, box, cylinder, torus, instanced, compound: crt_shaderec.h */
                crt_shade_rec sr;
                sr.written = 0;
                sr.normal[0] = sr.normal[1] = sr.normal[2] = 0.0f; sr.dpdu[0] = sr.dpdu[1] = sr.dpdu[2] = 0.0f;
                sr.wo[0] = wo.x; sr.wo[1] = wo.y; sr.wo[2] = wo.z;
                sr_fill_shape_object(rd.shapes, rd.num_shapes, __float_as_int(so.p[0]), &ray, &hit, 1, &sr);
                normal = v3p(sr.normal); dpdu = v3p(sr.dpdu); wo_sr = v3p(sr.wo);
            } else {                                              /* SPHERE, sphere.cpp:7-10,:25-29 */
                const V3 c = v3p(so.p);
                normal = normalize(sub(hp, c));
                const double phi_max = 2.0 * RD_PI;
                dpdu = normalize(v3((float)(-phi_max * (double)(hp.z - c.z)), 0.0f, (float)(phi_max * (double)(hp.x - c.x))));
            }
            /* ---- computeLocalBasis, trace.h:132-146 ---- */
            V3 tangent = dpdu;
            const float dp = dot(normal, tangent);
            if (dp != 0.0f) tangent = normalize(sub(tangent, scale(normal, dp)));
            const V3 binormal = normalize(cross(normal, tangent));
            /* ---- computeScatteringFunc: MATTE, sigma 0 -> one Lambertian; MIRROR -> one SpecularReflection (materials.cpp:116-136) ---- */
            V3 cd = v3p(mat.color);
            if (rd.mat_tex && rd.mat_tex[so.mat] >= 0 && (mat.type == CRT_MAT_MATTE || mat.type == CRT_MAT_PLASTIC)) {
                /* Matte_getDiffuseColor / Plastic.diffuse (materials.cpp:51-60,:146-149) at the hit's texcoord: interpTexcoord + the
                 * [0, 1] wrap of rayIntersectFlatTriangle / SmoothTriangle (triangle.cpp:191-198,:237-243); create() admits textured
                 * materials on these two kinds only */
                float uv[2], c[3];
                if (so.type == CRT_SPHERE || so.type == CRT_RECTANGLE) {
                    /* ShadeRec.uv as the object's own intersector writes it (sphere.cpp:14-19: phi / 2 pi and theta / pi of the root
                     * that was accepted; rect.cpp:28-37: the two edge coordinates), through the code crt_fill_shade_rec runs.  Not
                     * wrapped: getTexColor mirrors a negative u itself.  Device atan2f / acosf: class T4 (a texel boundary) */
                    crt_shade_rec sr;
                    sr.written = 0; sr.uv[0] = sr.uv[1] = 0.0f;
                    sr_fill_simple(&rd.shapes[__float_as_int(so.p[13])], ray.o, ray.d, hit.t, 1, &sr);
                    uv[0] = sr.uv[0]; uv[1] = sr.uv[1];
                } else {
                    const size_t tri = (size_t)__float_as_int(so.type == CRT_SMOOTH_TRIANGLE ? so.p[0] : so.p[13]);
                    sr_interp_uv(uv, hit.beta, hit.gamma, rd.vuv + 6 * tri);
                    sr_wrap_uv(uv);
                }
                const crt_texture t = rd.tex[rd.mat_tex[so.mat]];
                crt_tex_color(c, t.width, t.height, t.comp, t.is_float, t.data, uv);
                cd = v3p(c);
                mat.color[0] = c[0]; mat.color[1] = c[1]; mat.color[2] = c[2];           /* the lobes below read kd from the material */
            }
            const bool oren = mat.type == CRT_MAT_MATTE && mat.sigma != 0.0f;   /* BSDF_addOrenNayar; a, b sit in ior_in, ior_out of the device copy */
            const unsigned dim0 = DIM_BOUNCE0 + DIMS_PER_BOUNCE * (unsigned)bounce;
            #define RNG(d) crt_rng(rd.seed_lo, rd.seed_hi, pixel, samp, dim0 + (d))
            const float ls0 = RNG(DIM_LIGHT), ls1 = RNG(DIM_LIGHT + 1);
            /* scatter_sample (trace.h:469) is drawn and never used: dimensions DIM_SCATTER, +1 stay reserved */
            /* ---- uniformSampleOneLight, trace.h:359-397 ---- */
            if (rd.num_lights > 0 && mat.type != CRT_MAT_MIRROR && mat.type != CRT_MAT_TRANSPARENT && mat.type != CRT_MAT_GLASS &&
                mat.type != CRT_MAT_METAL) {                      /* trace.h:470; METAL's one lobe is GLOSSY, which BSDF_f excludes (trace.h:410): its estimate is black */
                const float u = RNG(DIM_PICK);
                int li = rd.num_lights - 1;                       /* the reference leaves light_index unset when no interval matches */
                float sel_pdf = rd.lights[li].power;
                float cdf = 0.0f;
                for (int k = 0; k < rd.num_lights; k++) {
                    const float pw = rd.lights[k].power;
                    if (cdf <= u && cdf + pw > u) { li = k; sel_pdf = pw; break; }
                    cdf += pw;
                }
                const LightRec lt = rd.lights[li];
                /* ---- estimateDirect, trace.h:221-357 ---- */
                V3 sp, sn, wi; float lpdf;
                if (lt.kind == LIGHT_AREA) {
                    if (lt.obj_type == CRT_SPHERE) {
                        const V3 c = v3p(lt.p); const float radius = lt.p[3];
                        const V3 z_axis = normalize(sub(hp, c));
                        V3 h = sample_hemisphere(ls0, ls1);
                        h = in_local_basis(h, z_axis);
                        sp = add(scale(h, radius), c);
                        sn = h;
                        lpdf = (float)(1.0f / (2.0f * RD_PI * (double)radius * (double)radius) * (double)fabsf(dot(h, z_axis)) * RD_INV_PI);
                    } else if (lt.obj_type == CRT_RECTANGLE) {
                        const V3 pt = v3p(lt.p), w = v3p(lt.p + 3), hgt = v3p(lt.p + 6);
                        sp = add(add(pt, scale(w, ls0)), scale(hgt, ls1));
                        sn = v3p(lt.p + 9);
                        lpdf = 1.0f / (length(w) * length(hgt));
                    } else {                                      /* DISK */
                        const V3 c = v3p(lt.p), dn = v3p(lt.p + 3); const float radius = lt.p[6];
                        float ds[2];
                        sample_disk(ds, ls0, ls1);
                        const V3 x_axis = normalize(cross(jittered_up(), dn));
                        const V3 y_axis = cross(x_axis, dn);
                        sp = add(add(scale(x_axis, ds[0] * radius), scale(y_axis, ds[1] * radius)), c);
                        sn = dn;
                        lpdf = (float)(1.0f / (RD_PI * (double)radius * (double)radius));
                    }
                    const V3 s2h = sub(sp, hp);
                    wi = normalize(s2h);
                    lpdf *= dot(s2h, s2h) / fabsf(dot(sn, neg(wi)));
                } else {                                          /* ENVLIGHT, trace.h:290-310 */
                    V3 h = sample_hemisphere(ls0, ls1);
                    h = ortho(tangent, binormal, normal, h);
                    h = env_transform(rd, h);                     /* mat3_mult_vec3(h_sample, env_light->transform, h_copy) */
                    sp = add(hp, scale(h, rd.env_world_radius));
                    sn = neg(h);
                    wi = h;
                    lpdf = (float)((double)fabsf(dot(h, normal)) * RD_INV_PI);
                }
                const V3 Li = lt.kind == LIGHT_ENV ? env_radiance(rd, wi) : scale(v3p(lt.color), lt.intensity);   /* trace.h:309 / :288 */
                const V3 s2h = sub(sp, hp);
                if (!(dot(s2h, sn) > 0.0f || dot(s2h, normal) < 0.0f) && lpdf > 0.0f && !is_black(Li)) {
                    V3 f = mat.type == CRT_MAT_PLASTIC ? fb_diffuse_f(wi, wo_sr, mat)             /* the DIFFUSE lobe of the two, on the WORLD vectors too */
                         : oren ? scale(cd, oren_nayar_scale(wi, wo_sr, mat.ior_in, mat.ior_out)) /* OrenNayar_f on the WORLD vectors */
                                : scale(cd, (float)RD_INV_PI);    /* Lambertian_f, reflection.cpp:159-163 */
                    f = scale(f, fabsf(dot(normal, wi)));
                    if (!is_black(f)) {
                        V3 radiance = mul(f, Li);
                        radiance = scale(radiance, 1.0f / lpdf);
                        V3 contrib = scale(radiance, 1.0f / sel_pdf);     /* trace.h:396 */
                        contrib = mul(contrib, beta);                      /* trace.h:474 */
                        push_shadow = true;
                        sdist = length(s2h);
                        sray.o[0] = hp.x; sray.o[1] = hp.y; sray.o[2] = hp.z;
                        sray.d[0] = wi.x; sray.d[1] = wi.y; sray.d[2] = wi.z;
                        scontrib = contrib;
                    }
                }
            }
            /* ---- BSDF_sample_f (reflection.cpp:750-811) with the material's one lobe ---- */
            const float bs0 = RNG(DIM_BSDF), bs1 = RNG(DIM_BSDF + 1);
            const V3 wo_l = v3(dot(wo, tangent), dot(wo, binormal), dot(wo, normal));    /* transposeTransform */
            alive = false;
            if (wo_l.z != 0.0f) {
                V3 f, wi_l; float pdf;
                if (mat.type == CRT_MAT_MIRROR) {                 /* SpecularReflection_sample_f, reflection.cpp:234-242 */
                    spec = true;                                  /* getBxDFFlagsFromType: REFLECTION | SPECULAR */
                    wi_l = v3(-wo_l.x, -wo_l.y, wo_l.z);
                    f = scale(cd, 1.0f / fabsf(wi_l.z));
                    pdf = 1.0f;
                } else if (mat.type == CRT_MAT_TRANSPARENT) {     /* SpecularTransmission_sample_f, `thin` branch, reflection.cpp:243-269 */
                    spec = true;                                  /* REFLECTION | TRANSMISSION | SPECULAR */
                    const float nz = wo_l.z < 0.0f ? -1.0f : 1.0f;
                    /* vec3_dot(normal, wo) with normal = (0, 0, 1) is 0*x + 0*y + 1*z: the sign test reads that sum */
                    const float kr = fresnel_dielectric(nz, wo_l.z, mat.ior_in, mat.ior_out);
                    const float u = RNG(DIM_SCATTER);             /* the reference's rand() (reflection.cpp:254) */
                    if (u <= kr) {
                        wi_l = v3(-wo_l.x, -wo_l.y, wo_l.z);
                        f = scale(v3(1.0f, 1.0f, 1.0f), kr / fabsf(wi_l.z));
                        pdf = kr;
                    } else {
                        const float eta = mat.ior_out / mat.ior_in;
                        wi_l = neg(wo_l);
                        f = scale(v3(1.0f, 1.0f, 1.0f), 1.0f - kr);
                        f = scale(f, (eta * eta) / fabsf(wi_l.z));
                        pdf = 1.0f - kr;
                    }
                } else if (mat.type == CRT_MAT_PLASTIC || mat.type == CRT_MAT_GLASS || mat.type == CRT_MAT_METAL) {
                    pdf = glossy_sample_f(mat, wo_l, bs0, bs1, RNG(DIM_SCATTER), &f, &wi_l, &spec);
                    if (pdf == 0.0f) { f = v3(0.f, 0.f, 0.f); wi_l = v3(0.f, 0.f, 1.0f); }
                } else {                                          /* Lambertian_sample_f, reflection.cpp:177-189 */
                    spec = false;
                    const float two_z = fabsf(wo_l.z);            /* tmp_wo flipped into the upper hemisphere */
                    wi_l = sample_hemisphere(bs0, bs1);
                    pdf = (wi_l.z * two_z > 0.0f) ? (float)((double)fabsf(wi_l.z) * RD_INV_PI) : 0.0f;   /* cosHemispherePdf */
                    f = oren ? scale(cd, oren_nayar_scale(wi_l, wo_l, mat.ior_in, mat.ior_out))          /* OrenNayar_sample_f, :221-232 */
                             : scale(cd, (float)RD_INV_PI);
                }
                if (pdf != 0.0f && !is_black(f)) {
                    const V3 wi = ortho(tangent, binormal, normal, wi_l);
                    f = scale(f, fabsf(dot(wi, normal)) / pdf);   /* trace.h:494 */
                    beta = mul(beta, f);
                    alive = true;
                    if (bounce > 3) {                             /* Russian roulette, trace.h:512-525 */
                        const float m12 = (beta.y > beta.z) ? beta.y : beta.z;
                        const float max_comp = (beta.x > m12) ? beta.x : m12;
                        const float omc = 1.0f - max_comp;
                        const float q = (0.05f > omc) ? 0.05f : omc;
                        const float rr = RNG(DIM_RR);
                        if (rr < q) alive = false;
                        else beta = scale(beta, 1.0f / (1.0f - q));
                    }
                    if (alive) {
                        push_next = true;
                        nray.o[0] = hp.x; nray.o[1] = hp.y; nray.o[2] = hp.z;
                        nray.d[0] = wi.x; nray.d[1] = wi.y; nray.d[2] = wi.z;
                    }
                }
            }
            #undef RNG
        }
        rd.beta[3 * (size_t)p] = beta.x; rd.beta[3 * (size_t)p + 1] = beta.y; rd.beta[3 * (size_t)p + 2] = beta.z;
        rd.L[3 * (size_t)p] = L.x; rd.L[3 * (size_t)p + 1] = L.y; rd.L[3 * (size_t)p + 2] = L.z;
        rd.good[p] = good | (spec ? 0x80000000u : 0u);
    }
    unsigned ns, ss;
    queue_push2((unsigned long long*)(rd.counters + 2 * (bounce + 1)), push_next, push_shadow, &ns, &ss);
    if (push_shadow) {
        rd.srays[ss] = sray; rd.sdist[ss] = sdist; rd.spath[ss] = p;
        rd.scontrib[3 * (size_t)ss] = scontrib.x; rd.scontrib[3 * (size_t)ss + 1] = scontrib.y; rd.scontrib[3 * (size_t)ss + 2] = scontrib.z;
    }
    if (push_next) { rd.rays[cur ^ 1][ns] = nray; rd.qpath[cur ^ 1][ns] = p; }
}

/* tail of estimateDirect / pathTrace's direct-light step: `if(!in_shadow)` (trace.h:348-355), then
 * L += contrib and good_paths++ when contrib is not black (trace.h:475-477) */
__global__ __launch_bounds__(256) void k_pt_accumulate(RenderDev rd, unsigned n_max, int bounce)
{
    unsigned n = rd.counters[2 * (bounce + 1) + 1];              /* shadow rays queued by this bounce */
    n = n < n_max ? n : n_max;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (rd.socc[i]) return;
    const unsigned p = rd.spath[i];
    const V3 c = v3p(rd.scontrib + 3 * (size_t)i);
    rd.L[3 * (size_t)p] += c.x; rd.L[3 * (size_t)p + 1] += c.y; rd.L[3 * (size_t)p + 2] += c.z;
    if (!is_black(c)) rd.good[p] += 1;
}

/* end of pathTrace (trace.h:528-530) and of threadFunc's per-pixel body (main.cpp:125-141), for the
 * `spp` samples of this pass in sample order */
__global__ __launch_bounds__(256) void k_pt_finish(RenderDev rd, unsigned spp, unsigned first_sample)
{
    const unsigned pixel = blockIdx.x * blockDim.x + threadIdx.x;
    if (pixel >= rd.npix) return;
    V3 buf = v3p(rd.color + 3 * (size_t)pixel);
    for (unsigned s = 0; s < spp; s++) {
        const size_t p = (size_t)s * rd.npix + pixel;
        V3 L = v3p(rd.L + 3 * p);
        const unsigned good = rd.good[p] & 0x7FFFFFFFu;
        if (good > 0) L = scale(L, 1.0f / (float)(int)good);
        V3 color = add(v3(0.f, 0.f, 0.f), L);
        if (isnan(color.x) || isnan(color.y) || isnan(color.z) || isinf(color.x) || isinf(color.y) || isinf(color.z))
            color = scale(buf, 1 / (float)(first_sample + s + 1));      /* main.cpp:133-135 */
        buf = add(buf, color);
    }
    rd.color[3 * (size_t)pixel] = buf.x; rd.color[3 * (size_t)pixel + 1] = buf.y; rd.color[3 * (size_t)pixel + 2] = buf.z;
}

} // namespace

struct crt_renderer {
    crt_scene* scene;
    int device;
    RenderDev rd;
    uint32_t spp_max;          /* samples in flight */
    size_t path_cap;
    uint64_t samples;
    crt_render_stats stats;
    hipStream_t stream;
    hipStream_t stream2; bool stream2_valid;          /* the shadow query and the accumulation of a bounce run here, beside the next bounce's extend query */
    int nobin_s_upto;
    int nobin_e_upto;                                 /* queries that go out unbinned: see crt_renderer_create */
    std::vector<hipEvent_t> ev_sync;                  /* 2 per bounce launched in a pass: shade done (stream), accumulate done (stream2) */
    std::vector<void*> allocs;
    /* the host follows the device two bounces behind: the queue sizes of each bounce land here (page-locked) */
    uint32_t* h_counts; size_t counts_words;
    std::vector<hipEvent_t> ev_count;                 /* ring of 4: bounce b's sizes have arrived */
    std::vector<hipEvent_t> ev_time;                  /* 4 per bounce launched in a pass: extend start / stop, shadow start / stop */
};

namespace {

struct DevGuard {
    int prev;
    explicit DevGuard(int dev) : prev(-1) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(dev); }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <class T>
int dev_alloc(crt_renderer* r, T** out, size_t count)
{
    void* p = 0;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 256;
    if (hipMalloc(&p, bytes) != hipSuccess) return CRT_E_NOMEM;
    r->allocs.push_back(p);
    *out = (T*)p;
    return CRT_OK;
}

void host_normalize(float* r, const float* a) { cam_normalize(r, a); }
void host_cross(float* r, const float* a, const float* b)
{
    r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
}
float host_length(const float* a) { return (float)sqrt((double)(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])); }

} // namespace

extern "C" {

void crt_renderer_destroy(crt_renderer* r)
{
    if (!r) return;
    DevGuard g(r->device);
    if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
    for (void* p : r->allocs) (void)hipFree(p);
    for (hipEvent_t e : r->ev_count) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->ev_time) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->ev_sync) (void)hipEventDestroy(e);
    if (r->stream2_valid) (void)hipStreamDestroy(r->stream2);
    if (r->h_counts) (void)hipHostFree(r->h_counts);
    delete r;
}

int crt_renderer_create(crt_scene* scene, const crt_render_desc* d, const crt_camera* cam, const crt_film* film, crt_renderer** out)
{
    if (!out) return CRT_E_INVALID;
    *out = 0;
    if (!scene || !d || !cam || !film || !d->objects) return CRT_E_INVALID;
    const crt_scene_input* in = d->objects;
    if (in->num_obj < 0 || d->num_materials < 0 || d->max_depth < 0) return CRT_E_INVALID;
    if (in->num_obj > 0 && (!d->obj_material || !d->materials)) return CRT_E_INVALID;
    if (film->frame_res_width < 1 || film->frame_res_height < 1) return CRT_E_INVALID;
    const unsigned long long npix64 = (unsigned long long)film->frame_res_width * (unsigned long long)film->frame_res_height;
    if (npix64 > (1ull << 28)) return CRT_E_TOOBIG;
    if (4 + 8ull * ((unsigned long long)d->max_depth + 1) > 0xFFFFFFFFull) return CRT_E_TOOBIG;
    for (int m = 0; m < d->num_materials; m++) {
        const crt_material& mt = d->materials[m];
        if (mt.type != CRT_MAT_MATTE && mt.type != CRT_MAT_EMISSIVE && mt.type != CRT_MAT_MIRROR && mt.type != CRT_MAT_TRANSPARENT &&
            mt.type != CRT_MAT_PLASTIC && mt.type != CRT_MAT_GLASS && mt.type != CRT_MAT_METAL) return CRT_E_UNSUPPORTED;
        /* pathTrace paints a path BLUE / RED when a component of its throughput is negative (debug overwrites, trace.h:452-453,
         * :478-479).  With non-negative colours and PLASTIC ks <= 1 (the Fresnel-blend diffuse lobe scales by 1 - ks,
         * reflection.cpp:479-496) no lobe value is negative, the throughput never is, and the overwrites cannot trigger;
         * materials outside those bounds are refused instead of painted */
        for (int k = 0; k < 3; k++) {
            if (mt.type != CRT_MAT_METAL && !(mt.color[k] >= 0.0f)) return CRT_E_UNSUPPORTED;      /* METAL: color = eta, k: any sign squared */
            if (mt.type == CRT_MAT_PLASTIC && !(mt.ks[k] >= 0.0f && mt.ks[k] <= 1.0f)) return CRT_E_UNSUPPORTED;
        }
    }
    if (d->num_textures < 0 || (d->num_textures > 0 && !d->textures)) return CRT_E_INVALID;
    for (int k = 0; k < d->num_textures; k++) {
        const crt_texture& t = d->textures[k];
        if (t.width < 1 || t.height < 1 || t.comp < 3 || t.comp > 4 || !t.data) return CRT_E_INVALID;    /* getTexColor reads three channels */
        if ((unsigned long long)t.width * (unsigned long long)t.height > (1ull << 28)) return CRT_E_TOOBIG;
        if (t.is_float) {                                         /* the same bound for float texels (8-bit ones cannot be negative) */
            const float* px = (const float*)t.data;
            const size_t count = (size_t)t.width * (size_t)t.height * (size_t)t.comp;
            for (size_t q = 0; q < count; q++) if (!(px[q] >= 0.0f)) return CRT_E_UNSUPPORTED;
        }
    }
    if (d->env_map < 0 || d->env_map > d->num_textures) return CRT_E_INVALID;
    if (d->env_map && !d->has_env_light) return CRT_E_INVALID;
    bool any_tex = false;
    if (d->material_diffuse_tex)
        for (int m = 0; m < d->num_materials; m++) {
            const int ti = d->material_diffuse_tex[m];
            if (ti < -1 || ti >= d->num_textures) return CRT_E_INVALID;
            if (ti >= 0) any_tex = true;
        }
    int row_begin = d->row_begin, row_end = d->row_end;
    if (row_begin == 0 && row_end == 0) row_end = film->frame_res_height;
    if (row_begin < 0 || row_end <= row_begin || row_end > film->frame_res_height) return CRT_E_INVALID;
    std::vector<ShadeObj> objs((size_t)in->num_obj);
    std::vector<LightRec> lights;
    const crt_mesh_shading* ms = d->mesh_shading;
    bool any_smooth = false, any_generic = false;
    if (d->has_env_light && d->env_intensity > 0.0f) {            /* the scene file's ENV_LIGHT precedes the objects: light 0; with intensity <= 0 it
                                                                     is seen (black) but is no light, buildscene.h:525-533 */
        LightRec l; memset(&l, 0, sizeof l);
        l.kind = LIGHT_ENV; l.obj_type = -1;
        memcpy(l.color, d->env_color, sizeof l.color);
        l.intensity = d->env_intensity;
        /* buildscene.h:894-895 averages EnvLight.color -- which the TEXTURE branch of the parser never sets (:479-483, malloc'd
         * memory); WHITE is taken there, as the parser's own fallback does (:486-487) */
        if (d->env_map) { l.color[0] = l.color[1] = l.color[2] = 1.0f; }
        l.power = (l.color[0] + l.color[1] + l.color[2]) / 3.0f;
        l.power *= d->env_intensity * d->env_world_radius;
        lights.push_back(l);
    }
    for (int i = 0; i < in->num_obj; i++) {
        const int type = in->obj_type ? in->obj_type[i] : CRT_SMOOTH_TRIANGLE;
        const int ref = in->obj_ref ? in->obj_ref[i] : i;
        const int mi = d->obj_material[i];
        if (mi < 0 || mi >= d->num_materials) return CRT_E_INVALID;
        const crt_material& mt = d->materials[mi];
        ShadeObj so; memset(&so, 0, sizeof so);
        so.type = type; so.mat = mi;
        if (any_tex && d->material_diffuse_tex[mi] >= 0 && (mt.type == CRT_MAT_MATTE || mt.type == CRT_MAT_PLASTIC)) {
            /* a texture needs a texcoord the reference defines: the mesh triangle kinds (triangle.cpp:191-198,:237-243), and the
             * sphere and the rectangle, whose intersectors write ShadeRec.uv (fillShadeRecSphere, sphere.cpp:14-19; rect.cpp:28-37) */
            if (type == CRT_FLAT_TRIANGLE || type == CRT_SMOOTH_TRIANGLE) { if (!ms || !ms->vertex_uv) return CRT_E_UNSUPPORTED; }
            else if (type == CRT_SPHERE || type == CRT_RECTANGLE) any_generic = true;      /* the shape table goes to the device */
            else return CRT_E_UNSUPPORTED;
        }
        /* EMISSIVE mesh triangles: initMeshLights (buildscene.h:749-833) registers them as MeshLights, which preprocessLights
         * gives no power (no MESHLIGHT case, buildscene.h:879-917) and estimateDirect cannot sample ("Invalid light type",
         * trace.h:311-313).  For pathTrace they are surfaces that emit when a camera ray or a specular / glossy bounce finds them;
         * no light record is made. */
        if (type == CRT_SMOOTH_TRIANGLE) {
            if (ref < 0 || ref >= in->num_tris || !in->tri_verts) return CRT_E_INVALID;
            if (!ms || !ms->vertex_normals || !ms->vertex_tangents || !ms->normal_mat_index || !ms->normal_mats) return CRT_E_UNSUPPORTED;
            const int nmi = ms->normal_mat_index[ref];
            if (nmi < 0 || nmi >= ms->num_normal_mats) return CRT_E_INVALID;
            memcpy(so.p, &ref, 4); memcpy(so.p + 1, &nmi, 4);
            any_smooth = true;
        } else if (type == CRT_TRIANGLE || type == CRT_FLAT_TRIANGLE) {
            if (ref < 0 || ref >= in->num_tris || !in->tri_verts) return CRT_E_INVALID;
            const float* v = in->tri_verts + 9 * (size_t)ref;
            const float e1[3] = { v[3] - v[0], v[4] - v[1], v[5] - v[2] }, e2[3] = { v[6] - v[0], v[7] - v[1], v[8] - v[2] };
            float c[3];
            host_cross(c, e1, e2);
            host_normalize(so.p, c);                              /* calcTriangleNormal, triangle.cpp:4-12 */
            if (type == CRT_FLAT_TRIANGLE && ms && ms->flat_normal) memcpy(so.p, ms->flat_normal + 3 * (size_t)ref, 12);   /* FlatTriangle.normal */
            host_normalize(so.p + 3, e1);                         /* dpdu: the reference leaves it unset for these two types; see DESIGN.md "N1" */
            memcpy(so.p + 13, &ref, 4);                           /* triangle index: texcoords of a textured FLAT_TRIANGLE */
        } else if (type == CRT_RECTANGLE || type == CRT_SPHERE || type == CRT_DISK) {
            if (ref < 0 || ref >= in->num_shapes || !in->shapes || in->shapes[ref].type != type) return CRT_E_INVALID;
            const float* sp = in->shapes[ref].p;
            memcpy(so.p + 13, &ref, 4);                           /* index into shapes[]: the uv of a textured sphere / rectangle */
            if (type == CRT_RECTANGLE) {
                memcpy(so.p, sp + 9, 12);                         /* normal */
                host_normalize(so.p + 3, sp + 3);                 /* dpdu = normalize(width), rect.cpp:35 */
                memcpy(so.p + 6, sp, 12);                         /* point */
            } else if (type == CRT_SPHERE) {
                memcpy(so.p, sp, 16);                             /* center, radius */
            } else {                                              /* DISK: shaded through crt_shaderec.h (its back side by convention) */
                memcpy(so.p, &ref, 4);
                any_generic = true;
            }
            if (mt.type == CRT_MAT_EMISSIVE) {                    /* initAreaLights, buildscene.h:567-608 */
                LightRec l; memset(&l, 0, sizeof l);
                l.kind = LIGHT_AREA; l.obj_type = type;
                memcpy(l.color, mt.color, sizeof l.color);
                l.intensity = mt.intensity;
                float area;
                if (type == CRT_SPHERE) { memcpy(l.p, sp, 16); area = (float)(4.0f * RD_PI * (double)sp[3] * (double)sp[3]); }
                else if (type == CRT_RECTANGLE) { memcpy(l.p, sp, 48); area = host_length(sp + 3) * host_length(sp + 6); }
                else { memcpy(l.p, sp, 28); area = (float)(RD_PI * (double)sp[6] * (double)sp[6]); }
                l.power = (l.color[0] * l.color[1] * l.color[2]) / 3.0f * l.intensity * area;   /* buildscene.h:914-915 (a product, as written there) */
                lights.push_back(l);
            }
        } else if (type == CRT_PLANE || type == CRT_AABOX || type == CRT_GENERICOPENCYLINDER || type == CRT_GENERICTORUS ||
                   type == CRT_INSTANCED || type == CRT_COMPOUND) {
            /* the shapes whose intersectors leave dpdu (and more) unset: shaded with the conventions of crt_fill_shade_rec
             * (DESIGN.md "a-13"); nesting as crt_scene_create allows it: instanced { simple | compound { simple } } */
            if (ref < 0 || ref >= in->num_shapes || !in->shapes || in->shapes[ref].type != type) return CRT_E_INVALID;
            memcpy(so.p, &ref, 4);
            any_generic = true;
        } else {
            return CRT_E_UNSUPPORTED;
        }
        objs[(size_t)i] = so;
    }
    if (lights.size() > 0x7FFFFFFFu) return CRT_E_TOOBIG;
    float total_power = 0.0f;
    for (const LightRec& l : lights) total_power += l.power;
    for (LightRec& l : lights) l.power /= total_power;            /* buildscene.h:921-922 */

    int device = 0, scene_objs = -1;
    int rc = crt_scene_device(scene, &device);
    if (rc != CRT_OK) return rc;
    if ((rc = crt_scene_num_objects(scene, &scene_objs)) != CRT_OK) return rc;
    if (scene_objs != in->num_obj) return CRT_E_INVALID;          /* hit ids index the per-object tables built above */
    crt_renderer* r = new (std::nothrow) crt_renderer();
    if (!r) return CRT_E_NOMEM;
    r->scene = scene; r->device = device; r->samples = 0; r->stream = 0; r->stream2 = 0; r->stream2_valid = false;
    r->h_counts = 0; r->counts_words = 0;
    /* The renderer's queries are not binned (round 3).  Its ray queues are in path order -- pixel by pixel, sample by sample,
     * compacted as paths die -- which keeps neighbouring rays near each other for as long as the surfaces do; sorting them once
     * more by the cell of their origin costs 7 % of the kernel time and was measured to lose against leaving them alone at
     * every bounce (4K courtyard, depth 5: 31.6 samples/s/pixel all binned, 32.1 camera rays and their shadow rays left alone,
     * 33.5 nothing binned).  A/B knobs: CRT_RENDER_NOBIN_EXTEND_UPTO / CRT_RENDER_NOBIN_SHADOW_UPTO = last bounce whose extend /
     * shadow query goes out unbinned (-1: every query is binned, as crt_intersect_batch_dev does for a caller's batch). */
    { const char* e = getenv("CRT_RENDER_NOBIN_EXTEND_UPTO"); r->nobin_e_upto = e ? atoi(e) : (1 << 30); }
    { const char* e = getenv("CRT_RENDER_NOBIN_SHADOW_UPTO"); r->nobin_s_upto = e ? atoi(e) : (1 << 30); }
    memset(&r->stats, 0, sizeof r->stats);
    memset(&r->rd, 0, sizeof r->rd);
    DevGuard g(device);
    const uint32_t npix = (uint32_t)(row_end - row_begin) * (uint32_t)film->frame_res_width;     /* the band this renderer renders */
    uint32_t spp = d->samples_in_flight;
    if (spp == 0) { spp = (uint32_t)((1u << 25) / npix); if (spp < 1) spp = 1; }    /* ~32 M paths per pass (150 bytes of state each): the
                                                                                     walk kernels' tails shrink with the batch, 22.2 -> 25.0 samples/s/pixel at 4K */
    if ((unsigned long long)spp * npix > 0x40000000ull) { delete r; return CRT_E_TOOBIG; }
    r->spp_max = spp;
    r->path_cap = (size_t)spp * npix;
    const size_t P = r->path_cap;
    RenderDev& rd = r->rd;
    ShadeObj* d_objs = 0; crt_material* d_mats = 0; LightRec* d_lights = 0;
    float* d_vn = 0; float* d_vt = 0; float* d_nm = 0;
    crt_shape* d_shapes = 0;
    if (any_generic) {                                            /* every index sr_fill_shape_object follows, checked before the kernel sees it */
        for (int i = 0; i < in->num_shapes; i++) {
            const crt_shape& sh = in->shapes[i];
            if (sh.type == CRT_INSTANCED) {
                if (sh.inner < 0 || sh.inner >= in->num_shapes) { delete r; return CRT_E_INVALID; }
                if (in->shapes[sh.inner].type == CRT_INSTANCED) { delete r; return CRT_E_UNSUPPORTED; }
            } else if (sh.type == CRT_COMPOUND) {
                if (sh.first_part < 0 || sh.num_parts < 0 || sh.first_part + sh.num_parts > in->num_shapes) { delete r; return CRT_E_INVALID; }
                for (int k = 0; k < sh.num_parts; k++) {
                    const int pt = in->shapes[sh.first_part + k].type;
                    if (pt == CRT_INSTANCED || pt == CRT_COMPOUND) { delete r; return CRT_E_UNSUPPORTED; }
                }
            }
        }
    }
    rc = dev_alloc(r, &d_objs, objs.size());
    if (rc == CRT_OK && any_generic) rc = dev_alloc(r, &d_shapes, (size_t)in->num_shapes);
    if (rc == CRT_OK && any_smooth) rc = dev_alloc(r, &d_vn, 9 * (size_t)in->num_tris);
    if (rc == CRT_OK && any_smooth) rc = dev_alloc(r, &d_vt, 9 * (size_t)in->num_tris);
    if (rc == CRT_OK && any_smooth) rc = dev_alloc(r, &d_nm, 9 * (size_t)ms->num_normal_mats);
    if (rc == CRT_OK) rc = dev_alloc(r, &d_mats, (size_t)d->num_materials);
    if (rc == CRT_OK) rc = dev_alloc(r, &d_lights, lights.size());
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.beta, 3 * P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.L, 3 * P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.good, P);
    for (int k = 0; k < 2 && rc == CRT_OK; k++) { rc = dev_alloc(r, &rd.rays[k], P); if (rc == CRT_OK) rc = dev_alloc(r, &rd.qpath[k], P); }
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.hits, P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.srays, P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.sdist, P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.spath, P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.scontrib, 3 * P);
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.socc, P);
    if (d->max_depth > (1 << 20)) { crt_renderer_destroy(r); return CRT_E_TOOBIG; }      /* two queue-size words per bounce, device and host */
    r->counts_words = 2 * ((size_t)d->max_depth + 2);
    if (r->counts_words < 64) r->counts_words = 64;
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.counters, r->counts_words);
    if (rc == CRT_OK && hipHostMalloc((void**)&r->h_counts, r->counts_words * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { r->h_counts = 0; rc = CRT_E_NOMEM; }
    for (int k = 0; k < 4 && rc == CRT_OK; k++) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = CRT_E_HIP; else r->ev_count.push_back(e);
    }
    if (rc == CRT_OK) rc = dev_alloc(r, &rd.color, 3 * (size_t)npix);
    bool ok = rc == CRT_OK;
    ok = ok && hipMemcpy(d_objs, objs.data(), objs.size() * sizeof(ShadeObj), hipMemcpyHostToDevice) == hipSuccess;
    std::vector<crt_material> dev_mats(d->materials, d->materials + d->num_materials);
    for (crt_material& m : dev_mats) {
        /* the device copy carries the Beckmann alphas in uroughness / vroughness: BSDF_addFresnelBlendSpecular and
         * BSDF_addMicrofacetReflectionMetal take the roughness as alpha, BSDF_addMicrofacetFresnel maps it through
         * BeckmannRoughnessToAlpha (reflection.cpp:879-955; microfacet.h:25-31 -- host code, glibc's logf as there) */
        if (m.type == CRT_MAT_PLASTIC) m.vroughness = m.uroughness;                   /* materials.cpp:168-169 passes roughness twice */
        if (m.type == CRT_MAT_GLASS) {
            float* rough[2] = { &m.uroughness, &m.vroughness };
            for (float* pr : rough) {
                const float rr = (*pr > (float)1e-3) ? *pr : (float)1e-3;
                const float x = logf(rr);
                *pr = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
            }
        }
        if (m.type != CRT_MAT_MATTE || m.sigma == 0.0f) continue;
        const float sigma_rad = m.sigma / 180.0f * (float)RD_PI;  /* BSDF_addOrenNayar, reflection.cpp:823-830; degToRad, util/math.h:65 */
        const float sigma2 = sigma_rad * sigma_rad;
        m.ior_in = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));   /* a */
        m.ior_out = 0.45f * sigma2 / (sigma2 + 0.09f);            /* b */
    }
    ok = ok && hipMemcpy(d_mats, dev_mats.data(), dev_mats.size() * sizeof(crt_material), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(d_lights, lights.data(), lights.size() * sizeof(LightRec), hipMemcpyHostToDevice) == hipSuccess;
    if (any_generic) ok = ok && hipMemcpy(d_shapes, in->shapes, (size_t)in->num_shapes * sizeof(crt_shape), hipMemcpyHostToDevice) == hipSuccess;
    if (any_smooth) {
        ok = ok && hipMemcpy(d_vn, ms->vertex_normals, 9 * (size_t)in->num_tris * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(d_vt, ms->vertex_tangents, 9 * (size_t)in->num_tris * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(d_nm, ms->normal_mats, 9 * (size_t)ms->num_normal_mats * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    /* textures: texel arrays copied as they are, Texture records with device pointers; texcoords of the mesh triangles */
    crt_texture* d_tex = 0; int32_t* d_mat_tex = 0; float* d_vuv = 0;
    if (ok && d->num_textures > 0 && (any_tex || d->env_map)) {
        std::vector<crt_texture> tex(d->textures, d->textures + d->num_textures);
        for (crt_texture& t : tex) {
            const size_t bytes = (size_t)t.width * (size_t)t.height * (size_t)t.comp * (t.is_float ? sizeof(float) : 1);
            unsigned char* dp = 0;
            ok = ok && dev_alloc(r, &dp, bytes) == CRT_OK;
            ok = ok && hipMemcpy(dp, t.data, bytes, hipMemcpyHostToDevice) == hipSuccess;
            t.data = dp;
        }
        ok = ok && dev_alloc(r, &d_tex, tex.size()) == CRT_OK;
        ok = ok && hipMemcpy(d_tex, tex.data(), tex.size() * sizeof(crt_texture), hipMemcpyHostToDevice) == hipSuccess;
        if (any_tex) {
            ok = ok && dev_alloc(r, &d_mat_tex, (size_t)d->num_materials) == CRT_OK;
            ok = ok && hipMemcpy(d_mat_tex, d->material_diffuse_tex, (size_t)d->num_materials * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess;
            if (ms && ms->vertex_uv && in->num_tris > 0) {
                ok = ok && dev_alloc(r, &d_vuv, 6 * (size_t)in->num_tris) == CRT_OK;
                ok = ok && hipMemcpy(d_vuv, ms->vertex_uv, 6 * (size_t)in->num_tris * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
            }
        }
    }
    ok = ok && hipMemset(rd.color, 0, 3 * (size_t)npix * sizeof(float)) == hipSuccess;
    ok = ok && hipMemset(rd.counters, 0, r->counts_words * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipStreamCreate(&r->stream) == hipSuccess;
    if (ok) { r->stream2_valid = hipStreamCreate(&r->stream2) == hipSuccess; ok = r->stream2_valid; }
    if (!ok) { crt_renderer_destroy(r); return rc != CRT_OK ? rc : CRT_E_HIP; }
    rd.objs = d_objs; rd.mats = d_mats; rd.lights = d_lights;
    rd.vnormals = d_vn; rd.vtangents = d_vt; rd.normal_mats = d_nm;
    rd.shapes = d_shapes; rd.num_shapes = any_generic ? in->num_shapes : 0;
    rd.num_lights = (int32_t)lights.size();
    rd.has_env = d->has_env_light ? 1 : 0;
    rd.tex = d_tex; rd.mat_tex = d_mat_tex; rd.vuv = d_vuv;
    rd.env_map = d->env_map;
    {
        bool zero = true;
        for (int k = 0; k < 9; k++) if (d->env_transform[k] != 0.0f) zero = false;
        for (int k = 0; k < 9; k++) rd.env_xf[k] = zero ? ((k % 4 == 0) ? 1.0f : 0.0f) : d->env_transform[k];
    }
    rd.pix_offset = (uint32_t)row_begin * (uint32_t)film->frame_res_width;
    rd.max_depth = d->max_depth;
    memcpy(rd.env_color, d->env_color, sizeof rd.env_color);
    rd.env_intensity = d->env_intensity; rd.env_world_radius = d->env_world_radius;
    rd.seed_lo = (uint32_t)d->seed; rd.seed_hi = (uint32_t)(d->seed >> 32);
    rd.npix = npix; rd.cam = *cam; rd.film = *film;
    *out = r;
    return CRT_OK;
}

int crt_renderer_render(crt_renderer* r, uint32_t num_samples)
{
    if (!r) return CRT_E_INVALID;
    if (r->samples + num_samples > 0xFFFFFFF0ull) return CRT_E_TOOBIG;
    DevGuard g(r->device);
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t st = r->stream;
    RenderDev& rd = r->rd;
    uint32_t done = 0;
    while (done < num_samples) {
        const uint32_t spp = num_samples - done < r->spp_max ? num_samples - done : r->spp_max;
        const uint32_t first_sample = (uint32_t)r->samples;
        const unsigned P = spp * rd.npix;
        /* The queue sizes never leave the device on the critical path: every launch of a bounce reads its ray count from
         * rd.counters (counted launches, sized for P).  The host only FOLLOWS, two bounces behind, through an 8-byte copy
         * into page-locked memory per bounce, to stop launching once the paths have died out (Russian roulette, misses)
         * -- so the GPU always has the next bounce queued and never waits for the host. */
        if (hipMemsetAsync(rd.counters, 0, r->counts_words * sizeof(uint32_t), st) != hipSuccess) return CRT_E_HIP;
        hipLaunchKernelGGL(k_pt_generate, dim3((P + 255) / 256), dim3(256), 0, st, rd, P, first_sample);
        int cur = 0, launched = 0;
        /* Two streams (round 3).  Within a bounce: extend -> shade on `st`; shadow -> accumulate on `st2`, after the shade that
         * filled the shadow queue.  The next bounce's extend query needs only the shade's ray queue, so it runs beside this
         * bounce's shadow query (the scene keeps two sets of launch scratch); the next SHADE waits for this bounce's
         * accumulate -- both add to L and good_paths of the same paths, in the reference's order (emitted light of bounce
         * b + 1 after the direct light of bounce b, trace.h:419-480), and by then the shadow queue is free again.
         * n_bound: the host follows the queue sizes two bounces behind; a bounce's queues are never longer than the ray queue
         * of the bounce before it, so that count bounds (and sizes) this bounce's launches -- small late bounces get small
         * grids and the walk instance the library keeps for small batches. */
        hipStream_t st2 = r->stream2;
        for (int bounce = 0; bounce <= rd.max_depth; bounce++) {
            unsigned n_bound = P;
            if (bounce >= 2) {
                if (hipEventSynchronize(r->ev_count[(bounce - 2) & 3]) != hipSuccess) return CRT_E_HIP;
                if (r->h_counts[2 * (bounce - 1)] == 0) break;                /* bounce - 2 queued nothing: the pass is over */
                if (r->h_counts[2 * (bounce - 1)] < n_bound) n_bound = r->h_counts[2 * (bounce - 1)];
            }
            while (r->ev_time.size() < 4 * (size_t)(bounce + 1)) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return CRT_E_HIP;
                r->ev_time.push_back(e);
            }
            while (r->ev_sync.size() < 2 * (size_t)(bounce + 1)) {
                hipEvent_t e;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return CRT_E_HIP;
                r->ev_sync.push_back(e);
            }
            hipEvent_t* ev = &r->ev_time[4 * (size_t)bounce];
            hipEvent_t* sy = &r->ev_sync[2 * (size_t)bounce];
            if (hipEventRecord(ev[0], st) != hipSuccess) return CRT_E_HIP;
            /* not binned (see crt_renderer_create): the queue is in path order */
            int rc = crt_internal_intersect_counted(r->scene, rd.rays[cur], rd.counters + 2 * bounce, n_bound, rd.hits, st, bounce > r->nobin_e_upto);
            if (rc != CRT_OK) return rc;
            if (hipEventRecord(ev[1], st) != hipSuccess) return CRT_E_HIP;
            if (bounce > 0 && hipStreamWaitEvent(st, r->ev_sync[2 * (size_t)(bounce - 1) + 1], 0) != hipSuccess) return CRT_E_HIP;   /* accumulate(bounce - 1) */
            hipLaunchKernelGGL(k_pt_shade, dim3((n_bound + 255) / 256), dim3(256), 0, st, rd, n_bound, bounce, cur, first_sample);
            if (hipMemcpyAsync(r->h_counts + 2 * (bounce + 1), rd.counters + 2 * (bounce + 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess) return CRT_E_HIP;
            if (hipEventRecord(r->ev_count[bounce & 3], st) != hipSuccess) return CRT_E_HIP;
            if (hipEventRecord(sy[0], st) != hipSuccess) return CRT_E_HIP;
            if (hipStreamWaitEvent(st2, sy[0], 0) != hipSuccess) return CRT_E_HIP;
            if (hipEventRecord(ev[2], st2) != hipSuccess) return CRT_E_HIP;
            rc = crt_internal_shadow_counted(r->scene, rd.srays, rd.sdist, rd.counters + 2 * (bounce + 1) + 1, n_bound, 0, rd.socc, st2, bounce > r->nobin_s_upto);
            if (rc != CRT_OK) return rc;
            if (hipEventRecord(ev[3], st2) != hipSuccess) return CRT_E_HIP;
            hipLaunchKernelGGL(k_pt_accumulate, dim3((n_bound + 255) / 256), dim3(256), 0, st2, rd, n_bound, bounce);
            if (hipEventRecord(sy[1], st2) != hipSuccess) return CRT_E_HIP;
            cur ^= 1; launched++;
        }
        if (launched > 0 && hipStreamWaitEvent(st, r->ev_sync[2 * (size_t)(launched - 1) + 1], 0) != hipSuccess) return CRT_E_HIP;
        hipLaunchKernelGGL(k_pt_finish, dim3((rd.npix + 255) / 256), dim3(256), 0, st, rd, spp, first_sample);
        if (hipGetLastError() != hipSuccess) return CRT_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return CRT_E_HIP;
        r->h_counts[0] = P;
        for (int b = 0; b < launched; b++) {
            const uint32_t n_ext = r->h_counts[2 * b], n_sh = r->h_counts[2 * (b + 1) + 1];
            if (n_ext > P || n_sh > P) return CRT_E_HIP;          /* cannot happen: one push of each kind per path */
            r->stats.rays_closest += n_ext;
            r->stats.rays_shadow += n_sh;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r->ev_time[4 * (size_t)b], r->ev_time[4 * (size_t)b + 1]) == hipSuccess) r->stats.ms_extend += ms;
            if (hipEventElapsedTime(&ms, r->ev_time[4 * (size_t)b + 2], r->ev_time[4 * (size_t)b + 3]) == hipSuccess) r->stats.ms_shadow += ms;
        }
        r->samples += spp; done += spp;
    }
    r->stats.samples = r->samples;
    r->stats.ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
}

int crt_renderer_read(crt_renderer* r, float* color_buffer, uint32_t* num_samples)
{
    if (!r) return CRT_E_INVALID;
    DevGuard g(r->device);
    if (color_buffer && hipMemcpy(color_buffer, r->rd.color, 3 * (size_t)r->rd.npix * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return CRT_E_HIP;
    if (num_samples) *num_samples = (uint32_t)r->samples;
    return CRT_OK;
}

int crt_renderer_reset(crt_renderer* r, uint32_t next_sample)
{
    if (!r) return CRT_E_INVALID;
    DevGuard g(r->device);
    if (hipMemsetAsync(r->rd.color, 0, 3 * (size_t)r->rd.npix * sizeof(float), r->stream) != hipSuccess) return CRT_E_HIP;
    if (hipStreamSynchronize(r->stream) != hipSuccess) return CRT_E_HIP;
    r->samples = next_sample;
    r->stats.samples = next_sample;
    return CRT_OK;
}

int crt_renderer_read_dev(crt_renderer* r, float* d_color_buffer, void* stream)
{
    if (!r || !d_color_buffer) return CRT_E_INVALID;
    DevGuard g(r->device);
    if (hipMemcpyAsync(d_color_buffer, r->rd.color, 3 * (size_t)r->rd.npix * sizeof(float), hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess) return CRT_E_HIP;
    return CRT_OK;
}

int crt_renderer_get_stats(crt_renderer* r, crt_render_stats* out)
{
    if (!r || !out) return CRT_E_INVALID;
    *out = r->stats;
    return CRT_OK;
}

} // extern "C"
