/*
 * crt_oracle_trace.cpp -- CPU restatement of the reference's path tracer for the restricted form of
 * SURVEY 8f N1 (checker for crt_renderer_*).  TEST INFRASTRUCTURE ONLY, like the rest of oracle/.
 *
 * Follows, function by function: threadFunc's per-pixel body (main.cpp:102-141), pathTrace
 * (trace.h:399-532), uniformSampleOneLight (trace.h:359-397), estimateDirect (trace.h:221-357),
 * computeLocalBasis (trace.h:132-146), computeScatteringFunc MATTE / MIRROR / TRANSPARENT (materials.cpp:116-141),
 * BSDF_f / BSDF_sample_f / Lambertian_* (reflection.cpp:719-811,:159-189,:6-21), the ShadeRec tails
 * of the intersectors (triangle.cpp:154-168,:185-204, rect.cpp:3-47, sphere.cpp:3-31), the sampling
 * maps (sampling.cpp:606-618), getVec3InLocalBasis (util/math.cpp:31-39), the light set-up
 * (buildscene.h:567-608,:875-922).  Scene queries go through oracle_intersect / oracle_shadow.
 *
 * PARITY UNPINNED for the control flow of trace.h: that header includes intersect.h -> <GLFW/glfw3.h>,
 * absent from this image, so pathTrace / estimateDirect cannot be compiled into oracle/_ref and the
 * reference holds no image or radiance fixture.  Pinned against oracle/_ref (tests/test_trace_pieces.py):
 * the sampling maps, getVec3InLocalBasis / orthoNormalTransform, and the ShadeRec fields of
 * rayIntersectTriangle / FlatTriangle / Rect / Sphere.
 *
 * Deliberate differences from the reference, shared with the product (DESIGN.md "N1"):
 *   - random numbers come from oracle_rng(seed, pixel, sample, dim) instead of the Sampler tables and
 *     libc rand() (trace.h:371,:516; reflection.cpp:254);
 *   - ShadeRec.dpdu of TRIANGLE / FLAT_TRIANGLE hits, which the reference leaves uninitialised
 *     (triangle.cpp:154-204 never write it), is normalize(v1 - v0);
 *   - when no CDF interval matches in uniformSampleOneLight (the reference leaves light_index
 *     unset), the last light is taken;
 *   - without an EnvLight an escaping primary ray adds black (the reference adds an uninitialised vec3).
 */
#include "crt_oracle.h"

#include <math.h>
#include <string.h>
#include <thread>
#include <vector>

namespace {

typedef float vec2[2];
typedef float vec3[3];

const double PI = 3.14159265358979323846264338327950288;      /* util/vec.h:6 */
const double INV_PI = 1.0 / PI;                                /* util/vec.h:10 */
const float K_EPSILON = CRT_K_EPSILON;
const float TMAX = CRT_TMAX;
const vec3 JITTERED_UP = {0.0072f, 1.0f, 0.0034f};             /* util/constants.h:27 */
const vec3 BLACK = {0.0f, 0.0f, 0.0f};

inline void vec3_copy(vec3 r, const vec3 a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
inline void vec3_assign(vec3 r, float x, float y, float z) { r[0] = x; r[1] = y; r[2] = z; }
inline void vec3_add(vec3 r, const vec3 a, const vec3 b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
inline void vec3_sub(vec3 r, const vec3 a, const vec3 b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
inline void vec3_mult(vec3 r, const vec3 a, const vec3 b) { r[0] = a[0] * b[0]; r[1] = a[1] * b[1]; r[2] = a[2] * b[2]; }
inline void vec3_scale(vec3 r, const vec3 a, float s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
inline void vec3_negate(vec3 r, const vec3 a) { r[0] = -a[0]; r[1] = -a[1]; r[2] = -a[2]; }
inline float vec3_dot(const vec3 a, const vec3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void vec3_cross(vec3 r, const vec3 a, const vec3 b)
{
    float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
inline float vec3_length(const vec3 a) { return (float)sqrt(vec3_dot(a, a)); }
inline void vec3_normalize(vec3 r, const vec3 a)
{
    float length = (float)sqrt(vec3_dot(a, a));
    if (length == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    vec3_scale(r, a, 1.0f / length);
}
inline bool vec3_equal(const vec3 a, const vec3 b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }
inline float maxf(float a, float b) { return (a > b) ? a : b; }   /* util/util.h:14 */

void orthoNormalTransform(vec3 r, const vec3 u, const vec3 v, const vec3 w, const vec3 a)
{
    vec3 u_comp, v_comp, w_comp;
    vec3_scale(u_comp, u, a[0]);
    vec3_scale(v_comp, v, a[1]);
    vec3_scale(w_comp, w, a[2]);
    vec3 t;
    vec3_add(t, u_comp, v_comp);
    vec3_add(r, t, w_comp);
}
void transposeTransform(vec3 r, const vec3 u, const vec3 v, const vec3 w, const vec3 a)
{
    vec3_assign(r, vec3_dot(a, u), vec3_dot(a, v), vec3_dot(a, w));
}
void getVec3InLocalBasis(vec3 r, const vec3 a, const vec3 normal)
{
    vec3 u, v, w;
    vec3_copy(w, normal);
    vec3_cross(v, w, JITTERED_UP);
    vec3_normalize(v, v);
    vec3_cross(u, v, w);
    vec3 in;
    vec3_copy(in, a);
    orthoNormalTransform(r, u, v, w, in);
}
void mapSampleToDisk(float* out, const vec2 in)
{
    float phi = 2.0f * M_PI * in[0];
    float radius = sqrtf(in[1]);
    out[0] = cosf(phi) * radius;
    out[1] = sinf(phi) * radius;
}
void mapSampleToHemisphere(vec3 out, const vec2 in)
{
    mapSampleToDisk(out, in);
    out[2] = sqrtf(maxf(0, 1.0f - out[0] * out[0] - out[1] * out[1]));
}

inline uint32_t fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

struct Light {                 /* AreaLight / EnvLight, lights.h */
    int kind;                  /* 0 ENVLIGHT, 1 AREALIGHT */
    int obj_type; const float* shape;
    vec3 color; float intensity;
    float power;
};
struct Setup {
    const oracle_scene* scene;
    const crt_render_desc* d;
    std::vector<Light> lights;
    std::vector<float> tri_normal, tri_dpdu;     /* per triangle */
    uint64_t seed;
};
struct ShadeRec {              /* util/shaderec.h, the fields this form reads */
    vec3 hit_point, normal, wo, dpdu;
    const crt_material* mat;
    /* BSDF (reflection.h) with its single Lambertian lobe */
    vec3 b_normal, b_tangent, b_binormal, cd;
};

struct Rng {
    uint64_t seed; uint32_t pixel, sample;
    float operator()(uint32_t dim) const { return oracle_rng(seed, pixel, sample, dim); }
};

/* interpTriangleVec3, triangle.cpp:206-216; mat3_mult_vec3, util/mat.h:181-187 (row-major 3 x 3) */
void interpTriangleVec3(vec3 out, const float beta, const float gamma, const vec3 v0, const vec3 v1, const vec3 v2)
{
    vec3 tmp;
    vec3_scale(out, v0, 1.0f - beta - gamma);
    vec3_scale(tmp, v1, beta);
    vec3_add(out, out, tmp);
    vec3_scale(tmp, v2, gamma);
    vec3_add(out, out, tmp);
    vec3_normalize(out, out);
}
void mat3_mult_vec3(vec3 r, const float* a, const vec3 b)
{
    vec3 t = { vec3_dot(a, b), vec3_dot(a + 3, b), vec3_dot(a + 6, b) };
    vec3_copy(r, t);
}

/* ShadeRec tail of the intersector that produced `hit` */
void fillShadeRec(const Setup& S, ShadeRec* sr, const crt_ray& ray, const crt_hit& hit)
{
    const crt_scene_input* in = S.d->objects;
    const int type = in->obj_type ? in->obj_type[hit.obj] : CRT_SMOOTH_TRIANGLE;
    const int ref = in->obj_ref ? in->obj_ref[hit.obj] : hit.obj;
    vec3 disp;
    vec3_scale(disp, ray.d, hit.t);
    vec3_add(sr->hit_point, ray.o, disp);                          /* getPointOnRay */
    vec3_negate(sr->wo, ray.d);
    sr->mat = S.d->obj_material ? &S.d->materials[S.d->obj_material[hit.obj]] : 0;
    if (type == CRT_TRIANGLE) {
        vec3_copy(sr->normal, &S.tri_normal[3 * (size_t)ref]);
        if (vec3_dot(sr->wo, sr->normal) < 0.0f) vec3_negate(sr->normal, sr->normal);
        vec3_copy(sr->dpdu, &S.tri_dpdu[3 * (size_t)ref]);
    } else if (type == CRT_FLAT_TRIANGLE) {
        const crt_mesh_shading* ms = S.d->mesh_shading;
        vec3_copy(sr->normal, (ms && ms->flat_normal) ? ms->flat_normal + 3 * (size_t)ref : &S.tri_normal[3 * (size_t)ref]);
        vec3_copy(sr->dpdu, &S.tri_dpdu[3 * (size_t)ref]);
    } else if (type == CRT_SMOOTH_TRIANGLE) {                      /* rayIntersectSmoothTriangle, triangle.cpp:226-251 */
        const crt_mesh_shading* ms = S.d->mesh_shading;
        const float* vn = ms->vertex_normals + 9 * (size_t)ref;
        const float* vt = ms->vertex_tangents + 9 * (size_t)ref;
        const float* nm = ms->normal_mats + 9 * (size_t)ms->normal_mat_index[ref];
        vec3 tmp_normal, tmp, tangent;
        interpTriangleVec3(tmp_normal, hit.beta, hit.gamma, vn, vn + 3, vn + 6);
        mat3_mult_vec3(sr->normal, nm, tmp_normal);
        vec3_normalize(sr->normal, sr->normal);
        interpTriangleVec3(tmp, hit.beta, hit.gamma, vt, vt + 3, vt + 6);
        mat3_mult_vec3(tangent, nm, tmp);
        vec3_normalize(sr->dpdu, tangent);
    } else if (type == CRT_RECTANGLE) {
        const float* p = in->shapes[ref].p;                        /* point, width, height, normal */
        vec3 displacement, normal;
        vec3_sub(displacement, p, ray.o);
        if (vec3_dot(displacement, p + 9) < 0.0f) vec3_negate(normal, p + 9);
        else vec3_copy(normal, p + 9);
        vec3_normalize(sr->dpdu, p + 3);
        vec3_copy(sr->normal, normal);
        if (vec3_dot(sr->wo, sr->normal) < 0.0f) {
            vec3_negate(sr->normal, sr->normal);
            vec3_negate(sr->dpdu, sr->dpdu);
        }
    } else {                                                       /* SPHERE: fillShadeRecSphere */
        const float* p = in->shapes[ref].p;                        /* center, radius */
        vec3 hit_point_to_center;
        vec3_sub(hit_point_to_center, sr->hit_point, p);
        vec3_normalize(sr->normal, hit_point_to_center);
        const double phi_max = 2.0 * PI;
        vec3_assign(sr->dpdu, -phi_max * (sr->hit_point[2] - p[2]), 0.0f, phi_max * (sr->hit_point[0] - p[0]));
        vec3_normalize(sr->dpdu, sr->dpdu);
    }
}

void computeLocalBasis(ShadeRec* sr)
{
    vec3_copy(sr->b_normal, sr->normal);
    vec3_copy(sr->b_tangent, sr->dpdu);
    float dot_product = vec3_dot(sr->b_normal, sr->b_tangent);
    if (dot_product != 0.0f) {
        vec3 adjustment;
        vec3_scale(adjustment, sr->b_normal, dot_product);
        vec3_sub(sr->b_tangent, sr->b_tangent, adjustment);
        vec3_normalize(sr->b_tangent, sr->b_tangent);
    }
    vec3_cross(sr->b_binormal, sr->b_normal, sr->b_tangent);
    vec3_normalize(sr->b_binormal, sr->b_binormal);
}

float cosHemispherePdf(const vec3 wi, const vec3 wo)
{
    return (wi[2] * wo[2] > 0.0f) ? fabsf(wi[2]) * INV_PI : 0.0f;
}

enum { BSDF_NONE = 0, BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16 };   /* reflection.h:8-17 */

/* calcFresnelDielectric, reflection.cpp:52-76 */
float calcFresnelDielectric(const vec3 normal, const vec3 wo, float etaT, float etaI)
{
    vec3 n;
    vec3_copy(n, normal);
    float cos_theta_i = vec3_dot(n, wo);
    if (cos_theta_i < 0.0f) {
        float tmp = etaI; etaI = etaT; etaT = tmp;
        vec3_negate(n, n);
        cos_theta_i = fabsf(cos_theta_i);
    }
    float sin_theta_i = sqrtf(maxf(0.0f, 1.0f - cos_theta_i * cos_theta_i));
    float sin_theta_t = etaI / etaT * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = sqrtf(maxf(0.0f, 1 - sin_theta_t * sin_theta_t));
    float r_parl = ((etaT * cos_theta_i) - (etaI * cos_theta_t)) / ((etaT * cos_theta_i) + (etaI * cos_theta_t));
    float r_perp = ((etaI * cos_theta_i) - (etaT * cos_theta_t)) / ((etaI * cos_theta_i) + (etaT * cos_theta_t));
    return (r_parl * r_parl + r_perp * r_perp) / 2;
}

/* util/math.h:13-36 */
inline float sinTheta(const vec3 w) { return sqrtf(maxf((float)0, (float)1 - w[2] * w[2])); }
inline float clampf(float x, float mn, float mx) { return (x < mn ? mn : (x > mx ? mx : x)); }
inline float cosPhi(const vec3 w) { float sin_theta = sinTheta(w); return (sin_theta == 0) ? 1 : clampf(w[0] / sin_theta, -1.0f, 1.0f); }
inline float sinPhi(const vec3 w) { float sin_theta = sinTheta(w); return (sin_theta == 0) ? 0 : clampf(w[1] / sin_theta, -1.0f, 1.0f); }

/* OrenNayar_f, reflection.cpp:191-214 */
void OrenNayar_f(vec3 f, const vec3 wi, const vec3 wo, const vec3 r, float a, float b)
{
    float sin_theta_i = sinTheta(wi);
    float sin_theta_o = sinTheta(wo);
    float max_cos = 0.0f;
    if (sin_theta_i > 1e-4 && sin_theta_o > 1e-4) {
        float sin_phi_i = sinPhi(wi), cos_phi_i = cosPhi(wi);
        float sin_phi_o = sinPhi(wo), cos_phi_o = cosPhi(wo);
        float d_cos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
        max_cos = maxf((float)0, d_cos);
    }
    float sin_alpha, tan_beta;
    if (fabsf(wi[2]) > fabsf(wo[2])) {
        sin_alpha = sin_theta_o;
        tan_beta = sin_theta_i / fabsf(wi[2]);
    } else {
        sin_alpha = sin_theta_i;
        tan_beta = sin_theta_o / fabsf(wo[2]);
    }
    vec3_scale(f, r, (a + b * max_cos * sin_alpha * tan_beta) * INV_PI);
}
/* BSDF_addOrenNayar, reflection.cpp:823-830 */
void orenNayarAB(float sigma, float* a, float* b)
{
    float sigma_rad = sigma / 180.0f * (float)PI;                  /* degToRad, util/math.h:65-68 */
    float sigma2 = sigma_rad * sigma_rad;
    *a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
    *b = 0.45f * sigma2 / (sigma2 + 0.09f);
}

enum { LOBE_LAMBERTIAN = 0, LOBE_SPECULAR_REFLECTION = 1, LOBE_SPECULAR_TRANSMISSION = 2, LOBE_ORENNAYAR = 3 };
struct Lobe { int kind; float ior_in, ior_out; float rand_float; float a, b; };   /* rand_float: the value SpecularTransmission_sample_f draws from rand() */

/* BSDF_sample_f with num_bxdf == 1 */
float BSDF_sample_f(vec3 f, vec3 wi, int* sampled_flags, const vec3 wo, const vec2 sample, const ShadeRec* sr, const Lobe& lobe)
{
    const bool mirror = lobe.kind == LOBE_SPECULAR_REFLECTION;
    *sampled_flags = mirror ? (BSDF_REFLECTION | BSDF_SPECULAR)                                        /* getBxDFFlagsFromType */
                   : lobe.kind == LOBE_SPECULAR_TRANSMISSION ? (BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR)   /* the lobe also reflects */
                   : (BSDF_REFLECTION | BSDF_DIFFUSE);
    vec3 wi_local, wo_local;
    transposeTransform(wo_local, sr->b_tangent, sr->b_binormal, sr->b_normal, wo);
    if (wo_local[2] == 0.0f) { vec3_copy(f, BLACK); return 0.0f; }   /* the reference returns with f unset; the caller only tests pdf */
    vec3_copy(f, BLACK);
    float pdf;
    if (mirror) {                                                  /* SpecularReflection_sample_f, reflection.cpp:234-242 */
        vec3_assign(wi_local, -wo_local[0], -wo_local[1], wo_local[2]);
        vec3_scale(f, sr->cd, 1.0f / fabsf(wi_local[2]));
        pdf = 1.0f;
    } else if (lobe.kind == LOBE_SPECULAR_TRANSMISSION) {          /* SpecularTransmission_sample_f, `thin` (reflection.cpp:243-269) */
        const vec3 WHITE = {1.0f, 1.0f, 1.0f};
        vec3 normal = {0.0f, 0.0f, 1.0f};
        float cos_theta_o = vec3_dot(normal, wo_local);
        if (cos_theta_o < 0.0f) vec3_negate(normal, normal);
        float kr = calcFresnelDielectric(normal, wo_local, lobe.ior_in, lobe.ior_out);
        if (lobe.rand_float <= kr) {
            vec3_assign(wi_local, -wo_local[0], -wo_local[1], wo_local[2]);
            vec3_scale(f, WHITE, kr / fabsf(wi_local[2]));
            pdf = kr;
        } else {
            float eta = lobe.ior_out / lobe.ior_in;
            vec3_negate(wi_local, wo_local);
            vec3_scale(f, WHITE, 1.0f - kr);
            vec3_scale(f, f, (eta * eta) / fabsf(wi_local[2]));
            pdf = 1.0f - kr;
        }
    } else if (lobe.kind == LOBE_ORENNAYAR) {                      /* OrenNayar_sample_f, reflection.cpp:221-232 */
        vec3 tmp_wo;
        vec3_copy(tmp_wo, wo_local);
        if (tmp_wo[2] < 0.0f) tmp_wo[2] *= -1.0f;
        mapSampleToHemisphere(wi_local, sample);
        pdf = cosHemispherePdf(wi_local, tmp_wo);
        OrenNayar_f(f, wi_local, wo_local, sr->cd, lobe.a, lobe.b);
    } else {                                                       /* Lambertian_sample_f */
        vec3 tmp_wo;
        vec3_copy(tmp_wo, wo_local);
        if (tmp_wo[2] < 0.0f) tmp_wo[2] *= -1.0f;
        vec3_copy(f, sr->cd);
        vec3_scale(f, f, INV_PI);
        mapSampleToHemisphere(wi_local, sample);
        pdf = cosHemispherePdf(wi_local, tmp_wo);
    }
    if (pdf == 0.0f) { vec3_copy(f, BLACK); return pdf; }
    orthoNormalTransform(wi, sr->b_tangent, sr->b_binormal, sr->b_normal, wi_local);
    return pdf;
}

bool estimateDirect(const Setup& S, vec3 L, const vec2 light_sample, const ShadeRec* sr, const Light& lt)
{
    vec3_copy(L, BLACK);
    vec3 Li, wi;
    float light_pdf;
    vec3 sample_point, sample_normal;
    if (lt.kind == 1) {
        if (lt.obj_type == CRT_SPHERE) {
            const float* center = lt.shape; const float radius = lt.shape[3];
            vec3 hit_point_to_center, z_axis, h_sample;
            vec3_sub(hit_point_to_center, sr->hit_point, center);
            vec3_normalize(z_axis, hit_point_to_center);
            mapSampleToHemisphere(h_sample, light_sample);
            getVec3InLocalBasis(h_sample, h_sample, z_axis);
            vec3_scale(sample_point, h_sample, radius);
            vec3_add(sample_point, sample_point, center);
            vec3_copy(sample_normal, h_sample);
            light_pdf = 1.0f / (2.0f * PI * radius * radius) * fabsf(vec3_dot(h_sample, z_axis)) * INV_PI;
        } else if (lt.obj_type == CRT_RECTANGLE) {
            const float* point = lt.shape; const float* width = lt.shape + 3; const float* height = lt.shape + 6;
            vec3 u, v;
            vec3_scale(u, width, light_sample[0]);
            vec3_scale(v, height, light_sample[1]);
            vec3_add(sample_point, point, u);
            vec3_add(sample_point, sample_point, v);
            vec3_copy(sample_normal, lt.shape + 9);
            light_pdf = 1.0f / (vec3_length(width) * vec3_length(height));
        } else {
            const float* center = lt.shape; const float* normal = lt.shape + 3; const float radius = lt.shape[6];
            vec2 disk_sample;
            mapSampleToDisk(disk_sample, light_sample);
            vec3 x_axis, y_axis, x_disp, y_disp;
            vec3_cross(x_axis, JITTERED_UP, normal);
            vec3_normalize(x_axis, x_axis);
            vec3_cross(y_axis, x_axis, normal);
            vec3_scale(x_disp, x_axis, disk_sample[0] * radius);
            vec3_scale(y_disp, y_axis, disk_sample[1] * radius);
            vec3_add(sample_point, x_disp, y_disp);
            vec3_add(sample_point, sample_point, center);
            vec3_copy(sample_normal, normal);
            light_pdf = 1.0f / (PI * radius * radius);
        }
        vec3 sample_to_hit_point, neg_wi;
        vec3_sub(sample_to_hit_point, sample_point, sr->hit_point);
        vec3_normalize(wi, sample_to_hit_point);
        vec3_negate(neg_wi, wi);
        light_pdf *= vec3_dot(sample_to_hit_point, sample_to_hit_point) / fabsf(vec3_dot(sample_normal, neg_wi));
        vec3_scale(Li, lt.color, lt.intensity);
    } else {
        vec3 h_sample;
        mapSampleToHemisphere(h_sample, light_sample);
        vec3 h_in;
        vec3_copy(h_in, h_sample);
        orthoNormalTransform(h_sample, sr->b_tangent, sr->b_binormal, sr->b_normal, h_in);   /* transformToLocalBasis; env transform = identity */
        vec3 displacement;
        vec3_scale(displacement, h_sample, S.d->env_world_radius);
        vec3_add(sample_point, sr->hit_point, displacement);
        vec3_negate(sample_normal, h_sample);
        vec3_copy(wi, h_sample);
        light_pdf = fabsf(vec3_dot(h_sample, sr->normal)) * INV_PI;
        vec3_scale(Li, lt.color, lt.intensity);
    }
    vec3 sample_to_hit_point;
    vec3_sub(sample_to_hit_point, sample_point, sr->hit_point);
    if (vec3_dot(sample_to_hit_point, sample_normal) > 0.0f || vec3_dot(sample_to_hit_point, sr->normal) < 0.0f) return false;
    vec3 f;
    if (light_pdf > 0.0f && !vec3_equal(Li, BLACK)) {
        /* BSDF_f: BLACK + Lambertian_f / OrenNayar_f.  BSDF_f hands the lobe the WORLD-space wi, wo
         * (reflection.cpp:731), which only Oren-Nayar notices */
        vec3 cur_f = {0.0f, 0.0f, 0.0f};
        if (sr->mat->type == CRT_MAT_MATTE && sr->mat->sigma != 0.0f) {
            float a, b;
            orenNayarAB(sr->mat->sigma, &a, &b);
            OrenNayar_f(cur_f, wi, sr->wo, sr->cd, a, b);
        } else {
            vec3_copy(cur_f, sr->cd);
            vec3_scale(cur_f, cur_f, INV_PI);
        }
        vec3_copy(f, BLACK);
        vec3_add(f, f, cur_f);
        vec3_scale(f, f, fabsf(vec3_dot(sr->normal, wi)));
    } else
        return false;
    if (!vec3_equal(f, BLACK)) {
        float distance = vec3_length(sample_to_hit_point);
        crt_ray shadow_ray;
        vec3_copy(shadow_ray.o, sr->hit_point);
        vec3_copy(shadow_ray.d, wi);
        float t = TMAX; uint8_t occ = 0;
        oracle_shadow(S.scene, &shadow_ray, &distance, 1, &t, &occ, 0, 1);
        bool in_shadow = t < distance - K_EPSILON;
        if (!in_shadow) {
            vec3 radiance;
            vec3_mult(radiance, f, Li);
            vec3_scale(radiance, radiance, 1.0f / light_pdf);
            vec3_add(L, L, radiance);
        }
    }
    return true;
}

void uniformSampleOneLight(const Setup& S, vec3 L, const vec2 light_sample, const ShadeRec* sr, float rand_float)
{
    vec3_copy(L, BLACK);
    int num_lights = (int)S.lights.size();
    if (num_lights == 0) return;
    int light_index = num_lights - 1;
    float light_pdf = S.lights[light_index].power;
    float cdf = 0.0f;
    for (int i = 0; i < num_lights; i++) {
        if (cdf <= rand_float && cdf + S.lights[i].power > rand_float) {
            light_index = i;
            light_pdf = S.lights[i].power;
            break;
        }
        cdf += S.lights[i].power;
    }
    vec3 Ld;
    estimateDirect(S, Ld, light_sample, sr, S.lights[light_index]);
    vec3_scale(L, Ld, 1.0f / light_pdf);
}

void pathTrace(const Setup& S, vec3 radiance, int depth, const crt_ray primary_ray, const Rng& rng)
{
    vec3 L = {0.0f, 0.0f, 0.0f};
    vec3 beta = {1.0f, 1.0f, 1.0f};
    crt_ray ray = primary_ray;
    int good_paths = 0;
    int sampled_flags = BSDF_NONE;
    const int excluded_from_direct = BSDF_SPECULAR | BSDF_GLOSSY;
    for (int bounces = 0; ; bounces++) {
        ShadeRec sr;
        crt_hit hit;
        oracle_intersect(S.scene, &ray, 1, &hit, CRT_TRAVERSE_EXACT, 0, 1);
        float t = hit.t;
        if (t < TMAX) fillShadeRec(S, &sr, ray, hit);
        if (bounces == 0 || (sampled_flags & excluded_from_direct)) {
            if (t < TMAX) {
                if (sr.mat->type == CRT_MAT_EMISSIVE) {
                    vec3 inc_radiance;
                    vec3_scale(inc_radiance, sr.mat->color, sr.mat->intensity);
                    vec3_mult(inc_radiance, inc_radiance, beta);
                    vec3_add(L, L, inc_radiance);
                    good_paths++;
                }
            } else {
                vec3 env_inc_radiance = {0.0f, 0.0f, 0.0f};
                if (S.d->has_env_light) vec3_scale(env_inc_radiance, S.d->env_color, S.d->env_intensity);
                vec3_mult(env_inc_radiance, env_inc_radiance, beta);
                vec3_add(L, L, env_inc_radiance);
                good_paths++;
            }
        }
        if (t == TMAX || bounces >= depth || sr.mat->type == CRT_MAT_EMISSIVE) break;

        computeLocalBasis(&sr);
        vec3_copy(sr.cd, sr.mat->color);                           /* computeScatteringFunc: BSDF_addLambertian(color) */

        const uint32_t dim0 = 4 + 8 * (uint32_t)bounces;
        vec2 light_sample = { rng(dim0 + 0), rng(dim0 + 1) };
        /* scatter_sample: dim0 + 2, dim0 + 3, drawn by the reference and unused */
        if (!(sr.mat->type == CRT_MAT_MIRROR || sr.mat->type == CRT_MAT_TRANSPARENT)) {
            vec3 contrib;
            uniformSampleOneLight(S, contrib, light_sample, &sr, rng(dim0 + 6));
            vec3_mult(contrib, contrib, beta);
            vec3_add(L, L, contrib);
            if (!vec3_equal(contrib, BLACK)) good_paths++;
        }
        vec3 wo, wi, f;
        vec3_negate(wo, ray.d);
        vec2 sample = { rng(dim0 + 4), rng(dim0 + 5) };
        Lobe lobe = { LOBE_LAMBERTIAN, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f };
        if (sr.mat->type == CRT_MAT_MATTE && sr.mat->sigma != 0.0f) { lobe.kind = LOBE_ORENNAYAR; orenNayarAB(sr.mat->sigma, &lobe.a, &lobe.b); }
        if (sr.mat->type == CRT_MAT_MIRROR) lobe.kind = LOBE_SPECULAR_REFLECTION;
        if (sr.mat->type == CRT_MAT_TRANSPARENT) {
            lobe.kind = LOBE_SPECULAR_TRANSMISSION; lobe.ior_in = sr.mat->ior_in; lobe.ior_out = sr.mat->ior_out;
            lobe.rand_float = rng(dim0 + 2);                       /* in place of rand() inside the lobe */
        }
        float pdf = BSDF_sample_f(f, wi, &sampled_flags, wo, sample, &sr, lobe);
        if (vec3_equal(f, BLACK) || pdf == 0.0f) break;
        vec3_scale(f, f, fabsf(vec3_dot(wi, sr.normal)) / pdf);
        vec3_mult(beta, beta, f);
        vec3_copy(ray.o, sr.hit_point);
        vec3_copy(ray.d, wi);
        float max_comp = maxf(beta[0], maxf(beta[1], beta[2]));
        if (bounces > 3) {
            float q = maxf(0.05, 1.0f - max_comp);
            float rand_float = rng(dim0 + 7);
            if (rand_float < q) break;
            vec3_scale(beta, beta, 1.0f / (1.0f - q));
        }
    }
    if (good_paths > 0) vec3_scale(L, L, 1.0f / good_paths);
    vec3_copy(radiance, L);
}

void triangleSetup(Setup* S, const crt_scene_input* in)
{
    S->tri_normal.resize(3 * (size_t)in->num_tris);
    S->tri_dpdu.resize(3 * (size_t)in->num_tris);
    for (int k = 0; k < in->num_tris; k++) {                       /* calcTriangleNormal, triangle.cpp:4-12 */
        const float* v = in->tri_verts + 9 * (size_t)k;
        vec3 tmp1, tmp2, tmp3;
        vec3_sub(tmp1, v + 3, v);
        vec3_sub(tmp2, v + 6, v);
        vec3_cross(tmp3, tmp1, tmp2);
        vec3_normalize(&S->tri_normal[3 * (size_t)k], tmp3);
        vec3_normalize(&S->tri_dpdu[3 * (size_t)k], tmp1);
    }
}

bool vec3_hasNanOrInf(const vec3 a)
{
    for (int k = 0; k < 3; k++) if (isnan(a[k]) || isinf(a[k])) return true;
    return false;
}

} // namespace

extern "C" {

float oracle_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim)
{
    uint32_t h = fmix32(pixel ^ (uint32_t)seed);
    h = fmix32(h + sample * 0x9E3779B1u + (uint32_t)(seed >> 32));
    h = fmix32(h ^ (dim * 0x85EBCA77u + 0x165667B1u));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

int oracle_render(const oracle_scene* scene, const crt_render_desc* d, const crt_camera* cam, const crt_film* film,
                  uint32_t first_sample, uint32_t num_samples, float* color_buffer, int nthreads)
{
    Setup S;
    S.scene = scene; S.d = d; S.seed = d->seed;
    const crt_scene_input* in = d->objects;
    triangleSetup(&S, in);
    if (d->has_env_light) {
        Light l; l.kind = 0; l.obj_type = -1; l.shape = 0;
        vec3_copy(l.color, d->env_color); l.intensity = d->env_intensity;
        float power = (l.color[0] + l.color[1] + l.color[2]) / 3.0f;
        power *= d->env_intensity * d->env_world_radius;
        l.power = power;
        S.lights.push_back(l);
    }
    for (int i = 0; i < in->num_obj; i++) {                        /* initAreaLights */
        const int type = in->obj_type ? in->obj_type[i] : CRT_SMOOTH_TRIANGLE;
        const crt_material& mat = d->materials[d->obj_material[i]];
        if (type == CRT_SMOOTH_TRIANGLE && !(d->mesh_shading && d->mesh_shading->vertex_normals && d->mesh_shading->vertex_tangents &&
                                             d->mesh_shading->normal_mat_index && d->mesh_shading->normal_mats)) return -1;
        if (type != CRT_TRIANGLE && type != CRT_FLAT_TRIANGLE && type != CRT_SMOOTH_TRIANGLE && type != CRT_RECTANGLE && type != CRT_SPHERE &&
            type != CRT_DISK) return -1;
        if (type == CRT_DISK && mat.type != CRT_MAT_EMISSIVE) return -1;
        if (mat.type == CRT_MAT_EMISSIVE && (type == CRT_FLAT_TRIANGLE || type == CRT_SMOOTH_TRIANGLE)) return -1;   /* MeshLight */
        if ((type == CRT_RECTANGLE || type == CRT_SPHERE || type == CRT_DISK) && mat.type == CRT_MAT_EMISSIVE) {
            Light l; l.kind = 1; l.obj_type = type; l.shape = in->shapes[in->obj_ref[i]].p;
            vec3_copy(l.color, mat.color); l.intensity = mat.intensity;
            float area = 0.0f;
            if (type == CRT_SPHERE) area = 4.0f * PI * l.shape[3] * l.shape[3];
            else if (type == CRT_RECTANGLE) area = vec3_length(l.shape + 3) * vec3_length(l.shape + 6);
            else area = PI * l.shape[6] * l.shape[6];
            l.power = (l.color[0] * l.color[1] * l.color[2]) / 3.0f * l.intensity * area;
            S.lights.push_back(l);
        }
    }
    float total_power = 0.0f;
    for (size_t i = 0; i < S.lights.size(); i++) total_power += S.lights[i].power;
    for (size_t i = 0; i < S.lights.size(); i++) S.lights[i].power /= total_power;

    const uint32_t npix = (uint32_t)film->frame_res_width * (uint32_t)film->frame_res_height;
    auto work = [&](uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) {
            for (uint32_t p = first_sample; p < first_sample + num_samples; p++) {
                Rng rng = { S.seed, i, p };
                float isample[2] = { rng(0), rng(1) }, lsample[2] = { rng(2), rng(3) };
                crt_ray ray;
                oracle_primary_rays(cam, film, i, 1, isample, lsample, &ray);
                vec3 color = {0.0f, 0.0f, 0.0f}, radiance = {0.0f, 0.0f, 0.0f};
                pathTrace(S, radiance, d->max_depth, ray, rng);
                vec3_add(color, color, radiance);
                if (vec3_hasNanOrInf(color)) {
                    vec3_assign(color, color_buffer[i * 3], color_buffer[i * 3 + 1], color_buffer[i * 3 + 2]);
                    vec3_scale(color, color, 1 / (float)(p + 1));
                }
                color_buffer[i * 3] += color[0];
                color_buffer[i * 3 + 1] += color[1];
                color_buffer[i * 3 + 2] += color[2];
            }
        }
    };
    if (nthreads <= 1) { work(0, npix); return 0; }
    std::vector<std::thread> th;
    /* interleaved strips so that threads get similar work */
    const uint32_t strip = 256;
    for (int k = 0; k < nthreads; k++)
        th.emplace_back([&, k]() { for (uint32_t b = (uint32_t)k * strip; b < npix; b += strip * (uint32_t)nthreads) work(b, b + strip < npix ? b + strip : npix); });
    for (auto& t : th) t.join();
    return 0;
}

/* ---- the pieces tests/test_trace_pieces.py compares with the reference's own code (oracle/_ref) ---- */
void oracle_trace_shade_rec_mesh(const crt_scene_input* in, const crt_mesh_shading* ms, int obj, const crt_ray* ray, float t,
                                 float beta, float gamma, float* out)
{
    crt_render_desc d; memset(&d, 0, sizeof d);
    d.objects = in; d.mesh_shading = ms;
    Setup S; S.scene = 0; S.d = &d; S.seed = 0;
    triangleSetup(&S, in);
    crt_hit hit; hit.t = t; hit.obj = obj; hit.part = 0; hit.beta = beta; hit.gamma = gamma;
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    fillShadeRec(S, &sr, *ray, hit);
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.wo, 12); memcpy(out + 9, sr.dpdu, 12);
}
void oracle_trace_shade_rec(const crt_scene_input* in, int obj, const crt_ray* ray, float t, float* out)
{
    crt_render_desc d; memset(&d, 0, sizeof d);
    d.objects = in;
    Setup S; S.scene = 0; S.d = &d; S.seed = 0;
    triangleSetup(&S, in);
    crt_hit hit; hit.t = t; hit.obj = obj; hit.part = 0; hit.beta = hit.gamma = 0.0f;
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    fillShadeRec(S, &sr, *ray, hit);
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.wo, 12); memcpy(out + 9, sr.dpdu, 12);
}
void oracle_trace_disk(const float* in, float* out) { mapSampleToDisk(out, in); }
void oracle_trace_hemisphere(const float* in, float* out) { mapSampleToHemisphere(out, in); }
void oracle_trace_local_basis(const float* a, const float* normal, float* out) { getVec3InLocalBasis(out, a, normal); }
float oracle_trace_lambertian(const float* cd, const float* basis, const float* wo, const float* sample,
                              float* f_out, float* wi_out, float* f_eval)
{
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    vec3_copy(sr.b_tangent, basis); vec3_copy(sr.b_binormal, basis + 3); vec3_copy(sr.b_normal, basis + 6);
    vec3_copy(sr.cd, cd);
    vec3 wi = {0.0f, 0.0f, 0.0f};
    int flags = 0;
    const Lobe lobe = { LOBE_LAMBERTIAN, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f };
    float pdf = BSDF_sample_f(f_out, wi, &flags, wo, sample, &sr, lobe);
    vec3_copy(wi_out, wi);
    vec3 cur_f;                                                    /* BSDF_f as estimateDirect uses it */
    vec3_copy(cur_f, sr.cd);
    vec3_scale(cur_f, cur_f, INV_PI);
    vec3_copy(f_eval, BLACK);
    vec3_add(f_eval, f_eval, cur_f);
    return pdf;
}

float oracle_trace_mirror(const float* cr, const float* basis, const float* wo, float* f_out, float* wi_out, int* flags)
{
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    vec3_copy(sr.b_tangent, basis); vec3_copy(sr.b_binormal, basis + 3); vec3_copy(sr.b_normal, basis + 6);
    vec3_copy(sr.cd, cr);
    vec3 wi = {0.0f, 0.0f, 0.0f};
    const vec2 sample = {0.5f, 0.5f};
    const Lobe lobe = { LOBE_SPECULAR_REFLECTION, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f };
    float pdf = BSDF_sample_f(f_out, wi, flags, wo, sample, &sr, lobe);
    vec3_copy(wi_out, wi);
    return pdf;
}
/* Oren-Nayar: BSDF_sample_f (tangent space) and BSDF_f evaluated for (wi_eval, wo) as given (the reference gives it world vectors) */
float oracle_trace_orennayar(const float* r, float sigma, const float* basis, const float* wo, const float* sample,
                             const float* wi_eval, float* f_out, float* wi_out, float* f_eval)
{
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    vec3_copy(sr.b_tangent, basis); vec3_copy(sr.b_binormal, basis + 3); vec3_copy(sr.b_normal, basis + 6);
    vec3_copy(sr.cd, r);
    Lobe lobe = { LOBE_ORENNAYAR, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f };
    orenNayarAB(sigma, &lobe.a, &lobe.b);
    vec3 wi = {0.0f, 0.0f, 0.0f};
    int flags = 0;
    float pdf = BSDF_sample_f(f_out, wi, &flags, wo, sample, &sr, lobe);
    vec3_copy(wi_out, wi);
    vec3 cur_f;
    OrenNayar_f(cur_f, wi_eval, wo, r, lobe.a, lobe.b);
    vec3_copy(f_eval, BLACK);
    vec3_add(f_eval, f_eval, cur_f);
    return pdf;
}
float oracle_trace_transparent(float ior_in, float ior_out, float rand_float, const float* basis, const float* wo,
                               float* f_out, float* wi_out, int* flags)
{
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    vec3_copy(sr.b_tangent, basis); vec3_copy(sr.b_binormal, basis + 3); vec3_copy(sr.b_normal, basis + 6);
    vec3 wi = {0.0f, 0.0f, 0.0f};
    const vec2 sample = {0.5f, 0.5f};
    const Lobe lobe = { LOBE_SPECULAR_TRANSMISSION, ior_in, ior_out, rand_float, 0.0f, 0.0f };
    float pdf = BSDF_sample_f(f_out, wi, flags, wo, sample, &sr, lobe);
    vec3_copy(wi_out, wi);
    return pdf;
}

} // extern "C"
