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
enum { BSDF_NONE = 0, BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16 };   /* reflection.h:8-17 */
enum { LAMBERTIAN, ORENNAYAR, SPECULAR_REFLECTION, SPECULAR_TRANSMISSION, MICROFACET_REFLECTION, MICROFACET_FRESNEL,
       FRESNEL_BLEND, FRESNEL_BLEND_DIFFUSE, FRESNEL_BLEND_SPECULAR };                                                       /* BxDFType, reflection.h:19-30 */
struct MicrofacetDistribution { float alphax, alphay; };       /* microfacet.h:10-16; every material uses BECKMANN (materials.cpp:161-185) */
struct BxDF {                  /* the union of reflection.h's lobe structs */
    int type;
    vec3 color;                /* Lambertian.cd, OrenNayar.r, SpecularReflection.cr, Microfacet*.color, FresnelBlendDiffuse.kd */
    vec3 ks;                   /* FresnelBlendDiffuse / FresnelBlendSpecular */
    float a, b;                /* OrenNayar */
    float ior_in, ior_out;
    bool is_metal; vec3 k, etaI, etaT;
    MicrofacetDistribution distrib;
    float rand_float;          /* the value the lobe would draw from rand() (reflection.cpp:254,:419) */
};
struct ShadeRec {              /* util/shaderec.h, the fields this form reads */
    vec3 hit_point, normal, wo, dpdu;
    vec2 uv; bool has_uv;      /* written for FLAT / SMOOTH mesh triangles with texcoords (triangle.cpp:191-198,:237-243) */
    int mat_index;
    const crt_material* mat;
    /* BSDF (reflection.h:157-165) */
    vec3 b_normal, b_tangent, b_binormal;
    int num_bxdf; BxDF bxdfs[2];
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

/* ---- analytic shapes: the reference's intersectors run again on the object that was hit, this time with their record
 *      writes (rayIntersectSphere + fillShadeRecSphere sphere.cpp:3-86, rayIntersectPlane plane.cpp:4-19, rayIntersectRect rect.cpp:3-52,
 *      rayIntersectDisk disk.cpp:3-29, rayIntersectAABox generic.cpp:311-404, rayIntersectGenericOpenCylinder + fill :3-80,
 *      rayIntersectGenericTorus :156-221 (its root comes from the hit: the solver lives in crt_oracle.cpp), rayIntersectInstanced
 *      instanced.cpp:48-104, rayIntersectCompound :155-216).  Where the reference leaves a field unset the conventions of DESIGN.md
 *      "a-13" apply: a disk hit from behind keeps the disk's normal, a torus hit outside its phi window is filled as if inside. ---- */
void getPointOnRay(vec3 r, const vec3 o, const vec3 d, float t) { vec3 disp; vec3_scale(disp, d, t); vec3_add(r, o, disp); }

float simpleShapeRecord(const crt_shape& sh, const vec3 o, const vec3 d, float hit_t, ShadeRec* sr, bool* has_normal, bool* has_dpdu)
{
    const float* p = sh.p;
    switch (sh.type) {
    case CRT_SPHERE: {
        const float radius = p[3];
        float a = vec3_dot(d, d);
        vec3 origin_to_center, tmp;
        vec3_sub(origin_to_center, o, p);
        vec3_scale(tmp, origin_to_center, 2);
        float b = vec3_dot(tmp, d);
        float c = vec3_dot(origin_to_center, origin_to_center) - radius * radius;
        float disc = b * b - 4 * a * c;
        if (disc < 0.0f) return TMAX;
        float e = (float)sqrt(disc);
        float denom = 2 * a;
        for (int root = 0; root < 2; root++) {
            float t = root == 0 ? (-b - e) / denom : (-b + e) / denom;
            if (!(t > K_EPSILON)) continue;
            vec3 hit_point;
            getPointOnRay(hit_point, o, d, t);
            float phi = root == 0 ? atan2f(hit_point[2] - p[2], hit_point[0] - p[0]) : atan2f(hit_point[0] - p[0], hit_point[2] - p[2]);
            float theta = acosf((hit_point[1] - p[1]) / radius);
            if (!(fabsf(phi) <= p[6] && theta >= p[4] && theta <= p[5])) continue;
            vec3_copy(sr->hit_point, hit_point);                   /* fillShadeRecSphere */
            vec3 hit_point_to_center;
            vec3_sub(hit_point_to_center, sr->hit_point, p);
            vec3_normalize(sr->normal, hit_point_to_center);
            vec3_negate(sr->wo, d);
            const double phi_max = 2.0 * PI;
            {                                                      /* sphere.cpp:14-19 */
                const double theta_max = PI, theta_min = 0.0;
                sr->uv[0] = (float)(phi / phi_max);
                sr->uv[1] = (float)((theta - theta_min) / (theta_max - theta_min));
                sr->has_uv = true;
            }
            vec3_assign(sr->dpdu, -phi_max * (hit_point[2] - p[2]), 0.0f, phi_max * (hit_point[0] - p[0]));
            vec3_normalize(sr->dpdu, sr->dpdu);
            *has_normal = *has_dpdu = true;
            return t;
        }
        return TMAX;
    }
    case CRT_PLANE: {
        vec3 displacement;
        vec3_sub(displacement, p, o);
        float t = vec3_dot(displacement, p + 3) / vec3_dot(d, p + 3);
        if (t > K_EPSILON) {
            vec3_copy(sr->normal, p + 3);
            getPointOnRay(sr->hit_point, o, d, t);
            vec3_negate(sr->wo, d);
            *has_normal = true;
            return t;
        }
        return TMAX;
    }
    case CRT_RECTANGLE: {
        vec3 displacement, normal;
        vec3_sub(displacement, p, o);
        if (vec3_dot(displacement, p + 9) < 0.0f) vec3_negate(normal, p + 9); else vec3_copy(normal, p + 9);
        float t = vec3_dot(displacement, normal) / vec3_dot(d, normal);
        if (t > K_EPSILON) {
            vec3 point;
            getPointOnRay(point, o, d, t);
            vec3_sub(displacement, point, p);
            float u = vec3_dot(displacement, p + 3) / vec3_dot(p + 3, p + 3);
            if (u >= 0.0f && u <= 1.0f) {
                float v = vec3_dot(displacement, p + 6) / vec3_dot(p + 6, p + 6);
                if (v >= 0.0f && v <= 1.0f) {
                    vec3_normalize(sr->dpdu, p + 3);
                    vec3_copy(sr->normal, normal);
                    vec3_copy(sr->hit_point, point);
                    vec3_negate(sr->wo, d);
                    if (vec3_dot(sr->wo, sr->normal) < 0.0f) { vec3_negate(sr->normal, sr->normal); vec3_negate(sr->dpdu, sr->dpdu); }
                    *has_normal = *has_dpdu = true;
                    return t;
                }
            }
        }
        return TMAX;
    }
    case CRT_DISK: {
        vec3 displacement;
        vec3_sub(displacement, p, o);
        float t = vec3_dot(displacement, p + 3) / vec3_dot(d, p + 3);
        if (t > K_EPSILON) {
            vec3 hit_point;
            getPointOnRay(hit_point, o, d, t);
            vec3_sub(displacement, hit_point, p);
            if (vec3_length(displacement) <= p[6]) {
                vec3_negate(sr->wo, d);
                if (vec3_dot(sr->wo, p + 3) < 0) vec3_negate(sr->wo, p + 3);      /* disk.cpp:16-19, as written */
                vec3_copy(sr->normal, p + 3);                      /* front side: the reference's write; back side: the convention */
                vec3_copy(sr->hit_point, hit_point);
                *has_normal = true;
                return t;
            }
        }
        return TMAX;
    }
    case CRT_AABOX: {
        float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
        float a = 1.0f / d[0];
        if (a >= 0) { tx_min = (p[0] - o[0]) * a; tx_max = (p[3] - o[0]) * a; } else { tx_min = (p[3] - o[0]) * a; tx_max = (p[0] - o[0]) * a; }
        float b = 1.0f / d[1];
        if (b >= 0) { ty_min = (p[1] - o[1]) * b; ty_max = (p[4] - o[1]) * b; } else { ty_min = (p[4] - o[1]) * b; ty_max = (p[1] - o[1]) * b; }
        float c = 1.0f / d[2];
        if (c >= 0) { tz_min = (p[2] - o[2]) * c; tz_max = (p[5] - o[2]) * c; } else { tz_min = (p[5] - o[2]) * c; tz_max = (p[2] - o[2]) * c; }
        float t0, t1;
        int face_in, face_out;
        if (tx_min > ty_min) { t0 = tx_min; face_in = (a >= 0.0f) ? 0 : 3; } else { t0 = ty_min; face_in = (b >= 0.0f) ? 1 : 4; }
        if (tz_min > t0) { t0 = tz_min; face_in = (c >= 0.0f) ? 2 : 5; }
        if (tx_max < ty_max) { t1 = tx_max; face_out = (a >= 0.0f) ? 3 : 0; } else { t1 = ty_max; face_out = (b >= 0.0f) ? 4 : 1; }
        if (tz_max < t1) { t1 = tz_max; face_out = (c >= 0.0f) ? 5 : 2; }
        float t_min = TMAX;
        if (t0 < t1 && t1 > K_EPSILON) {
            int face;
            if (t0 > K_EPSILON) { t_min = t0; face = face_in; } else { t_min = t1; face = face_out; }
            static const float normals[6][3] = { {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1} };   /* getAABoxNormal */
            vec3_copy(sr->normal, normals[face]);
            if (vec3_dot(sr->normal, d) > 0.0f) vec3_negate(sr->normal, sr->normal);
            getPointOnRay(sr->hit_point, o, d, t_min);
            vec3_negate(sr->wo, d);
            *has_normal = true;
        }
        return t_min;
    }
    case CRT_GENERICOPENCYLINDER: {
        const float half_height = p[0], radius = p[1];
        float a = d[0] * d[0] + d[2] * d[2];
        float b = 2 * (o[0] * d[0] + o[2] * d[2]);
        float c = o[0] * o[0] + o[2] * o[2] - radius * radius;
        float disc = b * b - 4 * a * c;
        if (disc < 0.0f) return TMAX;
        float e = (float)sqrt(disc);
        float denom = 2 * a;
        for (int root = 0; root < 2; root++) {
            float t = root == 0 ? (-b - e) / denom : (-b + e) / denom;
            if (!(t > K_EPSILON)) continue;
            vec3 hit_point;
            getPointOnRay(hit_point, o, d, t);
            if (!(fabsf(hit_point[1]) <= half_height)) continue;
            float phi = atan2f(hit_point[0], hit_point[2]);
            if (!(fabsf(phi) <= p[2])) continue;
            vec3_negate(sr->wo, d);                                /* fillShadeRecGenericOpenCylinder */
            vec3_copy(sr->hit_point, hit_point);
            const int normal_type = (int)p[3];
            if (normal_type == 2) vec3_assign(sr->normal, (-hit_point[0]) / radius, 0.0f, (-hit_point[2]) / radius);
            else {
                vec3_assign(sr->normal, hit_point[0] / radius, 0.0f, hit_point[2] / radius);
                if (normal_type == 0 && vec3_dot(sr->wo, sr->normal) < 0) vec3_negate(sr->normal, sr->normal);
            }
            *has_normal = true;
            return t;
        }
        return TMAX;
    }
    case CRT_GENERICTORUS: {
        const float t = hit_t;                                     /* the smallest root > eps of the quartic, as found by the query */
        vec3 hit_point;
        getPointOnRay(hit_point, o, d, t);
        vec3_copy(sr->hit_point, hit_point);                       /* inside the phi window: generic.cpp:207-217; outside: the convention */
        float param_squared = p[0] * p[0] + p[1] * p[1];           /* computeGenericTorusNormal */
        float x = hit_point[0], y = hit_point[1], z = hit_point[2];
        float sum_squared = x * x + y * y + z * z;
        sr->normal[0] = 4.0f * x * (sum_squared - param_squared);
        sr->normal[1] = 4.0f * y * (sum_squared - param_squared + 2.0f * p[0] * p[0]);
        sr->normal[2] = 4.0f * z * (sum_squared - param_squared);
        vec3_normalize(sr->normal, sr->normal);
        vec3_negate(sr->wo, d);
        if (vec3_dot(sr->wo, sr->normal) < 0) vec3_negate(sr->normal, sr->normal);
        *has_normal = true;
        return t;
    }
    default: return TMAX;
    }
}

void intersectShapeRecord(const crt_scene_input* in, const crt_shape& sh, const crt_ray& ray, const crt_hit& hit, ShadeRec* sr,
                          bool* has_normal, bool* has_dpdu)
{
    if (sh.type == CRT_INSTANCED) {
        const float* m = sh.inv_xform;                             /* transformRay, util/math.cpp:41-51 */
        vec3 o, d;
        for (int r = 0; r < 3; r++) {
            o[r] = m[4 * r] * ray.o[0] + m[4 * r + 1] * ray.o[1] + m[4 * r + 2] * ray.o[2] + m[4 * r + 3] * 1.0f;
            d[r] = m[4 * r] * ray.d[0] + m[4 * r + 1] * ray.d[1] + m[4 * r + 2] * ray.d[2] + m[4 * r + 3] * 0.0f;
        }
        const crt_shape* inner = &in->shapes[sh.inner];
        if (inner->type == CRT_COMPOUND) inner = &in->shapes[inner->first_part + hit.part];
        simpleShapeRecord(*inner, o, d, hit.t, sr, has_normal, has_dpdu);
        vec3 n;                                                    /* transpose(inv_transform) * (normal, 0), instanced.cpp:92-99 */
        for (int c = 0; c < 3; c++) n[c] = m[c] * sr->normal[0] + m[4 + c] * sr->normal[1] + m[8 + c] * sr->normal[2] + m[12 + c] * 0.0f;
        vec3_copy(sr->normal, n);
        getPointOnRay(sr->hit_point, ray.o, ray.d, hit.t);
    } else if (sh.type == CRT_COMPOUND) {
        simpleShapeRecord(in->shapes[sh.first_part + hit.part], ray.o, ray.d, hit.t, sr, has_normal, has_dpdu);
    } else {
        simpleShapeRecord(sh, ray.o, ray.d, hit.t, sr, has_normal, has_dpdu);
    }
}

/* ShadeRec tail of the intersector that produced `hit` */
/* getTexColor, texture.cpp:27-86: nearest texel, mirrored about 0, reflected beyond 1, v flipped; both kinds of texel
 * are divided by 255.  PARITY UNPINNED: texture.cpp includes stb_image.h, which the reference does not ship. */
bool getTexColor(vec3 color_out, const crt_texture* texture, const vec2 texcoord)
{
    float u_float = texcoord[0];
    if (u_float < 0.0f) u_float = -u_float;
    if (u_float > 1.0f) {
        u_float = u_float - (int)u_float;
        u_float = 1.0f - u_float;
    }
    u_float = u_float * texture->width;
    int u;
    if (u_float - (int)u_float > 0.5f) u = (int)u_float + 1;
    else u = (int)u_float;
    u = u % texture->width;

    float v_float = texcoord[1];
    if (v_float < 0.0f) v_float = -v_float;
    if (v_float > 1.0f) {
        v_float = v_float - (int)v_float;
        v_float = 1.0f - v_float;
    }
    v_float = 1.0f - v_float;
    v_float = v_float * texture->height;
    int v;
    if (v_float - (int)v_float > 0.5f) v = (int)v_float + 1;
    else v = (int)v_float;
    v = v % texture->height;

    if (texture->is_float) {
        const size_t row_len = (size_t)texture->width * texture->comp * sizeof(float);
        const float* texel_ptr = (const float*)((const unsigned char*)texture->data + v * row_len + (size_t)u * texture->comp * sizeof(float));
        vec3_assign(color_out, texel_ptr[0] / 255.0f, texel_ptr[1] / 255.0f, texel_ptr[2] / 255.0f);
    } else {
        const size_t row_len = (size_t)texture->width * texture->comp;
        const unsigned char* texel_ptr = (const unsigned char*)texture->data + v * row_len + (size_t)u * texture->comp;
        vec3_assign(color_out, texel_ptr[0] / 255.0f, texel_ptr[1] / 255.0f, texel_ptr[2] / 255.0f);
    }
    return true;
}
/* cartesianToSpherical + sphericalToUV, util/math.h:95-107.  PI there is a double (util/vec.h:6), so `out[0] += PI` and the two
 * quotients are double arithmetic rounded when stored.  atan2f / acosf are taken as the double function rounded once (shared with
 * the product: what host and device agree on); pinned against util/math.h through oracle/_ref to that allowance. */
void envDirToUV(vec2 uv, const vec3 in)
{
    const double PI_D = 3.14159265358979323846264338327950288;
    vec2 spherical;
    spherical[0] = (float)atan2((double)in[2], (double)in[0]);
    spherical[0] = (float)((double)spherical[0] + PI_D);
    const float y = in[1] < -1.0f ? -1.0f : (in[1] > 1.0f ? 1.0f : in[1]);
    spherical[1] = (float)acos((double)y);
    uv[0] = (float)((double)spherical[0] / (2.0f * PI_D));
    uv[1] = (float)(1.0f - (double)spherical[1] / PI_D);
}
/* getEnvLightIncRadiance, lights.cpp:233-248 */
void getEnvLightIncRadiance(const Setup& S, vec3 r, const vec3 dir)
{
    const crt_render_desc* d = S.d;
    if (d->env_map == 0) { vec3_scale(r, d->env_color, d->env_intensity); return; }
    vec2 uv;
    envDirToUV(uv, dir);
    vec3 tex_color;
    getTexColor(tex_color, &d->textures[d->env_map - 1], uv);
    vec3_scale(r, tex_color, d->env_intensity);
}
/* mat3_mult_vec3(out, env_light->transform, v); an all-zero env_transform stands for the identity */
void envTransform(const Setup& S, vec3 out, const vec3 v)
{
    const float* m = S.d->env_transform;
    bool zero = true;
    for (int k = 0; k < 9; k++) if (m[k] != 0.0f) zero = false;
    const float id[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
    if (zero) m = id;
    vec3 t = { vec3_dot(m, v), vec3_dot(m + 3, v), vec3_dot(m + 6, v) };
    vec3_copy(out, t);
}
/* interpTexcoord, triangle.cpp:170-183, and the wrap loops of :195-198 / :240-243 */
void interpTexcoordWrapped(vec2 uv_out, float beta, float gamma, const float* t)
{
    const float w = 1.0f - beta - gamma;
    float u = t[0] * w, v = t[1] * w;
    u = u + t[2] * beta; v = v + t[3] * beta;
    u = u + t[4] * gamma; v = v + t[5] * gamma;
    uv_out[0] = u; uv_out[1] = v;
    for (int k = 0; k < 2; k++) {
        float x = uv_out[k];
        if (!(x == x) || x > 1.0e6f || x < -1.0e6f) continue;     /* the reference's loops would spin */
        while (x > 1.0f) x -= 1.0f;
        while (x < 0.0f) x += 1.0f;
        uv_out[k] = x;
    }
}

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
    sr->mat_index = S.d->obj_material ? S.d->obj_material[hit.obj] : -1;
    sr->has_uv = false; sr->uv[0] = sr->uv[1] = 0.0f;
    if ((type == CRT_FLAT_TRIANGLE || type == CRT_SMOOTH_TRIANGLE) && S.d->mesh_shading && S.d->mesh_shading->vertex_uv) {
        interpTexcoordWrapped(sr->uv, hit.beta, hit.gamma, S.d->mesh_shading->vertex_uv + 6 * (size_t)ref);
        sr->has_uv = true;
    }
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
        {                                                          /* rect.cpp:22-37: the edge coordinates of the hit point */
            vec3 d2;
            vec3_sub(d2, sr->hit_point, p);
            sr->uv[0] = vec3_dot(d2, p + 3) / vec3_dot(p + 3, p + 3);
            sr->uv[1] = vec3_dot(d2, p + 6) / vec3_dot(p + 6, p + 6);
            sr->has_uv = true;
        }
    } else {                                                       /* every other kind: the hit object's own intersector, re-run */
        bool has_normal = false, has_dpdu = false;
        intersectShapeRecord(in, in->shapes[ref], ray, hit, sr, &has_normal, &has_dpdu);
        if (!has_dpdu) {                                           /* shared convention: dpdu where the reference writes none */
            vec3 c;
            vec3_cross(c, JITTERED_UP, sr->normal);
            vec3_normalize(sr->dpdu, c);
        }
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

/* exp / log / tan / atan of the microfacet code.  The reference calls glibc's float routines (microfacet.cpp:19,:111,
 * :121-122), which are within an ulp but not correctly rounded, and no GPU libm returns the same bits; product and oracle
 * both take the double-precision function and round once.  Against the reference's own lobes that is a last-ulp
 * difference in a small fraction of the calls (tests/test_trace_pieces.py states the bound and the observed count). */
inline float r_expf(float x) { return (float)exp((double)x); }
inline float r_logf(float x) { return (float)log((double)x); }
inline float r_tanf(float x) { return (float)tan((double)x); }
inline float r_atanf(float x) { return (float)atan((double)x); }

/* util/math.h:13-52 */
inline float cosTheta(const vec3 w) { return w[2]; }
inline float cos2Theta(const vec3 w) { return w[2] * w[2]; }
inline float absCosTheta(const vec3 w) { return fabsf(w[2]); }
inline float sin2Theta(const vec3 w) { return maxf((float)0, (float)1 - cos2Theta(w)); }
inline float tanTheta(const vec3 w) { return sinTheta(w) / cosTheta(w); }
inline float tan2Theta(const vec3 w) { return sin2Theta(w) / cos2Theta(w); }
inline float cos2Phi(const vec3 w) { return cosPhi(w) * cosPhi(w); }
inline float sin2Phi(const vec3 w) { return sinPhi(w) * sinPhi(w); }
inline bool sameHemisphere(const vec3 a, const vec3 b) { return a[2] * b[2] > 0.0f; }

/* microfacet.cpp:4-21, BECKMANN */
float MicrofacetDistribution_D(const vec3 wh, const MicrofacetDistribution* distrib)
{
    float tan_2_theta = tan2Theta(wh);
    if (isinf(tan_2_theta)) return 0.0f;
    float cos_4_theta = cos2Theta(wh) * cos2Theta(wh);
    return r_expf(-tan_2_theta * (cos2Phi(wh) / (distrib->alphax * distrib->alphax) +
                                  sin2Phi(wh) / (distrib->alphay * distrib->alphay))) /
        (PI * distrib->alphax * distrib->alphay * cos_4_theta);
}
/* microfacet.cpp:33-55 */
float MicrofacetDistribution_Lambda(const vec3 w, const MicrofacetDistribution* distrib)
{
    float abs_tan_theta = fabsf(tanTheta(w));
    if (isinf(abs_tan_theta)) return 0.0f;
    float alpha = sqrtf(cos2Phi(w) * distrib->alphax * distrib->alphax + sin2Phi(w) * distrib->alphay * distrib->alphay);
    float a = 1.0f / (alpha * abs_tan_theta);
    if (a >= 1.6f) return 0.0f;
    return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}
/* microfacet.cpp:72-75 */
float MicrofacetDistribution_G(const vec3 wo, const vec3 wi, const MicrofacetDistribution* distrib)
{
    return 1.0f / (1.0f + MicrofacetDistribution_Lambda(wo, distrib) + MicrofacetDistribution_Lambda(wi, distrib));
}
/* microfacet.cpp:77-135; sphericalDirection, util/math.h:91-93 */
void MicrofacetDistribution_sample_wh(vec3 wh, const vec3 wo, const vec2 sample, const MicrofacetDistribution* distrib)
{
    float tan_2_theta, phi;
    if (distrib->alphax == distrib->alphay) {
        float log_sample = r_logf(sample[0]);
        if (isinf(log_sample)) log_sample = 0.0f;
        tan_2_theta = -distrib->alphax * distrib->alphax * log_sample;
        phi = sample[1] * 2.0f * PI;
    } else {
        float log_sample = r_logf(sample[0]);
        if (isinf(log_sample)) log_sample = 0.0f;
        phi = r_atanf(distrib->alphay / distrib->alphax * r_tanf(2.0f * PI * sample[1] + 0.5f * PI));
        if (sample[1] > 0.5f) phi += PI;
        float sin_phi = sinf(phi), cos_phi = cosf(phi);
        float alphax2 = distrib->alphax * distrib->alphax;
        float alphay2 = distrib->alphay * distrib->alphay;
        tan_2_theta = -log_sample / (cos_phi * cos_phi / alphax2 + sin_phi * sin_phi / alphay2);
    }
    float cos_theta = 1.0f / sqrtf(1.0f + tan_2_theta);
    float sin_theta = sqrtf(maxf(0.0f, 1.0f - cos_theta * cos_theta));
    vec3_assign(wh, sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    if (!sameHemisphere(wo, wh)) vec3_negate(wh, wh);
}
/* microfacet.cpp:137-140 */
float MicrofacetDistribution_pdf(const vec3 wo, const vec3 wh, const MicrofacetDistribution* distrib)
{
    (void)wo;
    return MicrofacetDistribution_D(wh, distrib) * absCosTheta(wh);
}
/* microfacet.h:25-31; runs once per material, on the host in the product too, so glibc's logf as in the reference */
float BeckmannRoughnessToAlpha(float roughness)
{
    roughness = maxf(roughness, 1e-3);
    float x = logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

/* calcFresnelConductor, reflection.cpp:78-157 */
void calcFresnelConductor(vec3 fresnel, const vec3 normal, const vec3 wo, const vec3 etaT, const vec3 etaI, const vec3 k)
{
    float cos_theta_i = vec3_dot(wo, normal);
    float cos_theta_i_2 = cos_theta_i * cos_theta_i;
    float sin_theta_i_2 = 1.0f - cos_theta_i_2;
    for (int c = 0; c < 3; c++) {                                  /* the vec3_* calls of the reference, per component */
        float eta = etaT[c] / etaI[c];
        float etaK = k[c] / etaI[c];
        float eta2 = eta * eta;
        float etaK2 = etaK * etaK;
        float t0 = eta2 - etaK2;
        t0 = t0 - sin_theta_i_2;
        float t0_2 = t0 * t0;
        float eta2_2 = eta2 * etaK2;
        eta2_2 = eta2_2 * 4.0f;
        float a2plusb2 = sqrtf(t0_2 + eta2_2);
        float t1 = a2plusb2 + cos_theta_i_2;
        float a = a2plusb2 + t0;
        a = a * 0.5f;
        a = sqrtf(a);
        float t2 = a * (2.0f * cos_theta_i);
        float rs = (t1 - t2) / (t1 + t2);
        float t3 = a2plusb2 * cos_theta_i_2;
        t3 = t3 + sin_theta_i_2 * sin_theta_i_2;
        float t4 = t2 * sin_theta_i_2;
        float rp = (t3 - t4) / (t3 + t4);
        rp = rp * rs;
        fresnel[c] = (rp + rs) * 0.5f;
    }
}

/* util/ray.cpp:11-17,:42-59 */
void calcReflectRayDir(vec3 reflect_dir, const vec3 normal, const vec3 incident_dir)
{
    float magnitude = -2.0f * vec3_dot(incident_dir, normal);
    vec3 adjusted_normal;
    vec3_scale(adjusted_normal, normal, magnitude);
    vec3_add(reflect_dir, incident_dir, adjusted_normal);
}
bool refract(vec3 wi, const vec3 wo, const vec3 n, const float eta)
{
    float cos_theta_i = vec3_dot(n, wo);
    float sin_2_theta_i = maxf(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    float sin_2_theta_t = eta * eta * sin_2_theta_i;
    if (sin_2_theta_t >= 1.0f) return false;
    float cos_theta_t = sqrtf(1.0f - sin_2_theta_t);
    vec3 perp_comp, neg_wo;
    vec3_negate(neg_wo, wo);
    vec3_scale(perp_comp, neg_wo, eta);
    float scale_factor = eta * cos_theta_i - cos_theta_t;
    vec3 parallel_comp;
    vec3_scale(parallel_comp, n, scale_factor);
    vec3_add(wi, perp_comp, parallel_comp);
    return true;
}

int getBxDFFlagsFromType(int type)                                 /* reflection.h:32-57 */
{
    switch (type) {
    case LAMBERTIAN: case ORENNAYAR: return BSDF_REFLECTION | BSDF_DIFFUSE;
    case SPECULAR_REFLECTION: return BSDF_REFLECTION | BSDF_SPECULAR;
    case SPECULAR_TRANSMISSION: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
    case MICROFACET_REFLECTION: return BSDF_REFLECTION | BSDF_GLOSSY;
    case MICROFACET_FRESNEL: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_GLOSSY;
    case FRESNEL_BLEND: case FRESNEL_BLEND_DIFFUSE: return BSDF_REFLECTION | BSDF_DIFFUSE;
    case FRESNEL_BLEND_SPECULAR: return BSDF_REFLECTION | BSDF_GLOSSY;
    default: return BSDF_NONE;
    }
}

/* MicrofacetReflection_f, reflection.cpp:297-327.  f is written only when the lobe has something to say. */
void MicrofacetReflection_f(vec3 f, const vec3 wi, const vec3 wo, const BxDF* mf)
{
    float cos_theta_o = absCosTheta(wo), cos_theta_i = absCosTheta(wi);
    vec3 wh;
    vec3_add(wh, wi, wo);
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return;
    if (wh[0] == 0.0f && wh[1] == 0.0f && wh[2] == 0.0f) return;
    vec3_normalize(wh, wh);
    vec3 fresnel_reflectance;
    if (!mf->is_metal) {
        float kr = 1.0f - calcFresnelDielectric(wh, wi, mf->ior_in, mf->ior_out);      /* as written there: 1 - kr */
        vec3_assign(fresnel_reflectance, kr, kr, kr);
    } else {
        calcFresnelConductor(fresnel_reflectance, wh, wi, mf->etaT, mf->etaI, mf->k);
    }
    float scale_factor = MicrofacetDistribution_D(wh, &mf->distrib) * MicrofacetDistribution_G(wo, wi, &mf->distrib) /
        (4.0f * cos_theta_i * cos_theta_o);
    vec3_scale(f, mf->color, scale_factor);
    vec3_mult(f, f, fresnel_reflectance);
}
/* reflection.cpp:329-344 */
float MicrofacetReflection_sample_f(vec3 f, vec3 wi, const vec3 wo, const vec2 sample, const BxDF* mr)
{
    if (wo[2] == 0.0f) return 0.0f;
    vec3 wh;
    MicrofacetDistribution_sample_wh(wh, wo, sample, &mr->distrib);
    vec3 neg_wo;
    vec3_negate(neg_wo, wo);
    calcReflectRayDir(wi, wh, neg_wo);
    if (!sameHemisphere(wo, wi)) return 0.0f;
    MicrofacetReflection_f(f, wi, wo, mr);
    return MicrofacetDistribution_pdf(wo, wh, &mr->distrib) / (4.0f * vec3_dot(wo, wh));
}
/* reflection.cpp:346-353 */
float MicrofacetReflection_pdf(const vec3 wi, const vec3 wo, const BxDF* mr)
{
    if (!sameHemisphere(wi, wo)) return 0.0f;
    vec3 wh;
    vec3_add(wh, wo, wi);
    vec3_normalize(wh, wh);
    return MicrofacetDistribution_pdf(wo, wh, &mr->distrib) / (4.0f * vec3_dot(wo, wh));
}
/* reflection.cpp:356-387 */
void MicrofacetFresnel_f(vec3 f, const vec3 wi, const vec3 wo, const BxDF* mt)
{
    if (sameHemisphere(wo, wi)) return;
    float cos_theta_o = cosTheta(wo);
    float cos_theta_i = cosTheta(wi);
    if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) { vec3_copy(f, BLACK); return; }
    float eta = cos_theta_o > 0.0f ? (mt->ior_in / mt->ior_out) : (mt->ior_out / mt->ior_in);
    vec3 wh, scaled_wi;
    vec3_scale(scaled_wi, wi, eta);
    vec3_add(wh, wo, scaled_wi);
    vec3_normalize(wh, wh);
    if (wh[2] < 0.0f) vec3_negate(wh, wh);
    float fresnel_reflectance = calcFresnelDielectric(wh, wo, mt->ior_in, mt->ior_out);
    float sqrt_denom = vec3_dot(wo, wh) + eta * vec3_dot(wi, wh);
    float denom = cos_theta_i * cos_theta_o * sqrt_denom * sqrt_denom;
    float factor = 1.0f / eta;
    float numerator = MicrofacetDistribution_D(wh, &mt->distrib) * MicrofacetDistribution_G(wo, wi, &mt->distrib) * eta * eta *
        fabsf(vec3_dot(wi, wh)) * fabsf(vec3_dot(wo, wh)) * factor * factor;
    vec3 tmp;
    vec3_scale(tmp, mt->color, 1.0f - fresnel_reflectance);
    vec3_scale(f, tmp, fabsf(numerator / denom));
}
/* reflection.cpp:447-458 */
float MicrofacetFresnel_pdf(const vec3 wi, const vec3 wo, const BxDF* mt)
{
    if (sameHemisphere(wo, wi)) return 0.0f;
    float eta = cosTheta(wo) > 0.0f ? (mt->ior_in / mt->ior_out) : (mt->ior_out / mt->ior_in);
    vec3 wh, scaled_wi;
    vec3_scale(scaled_wi, wi, eta);
    vec3_add(wh, wo, scaled_wi);
    vec3_normalize(wh, wh);
    float sqrt_denom = vec3_dot(wo, wh) + eta * vec3_dot(wi, wh);
    float dwh_dwi = fabsf((eta * eta * vec3_dot(wi, wh)) / (sqrt_denom * sqrt_denom));
    return MicrofacetDistribution_pdf(wo, wh, &mt->distrib) * dwh_dwi;
}
/* reflection.cpp:389-445.  The MicrofacetReflection it builds on the stack for the reflected branch leaves is_metal
 * unset in the reference; it is the dielectric form here (DESIGN.md "N1", deliberate differences). */
float MicrofacetFresnel_sample_f(vec3 f, vec3 wi, const vec3 wo, const vec2 sample, const BxDF* mt)
{
    if (wo[2] == 0.0f) return 0.0f;
    vec3 wh;
    MicrofacetDistribution_sample_wh(wh, wo, sample, &mt->distrib);
    float kr = calcFresnelDielectric(wh, wo, mt->ior_in, mt->ior_out);
    float rand_float = mt->rand_float;
    if (rand_float <= kr) {
        vec3 neg_wo;
        vec3_negate(neg_wo, wo);
        calcReflectRayDir(wi, wh, neg_wo);
        if (!sameHemisphere(wo, wi)) return 0.0f;
        BxDF mr = *mt;
        mr.is_metal = false;
        MicrofacetReflection_f(f, wi, wo, &mr);
        return MicrofacetDistribution_pdf(wo, wh, &mr.distrib) / (4.0f * vec3_dot(wo, wh));
    } else {
        float eta = cosTheta(wo) > 0.0f ? (mt->ior_out / mt->ior_in) : (mt->ior_in / mt->ior_out);
        if (!refract(wi, wo, wh, eta)) return 0.0f;
        MicrofacetFresnel_f(f, wi, wo, mt);
        return MicrofacetFresnel_pdf(wi, wo, mt);
    }
}

inline float pow5(float v) { return (v * v) * (v * v) * v; }       /* reflection.cpp:460-463 */
/* reflection.cpp:465-477 */
void schlickFresnel(vec3 ret, const float cos_theta, const vec3 Rs)
{
    vec3 ones = {1.0f, 1.0f, 1.0f};
    vec3_sub(ret, ones, Rs);
    vec3_scale(ret, ret, pow5(1.0f - cos_theta));
    vec3_add(ret, ret, Rs);
}
/* reflection.cpp:479-490 */
void FresnelBlendDiffuse_f(vec3 f, const vec3 wi, const vec3 wo, const BxDF* fbd)
{
    vec3 diffuse;
    vec3_scale(diffuse, fbd->color, 28.0f / (23.0f * PI));
    vec3 ones = {1.0f, 1.0f, 1.0f};
    vec3 one_minus_ks;
    vec3_sub(one_minus_ks, ones, fbd->ks);
    vec3_mult(diffuse, diffuse, one_minus_ks);
    vec3_scale(diffuse, diffuse, 1.0f - pow5(1.0f - 0.5f * absCosTheta(wi)));
    vec3_scale(diffuse, diffuse, 1.0f - pow5(1.0f - 0.5f * absCosTheta(wo)));
    vec3_copy(f, diffuse);
}
/* reflection.cpp:492-499 */
float FresnelBlendDiffuse_sample_f(vec3 f, vec3 wi, const vec3 wo, const vec2 sample, const BxDF* fbd)
{
    mapSampleToHemisphere(wi, sample);                             /* cosHemisphereSample; its pdf is not used */
    if (wo[2] < 0.0f) wi[2] = -wi[2];
    FresnelBlendDiffuse_f(f, wi, wo, fbd);
    return cosHemispherePdf(wi, wo);
}
/* reflection.cpp:507-522 */
void FresnelBlendSpecular_f(vec3 f, const vec3 wi, const vec3 wo, const BxDF* fbs)
{
    vec3 wh;
    vec3_add(wh, wi, wo);
    if (vec3_equal(wh, BLACK)) return;
    vec3_normalize(wh, wh);
    vec3 specular, fresnel;
    schlickFresnel(fresnel, vec3_dot(wi, wh), fbs->ks);
    float cos_theta = vec3_dot(wi, wh);
    vec3_scale(specular, fresnel, MicrofacetDistribution_D(wh, &fbs->distrib) /
               (4.0f * fabsf(cos_theta) * maxf(absCosTheta(wi), absCosTheta(wo))));
    vec3_copy(f, specular);
}
/* reflection.cpp:537-547 */
float FresnelBlendSpecular_pdf(const vec3 wi, const vec3 wo, const BxDF* fbs)
{
    if (!sameHemisphere(wo, wi)) return 0.0f;
    vec3 wh;
    vec3_add(wh, wo, wi);
    vec3_normalize(wh, wh);
    float pdf_wh = MicrofacetDistribution_D(wh, &fbs->distrib);
    return pdf_wh / (2.0f * vec3_dot(wo, wh));
}
/* reflection.cpp:524-535 */
float FresnelBlendSpecular_sample_f(vec3 f, vec3 wi, const vec3 wo, const vec2 sample, const BxDF* fbs)
{
    vec3 wh;
    MicrofacetDistribution_sample_wh(wh, wo, sample, &fbs->distrib);
    vec3 neg_wo;
    vec3_negate(neg_wo, wo);
    calcReflectRayDir(wi, wh, neg_wo);
    if (!sameHemisphere(wo, wi)) return 0.0f;
    FresnelBlendSpecular_f(f, wi, wo, fbs);
    return FresnelBlendSpecular_pdf(wi, wo, fbs);
}

/* BxDF_f, reflection.cpp:643-665 */
void BxDF_f(vec3 f, const vec3 wi, const vec3 wo, const BxDF* bxdf)
{
    switch (bxdf->type) {
    case LAMBERTIAN: vec3_copy(f, bxdf->color); vec3_scale(f, f, INV_PI); return;       /* Lambertian_f, :159-163 */
    case ORENNAYAR: OrenNayar_f(f, wi, wo, bxdf->color, bxdf->a, bxdf->b); return;
    case SPECULAR_REFLECTION: case SPECULAR_TRANSMISSION: break;
    case MICROFACET_REFLECTION: MicrofacetReflection_f(f, wi, wo, bxdf); return;
    case MICROFACET_FRESNEL: MicrofacetFresnel_f(f, wi, wo, bxdf); return;
    case FRESNEL_BLEND_DIFFUSE: FresnelBlendDiffuse_f(f, wi, wo, bxdf); return;
    case FRESNEL_BLEND_SPECULAR: FresnelBlendSpecular_f(f, wi, wo, bxdf); return;
    }
}
/* BxDF_pdf, reflection.cpp:693-717 */
float BxDF_pdf(const vec3 wi, const vec3 wo, const BxDF* bxdf)
{
    switch (bxdf->type) {
    case LAMBERTIAN: case ORENNAYAR: return cosHemispherePdf(wi, wo);
    case SPECULAR_REFLECTION: case SPECULAR_TRANSMISSION: return 0.0f;
    case MICROFACET_REFLECTION: return MicrofacetReflection_pdf(wi, wo, bxdf);
    case MICROFACET_FRESNEL: return MicrofacetFresnel_pdf(wi, wo, bxdf);
    case FRESNEL_BLEND_DIFFUSE: return cosHemispherePdf(wi, wo);
    case FRESNEL_BLEND_SPECULAR: return FresnelBlendSpecular_pdf(wi, wo, bxdf);
    }
    return 0.0f;
}
/* BxDF_sample_f, reflection.cpp:667-691 */
float BxDF_sample_f(vec3 f, vec3 wi, const vec3 wo, const vec2 sample, const BxDF* bxdf)
{
    switch (bxdf->type) {
    case LAMBERTIAN: {                                             /* Lambertian_sample_f, :177-189 */
        vec3 tmp_wo;
        vec3_copy(tmp_wo, wo);
        if (tmp_wo[2] < 0.0f) tmp_wo[2] *= -1.0f;
        vec3_copy(f, bxdf->color);
        vec3_scale(f, f, INV_PI);
        mapSampleToHemisphere(wi, sample);
        return cosHemispherePdf(wi, tmp_wo);
    }
    case ORENNAYAR: {                                              /* OrenNayar_sample_f, :221-232 */
        vec3 tmp_wo;
        vec3_copy(tmp_wo, wo);
        if (tmp_wo[2] < 0.0f) tmp_wo[2] *= -1.0f;
        mapSampleToHemisphere(wi, sample);
        float pdf = cosHemispherePdf(wi, tmp_wo);
        OrenNayar_f(f, wi, wo, bxdf->color, bxdf->a, bxdf->b);
        return pdf;
    }
    case SPECULAR_REFLECTION:                                      /* SpecularReflection_sample_f, :234-241 */
        vec3_assign(wi, -wo[0], -wo[1], wo[2]);
        vec3_scale(f, bxdf->color, 1.0f / fabsf(cosTheta(wi)));
        return 1.0f;
    case SPECULAR_TRANSMISSION: {                                  /* SpecularTransmission_sample_f, `thin`, :243-270 */
        const vec3 WHITE = {1.0f, 1.0f, 1.0f};
        vec3 normal = {0.0f, 0.0f, 1.0f};
        float cos_theta_o = vec3_dot(normal, wo);
        if (cos_theta_o < 0.0f) vec3_negate(normal, normal);
        float kr = calcFresnelDielectric(normal, wo, bxdf->ior_in, bxdf->ior_out);
        if (bxdf->rand_float <= kr) {
            vec3_assign(wi, -wo[0], -wo[1], wo[2]);
            vec3_scale(f, WHITE, kr / fabsf(cosTheta(wi)));
            return kr;
        } else {
            float eta = bxdf->ior_out / bxdf->ior_in;
            vec3_negate(wi, wo);
            vec3_scale(f, WHITE, 1.0f - kr);
            vec3_scale(f, f, (eta * eta) / fabsf(cosTheta(wi)));
            return 1.0f - kr;
        }
    }
    case MICROFACET_REFLECTION: return MicrofacetReflection_sample_f(f, wi, wo, sample, bxdf);
    case MICROFACET_FRESNEL: return MicrofacetFresnel_sample_f(f, wi, wo, sample, bxdf);
    case FRESNEL_BLEND_DIFFUSE: return FresnelBlendDiffuse_sample_f(f, wi, wo, sample, bxdf);
    case FRESNEL_BLEND_SPECULAR: return FresnelBlendSpecular_sample_f(f, wi, wo, sample, bxdf);
    }
    return 0.0f;
}

/* BSDF_f, reflection.cpp:719-735: the lobes receive the WORLD-space wi, wo (line 731 passes wi, wo, not the
 * wi_local, wo_local computed above it) */
void BSDF_f(vec3 f, const vec3 wi, const vec3 wo, const ShadeRec* sr, const int excluded)
{
    vec3_copy(f, BLACK);
    for (int i = 0; i < sr->num_bxdf; i++) {
        if (!(getBxDFFlagsFromType(sr->bxdfs[i].type) & excluded)) {
            vec3 cur_f = {0.0f, 0.0f, 0.0f};
            BxDF_f(cur_f, wi, wo, &sr->bxdfs[i]);
            vec3_add(f, f, cur_f);
        }
    }
}

/* BSDF_sample_f, reflection.cpp:750-811 */
float BSDF_sample_f(vec3 f, vec3 wi, int* sampled_flags, const vec3 wo, const vec2 sample, const ShadeRec* sr)
{
    if (sr->num_bxdf == 0) { vec3_copy(f, BLACK); *sampled_flags = BSDF_NONE; return 0.0f; }
    int bxdf_index = (int)(sample[0] * sr->num_bxdf);
    *sampled_flags = getBxDFFlagsFromType(sr->bxdfs[bxdf_index].type);
    vec2 remapped_sample = {sample[0] * sr->num_bxdf - bxdf_index, sample[1]};
    vec3 wi_local, wo_local;
    transposeTransform(wo_local, sr->b_tangent, sr->b_binormal, sr->b_normal, wo);
    if (wo_local[2] == 0.0f) { vec3_copy(f, BLACK); return 0.0f; }   /* the reference returns with f unset; the caller only tests pdf */
    vec3_copy(f, BLACK);
    vec3_assign(wi_local, 0.0f, 0.0f, 0.0f);
    float pdf = BxDF_sample_f(f, wi_local, wo_local, remapped_sample, &sr->bxdfs[bxdf_index]);
    if (pdf == 0.0f) { vec3_copy(f, BLACK); return pdf; }
    orthoNormalTransform(wi, sr->b_tangent, sr->b_binormal, sr->b_normal, wi_local);
    if (!(*sampled_flags & BSDF_SPECULAR)) {
        for (int i = 0; i < sr->num_bxdf; i++)
            if (i != bxdf_index) pdf += BxDF_pdf(wi_local, wo_local, &sr->bxdfs[i]);
    }
    for (int i = 0; i < sr->num_bxdf; i++) {
        if (i != bxdf_index) {
            vec3 cur_f = {0.0f, 0.0f, 0.0f};
            BxDF_f(cur_f, wi_local, wo_local, &sr->bxdfs[i]);
            vec3_add(f, f, cur_f);
        }
    }
    return pdf;
}

/* computeScatteringFunc, materials.cpp:111-188, without textures */
void computeScatteringFunc(ShadeRec* sr, const crt_material* mat_in, float rand_float, const crt_render_desc* d = 0)
{
    sr->num_bxdf = 0;
    BxDF b; memset(&b, 0, sizeof b);
    b.rand_float = rand_float;
    const vec3 WHITE = {1.0f, 1.0f, 1.0f};
    /* Matte_getDiffuseColor / Plastic.diffuse (materials.cpp:51-60,:146-149): the texel at sr->uv replaces color / kd */
    crt_material textured = *mat_in;
    const crt_material* mat = mat_in;
    if (d && d->material_diffuse_tex && sr->mat_index >= 0 && d->material_diffuse_tex[sr->mat_index] >= 0 && sr->has_uv &&
        (mat_in->type == CRT_MAT_MATTE || mat_in->type == CRT_MAT_PLASTIC)) {
        getTexColor(textured.color, &d->textures[d->material_diffuse_tex[sr->mat_index]], sr->uv);
        mat = &textured;
    }
    switch (mat->type) {
    case CRT_MAT_MATTE:
        vec3_copy(b.color, mat->color);
        if (mat->sigma == 0.0f) b.type = LAMBERTIAN;               /* BSDF_addLambertian */
        else { b.type = ORENNAYAR; orenNayarAB(mat->sigma, &b.a, &b.b); }
        sr->bxdfs[sr->num_bxdf++] = b;
        break;
    case CRT_MAT_MIRROR:
        b.type = SPECULAR_REFLECTION; vec3_copy(b.color, mat->color);
        sr->bxdfs[sr->num_bxdf++] = b;
        break;
    case CRT_MAT_TRANSPARENT:
        b.type = SPECULAR_TRANSMISSION; b.ior_in = mat->ior_in; b.ior_out = mat->ior_out;
        sr->bxdfs[sr->num_bxdf++] = b;
        break;
    case CRT_MAT_PLASTIC: {                                        /* materials.cpp:142-171 */
        b.type = FRESNEL_BLEND_DIFFUSE; vec3_copy(b.color, mat->color); vec3_copy(b.ks, mat->ks);      /* BSDF_addFresnelBlendDiffuse */
        sr->bxdfs[sr->num_bxdf++] = b;
        BxDF s; memset(&s, 0, sizeof s);                           /* BSDF_addFresnelBlendSpecular, reflection.cpp:940-955: alpha = roughness */
        s.type = FRESNEL_BLEND_SPECULAR; vec3_copy(s.ks, mat->ks);
        s.distrib.alphax = mat->uroughness; s.distrib.alphay = mat->uroughness;
        s.ior_in = 1.5f; s.ior_out = 1.0f;
        sr->bxdfs[sr->num_bxdf++] = s;
    } break;
    case CRT_MAT_GLASS:                                            /* materials.cpp:172-178; BSDF_addMicrofacetFresnel, reflection.cpp:899-912 */
        b.type = MICROFACET_FRESNEL; vec3_copy(b.color, WHITE);
        b.distrib.alphax = BeckmannRoughnessToAlpha(mat->uroughness);
        b.distrib.alphay = BeckmannRoughnessToAlpha(mat->vroughness);
        b.ior_in = mat->ior_in; b.ior_out = mat->ior_out;
        sr->bxdfs[sr->num_bxdf++] = b;
        break;
    case CRT_MAT_METAL:                                            /* materials.cpp:179-186; BSDF_addMicrofacetReflectionMetal, reflection.cpp:879-897 */
        b.type = MICROFACET_REFLECTION; vec3_copy(b.color, WHITE);
        b.distrib.alphax = mat->uroughness; b.distrib.alphay = mat->vroughness;
        vec3_copy(b.etaT, mat->color); vec3_copy(b.etaI, WHITE); vec3_copy(b.k, mat->k);
        b.is_metal = true;
        sr->bxdfs[sr->num_bxdf++] = b;
        break;
    default: break;
    }
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
        orthoNormalTransform(h_sample, sr->b_tangent, sr->b_binormal, sr->b_normal, h_in);   /* transformToLocalBasis */
        vec3 h_copy;
        vec3_copy(h_copy, h_sample);
        envTransform(S, h_sample, h_copy);                         /* mat3_mult_vec3(h_sample, env_light->transform, h_copy), trace.h:298-300 */
        vec3 displacement;
        vec3_scale(displacement, h_sample, S.d->env_world_radius);
        vec3_add(sample_point, sr->hit_point, displacement);
        vec3_negate(sample_normal, h_sample);
        vec3_copy(wi, h_sample);
        light_pdf = fabsf(vec3_dot(h_sample, sr->normal)) * INV_PI;
        getEnvLightIncRadiance(S, Li, wi);                         /* trace.h:309 */
    }
    vec3 sample_to_hit_point;
    vec3_sub(sample_to_hit_point, sample_point, sr->hit_point);
    if (vec3_dot(sample_to_hit_point, sample_normal) > 0.0f || vec3_dot(sample_to_hit_point, sr->normal) < 0.0f) return false;
    vec3 f;
    if (light_pdf > 0.0f && !vec3_equal(Li, BLACK)) {
        BSDF_f(f, wi, sr->wo, sr, BSDF_SPECULAR | BSDF_GLOSSY);    /* excluded_from_direct, trace.h:410 */
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
                if (S.d->has_env_light) {                          /* trace.h:436-441 */
                    vec3 dir;
                    envTransform(S, dir, ray.d);
                    getEnvLightIncRadiance(S, env_inc_radiance, dir);
                }
                vec3_mult(env_inc_radiance, env_inc_radiance, beta);
                vec3_add(L, L, env_inc_radiance);
                good_paths++;
            }
        }
        if (t == TMAX || bounces >= depth || sr.mat->type == CRT_MAT_EMISSIVE) break;

        computeLocalBasis(&sr);
        const uint32_t dim0 = 4 + 8 * (uint32_t)bounces;
        computeScatteringFunc(&sr, sr.mat, rng(dim0 + 2), S.d);    /* dim0 + 2 stands in for rand() inside the lobes */

        vec2 light_sample = { rng(dim0 + 0), rng(dim0 + 1) };
        /* scatter_sample: dim0 + 2, dim0 + 3, drawn by the reference and unused */
        if (!(sr.mat->type == CRT_MAT_MIRROR || sr.mat->type == CRT_MAT_TRANSPARENT || sr.mat->type == CRT_MAT_GLASS)) {
            vec3 contrib;
            uniformSampleOneLight(S, contrib, light_sample, &sr, rng(dim0 + 6));
            vec3_mult(contrib, contrib, beta);
            vec3_add(L, L, contrib);
            if (!vec3_equal(contrib, BLACK)) good_paths++;
        }
        vec3 wo, wi, f;
        vec3_negate(wo, ray.d);
        vec2 sample = { rng(dim0 + 4), rng(dim0 + 5) };
        float pdf = BSDF_sample_f(f, wi, &sampled_flags, wo, sample, &sr);
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
    if (d->has_env_light && d->env_intensity > 0.0f) {             /* buildscene.h:525-533 */
        Light l; l.kind = 0; l.obj_type = -1; l.shape = 0;
        vec3_copy(l.color, d->env_color); l.intensity = d->env_intensity;
        if (d->env_map) { l.color[0] = l.color[1] = l.color[2] = 1.0f; }   /* EnvLight.color is never set for a map (buildscene.h:479-483): WHITE, shared convention */
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
        if (type < 0 || type > CRT_SMOOTH_TRIANGLE) return -1;
        /* EMISSIVE on FLAT / SMOOTH triangles: a MeshLight in the reference, with zero power (preprocessLights has no MESHLIGHT case)
         * and not sampleable by estimateDirect (trace.h:311-313): emits when found, never chosen */
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

    /* image rows [row_begin, row_end) (0 / 0 = all): color_buffer holds that band, pixel indices stay those of the frame */
    int row_begin = d->row_begin, row_end = d->row_end;
    if (row_begin == 0 && row_end == 0) row_end = film->frame_res_height;
    if (row_begin < 0 || row_end <= row_begin || row_end > film->frame_res_height) return -1;
    const uint32_t pix_offset = (uint32_t)row_begin * (uint32_t)film->frame_res_width;
    const uint32_t npix = (uint32_t)(row_end - row_begin) * (uint32_t)film->frame_res_width;
    auto work = [&](uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) {
            for (uint32_t p = first_sample; p < first_sample + num_samples; p++) {
                Rng rng = { S.seed, pix_offset + i, p };
                float isample[2] = { rng(0), rng(1) }, lsample[2] = { rng(2), rng(3) };
                crt_ray ray;
                oracle_primary_rays(cam, film, pix_offset + i, 1, isample, lsample, &ray);
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
/* any object kind: out = hit_point, normal, wo, dpdu (with the conventions applied where the reference writes nothing) */
void oracle_trace_shade_rec_any(const crt_scene_input* in, const crt_hit* hit, const crt_ray* ray, float* out)
{
    crt_render_desc d; memset(&d, 0, sizeof d);
    d.objects = in;
    Setup S; S.scene = 0; S.d = &d; S.seed = 0;
    triangleSetup(&S, in);
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    fillShadeRec(S, &sr, *ray, *hit);
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.wo, 12); memcpy(out + 9, sr.dpdu, 12);
}
void oracle_trace_disk(const float* in, float* out) { mapSampleToDisk(out, in); }
void oracle_trace_hemisphere(const float* in, float* out) { mapSampleToHemisphere(out, in); }
void oracle_trace_local_basis(const float* a, const float* normal, float* out) { getVec3InLocalBasis(out, a, normal); }
/* One entry for every BSDF piece test: the BSDF computeScatteringFunc builds for `mat`, BSDF_sample_f for (wo, sample)
 * with rand_float standing in for the lobe's rand(), and BSDF_f for (wi_eval, wo) as estimateDirect calls it (world
 * vectors, SPECULAR | GLOSSY excluded). */
float oracle_trace_material_bsdf(const crt_material* mat, float rand_float, const float* basis, const float* wo, const float* sample,
                                 const float* wi_eval, float* f_out, float* wi_out, int* flags_out, float* f_eval)
{
    ShadeRec sr; memset(&sr, 0, sizeof sr);
    vec3_copy(sr.b_tangent, basis); vec3_copy(sr.b_binormal, basis + 3); vec3_copy(sr.b_normal, basis + 6);
    computeScatteringFunc(&sr, mat, rand_float);
    vec3 wi = {0.0f, 0.0f, 0.0f};
    int flags = 0;
    vec3_copy(f_out, BLACK);
    float pdf = BSDF_sample_f(f_out, wi, &flags, wo, sample, &sr);
    vec3_copy(wi_out, wi);
    *flags_out = flags;
    BSDF_f(f_eval, wi_eval, wo, &sr, BSDF_SPECULAR | BSDF_GLOSSY);
    return pdf;
}
float oracle_trace_lambertian(const float* cd, const float* basis, const float* wo, const float* sample,
                              float* f_out, float* wi_out, float* f_eval)
{
    crt_material m; memset(&m, 0, sizeof m); m.type = CRT_MAT_MATTE; vec3_copy(m.color, cd);
    int flags = 0;
    return oracle_trace_material_bsdf(&m, 0.0f, basis, wo, sample, wo, f_out, wi_out, &flags, f_eval);
}
float oracle_trace_mirror(const float* cr, const float* basis, const float* wo, float* f_out, float* wi_out, int* flags)
{
    crt_material m; memset(&m, 0, sizeof m); m.type = CRT_MAT_MIRROR; vec3_copy(m.color, cr);
    const vec2 sample = {0.5f, 0.5f};
    vec3 fe;
    return oracle_trace_material_bsdf(&m, 0.0f, basis, wo, sample, wo, f_out, wi_out, flags, fe);
}
/* Oren-Nayar: BSDF_sample_f (tangent space) and BSDF_f evaluated for (wi_eval, wo) as given (the reference gives it world vectors) */
float oracle_trace_orennayar(const float* r, float sigma, const float* basis, const float* wo, const float* sample,
                             const float* wi_eval, float* f_out, float* wi_out, float* f_eval)
{
    crt_material m; memset(&m, 0, sizeof m); m.type = CRT_MAT_MATTE; vec3_copy(m.color, r); m.sigma = sigma;
    int flags = 0;
    return oracle_trace_material_bsdf(&m, 0.0f, basis, wo, sample, wi_eval, f_out, wi_out, &flags, f_eval);
}
float oracle_trace_transparent(float ior_in, float ior_out, float rand_float, const float* basis, const float* wo,
                               float* f_out, float* wi_out, int* flags)
{
    crt_material m; memset(&m, 0, sizeof m); m.type = CRT_MAT_TRANSPARENT; m.ior_in = ior_in; m.ior_out = ior_out;
    const vec2 sample = {0.5f, 0.5f};
    vec3 fe;
    return oracle_trace_material_bsdf(&m, rand_float, basis, wo, sample, wo, f_out, wi_out, flags, fe);
}
/* Beckmann pieces: out = { D(wh), G(wo, wi), Lambda(wo), sample_wh(wo, sample)[3] } */
void oracle_trace_beckmann(float alphax, float alphay, const float* wo, const float* wi, const float* wh, const float* sample, float* out)
{
    MicrofacetDistribution d = { alphax, alphay };
    out[0] = MicrofacetDistribution_D(wh, &d);
    out[1] = MicrofacetDistribution_G(wo, wi, &d);
    out[2] = MicrofacetDistribution_Lambda(wo, &d);
    MicrofacetDistribution_sample_wh(out + 3, wo, sample, &d);
}
void oracle_trace_fresnel_conductor(const float* normal, const float* wo, const float* etaT, const float* etaI, const float* k, float* out)
{
    calcFresnelConductor(out, normal, wo, etaT, etaI, k);
}
void oracle_trace_tex_color(const crt_texture* tex, const float* uv, float* out) { getTexColor(out, tex, uv); }
void oracle_trace_env_uv(const float* dir, float* uv) { envDirToUV(uv, dir); }
float oracle_trace_roughness_to_alpha(float roughness) { return BeckmannRoughnessToAlpha(roughness); }

} // extern "C"
