/*
 * crt_shaderec.h -- hit record -> ShadeRec, host and device.
 *
 * The reference's intersectors fill a 352-byte ShadeRec (util/shaderec.h:7-19) in their tails and copy it
 * up the recursion on every accepted hit.  The traversal kernels return five words per ray instead
 * ({t, obj, part, beta, gamma}); the functions here rebuild, for ONE hit, what the intersector of the hit
 * object left in the record -- field by field, operation by operation:
 *   rayIntersectTriangle / FlatTriangle / SmoothTriangle   shapes/triangle.cpp:154-258
 *   fillShadeRecSphere, rayIntersectSphere                  shapes/sphere.cpp:3-86
 *   rayIntersectPlane                                       shapes/plane.cpp:4-19
 *   rayIntersectRect                                        shapes/rect.cpp:3-52
 *   rayIntersectDisk                                        shapes/disk.cpp:3-29
 *   rayIntersectAABox, fillShadeRecGenericOpenCylinder,
 *   rayIntersectGenericTorus, computeGenericTorusNormal     shapes/generic.cpp:3-32,:127-142,:156-221,:311-404
 *   rayIntersectInstanced, rayIntersectCompound             shapes/instanced.cpp:48-104,:155-216
 * Everything follows from the hit's t (hit_point = o + t d, as getPointOnRay, util/ray.cpp:4-9); the only
 * things recomputed are which root of the sphere's quadratic t is (the two roots use different phi
 * formulas, sphere.cpp:58 vs :74) and which face of an AABox it lies on (generic.cpp:350-382).
 *
 * `written` says which fields the reference's intersector writes for that shape; the others are left as
 * the caller passed them (the reference leaves them holding whatever the record held before).  With
 * `conventions` set, the fields a path tracer reads but the reference never writes are filled by this
 * library's documented choices (DESIGN.md "a-13") and flagged CRT_SR_BY_CONVENTION.
 *
 * Compiled twice: as host C++ (crt_shaderec.cpp -> crt_fill_shade_rec, glibc's libm: the reference's bits)
 * and as device code (crt_render.hip; device libm: class T4 for sphere / cylinder / torus uv and seams).
 * float32, every multiply and add rounded separately (-ffp-contract=off).
 */
#pragma once
#include <math.h>
#include <stdint.h>
#include "../../include/crt_gpu.h"

#ifdef __HIP__
#define SR_HD __host__ __device__ inline
#else
#define SR_HD inline
#endif

#define SR_PI 3.14159265358979323846264338327950288          /* util/vec.h:6 (double) */

SR_HD float sr_dot(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }      /* util/vec.h:154 */
SR_HD void sr_set(float* r, float x, float y, float z) { r[0] = x; r[1] = y; r[2] = z; }
SR_HD void sr_copy(float* r, const float* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
SR_HD void sr_neg(float* r, const float* a) { r[0] = -a[0]; r[1] = -a[1]; r[2] = -a[2]; }
SR_HD void sr_sub(float* r, const float* a, const float* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
SR_HD void sr_scale(float* r, const float* a, float s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
SR_HD void sr_cross(float* r, const float* a, const float* b)                                               /* util/vec.h:229-234 */
{
    const float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
SR_HD float sr_length(const float* a) { return (float)sqrt((double)sr_dot(a, a)); }                        /* util/vec.h:241 */
SR_HD void sr_normalize(float* r, const float* a)                                                           /* util/vec.h:206-215 */
{
    const float len = (float)sqrt((double)sr_dot(a, a));
    if (len == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    sr_scale(r, a, 1.0f / len);
}
SR_HD void sr_point_on_ray(float* r, const float* o, const float* d, float t)                               /* util/ray.cpp:4-9 */
{
    r[0] = o[0] + d[0] * t; r[1] = o[1] + d[1] * t; r[2] = o[2] + d[2] * t;
}
/* calcTriangleNormal, shapes/triangle.cpp:4-12 */
SR_HD void sr_triangle_normal(float* r, const float* v)
{
    float e1[3], e2[3], c[3];
    sr_sub(e1, v + 3, v); sr_sub(e2, v + 6, v);
    sr_cross(c, e1, e2);
    sr_normalize(r, c);
}
/* interpTriangleVec3, triangle.cpp:206-216 */
SR_HD void sr_interp3(float* out, float beta, float gamma, const float* v0, const float* v1, const float* v2)
{
    const float w = 1.0f - beta - gamma;
    float a[3] = { v0[0] * w, v0[1] * w, v0[2] * w };
    a[0] = a[0] + v1[0] * beta; a[1] = a[1] + v1[1] * beta; a[2] = a[2] + v1[2] * beta;
    a[0] = a[0] + v2[0] * gamma; a[1] = a[1] + v2[1] * gamma; a[2] = a[2] + v2[2] * gamma;
    sr_normalize(out, a);
}
/* mat3_mult_vec3, util/mat.h:181-187, row-major */
SR_HD void sr_mat3_vec3(float* r, const float* m, const float* b)
{
    const float x = sr_dot(m, b), y = sr_dot(m + 3, b), z = sr_dot(m + 6, b);
    r[0] = x; r[1] = y; r[2] = z;
}
/* interpTexcoord (triangle.cpp:170-183) and the [0, 1] wrap loops of :195-198 / :240-243 */
SR_HD void sr_interp_uv(float* uv, float beta, float gamma, const float* t)
{
    const float w = 1.0f - beta - gamma;
    float u = t[0] * w, v = t[1] * w;
    u = u + t[2] * beta; v = v + t[3] * beta;
    u = u + t[4] * gamma; v = v + t[5] * gamma;
    uv[0] = u; uv[1] = v;
}
SR_HD void sr_wrap_uv(float* uv)
{
    for (int k = 0; k < 2; k++) {
        float x = uv[k];
        if (!(x == x) || x > 1.0e6f || x < -1.0e6f) continue;      /* the reference's loops would spin (NaN: fall through) */
        while (x > 1.0f) x -= 1.0f;
        while (x < 0.0f) x += 1.0f;
        uv[k] = x;
    }
}
/* JITTERED_UP, util/constants.h:27 */
SR_HD void sr_tangent_from_normal(float* dpdu, const float* normal)
{
    const float up[3] = { 0.0072f, 1.0f, 0.0034f };
    float c[3];
    sr_cross(c, up, normal);
    sr_normalize(dpdu, c);
}

/* ---- one simple shape, in the space its fields are given in --------------------------------------
 * `o`, `d`: the ray in that space; `t`: the hit distance.  Writes what rayIntersect<Shape> writes. */
SR_HD void sr_fill_simple(const crt_shape* sh, const float* o, const float* d, float t, int conventions, crt_shade_rec* sr)
{
    const float* p = sh->p;
    float hp[3];
    sr_point_on_ray(hp, o, d, t);
    switch (sh->type) {
    case CRT_SPHERE: {                                            /* rayIntersectSphere + fillShadeRecSphere */
        const float radius = p[3];
        /* which root is t?  (-b - e) / denom first, sphere.cpp:37-52 */
        float oc[3], tmp[3];
        const float a = sr_dot(d, d);
        sr_sub(oc, o, p);
        sr_scale(tmp, oc, 2);
        const float b = sr_dot(tmp, d);
        const float c = sr_dot(oc, oc) - radius * radius;
        const float disc = b * b - 4 * a * c;
        bool first = false;
        float phi = 0.0f, theta = 0.0f;
        if (!(disc < 0.0f)) {
            const float e = sqrtf(disc);
            const float denom = 2 * a;
            const float t1 = (-b - e) / denom;
            if (t1 == t && t1 > CRT_K_EPSILON) {
                float h1[3];
                sr_point_on_ray(h1, o, d, t1);
                phi = atan2f(h1[2] - p[2], h1[0] - p[0]);          /* sphere.cpp:58 */
                theta = acosf((h1[1] - p[1]) / radius);
                first = fabsf(phi) <= p[6] && theta >= p[4] && theta <= p[5];
            }
        }
        if (!first) {
            phi = atan2f(hp[0] - p[0], hp[2] - p[2]);              /* sphere.cpp:74 */
            theta = acosf((hp[1] - p[1]) / radius);
        }
        sr_copy(sr->hit_point, hp);
        float h2c[3];
        sr_sub(h2c, hp, p);
        sr_normalize(sr->normal, h2c);
        sr_neg(sr->wo, d);
        const double theta_max = SR_PI, theta_min = 0.0, phi_max = 2.0 * SR_PI;
        sr->uv[0] = (float)((double)phi / phi_max);
        sr->uv[1] = (float)(((double)theta - theta_min) / (theta_max - theta_min));
        const float y_radius = sqrtf(hp[0] * hp[0] + hp[2] * hp[2]);
        const float inv_y_radius = 1.0f / y_radius;
        const float cos_phi = hp[0] * inv_y_radius, sin_phi = hp[2] * inv_y_radius;
        sr_set(sr->dpdu, (float)(-phi_max * (double)(hp[2] - p[2])), 0.0f, (float)(phi_max * (double)(hp[0] - p[0])));
        float tv[3] = { hp[1] * cos_phi, -radius * sinf(theta), hp[1] * sin_phi };
        sr_scale(sr->dpdv, tv, (float)(theta_max * theta_max));
        sr_normalize(sr->dpdu, sr->dpdu);
        sr_normalize(sr->dpdv, sr->dpdv);
        sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO | CRT_SR_UV | CRT_SR_DPDU | CRT_SR_DPDV;
    } break;
    case CRT_PLANE:                                               /* plane.cpp:10-16 */
        sr_copy(sr->normal, p + 3);
        sr_copy(sr->hit_point, hp);
        sr_neg(sr->wo, d);
        sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO;
        break;
    case CRT_RECTANGLE: {                                         /* rect.cpp:5-46 */
        float disp[3], normal[3];
        sr_sub(disp, p, o);
        if (sr_dot(disp, p + 9) < 0.0f) sr_neg(normal, p + 9); else sr_copy(normal, p + 9);
        sr_sub(disp, hp, p);
        sr->uv[0] = sr_dot(disp, p + 3) / sr_dot(p + 3, p + 3);
        sr->uv[1] = sr_dot(disp, p + 6) / sr_dot(p + 6, p + 6);
        sr_normalize(sr->dpdu, p + 3);
        sr_normalize(sr->dpdv, p + 6);
        sr_copy(sr->normal, normal);
        sr_copy(sr->hit_point, hp);
        sr_neg(sr->wo, d);
        if (sr_dot(sr->wo, sr->normal) < 0.0f) { sr_neg(sr->normal, sr->normal); sr_neg(sr->dpdu, sr->dpdu); }
        sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO | CRT_SR_UV | CRT_SR_DPDU | CRT_SR_DPDV;
    } break;
    case CRT_AABOX: {                                             /* generic.cpp:311-404: the face t lies on */
        float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
        const float a = 1.0f / d[0];
        if (a >= 0) { tx_min = (p[0] - o[0]) * a; tx_max = (p[3] - o[0]) * a; } else { tx_min = (p[3] - o[0]) * a; tx_max = (p[0] - o[0]) * a; }
        const float b = 1.0f / d[1];
        if (b >= 0) { ty_min = (p[1] - o[1]) * b; ty_max = (p[4] - o[1]) * b; } else { ty_min = (p[4] - o[1]) * b; ty_max = (p[1] - o[1]) * b; }
        const float c = 1.0f / d[2];
        if (c >= 0) { tz_min = (p[2] - o[2]) * c; tz_max = (p[5] - o[2]) * c; } else { tz_min = (p[5] - o[2]) * c; tz_max = (p[2] - o[2]) * c; }
        float t0, t1;
        int face_in, face_out;
        if (tx_min > ty_min) { t0 = tx_min; face_in = (a >= 0.0f) ? 0 : 3; } else { t0 = ty_min; face_in = (b >= 0.0f) ? 1 : 4; }
        if (tz_min > t0) { t0 = tz_min; face_in = (c >= 0.0f) ? 2 : 5; }
        if (tx_max < ty_max) { t1 = tx_max; face_out = (a >= 0.0f) ? 3 : 0; } else { t1 = ty_max; face_out = (b >= 0.0f) ? 4 : 1; }
        if (tz_max < t1) { t1 = tz_max; face_out = (c >= 0.0f) ? 5 : 2; }
        const int face = (t0 > CRT_K_EPSILON) ? face_in : face_out;
        (void)t1;
        sr_set(sr->normal, face == 0 ? -1.0f : (face == 3 ? 1.0f : 0.0f), face == 1 ? -1.0f : (face == 4 ? 1.0f : 0.0f),
               face == 2 ? -1.0f : (face == 5 ? 1.0f : 0.0f));   /* getAABoxNormal, generic.cpp:280-309 */
        if (sr_dot(sr->normal, d) > 0.0f) sr_neg(sr->normal, sr->normal);
        sr_copy(sr->hit_point, hp);
        sr_neg(sr->wo, d);
        sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO;
    } break;
    case CRT_GENERICOPENCYLINDER: {                               /* fillShadeRecGenericOpenCylinder, generic.cpp:3-32 */
        const float half_height = p[0], radius = p[1], cphi = p[2];
        const int normal_type = (int)p[3];                        /* NormalType OPEN 0 / CONVEX 1 / CONCAVE 2 (generic.h:13-18) */
        sr_neg(sr->wo, d);
        sr_copy(sr->hit_point, hp);
        const float phi = atan2f(hp[2], hp[0]);
        sr->uv[0] = phi / cphi;
        sr->uv[1] = (hp[1] + half_height) / (2.0f * half_height);
        if (normal_type == 2) sr_set(sr->normal, (-hp[0]) / radius, 0.0f, (-hp[2]) / radius);
        else {
            sr_set(sr->normal, hp[0] / radius, 0.0f, hp[2] / radius);
            if (normal_type == 0 && sr_dot(sr->wo, sr->normal) < 0) sr_neg(sr->normal, sr->normal);
        }
        sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO | CRT_SR_UV;
    } break;
    case CRT_DISK: {                                              /* disk.cpp:13-24 */
        sr_neg(sr->wo, d);
        if (sr_dot(sr->wo, p + 3) < 0) {
            sr_neg(sr->wo, p + 3);                                /* as written there: wo, not normal */
            if (conventions) { sr_copy(sr->normal, p + 3); sr->written |= CRT_SR_BY_CONVENTION; }
        } else {
            sr_copy(sr->normal, p + 3);
            sr->written |= CRT_SR_NORMAL;
        }
        sr_copy(sr->hit_point, hp);
        sr->written |= CRT_SR_HIT_POINT | CRT_SR_WO;
    } break;
    case CRT_GENERICTORUS: {                                      /* generic.cpp:199-220 */
        const float swept = p[0], tube = p[1];
        const float phi = atan2f(hp[0], hp[2]);
        const bool accepted = fabsf(phi) <= p[2];
        if (accepted || conventions) {                            /* outside the phi window the reference returns t and writes nothing */
            sr_copy(sr->hit_point, hp);
            const float param_squared = swept * swept + tube * tube;                 /* computeGenericTorusNormal, :127-142 */
            const float x = hp[0], y = hp[1], z = hp[2];
            const float sum_squared = x * x + y * y + z * z;
            float n[3] = { 4.0f * x * (sum_squared - param_squared),
                           4.0f * y * (sum_squared - param_squared + 2.0f * swept * swept),
                           4.0f * z * (sum_squared - param_squared) };
            sr_normalize(sr->normal, n);
            sr_neg(sr->wo, d);
            if (sr_dot(sr->wo, sr->normal) < 0) sr_neg(sr->normal, sr->normal);
            sr->written |= accepted ? (CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO) : CRT_SR_BY_CONVENTION;
        }
    } break;
    default: break;
    }
}

/* ---- one hit on a shape-table object: simple | compound | instanced { simple | compound } -------------- */
SR_HD void sr_fill_shape_object(const crt_shape* shapes, int num_shapes, int ref, const crt_ray* ray, const crt_hit* hit, int conventions,
                                crt_shade_rec* sr)
{
    if (ref < 0 || ref >= num_shapes) return;
    const float* o = ray->o;
    const float* d = ray->d;
    const float t = hit->t;
    const crt_shape* sh = &shapes[ref];
    if (sh->type == CRT_INSTANCED) {                              /* rayIntersectInstanced, instanced.cpp:48-104 */
        const float* m = sh->inv_xform;
        float o2[3], d2[3];                                       /* transformRay, util/math.cpp:41-51 */
        o2[0] = m[0] * o[0] + m[1] * o[1] + m[2] * o[2] + m[3] * 1.0f;
        o2[1] = m[4] * o[0] + m[5] * o[1] + m[6] * o[2] + m[7] * 1.0f;
        o2[2] = m[8] * o[0] + m[9] * o[1] + m[10] * o[2] + m[11] * 1.0f;
        d2[0] = m[0] * d[0] + m[1] * d[1] + m[2] * d[2] + m[3] * 0.0f;
        d2[1] = m[4] * d[0] + m[5] * d[1] + m[6] * d[2] + m[7] * 0.0f;
        d2[2] = m[8] * d[0] + m[9] * d[1] + m[10] * d[2] + m[11] * 0.0f;
        if (sh->inner < 0 || sh->inner >= num_shapes) return;
        const crt_shape* inner = &shapes[sh->inner];
        if (inner->type == CRT_COMPOUND) {
            if (hit->part < 0 || hit->part >= inner->num_parts) return;
            inner = &shapes[inner->first_part + hit->part];
        }
        sr_fill_simple(inner, o2, d2, t, conventions, sr);
        /* normal = transpose(inv_transform) * (normal, 0), NOT renormalised; hit point on the source ray (:92-100) */
        const float nx = sr->normal[0], ny = sr->normal[1], nz = sr->normal[2];
        sr->normal[0] = m[0] * nx + m[4] * ny + m[8] * nz + m[12] * 0.0f;
        sr->normal[1] = m[1] * nx + m[5] * ny + m[9] * nz + m[13] * 0.0f;
        sr->normal[2] = m[2] * nx + m[6] * ny + m[10] * nz + m[14] * 0.0f;
        sr_point_on_ray(sr->hit_point, o, d, t);
        sr->written |= CRT_SR_HIT_POINT;
    } else if (sh->type == CRT_COMPOUND) {                        /* rayIntersectCompound, instanced.cpp:155-216: the winning part's record */
        if (hit->part < 0 || hit->part >= sh->num_parts) return;
        sr_fill_simple(&shapes[sh->first_part + hit->part], o, d, t, conventions, sr);
    } else {
        sr_fill_simple(sh, o, d, t, conventions, sr);
    }
    if (conventions && !(sr->written & CRT_SR_DPDU) && (sr->written & (CRT_SR_NORMAL | CRT_SR_BY_CONVENTION))) {
        sr_tangent_from_normal(sr->dpdu, sr->normal);             /* dpdu where the reference never writes one: normalize(JITTERED_UP x normal) */
        sr->written |= CRT_SR_BY_CONVENTION;
    }
}

/* ---- one hit ------------------------------------------------------------------------------------
 * in: the insertion-order object arrays the scene was built from; ms: mesh shading arrays or null. */
SR_HD void sr_fill(const crt_scene_input* in, const crt_mesh_shading* ms, const crt_ray* ray, const crt_hit* hit, int conventions,
                   crt_shade_rec* sr)
{
    sr->written = 0;
    if (hit->obj < 0 || hit->obj >= in->num_obj) return;
    const int type = in->obj_type ? in->obj_type[hit->obj] : CRT_SMOOTH_TRIANGLE;
    const int ref = in->obj_ref ? in->obj_ref[hit->obj] : hit->obj;
    const float* o = ray->o;
    const float* d = ray->d;
    const float t = hit->t;
    if (type == CRT_TRIANGLE || type == CRT_FLAT_TRIANGLE || type == CRT_SMOOTH_TRIANGLE) {
        const float* v = in->tri_verts + 9 * (size_t)ref;
        if (type == CRT_TRIANGLE) {                               /* triangle.cpp:159-166 */
            sr_triangle_normal(sr->normal, v);
            sr_point_on_ray(sr->hit_point, o, d, t);
            sr_neg(sr->wo, d);
            if (sr_dot(sr->wo, sr->normal) < 0.0f) sr_neg(sr->normal, sr->normal);
            sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO;
        } else if (type == CRT_FLAT_TRIANGLE) {                   /* triangle.cpp:190-203 */
            if (ms && ms->vertex_uv) {
                sr_interp_uv(sr->uv, hit->beta, hit->gamma, ms->vertex_uv + 6 * (size_t)ref);
                sr_wrap_uv(sr->uv);
                sr->written |= CRT_SR_UV;
            }
            if (ms && ms->flat_normal) sr_copy(sr->normal, ms->flat_normal + 3 * (size_t)ref);
            else sr_triangle_normal(sr->normal, v);
            sr_point_on_ray(sr->hit_point, o, d, t);
            sr_neg(sr->wo, d);
            sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO;
        } else if (ms && ms->vertex_normals && ms->vertex_tangents && ms->normal_mat_index && ms->normal_mats) {   /* triangle.cpp:226-256 */
            const float* vn = ms->vertex_normals + 9 * (size_t)ref;
            const float* vt = ms->vertex_tangents + 9 * (size_t)ref;
            const float* nm = ms->normal_mats + 9 * (size_t)ms->normal_mat_index[ref];
            float tmp[3], tangent[3], binormal[3];
            sr_interp3(tmp, hit->beta, hit->gamma, vn, vn + 3, vn + 6);
            sr_mat3_vec3(sr->normal, nm, tmp);
            sr_normalize(sr->normal, sr->normal);
            if (ms->vertex_uv) {
                sr_interp_uv(sr->uv, hit->beta, hit->gamma, ms->vertex_uv + 6 * (size_t)ref);
                sr_wrap_uv(sr->uv);
                sr->written |= CRT_SR_UV;
            }
            sr_interp3(tmp, hit->beta, hit->gamma, vt, vt + 3, vt + 6);
            sr_mat3_vec3(tangent, nm, tmp);
            sr_normalize(sr->dpdu, tangent);
            sr_cross(binormal, sr->normal, tangent);
            sr_normalize(sr->dpdv, binormal);
            sr_point_on_ray(sr->hit_point, o, d, t);
            sr_neg(sr->wo, d);
            sr->written |= CRT_SR_HIT_POINT | CRT_SR_NORMAL | CRT_SR_WO | CRT_SR_DPDU | CRT_SR_DPDV;
        }
        if (conventions && !(sr->written & CRT_SR_DPDU) && (sr->written & CRT_SR_NORMAL)) {
            float e1[3];                                          /* dpdu of TRIANGLE / FLAT_TRIANGLE: normalize(v1 - v0) */
            sr_sub(e1, v + 3, v);
            sr_normalize(sr->dpdu, e1);
            sr->written |= CRT_SR_BY_CONVENTION;
        }
        return;
    }
    sr_fill_shape_object(in->shapes, in->num_shapes, ref, ray, hit, conventions, sr);
}
