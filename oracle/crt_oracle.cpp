/*
 * crt_oracle.cpp -- CPU restatement of craytracer's ray traversal + intersection path.
 *
 * TEST INFRASTRUCTURE ONLY (see crt_oracle.h).  Scalar float32, one ray at a time,
 * written to follow the reference's arithmetic operation by operation so that results
 * are bit-identical to the reference compiled without FMA contraction
 * (build with -ffp-contract=off).  Citations are file:line under /root/reference.
 *
 * Pinning status: primitive intersectors, box tests, the three builders and the QBVH
 * traversals are checked against oracle/_ref (the reference's own code) and the golden
 * vectors in tests/golden/.  gridIntersectTest / BVHIntersectTest / dispatch live in
 * intersect.h, which cannot be compiled here (needs GLFW): restated from text only.
 */
#include "crt_oracle.h"

#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <thread>

namespace {

const float K_EPSILON = 0.000007f;      /* util/constants.h:45 */
const float TMAX = FLT_MAX;             /* util/constants.h:48 */
const float HUGEVALUE = 1.0E10;         /* util/constants.h:49 */
const float K_FLAT_AABB = 0.001f;       /* util/constants.h:50 */

struct Ray { float o[3]; float d[3]; };
struct AABB { float min[3]; float max[3]; };
struct HitInfo { int obj; int part; float beta; float gamma; };

/* util/util.h:9-37 */
inline float fminr(float a, float b) { return (a < b) ? a : b; }
inline float fmaxr(float a, float b) { return (a > b) ? a : b; }
inline float max3(float a, float b, float c) { return fmaxr(a, fmaxr(b, c)); }
inline float min3(float a, float b, float c) { return fminr(a, fminr(b, c)); }
inline float clampr(float x, float mn, float mx) { return (x < mn ? mn : (x > mx ? mx : x)); }

/* util/vec.h:154-157, :228-233 */
inline float dot3(const float* a, const float* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
inline float dot4(const float* a, const float* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3]; }
inline void cross3(float* r, const float* a, const float* b)
{
    r[0] = a[1]*b[2] - a[2]*b[1];
    r[1] = a[2]*b[0] - a[0]*b[2];
    r[2] = a[0]*b[1] - a[1]*b[0];
}
inline void sub3(float* r, const float* a, const float* b) { r[0]=a[0]-b[0]; r[1]=a[1]-b[1]; r[2]=a[2]-b[2]; }
inline void add3(float* r, const float* a, const float* b) { r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; }
inline void scale3(float* r, const float* a, float s) { r[0]=a[0]*s; r[1]=a[1]*s; r[2]=a[2]*s; }
inline float length3(const float* a) { return (float)sqrt(dot3(a, a)); }         /* vec.h:240 */
inline void normalize3(float* r, const float* a)                                   /* vec.h:206-215 */
{
    float length = (float)sqrt(dot3(a, a));
    if (length == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    scale3(r, a, 1.0f/length);
}
/* util/ray.cpp:4-9 */
inline void pointOnRay(float* r, const Ray& ray, float t)
{
    float disp[3];
    scale3(disp, ray.d, t);
    add3(r, ray.o, disp);
}

/* ------------------------------------------------------------------ box tests */

/* aabb.cpp:89-168.  Also returns t0/t1 and the hit flag for the pruned traversals. */
inline float rayAABB(const float* bmin, const float* bmax, const Ray& ray, float* t0_out = 0)
{
    float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
    float a = 1.0f/ray.d[0];
    if (a >= 0) { tx_min = (bmin[0] - ray.o[0]) * a; tx_max = (bmax[0] - ray.o[0]) * a; }
    else        { tx_min = (bmax[0] - ray.o[0]) * a; tx_max = (bmin[0] - ray.o[0]) * a; }
    float b = 1.0f/ray.d[1];
    if (b >= 0) { ty_min = (bmin[1] - ray.o[1]) * b; ty_max = (bmax[1] - ray.o[1]) * b; }
    else        { ty_min = (bmax[1] - ray.o[1]) * b; ty_max = (bmin[1] - ray.o[1]) * b; }
    float c = 1.0f/ray.d[2];
    if (c >= 0) { tz_min = (bmin[2] - ray.o[2]) * c; tz_max = (bmax[2] - ray.o[2]) * c; }
    else        { tz_min = (bmax[2] - ray.o[2]) * c; tz_max = (bmin[2] - ray.o[2]) * c; }
    float t0, t1;
    if (tx_min > ty_min) t0 = tx_min; else t0 = ty_min;
    if (tz_min > t0) t0 = tz_min;
    if (tx_max < ty_max) t1 = tx_max; else t1 = ty_max;
    if (tz_max < t1) t1 = tz_max;
    if (t0_out) *t0_out = t0;
    if (t0 < t1 && t1 > K_EPSILON) {
        if (t0 > K_EPSILON) return t0; else return t1;
    }
    return TMAX;
}

/* SSE semantics of _mm_max_ps / _mm_min_ps: second operand when unordered. */
inline float sse_max(float a, float b) { return a > b ? a : b; }
inline float sse_min(float a, float b) { return a < b ? a : b; }

/* bvh4.h:23-97.  less_than[i] is the C truth value of the stored mask word (:358):
 * all-ones (NaN) when t0 < eps, else 0. */
inline void rayAABB4(const float bbox[24], const Ray& ray, float t_out[4], bool less_than[4])
{
    float a = 1.0f / ray.d[0];
    float b = 1.0f / ray.d[1];
    float c = 1.0f / ray.d[2];
    for (int i = 0; i < 4; i++) {
        float min_x = bbox[0+i], min_y = bbox[4+i], min_z = bbox[8+i];
        float max_x = bbox[12+i], max_y = bbox[16+i], max_z = bbox[20+i];
        float tx_min, tx_max, ty_min, ty_max, tz_min, tz_max;
        if (ray.d[0] >= 0.0f) { tx_min = (min_x - ray.o[0]) * a; tx_max = (max_x - ray.o[0]) * a; }
        else                  { tx_min = (max_x - ray.o[0]) * a; tx_max = (min_x - ray.o[0]) * a; }
        if (ray.d[1] >= 0.0f) { ty_min = (min_y - ray.o[1]) * b; ty_max = (max_y - ray.o[1]) * b; }
        else                  { ty_min = (max_y - ray.o[1]) * b; ty_max = (min_y - ray.o[1]) * b; }
        if (ray.d[2] >= 0.0f) { tz_min = (min_z - ray.o[2]) * c; tz_max = (max_z - ray.o[2]) * c; }
        else                  { tz_min = (max_z - ray.o[2]) * c; tz_max = (min_z - ray.o[2]) * c; }
        float t0 = sse_max(sse_max(tx_min, ty_min), tz_min);
        float t1 = sse_min(sse_min(tx_max, ty_max), tz_max);
        bool hit = (t0 < t1) && (t1 > K_EPSILON);
        less_than[i] = (t0 < K_EPSILON);
        float t = (t0 > K_EPSILON) ? t0 : t1;
        t_out[i] = hit ? t : TMAX;
    }
}

/* aabb.cpp:171-181 */
inline bool insideAABB(const AABB& b, const float* p)
{
    for (int i = 0; i < 3; i++) if (p[i] < b.min[i] || p[i] > b.max[i]) return false;
    return true;
}
/* aabb.cpp:183-192 */
inline void addToAABB(AABB* r, const AABB* a)
{
    if (a->min[0] < r->min[0]) r->min[0] = a->min[0];
    if (a->min[1] < r->min[1]) r->min[1] = a->min[1];
    if (a->min[2] < r->min[2]) r->min[2] = a->min[2];
    if (a->max[0] > r->max[0]) r->max[0] = a->max[0];
    if (a->max[1] > r->max[1]) r->max[1] = a->max[1];
    if (a->max[2] > r->max[2]) r->max[2] = a->max[2];
}

/* ------------------------------------------------------------------ triangles */

/* triangle.cpp:14-77 */
inline float triIntersect(float* beta_out, float* gamma_out, const float* v0, const float* v1,
                          const float* v2, const Ray& ray)
{
    float a = v0[0] - v1[0], b = v0[0] - v2[0], c = ray.d[0], d = v0[0] - ray.o[0];
    float e = v0[1] - v1[1], f = v0[1] - v2[1], g = ray.d[1], h = v0[1] - ray.o[1];
    float i = v0[2] - v1[2], j = v0[2] - v2[2], k = ray.d[2], l = v0[2] - ray.o[2];

    float m = f*k - g*j, n = h*k - g*l, p = f*l - h*j;
    float q = g*i - e*k, s = e*j - f*i;

    float inv_denom = 1.0f / (a*m + b*q + c*s);

    float e1 = d*m - b*n - c*p;
    float beta = e1 * inv_denom;

    float r = e*l - h*i;
    float e2 = a*n + d*q + c*r;
    float gamma = e2 * inv_denom;
    *beta_out = beta;
    *gamma_out = gamma;

    if (beta < 0.0f) return TMAX;
    if (gamma < 0.0f) return TMAX;
    if (beta + gamma > 1.0f) return TMAX;

    float e3 = a*p - b*r + d*s;
    float t = e3 * inv_denom;
    if (t < K_EPSILON) return TMAX;
    return t;
}

/* ------------------------------------------------------------------ analytic shapes */

/* sphere.cpp:33-86 (closest) and :88-134 (shadow).  The first root of the closest-hit
 * variant uses atan2(z, x); every other site uses atan2(x, z) (sphere.cpp:58 vs :74,:110,:124). */
inline float sphereIntersect(const float* p, const Ray& ray, bool shadow)
{
    const float* center = p; float radius = p[3], min_theta = p[4], max_theta = p[5], sphi = p[6];
    float a = dot3(ray.d, ray.d);
    float otc[3], tmp[3];
    sub3(otc, ray.o, center);
    scale3(tmp, otc, 2);
    float b = dot3(tmp, ray.d);
    float c = dot3(otc, otc) - radius*radius;
    float disc = b*b - 4*a*c;
    if (disc < 0.0f) return TMAX;
    float e = (float)sqrt(disc);
    float denom = 2*a;
    float t = (-b - e)/denom;
    if (t > K_EPSILON) {
        float hp[3];
        pointOnRay(hp, ray, t);
        float phi = shadow ? (float)atan2(hp[0] - center[0], hp[2] - center[2])
                           : (float)atan2(hp[2] - center[2], hp[0] - center[0]);
        float theta = (float)acos(((hp[1] - center[1]) / radius));
        if (fabs(phi) <= sphi && theta >= min_theta && theta <= max_theta) return t;
    }
    t = (-b + e)/denom;
    if (t > K_EPSILON) {
        float hp[3];
        pointOnRay(hp, ray, t);
        float phi = (float)atan2(hp[0] - center[0], hp[2] - center[2]);
        float theta = (float)acos(((hp[1] - center[1]) / radius));
        if (fabs(phi) <= sphi && theta >= min_theta && theta <= max_theta) return t;
    }
    return TMAX;
}

/* plane.cpp:4-31 (closest and shadow compute the same t) */
inline float planeIntersect(const float* p, const Ray& ray)
{
    float disp[3];
    sub3(disp, p, ray.o);
    float t = dot3(disp, p + 3) / dot3(ray.d, p + 3);
    if (t > K_EPSILON) return t;
    return TMAX;
}

/* rect.cpp:3-52 */
inline float rectIntersect(const float* p, const Ray& ray)
{
    const float* point = p; const float* width = p + 3; const float* height = p + 6; const float* rnormal = p + 9;
    float disp[3], normal[3];
    sub3(disp, point, ray.o);
    if (dot3(disp, rnormal) < 0.0f) { normal[0] = -rnormal[0]; normal[1] = -rnormal[1]; normal[2] = -rnormal[2]; }
    else                            { normal[0] = rnormal[0];  normal[1] = rnormal[1];  normal[2] = rnormal[2]; }
    float t = dot3(disp, normal) / dot3(ray.d, normal);
    if (t > K_EPSILON) {
        float pt[3];
        pointOnRay(pt, ray, t);
        sub3(disp, pt, point);
        float u = dot3(disp, width) / dot3(width, width);
        if (u >= 0.0f && u <= 1.0f) {
            float v = dot3(disp, height) / dot3(height, height);
            if (v >= 0.0f && v <= 1.0f) return t;
        }
    }
    return TMAX;
}
/* rect.cpp:54-83: normal always negated, un-normalised dot-product bounds */
inline float rectShadow(const float* p, const Ray& ray)
{
    const float* point = p; const float* width = p + 3; const float* height = p + 6; const float* rnormal = p + 9;
    float disp[3], normal[3];
    sub3(disp, point, ray.o);
    normal[0] = -rnormal[0]; normal[1] = -rnormal[1]; normal[2] = -rnormal[2];
    float t = dot3(disp, normal) / dot3(ray.d, normal);
    if (t > K_EPSILON) {
        float pt[3];
        pointOnRay(pt, ray, t);
        sub3(disp, pt, point);
        float dp = dot3(disp, width);
        if (dp >= 0 && dp <= dot3(width, width)) {
            dp = dot3(disp, height);
            if (dp >= 0 && dp <= dot3(height, height)) return t;
        }
    }
    return TMAX;
}

/* disk.cpp:3-47 */
inline float diskIntersect(const float* p, const Ray& ray)
{
    const float* center = p; const float* normal = p + 3; float radius = p[6];
    float disp[3];
    sub3(disp, center, ray.o);
    float t = dot3(disp, normal) / dot3(ray.d, normal);
    if (t > K_EPSILON) {
        float hp[3];
        pointOnRay(hp, ray, t);
        sub3(disp, hp, center);
        if (length3(disp) <= radius) return t;
    }
    return TMAX;
}

/* generic.cpp:34-125 (closest and shadow accept the same roots) */
inline float cylinderIntersect(const float* p, const Ray& ray)
{
    float half_height = p[0], radius = p[1], cphi = p[2];
    float a = ray.d[0]*ray.d[0] + ray.d[2]*ray.d[2];
    float b = 2*(ray.o[0]*ray.d[0] + ray.o[2]*ray.d[2]);
    float c = ray.o[0]*ray.o[0] + ray.o[2]*ray.o[2] - radius*radius;
    float disc = b*b - 4*a*c;
    if (disc < 0.0f) return TMAX;
    float e = (float)sqrt(disc);
    float denom = 2*a;
    float t = (-b - e)/denom;
    float hp[3];
    if (t > K_EPSILON) {
        pointOnRay(hp, ray, t);
        if (fabs(hp[1]) <= half_height) {
            float phi = (float)atan2(hp[0], hp[2]);
            if (fabs(phi) <= cphi) return t;
        }
    }
    t = (-b + e)/denom;
    if (t > K_EPSILON) {
        pointOnRay(hp, ray, t);
        if (fabs(hp[1]) <= half_height) {
            float phi = (float)atan2(hp[0], hp[2]);
            if (fabs(phi) <= cphi) return t;
        }
    }
    return TMAX;
}

/* util/math.cpp:147-341 -- Schwarze's quadric/cubic/quartic solver as the reference carries it
 * (EQN_EPS 1e-90, float literals 1.0f/3 etc. kept because they change the double results). */
#define EQN_EPS 1e-90
#define IsZero(x) ((x) > -EQN_EPS && (x) < EQN_EPS)
const double PI_D = 3.14159265358979323846264338327950288;   /* util/vec.h:6 */

int solveQuadric(double c[3], double s[2])
{
    double p, q, D;
    p = c[1] / (2 * c[2]);
    q = c[0] / c[2];
    D = p * p - q;
    if (IsZero(D)) { s[0] = -p; return 1; }
    else if (D > 0) { double sqrt_D = (double)sqrt(D); s[0] = sqrt_D - p; s[1] = -sqrt_D - p; return 2; }
    return 0;
}
int solveCubic(double c[4], double s[3])
{
    int i, num;
    double sub, A, B, C, sq_A, p, q, cb_p, D;
    A = c[2] / c[3];
    B = c[1] / c[3];
    C = c[0] / c[3];
    sq_A = A * A;
    p = 1.0f/3 * (- 1.0f/3 * sq_A + B);
    q = 1.0f/2 * (2.0f/27 * A * sq_A - 1.0f/3 * A * B + C);
    cb_p = p * p * p;
    D = q * q + cb_p;
    if (IsZero(D)) {
        if (IsZero(q)) { s[0] = 0; num = 1; }
        else { double u = (double)cbrt(-q); s[0] = 2 * u; s[1] = -u; num = 2; }
    } else if (D < 0) {
        double phi = 1.0f/3 * (double)acos(-q / (double)sqrt(-cb_p));
        double t = 2.0f * (double)sqrt(-p);
        s[0] =   t * (double)cos(phi);
        s[1] = - t * (double)cos(phi + PI_D / 3);
        s[2] = - t * (double)cos(phi - PI_D / 3);
        num = 3;
    } else {
        double sqrt_D = (double)sqrt(D);
        double u = (double)cbrt(sqrt_D - q);
        double v = - (double)cbrt(sqrt_D + q);
        s[0] = u + v;
        num = 1;
    }
    sub = 1.0f/3 * A;
    for (i = 0; i < num; ++i) s[i] -= sub;
    return num;
}
int solveQuartic(double c[5], double s[4])
{
    double coeffs[4];
    double z, u, v, sub, A, B, C, D, sq_A, p, q, r;
    int i, num;
    A = c[3] / c[4];
    B = c[2] / c[4];
    C = c[1] / c[4];
    D = c[0] / c[4];
    sq_A = A * A;
    p = - 3.0f/8 * sq_A + B;
    q = 1.0f/8 * sq_A * A - 1.0f/2 * A * B + C;
    r = - 3.0f/256*sq_A*sq_A + 1.0f/16*sq_A*B - 1.0f/4*A*C + D;
    if (IsZero(r)) {
        coeffs[0] = q; coeffs[1] = p; coeffs[2] = 0; coeffs[3] = 1;
        num = solveCubic(coeffs, s);
        s[num++] = 0;
    } else {
        coeffs[0] = 1.0f/2 * r * p - 1.0f/8 * q * q;
        coeffs[1] = - r;
        coeffs[2] = - 1.0f/2 * p;
        coeffs[3] = 1;
        (void) solveCubic(coeffs, s);
        z = s[0];
        u = z * z - r;
        v = 2 * z - p;
        if (IsZero(u)) u = 0; else if (u > 0) u = sqrt(u); else return 0;
        if (IsZero(v)) v = 0; else if (v > 0) v = sqrt(v); else return 0;
        coeffs[0] = z - u;
        coeffs[1] = q < 0 ? -v : v;
        coeffs[2] = 1;
        num = solveQuadric(coeffs, s);
        coeffs[0] = z + u;
        coeffs[1] = q < 0 ? v : -v;
        coeffs[2] = 1;
        num += solveQuadric(coeffs, s + num);
    }
    sub = 1.0f/4 * A;
    for (i = 0; i < num; ++i) s[i] -= sub;
    return num;
}

/* generic.cpp:156-221: coefficients in double; returns t even when the phi window rejects (:220) */
inline float torusIntersect(const float* p, const Ray& ray)
{
    float swept_radius = p[0], tube_radius = p[1];
    double x1 = ray.o[0]; double y1 = ray.o[1]; double z1 = ray.o[2];
    double d1 = ray.d[0]; double d2 = ray.d[1]; double d3 = ray.d[2];
    double coeffs[5];
    double roots[4];
    double sum_d_sqrd = d1*d1 + d2*d2 + d3*d3;
    double e = x1*x1 + y1*y1 + z1*z1 - swept_radius*swept_radius - tube_radius*tube_radius;
    double f = x1*d1 + y1*d2 + z1*d3;
    double four_a_sqrd = 4.0f * swept_radius * swept_radius;
    coeffs[0] = e*e - four_a_sqrd*(tube_radius * tube_radius - y1*y1);
    coeffs[1] = 4.0f*f*e + 2.0f*four_a_sqrd*y1*d2;
    coeffs[2] = 2.0f*sum_d_sqrd*e + 4.0f*f*f + four_a_sqrd*d2*d2;
    coeffs[3] = 4.0f*sum_d_sqrd*f;
    coeffs[4] = sum_d_sqrd*sum_d_sqrd;
    int num_real_roots = solveQuartic(coeffs, roots);
    bool intersected = false;
    float t = FLT_MAX;
    if (num_real_roots == 0) return TMAX;
    for (int i = 0; i < num_real_roots; i++) {
        if (roots[i] > K_EPSILON) {
            intersected = true;
            if (roots[i] < t) t = (float)roots[i];
        }
    }
    if (!intersected) return TMAX;
    return (float)t;
}
/* generic.cpp:223-278: coefficients in float then widened; TMAX when the phi window rejects */
inline float torusShadow(const float* p, const Ray& ray)
{
    float swept_radius = p[0], tube_radius = p[1], tphi = p[2];
    float x1 = ray.o[0]; float y1 = ray.o[1]; float z1 = ray.o[2];
    float d1 = ray.d[0]; float d2 = ray.d[1]; float d3 = ray.d[2];
    double coeffs[5];
    double roots[4];
    float sum_d_sqrd = d1*d1 + d2*d2 + d3*d3;
    float e = x1*x1 + y1*y1 + z1*z1 - swept_radius*swept_radius - tube_radius*tube_radius;
    float f = x1*d1 + y1*d2 + z1*d3;
    float four_a_sqrd = 4.0f * swept_radius * swept_radius;
    coeffs[0] = e*e - four_a_sqrd*(tube_radius * tube_radius - y1*y1);
    coeffs[1] = 4.0f*f*e + 2.0f*four_a_sqrd*y1*d2;
    coeffs[2] = 2.0f*sum_d_sqrd*e + 4.0f*f*f + four_a_sqrd*d2*d2;
    coeffs[3] = 4.0f*sum_d_sqrd*f;
    coeffs[4] = sum_d_sqrd*sum_d_sqrd;
    int num_real_roots = solveQuartic(coeffs, roots);
    bool intersected = false;
    float t = FLT_MAX;
    if (num_real_roots == 0) return TMAX;
    for (int i = 0; i < num_real_roots; i++) {
        if (roots[i] > K_EPSILON) {
            intersected = true;
            if (roots[i] < t) t = (float)roots[i];
        }
    }
    if (!intersected) return TMAX;
    float hp[3];
    pointOnRay(hp, ray, t);
    float phi = (float)atan2(hp[0], hp[2]);
    if (fabs(phi) <= tphi) return t;
    return TMAX;
}

/* ------------------------------------------------------------------ scene */

struct OObj { int type; int ref; int id; };

struct ONode4 { float bbox[24]; int child[4]; signed char type[4]; unsigned char axis; };
struct ONode2 { AABB aabb; int left, right; int obj; bool leaf; };

} // namespace

struct oracle_scene {
    std::vector<OObj> objects;
    int num_non_grid = 0;
    int accel = 0;
    std::vector<float> tri;
    std::vector<crt_shape> shapes;
    std::vector<ONode4> nodes4;
    std::vector<ONode2> nodes2;
    /* grid (uniformgrid.h:10-17) */
    int nx = 0, ny = 0, nz = 0, num_cells = 0;
    float wx = 0, wy = 0, wz = 0;
    AABB gaabb;
    std::vector<std::vector<int> > cells;
    int sort_mode = 0;
};

namespace {

inline bool isTriType(int t) { return t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE; }

/* util/math.cpp:41-51 with mat4_mult_vec4 (mat.h:189-195): full 4-term dot products */
inline void transformRay(Ray* dst, const float* m, const Ray& src)
{
    float so[4] = { src.o[0], src.o[1], src.o[2], 1.0f };
    float sd[4] = { src.d[0], src.d[1], src.d[2], 0.0f };
    for (int i = 0; i < 3; i++) {
        dst->o[i] = dot4(m + 4*i, so);
        dst->d[i] = dot4(m + 4*i, sd);
    }
}

float shapeIntersect(const oracle_scene* s, int si, const Ray& ray, int* part);
float shapeShadow(const oracle_scene* s, int si, const Ray& ray);

/* shapes.cpp:4-48 / instanced.cpp:48-104,:155-216 for non-triangle shapes */
float shapeIntersect(const oracle_scene* s, int si, const Ray& ray, int* part)
{
    const crt_shape& sh = s->shapes[si];
    switch (sh.type) {
    case CRT_SPHERE: return sphereIntersect(sh.p, ray, false);
    case CRT_PLANE: return planeIntersect(sh.p, ray);
    case CRT_RECTANGLE: return rectIntersect(sh.p, ray);
    case CRT_AABOX: return rayAABB(sh.p, sh.p + 3, ray);           /* generic.cpp:311-404 */
    case CRT_GENERICOPENCYLINDER: return cylinderIntersect(sh.p, ray);
    case CRT_DISK: return diskIntersect(sh.p, ray);
    case CRT_GENERICTORUS: return torusIntersect(sh.p, ray);
    case CRT_INSTANCED: {
        Ray r2;
        transformRay(&r2, sh.inv_xform, ray);
        return shapeIntersect(s, sh.inner, r2, part);
    }
    case CRT_COMPOUND: {                                            /* instanced.cpp:155-216 */
        if (rayAABB(sh.p, sh.p + 3, ray) == TMAX) return TMAX;
        float min_t = TMAX, tmp_t = TMAX;
        for (int i = 0; i < sh.num_parts; i++) {
            int dummy = 0;
            if (s->shapes[sh.first_part + i].type != CRT_COMPOUND)   /* :202-203 nested compound: no-op */
                tmp_t = shapeIntersect(s, sh.first_part + i, ray, &dummy);
            if (tmp_t < min_t) { min_t = tmp_t; *part = i; }
        }
        return min_t;
    }
    }
    return TMAX;
}

/* shapes.cpp:50-93 / instanced.cpp:107-153,:218-273 */
float shapeShadow(const oracle_scene* s, int si, const Ray& ray)
{
    const crt_shape& sh = s->shapes[si];
    switch (sh.type) {
    case CRT_SPHERE: return sphereIntersect(sh.p, ray, true);
    case CRT_PLANE: return planeIntersect(sh.p, ray);
    case CRT_RECTANGLE: return rectShadow(sh.p, ray);
    case CRT_AABOX:                                                  /* generic.cpp:406-493 */
        if (!(sh.p[6] != 0.0f)) return TMAX;
        return rayAABB(sh.p, sh.p + 3, ray);
    case CRT_GENERICOPENCYLINDER: return cylinderIntersect(sh.p, ray);
    case CRT_DISK: return diskIntersect(sh.p, ray);
    case CRT_GENERICTORUS: return torusShadow(sh.p, ray);
    case CRT_INSTANCED: {
        Ray r2;
        transformRay(&r2, sh.inv_xform, ray);
        return shapeShadow(s, sh.inner, r2);
    }
    case CRT_COMPOUND: {
        if (rayAABB(sh.p, sh.p + 3, ray) == TMAX) return TMAX;
        float t = TMAX;
        for (int i = 0; i < sh.num_parts; i++) {
            if (s->shapes[sh.first_part + i].type != CRT_COMPOUND)
                t = shapeShadow(s, sh.first_part + i, ray);
            if (t < TMAX) return t;
        }
        return t;
    }
    }
    return TMAX;
}

/* rayIntersectObject, shapes.cpp:4-48.  hi->obj is set by the caller. */
inline float objIntersect(const oracle_scene* s, const OObj& o, const Ray& ray, HitInfo* hi)
{
    hi->part = 0; hi->beta = 0.0f; hi->gamma = 0.0f;
    if (isTriType(o.type)) {
        const float* v = &s->tri[(size_t)o.ref * 9];
        return triIntersect(&hi->beta, &hi->gamma, v, v + 3, v + 6, ray);
    }
    return shapeIntersect(s, o.ref, ray, &hi->part);
}
/* shadowRayIntersectObject, shapes.cpp:50-93 */
inline float objShadow(const oracle_scene* s, const OObj& o, const Ray& ray)
{
    if (isTriType(o.type)) {
        float b, g;
        const float* v = &s->tri[(size_t)o.ref * 9];
        return triIntersect(&b, &g, v, v + 3, v + 6, ray);
    }
    return shapeShadow(s, o.ref, ray);
}

/* ------------------------------------------------------------------ object AABBs (builders) */

/* getObjectAABB, shapes.cpp:141-314, for a shape-table entry */
void shapeAABB(const oracle_scene* s, int si, AABB* aabb)
{
    const crt_shape& sh = s->shapes[si];
    const float* p = sh.p;
    switch (sh.type) {
    case CRT_SPHERE:
        for (int i = 0; i < 3; i++) { aabb->min[i] = p[i] - p[3]; aabb->max[i] = p[i] + p[3]; }
        break;
    case CRT_RECTANGLE: {                                           /* shapes.cpp:158-186 */
        float pts[4][3], tmp[3];
        for (int i = 0; i < 3; i++) pts[0][i] = p[i];
        add3(pts[1], pts[0], p + 3);
        add3(pts[2], pts[0], p + 6);
        add3(tmp, p + 3, p + 6);
        add3(pts[3], pts[0], tmp);
        for (int j = 0; j < 3; j++) { aabb->min[j] = pts[0][j]; aabb->max[j] = pts[0][j]; }
        for (int i = 1; i < 4; i++)
            for (int j = 0; j < 3; j++) {
                if (pts[i][j] < aabb->min[j]) aabb->min[j] = pts[i][j];
                if (pts[i][j] > aabb->max[j]) aabb->max[j] = pts[i][j];
            }
        for (int i = 0; i < 3; i++) { aabb->min[i] -= K_EPSILON; aabb->max[i] += K_EPSILON; }
    } break;
    case CRT_AABOX: {                                               /* shapes.cpp:187-193 */
        float gap = 0.00001f;
        for (int i = 0; i < 3; i++) { aabb->min[i] = p[i] - gap; aabb->max[i] = p[3+i] + gap; }
    } break;
    case CRT_GENERICOPENCYLINDER:                                   /* shapes.cpp:251-260 */
        aabb->min[0] = -(p[1]); aabb->min[1] = -(p[0]); aabb->min[2] = -(p[1]);
        aabb->max[0] = p[1]; aabb->max[1] = p[0]; aabb->max[2] = p[1];
        break;
    case CRT_DISK:                                                  /* shapes.cpp:261-277 */
        for (int i = 0; i < 3; i++) { aabb->min[i] = p[i] - p[6]; aabb->max[i] = p[i] + p[6]; }
        for (int i = 0; i < 3; i++) { aabb->min[i] -= K_FLAT_AABB; aabb->max[i] += K_FLAT_AABB; }
        break;
    case CRT_GENERICTORUS:                                          /* generic.cpp:143-153 */
        aabb->min[0] = -(p[0] + p[1]); aabb->min[1] = -(p[1]); aabb->min[2] = aabb->min[0];
        aabb->max[0] = p[0] + p[1]; aabb->max[1] = p[1]; aabb->max[2] = aabb->max[0];
        break;
    case CRT_COMPOUND:
        for (int i = 0; i < 3; i++) { aabb->min[i] = p[i]; aabb->max[i] = p[3+i]; }
        break;
    case CRT_INSTANCED: {                                           /* shapes.cpp:284-304 + updateAABB :119-139 */
        shapeAABB(s, sh.inner, aabb);
        AABB tmp = *aabb;
        /* transform = affine_inverse(inv_transform) (mat.h:403-412); updateAABB then reads
         * transform[j][i], i.e. inv_transform[i][j], and transform[i][3] = -inv_transform[i][3]. */
        const float* m = sh.inv_xform;
        for (int i = 0; i < 3; i++) {
            aabb->min[i] = aabb->max[i] = -m[4*i + 3];
            for (int j = 0; j < 3; j++) {
                float e = m[4*i + j] * tmp.min[j];
                float f = m[4*i + j] * tmp.max[j];
                if (e < f) { aabb->min[i] += e; aabb->max[i] += f; }
                else       { aabb->min[i] += f; aabb->max[i] += e; }
            }
        }
    } break;
    default: break;   /* PLANE: leaves *aabb untouched (shapes.cpp:145-146) */
    }
}

void objAABB(const oracle_scene* s, const OObj& o, AABB* aabb)
{
    if (isTriType(o.type)) {                                        /* shapes.cpp:194-250 */
        const float* v = &s->tri[(size_t)o.ref * 9];
        for (int i = 0; i < 3; i++) { aabb->min[i] = v[i]; aabb->max[i] = v[i]; }
        for (int i = 0; i < 3; i++) {
            if (v[3+i] < aabb->min[i]) aabb->min[i] = v[3+i];
            if (v[3+i] > aabb->max[i]) aabb->max[i] = v[3+i];
            if (v[6+i] < aabb->min[i]) aabb->min[i] = v[6+i];
            if (v[6+i] > aabb->max[i]) aabb->max[i] = v[6+i];
        }
        for (int i = 0; i < 3; i++) { aabb->min[i] -= K_FLAT_AABB; aabb->max[i] += K_FLAT_AABB; }
        return;
    }
    shapeAABB(s, o.ref, aabb);
}

/* isGridObjType, shapes.cpp:95-117 (COMPOUND and PLANE fall through to false) */
bool isGridShape(const oracle_scene* s, int si)
{
    const crt_shape& sh = s->shapes[si];
    switch (sh.type) {
    case CRT_SPHERE: case CRT_RECTANGLE: case CRT_AABOX: case CRT_GENERICOPENCYLINDER:
    case CRT_DISK: case CRT_GENERICTORUS: return true;
    case CRT_INSTANCED: return isGridShape(s, sh.inner);
    }
    return false;
}
bool isGridObj(const oracle_scene* s, const OObj& o)
{
    if (isTriType(o.type)) return true;
    return isGridShape(s, o.ref);
}

/* calcBoundingVolume, bvh.h:28-39 */
AABB calcBoundingVolume(const oracle_scene* s, const OObj* objects, int num_obj)
{
    AABB r, tmp;
    for (int i = 0; i < 3; i++) { r.min[i] = FLT_MAX; r.max[i] = -HUGEVALUE; }
    for (int i = 0; i < num_obj; i++) {
        objAABB(s, objects[i], &tmp);
        addToAABB(&r, &tmp);
    }
    return r;
}

/* bvh.h:41-83 */
void exch(float* c, OObj* o, int a, int b) { float tc = c[a]; OObj to = o[a]; c[a] = c[b]; o[a] = o[b]; c[b] = tc; o[b] = to; }
int quicksortPartition(float* centroid, OObj* objects, int num_obj)
{
    int rand_index = rand() % num_obj;
    exch(centroid, objects, rand_index, num_obj-1);
    int i = -1;
    float pivot = centroid[num_obj-1];
    for (int j = 0; j < num_obj - 1; j++) {
        if (centroid[j] <= pivot) { i++; exch(centroid, objects, j, i); }
    }
    exch(centroid, objects, i+1, num_obj-1);
    return i+1;
}
void quicksort(float* centroid, OObj* objects, int num_obj)
{
    if (num_obj > 1) {
        int pivot_index = quicksortPartition(centroid, objects, num_obj);
        quicksort(centroid, objects, pivot_index);
        quicksort(centroid + pivot_index, objects + pivot_index, num_obj - pivot_index);
    }
}

/* partitionObjects, bvh.h:85-115 */
int partitionObjects(const oracle_scene* s, OObj* objects, int num_obj, int axis)
{
    std::vector<float> centroid(num_obj > 0 ? num_obj : 1);
    for (int i = 0; i < num_obj; ++i) {
        AABB tmp;
        objAABB(s, objects[i], &tmp);
        centroid[i] = (tmp.min[axis] + tmp.max[axis]) * 0.5f;
    }
    if (s->sort_mode == 1 && num_obj > 30) {
        quicksort(centroid.data(), objects, num_obj);
    } else if (num_obj <= 30) {
        for (int i = 1; i < num_obj; ++i)
            for (int j = i; j > 0 && centroid[j-1] > centroid[j]; --j) exch(centroid.data(), objects, j, j-1);
    } else {
        /* stable order for large ranges: what a stable insertion sort would give */
        std::vector<int> idx(num_obj);
        for (int i = 0; i < num_obj; i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return centroid[a] < centroid[b]; });
        std::vector<OObj> tmp(objects, objects + num_obj);
        for (int i = 0; i < num_obj; i++) objects[i] = tmp[idx[i]];
    }
    return num_obj / 2;
}

inline int longestAxis(const AABB& b)                                /* bvh4.h:143-157, bvh.h:125-138 */
{
    float x = b.max[0] - b.min[0], y = b.max[1] - b.min[1], z = b.max[2] - b.min[2];
    if (x > y && x > z) return 0;
    else if (y > z) return 1;
    return 2;
}

/* BVH4_build, bvh4.h:127-297.  `base` = index of objects[0] in the scene's full array. */
int buildBVH4(oracle_scene* s, OObj* objects, int num_obj, int base)
{
    int self = (int)s->nodes4.size();
    s->nodes4.push_back(ONode4());
    AABB tmp = calcBoundingVolume(s, objects, num_obj);
    int axis = longestAxis(tmp);
    int middle_split = partitionObjects(s, objects, num_obj, axis);

    tmp = calcBoundingVolume(s, objects, middle_split);
    axis = longestAxis(tmp);
    int left_split = partitionObjects(s, objects, middle_split, axis);

    tmp = calcBoundingVolume(s, objects + left_split, num_obj - middle_split);   /* :180 wrong range, kept */
    axis = longestAxis(tmp);
    unsigned char node_axis = (unsigned char)axis;                                /* :194-195 overwrite, not OR */
    int right_split = partitionObjects(s, objects + middle_split, num_obj - middle_split, axis);
    right_split += middle_split;

    int start[5] = { 0, left_split, middle_split, right_split, num_obj };
    ONode4 nd;
    memset(&nd, 0, sizeof nd);
    nd.axis = node_axis;
    for (int c = 0; c < 4; c++) {
        AABB b = calcBoundingVolume(s, objects + start[c], start[c+1] - start[c]);
        nd.bbox[c + 0] = b.min[0]; nd.bbox[c + 4] = b.min[1]; nd.bbox[c + 8] = b.min[2];
        nd.bbox[c + 12] = b.max[0]; nd.bbox[c + 16] = b.max[1]; nd.bbox[c + 20] = b.max[2];
    }
    for (int c = 0; c < 4; c++) {
        int cnt = start[c+1] - start[c];
        if (cnt > 4) {
            nd.type[c] = -1;
            nd.child[c] = buildBVH4(s, objects + start[c], cnt, base + start[c]);
        } else {
            nd.type[c] = (signed char)cnt;
            nd.child[c] = base + start[c];
        }
    }
    s->nodes4[self] = nd;
    return self;
}

/* BVH_build, bvh.h:117-154 */
int buildBVH2(oracle_scene* s, OObj* objects, int num_obj, int base)
{
    int self = (int)s->nodes2.size();
    s->nodes2.push_back(ONode2());
    ONode2 nd;
    nd.aabb = calcBoundingVolume(s, objects, num_obj);
    int axis = longestAxis(nd.aabb);
    if (num_obj <= 1) {
        nd.leaf = true; nd.obj = base; nd.left = nd.right = -1;
    } else {
        nd.leaf = false; nd.obj = -1;
        int k = partitionObjects(s, objects, num_obj, axis);
        nd.left = buildBVH2(s, objects, k, base);
        nd.right = buildBVH2(s, objects + k, num_obj - k, base + k);
    }
    s->nodes2[self] = nd;
    return self;
}

/* testAABBPlane, shapes.cpp:416-431 */
int testAABBPlane(const AABB* aabb, const float* plane_normal, float plane_d)
{
    float span[3], half_span[3], center[3];
    sub3(span, aabb->max, aabb->min);
    scale3(half_span, span, 0.5f);
    add3(center, aabb->min, half_span);
    float r = half_span[0] * fabs(plane_normal[0]) + half_span[1] * fabs(plane_normal[1])
        + half_span[2] * fabs(plane_normal[2]);
    float sd = dot3(plane_normal, center) - plane_d;
    return fabs(sd) <= r;
}

/* testTriangleAABB, shapes.cpp:434-565 (13-axis SAT + plane test) */
int testTriangleAABB(const float* tv0, const float* tv1, const float* tv2, const AABB* aabb)
{
    float span[3], half_span[3], center[3];
    sub3(span, aabb->max, aabb->min);
    scale3(half_span, span, 0.5f);
    add3(center, half_span, aabb->min);
    float x_extent = half_span[0], y_extent = half_span[1], z_extent = half_span[2];
    float v0[3], v1[3], v2[3];
    sub3(v0, tv0, center); sub3(v1, tv1, center); sub3(v2, tv2, center);
    float u[3][3] = { {1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f} };
    float e[3][3];
    sub3(e[0], v1, v0); sub3(e[1], v2, v1); sub3(e[2], v0, v2);
    for (int ui = 0; ui < 3; ui++) {
        for (int ei = 0; ei < 3; ei++) {
            float a[3];
            cross3(a, u[ui], e[ei]);
            float p0 = dot3(v0, a), p1 = dot3(v1, a), p2 = dot3(v2, a);
            float r = x_extent * fabs(a[0]) + y_extent * fabs(a[1]) + z_extent * fabs(a[2]);
            if (ui == 0 && ei == 0) {                                    /* :505 uses p0,p2 only */
                if (fmaxr(-fmaxr(p0, p2), fminr(p0, p2)) > r) return 0;
            } else {
                if (fmaxr(-max3(p0, p1, p2), min3(p0, p1, p2)) > r) return 0;
            }
        }
    }
    if (max3(v0[0], v1[0], v2[0]) < -x_extent || min3(v0[0], v1[0], v2[0]) > x_extent) return 0;
    if (max3(v0[1], v1[1], v2[1]) < -y_extent || min3(v0[1], v1[1], v2[1]) > y_extent) return 0;
    if (max3(v0[2], v1[2], v2[2]) < -z_extent || min3(v0[2], v1[2], v2[2]) > z_extent) return 0;
    float pn[3];
    cross3(pn, e[0], e[1]);
    normalize3(pn, pn);
    float plane_d = dot3(pn, tv0);
    return testAABBPlane(aabb, pn, plane_d);
}

/* UniformGrid_create, uniformgrid.h:54-154 (multiplier 3, buildscene.h:705) */
void buildGrid(oracle_scene* s)
{
    const int multiplier = 3;
    int num_obj = (int)s->objects.size();
    std::vector<AABB> aabb_array;
    for (int i = 0; i < num_obj; ++i)
        if (isGridObj(s, s->objects[i])) { AABB a; objAABB(s, s->objects[i], &a); aabb_array.push_back(a); }
    int num_aabb = (int)aabb_array.size();
    AABB r;
    for (int i = 0; i < 3; i++) { r.min[i] = FLT_MAX; r.max[i] = -HUGEVALUE; }
    for (int i = 0; i < num_aabb; ++i) addToAABB(&r, &aabb_array[i]);
    s->gaabb = r;
    s->wx = r.max[0] - r.min[0];
    s->wy = r.max[1] - r.min[1];
    s->wz = r.max[2] - r.min[2];
    float sz = powf(s->wx * s->wy * s->wz / num_obj, 0.33333f);
    s->nx = (int)(s->wx * multiplier / sz + 1);
    s->ny = (int)(s->wy * multiplier / sz + 1);
    s->nz = (int)(s->wz * multiplier / sz + 1);
    s->num_cells = s->nx * s->ny * s->nz;
    int nx = s->nx, ny = s->ny, nz = s->nz;
    float wx = s->wx, wy = s->wy, wz = s->wz;
    float unit_x = wx / nx, unit_y = wy / ny, unit_z = wz / nz;
    s->cells.assign(s->num_cells, std::vector<int>());
    for (int i = 0; i < num_aabb; ++i) {
        const AABB& a = aabb_array[i];
        int min_ix = (int)clampr(((a.min[0] - r.min[0]) / wx) * nx, 0.0f, (float)nx - 1);
        int min_iy = (int)clampr(((a.min[1] - r.min[1]) / wy) * ny, 0.0f, (float)ny - 1);
        int min_iz = (int)clampr(((a.min[2] - r.min[2]) / wz) * nz, 0.0f, (float)nz - 1);
        int max_ix = (int)clampr(((a.max[0] - r.min[0]) / wx) * nx, 0.0f, (float)nx - 1);
        int max_iy = (int)clampr(((a.max[1] - r.min[1]) / wy) * ny, 0.0f, (float)ny - 1);
        int max_iz = (int)clampr(((a.max[2] - r.min[2]) / wz) * nz, 0.0f, (float)nz - 1);
        const OObj& obj = s->objects[i + s->num_non_grid];
        for (int j = min_iy; j <= max_iy; ++j)
            for (int k = min_iz; k <= max_iz; ++k)
                for (int p = min_ix; p <= max_ix; ++p) {
                    if (isTriType(obj.type)) {
                        AABB cell;
                        float min_disp[3] = { p * unit_x, j * unit_y, k * unit_z };
                        add3(cell.min, r.min, min_disp);
                        float unit_span[3] = { unit_x, unit_y, unit_z };
                        add3(cell.max, cell.min, unit_span);
                        const float* v = &s->tri[(size_t)obj.ref * 9];
                        if (testTriangleAABB(v, v + 3, v + 6, &cell))
                            s->cells[j*nx*nz + k*nx + p].push_back(i + s->num_non_grid);
                    } else {
                        s->cells[j*nx*nz + k*nx + p].push_back(i + s->num_non_grid);
                    }
                }
    }
}

/* ------------------------------------------------------------------ traversal */

struct Counters { unsigned long long nodes = 0, prims = 0; };

/* BVH4IntersectTest, bvh4.h:299-389: axis decode bug (:308-310) gives axis0 = axis1 = 0 */
inline void childOrder(const ONode4& nd, const Ray& ray, int indices[4])
{
    unsigned char axis2 = nd.axis & 0x03;
    unsigned char axis1 = (nd.axis & 0x0C) >> 4;
    unsigned char axis0 = (nd.axis & 0x30) >> 8;
    if (ray.d[axis0] > 0.0f) {
        if (ray.d[axis1] > 0.0f) { indices[0] = 0; indices[1] = 1; } else { indices[0] = 1; indices[1] = 0; }
        if (ray.d[axis2] > 0.0f) { indices[2] = 2; indices[3] = 3; } else { indices[2] = 3; indices[3] = 2; }
    } else {
        if (ray.d[axis2] > 0.0f) { indices[0] = 2; indices[1] = 3; } else { indices[0] = 3; indices[1] = 2; }
        if (ray.d[axis1] > 0.0f) { indices[2] = 0; indices[3] = 1; } else { indices[2] = 1; indices[3] = 0; }
    }
}

float bvh4Exact(const oracle_scene* s, int ni, const Ray& ray, HitInfo* sr, Counters* cn)
{
    const ONode4& nd = s->nodes4[ni];
    float bbox_result[4]; bool less_than[4]; int indices[4];
    cn->nodes++;
    rayAABB4(nd.bbox, ray, bbox_result, less_than);
    childOrder(nd, ray, indices);
    float min_t = TMAX;                                              /* local to this visit, :354 */
    for (int i = 0; i < 4; i++) {
        int index = indices[i];
        if (bbox_result[index] < min_t || (less_than[index] && bbox_result[index] < TMAX)) {
            float t;
            HitInfo tmp_sr;
            if (nd.type[index] == -1) {
                t = bvh4Exact(s, nd.child[index], ray, &tmp_sr, cn);
                if (t < min_t) { min_t = t; *sr = tmp_sr; }
            } else {
                for (int j = 0; j < nd.type[index]; j++) {
                    int oi = nd.child[index] + j;
                    cn->prims++;
                    t = objIntersect(s, s->objects[oi], ray, &tmp_sr);
                    if (t < min_t) { min_t = t; tmp_sr.obj = oi; *sr = tmp_sr; }
                }
            }
        }
    }
    return min_t;
}

/* Same order and tests, but min_t is the ray's global best (what the GPU kernels do). */
void bvh4Pruned(const oracle_scene* s, int ni, const Ray& ray, float* min_t, HitInfo* sr, Counters* cn)
{
    const ONode4& nd = s->nodes4[ni];
    float bbox_result[4]; bool less_than[4]; int indices[4];
    cn->nodes++;
    rayAABB4(nd.bbox, ray, bbox_result, less_than);
    childOrder(nd, ray, indices);
    for (int i = 0; i < 4; i++) {
        int index = indices[i];
        if (bbox_result[index] < *min_t || (less_than[index] && bbox_result[index] < TMAX)) {
            if (nd.type[index] == -1) {
                bvh4Pruned(s, nd.child[index], ray, min_t, sr, cn);
            } else {
                for (int j = 0; j < nd.type[index]; j++) {
                    int oi = nd.child[index] + j;
                    HitInfo tmp_sr;
                    cn->prims++;
                    float t = objIntersect(s, s->objects[oi], ray, &tmp_sr);
                    if (t < *min_t) { *min_t = t; tmp_sr.obj = oi; *sr = tmp_sr; }
                }
            }
        }
    }
}

/* Nearest-first variant of the pruned walk (children of a node visited by ascending box key,
 * origin-inside children first; ties between children keep the reference's order).  Same min t;
 * *tie is set when a second object reaches exactly the best t, in which case the caller falls
 * back to the reference order to name the winner. */
void bvh4Nearest(const oracle_scene* s, int ni, const Ray& ray, float* min_t, HitInfo* sr, bool* tie, Counters* cn)
{
    const ONode4& nd = s->nodes4[ni];
    float bbox_result[4]; bool less_than[4]; int indices[4];
    cn->nodes++;
    rayAABB4(nd.bbox, ray, bbox_result, less_than);
    childOrder(nd, ray, indices);
    float key[4]; int ord[4];
    for (int i = 0; i < 4; i++) {
        int index = indices[i];
        key[i] = (less_than[index] && bbox_result[index] < TMAX) ? -INFINITY : bbox_result[index];
        ord[i] = index;
    }
    for (int i = 1; i < 4; i++)                       /* stable insertion sort by key */
        for (int j = i; j > 0 && key[j-1] > key[j]; --j) { std::swap(key[j-1], key[j]); std::swap(ord[j-1], ord[j]); }
    for (int i = 0; i < 4; i++) {
        int index = ord[i];
        if (key[i] < *min_t) {
            if (nd.type[index] == -1) {
                bvh4Nearest(s, nd.child[index], ray, min_t, sr, tie, cn);
            } else {
                for (int j = 0; j < nd.type[index]; j++) {
                    int oi = nd.child[index] + j;
                    HitInfo tmp_sr;
                    cn->prims++;
                    float t = objIntersect(s, s->objects[oi], ray, &tmp_sr);
                    if (t < *min_t) { *min_t = t; tmp_sr.obj = oi; *sr = tmp_sr; *tie = false; }
                    else if (t == *min_t && t < TMAX) *tie = true;
                }
            }
        }
    }
}

/* BVH4ShadowIntersectTest, bvh4.h:392-478 */
float bvh4Shadow(const oracle_scene* s, int ni, const Ray& ray, float light_dist, Counters* cn)
{
    const ONode4& nd = s->nodes4[ni];
    float bbox_result[4]; bool less_than[4]; int indices[4];
    cn->nodes++;
    rayAABB4(nd.bbox, ray, bbox_result, less_than);
    childOrder(nd, ray, indices);
    for (int i = 0; i < 4; i++) {
        int index = indices[i];
        if (bbox_result[index] < light_dist || (less_than[index] && bbox_result[index] < TMAX)) {
            float t = TMAX;
            if (nd.type[index] == -1) {
                t = bvh4Shadow(s, nd.child[index], ray, light_dist, cn);
                if (t < light_dist) return t;
            } else {
                for (int j = 0; j < nd.type[index]; j++) {
                    cn->prims++;
                    t = objShadow(s, s->objects[nd.child[index] + j], ray);
                    if (t < light_dist) return t;
                }
            }
        }
    }
    return TMAX;
}

/* BVHIntersectTest, intersect.h:310-342 */
float bvh2Exact(const oracle_scene* s, int ni, const Ray& ray, HitInfo* sr, Counters* cn)
{
    const ONode2& nd = s->nodes2[ni];
    if (nd.leaf) {
        cn->prims++;
        float t = objIntersect(s, s->objects[nd.obj], ray, sr);
        sr->obj = nd.obj;
        return t;
    }
    float t1 = TMAX, t2 = TMAX;
    HitInfo sr1, sr2;
    cn->nodes += 2;
    if (rayAABB(s->nodes2[nd.left].aabb.min, s->nodes2[nd.left].aabb.max, ray) < TMAX)
        t1 = bvh2Exact(s, nd.left, ray, &sr1, cn);
    if (rayAABB(s->nodes2[nd.right].aabb.min, s->nodes2[nd.right].aabb.max, ray) < TMAX)
        t2 = bvh2Exact(s, nd.right, ray, &sr2, cn);
    if (t1 < t2) { *sr = sr1; return t1; }
    if (t2 < TMAX) *sr = sr2;
    return t2;
}

/* Best-t pruned: left then right; a later equal-t hit replaces (right subtree wins ties,
 * intersect.h:327-338); a box is skipped only when its entry t0 lies beyond the best hit. */
void bvh2Pruned(const oracle_scene* s, int ni, const Ray& ray, float* min_t, HitInfo* sr, Counters* cn)
{
    const ONode2& nd = s->nodes2[ni];
    if (nd.leaf) {
        HitInfo tmp;
        cn->prims++;
        float t = objIntersect(s, s->objects[nd.obj], ray, &tmp);
        if (t <= *min_t && t < TMAX) { *min_t = t; tmp.obj = nd.obj; *sr = tmp; }
        return;
    }
    int kids[2] = { nd.left, nd.right };
    for (int k = 0; k < 2; k++) {
        float t0;
        cn->nodes++;
        float bt = rayAABB(s->nodes2[kids[k]].aabb.min, s->nodes2[kids[k]].aabb.max, ray, &t0);
        if (bt < TMAX && !(t0 > K_EPSILON && t0 > *min_t)) bvh2Pruned(s, kids[k], ray, min_t, sr, cn);
    }
}

/* BVHShadowIntersectTest, intersect.h:344-362 (no early-out, light_dist unused) */
float bvh2Shadow(const oracle_scene* s, int ni, const Ray& ray, Counters* cn)
{
    const ONode2& nd = s->nodes2[ni];
    if (nd.leaf) { cn->prims++; return objShadow(s, s->objects[nd.obj], ray); }
    float t1 = TMAX, t2 = TMAX;
    cn->nodes += 2;
    if (rayAABB(s->nodes2[nd.left].aabb.min, s->nodes2[nd.left].aabb.max, ray) < TMAX)
        t1 = bvh2Shadow(s, nd.left, ray, cn);
    if (rayAABB(s->nodes2[nd.right].aabb.min, s->nodes2[nd.right].aabb.max, ray) < TMAX)
        t2 = bvh2Shadow(s, nd.right, ray, cn);
    return t1 < t2 ? t1 : t2;
}

/* Shared DDA set-up of gridIntersectTest / gridShadowIntersectTest (intersect.h:61-129, :196-262) */
struct GridWalk {
    bool hits_grid;
    int ix, iy, iz, ix_step, iy_step, iz_step, ix_stop, iy_stop, iz_stop;
    float tx_next, ty_next, tz_next, dtx, dty, dtz;
};
inline void gridSetup(const oracle_scene* s, const Ray& ray, GridWalk* w)
{
    const AABB& gb = s->gaabb;
    int nx = s->nx, ny = s->ny, nz = s->nz;
    w->hits_grid = false;
    w->ix = w->iy = w->iz = 0;
    if (insideAABB(gb, ray.o)) {
        w->ix = (int)clampr((ray.o[0] - gb.min[0]) / s->wx * nx, 0.0f, (float)nx - 1);
        w->iy = (int)clampr((ray.o[1] - gb.min[1]) / s->wy * ny, 0.0f, (float)ny - 1);
        w->iz = (int)clampr((ray.o[2] - gb.min[2]) / s->wz * nz, 0.0f, (float)nz - 1);
        w->hits_grid = true;
    } else {
        float t0 = rayAABB(gb.min, gb.max, ray);
        if (t0 < TMAX) {
            float point[3];
            pointOnRay(point, ray, t0);
            w->ix = (int)clampr((point[0] - gb.min[0]) / s->wx * nx, 0.0f, (float)nx - 1);
            w->iy = (int)clampr((point[1] - gb.min[1]) / s->wy * ny, 0.0f, (float)ny - 1);
            w->iz = (int)clampr((point[2] - gb.min[2]) / s->wz * nz, 0.0f, (float)nz - 1);
            w->hits_grid = true;
        }
    }
    if (!w->hits_grid) return;
    float x_comp = ray.d[0], y_comp = ray.d[1], z_comp = ray.d[2];
    float x_min = x_comp > 0 ? gb.min[0] : gb.max[0];
    float y_min = y_comp > 0 ? gb.min[1] : gb.max[1];
    float z_min = z_comp > 0 ? gb.min[2] : gb.max[2];
    float x_max = x_comp > 0 ? gb.max[0] : gb.min[0];
    float y_max = y_comp > 0 ? gb.max[1] : gb.min[1];
    float z_max = z_comp > 0 ? gb.max[2] : gb.min[2];
    float tx_min = (x_min - ray.o[0]) / x_comp;
    float ty_min = (y_min - ray.o[1]) / y_comp;
    float tz_min = (z_min - ray.o[2]) / z_comp;
    float tx_max = (x_max - ray.o[0]) / x_comp;
    float ty_max = (y_max - ray.o[1]) / y_comp;
    float tz_max = (z_max - ray.o[2]) / z_comp;
    w->dtx = (tx_max - tx_min) / nx;
    w->dty = (ty_max - ty_min) / ny;
    w->dtz = (tz_max - tz_min) / nz;
    w->ix_step = x_comp > 0 ? 1 : -1;
    w->iy_step = y_comp > 0 ? 1 : -1;
    w->iz_step = z_comp > 0 ? 1 : -1;
    w->ix_stop = x_comp > 0 ? nx : -1;
    w->iy_stop = y_comp > 0 ? ny : -1;
    w->iz_stop = z_comp > 0 ? nz : -1;
    int tmp_ix = x_comp > 0 ? w->ix : nx - w->ix - 1;
    int tmp_iy = y_comp > 0 ? w->iy : ny - w->iy - 1;
    int tmp_iz = z_comp > 0 ? w->iz : nz - w->iz - 1;
    w->tx_next = tx_min + w->dtx * (tmp_ix + 1);
    w->ty_next = ty_min + w->dty * (tmp_iy + 1);
    w->tz_next = tz_min + w->dtz * (tmp_iz + 1);
}
inline void gridStep(GridWalk* w)                                    /* intersect.h:161-173 */
{
    if (w->tx_next < w->ty_next && w->tx_next < w->tz_next) { w->tx_next += w->dtx; w->ix += w->ix_step; }
    else if (w->ty_next < w->tz_next) { w->ty_next += w->dty; w->iy += w->iy_step; }
    else { w->tz_next += w->dtz; w->iz += w->iz_step; }
}

/* gridIntersectTest, intersect.h:61-194 */
float gridIntersect(const oracle_scene* s, const Ray& ray, HitInfo* sr, Counters* cn)
{
    GridWalk w;
    gridSetup(s, ray, &w);
    float t = TMAX;
    if (w.hits_grid) {
        while (w.ix != w.ix_stop && w.iy != w.iy_stop && w.iz != w.iz_stop) {
            int cell_index = w.iy*s->nz*s->nx + w.iz*s->nx + w.ix;
            cn->nodes++;
            float obj_t = TMAX;
            HitInfo obj_sr;
            const std::vector<int>& cell = s->cells[cell_index];
            for (size_t i = 0; i < cell.size(); i++) {
                HitInfo tmp_sr;
                cn->prims++;
                float tmp_t = objIntersect(s, s->objects[cell[i]], ray, &tmp_sr);
                if (tmp_t < obj_t) { obj_t = tmp_t; tmp_sr.obj = cell[i]; obj_sr = tmp_sr; }
            }
            if (fminr(obj_t, fminr(w.tx_next, fminr(w.ty_next, w.tz_next))) == obj_t) {
                if (obj_t < TMAX) { *sr = obj_sr; t = obj_t; break; }
            }
            gridStep(&w);
        }
    }
    float min_t = TMAX;
    HitInfo min_sr;
    for (int i = 0; i < s->num_non_grid; i++) {
        HitInfo tmp_sr;
        cn->prims++;
        float tmp_t = objIntersect(s, s->objects[i], ray, &tmp_sr);
        if (tmp_t < min_t) { min_t = tmp_t; tmp_sr.obj = i; min_sr = tmp_sr; }
    }
    if (min_t < t) { *sr = min_sr; t = min_t; }
    return t;
}

/* gridShadowIntersectTest, intersect.h:196-308 */
float gridShadow(const oracle_scene* s, const Ray& ray, float light_dist, Counters* cn)
{
    GridWalk w;
    gridSetup(s, ray, &w);
    if (w.hits_grid) {
        while (w.ix != w.ix_stop && w.iy != w.iy_stop && w.iz != w.iz_stop) {
            int cell_index = w.iy*s->nz*s->nx + w.iz*s->nx + w.ix;
            cn->nodes++;
            const std::vector<int>& cell = s->cells[cell_index];
            for (size_t i = 0; i < cell.size(); i++) {
                cn->prims++;
                float t = objShadow(s, s->objects[cell[i]], ray);
                if (t < (light_dist - K_EPSILON)) return t;
            }
            gridStep(&w);
        }
    }
    float t = TMAX;
    for (int i = 0; i < s->num_non_grid; i++) {
        cn->prims++;
        t = objShadow(s, s->objects[i], ray);
        if (t < (light_dist - K_EPSILON)) return t;
    }
    return t;
}

/* intersectTest, intersect.h:365-440 */
float intersectTest(const oracle_scene* s, const Ray& ray, HitInfo* sr, int mode, Counters* cn)
{
    float min_t = TMAX;
    sr->obj = -1; sr->part = -1; sr->beta = 0; sr->gamma = 0;
    if (s->accel == CRT_ACCEL_GRID) {
        return gridIntersect(s, ray, sr, cn);
    } else if (s->accel == CRT_ACCEL_BVH || s->accel == CRT_ACCEL_BVH4) {
        int n_accel = (int)s->objects.size() - s->num_non_grid;
        if (s->accel == CRT_ACCEL_BVH) {
            if (n_accel > 0) {
                cn->nodes++;
                if (rayAABB(s->nodes2[0].aabb.min, s->nodes2[0].aabb.max, ray) < TMAX) {
                    if (mode == CRT_TRAVERSE_PRUNED) bvh2Pruned(s, 0, ray, &min_t, sr, cn);
                    else { HitInfo h; h.obj = -1; h.part = -1; h.beta = h.gamma = 0;
                           min_t = bvh2Exact(s, 0, ray, &h, cn); if (min_t < TMAX) *sr = h; }
                }
            }
        } else if (n_accel > 0) {
            if (mode == 3) {                       /* experimental: nearest-first with tie fallback */
                bool tie = false;
                bvh4Nearest(s, 0, ray, &min_t, sr, &tie, cn);
                if (tie) { min_t = TMAX; bvh4Pruned(s, 0, ray, &min_t, sr, cn); }
            }
            else if (mode == CRT_TRAVERSE_PRUNED) bvh4Pruned(s, 0, ray, &min_t, sr, cn);
            else { HitInfo h; h.obj = -1; h.part = -1; h.beta = h.gamma = 0;
                   min_t = bvh4Exact(s, 0, ray, &h, cn); if (min_t < TMAX) *sr = h; }
        }
        for (int i = 0; i < s->num_non_grid; i++) {
            HitInfo tmp_sr;
            cn->prims++;
            float tmp_t = objIntersect(s, s->objects[i], ray, &tmp_sr);
            if (tmp_t < min_t) { min_t = tmp_t; tmp_sr.obj = i; *sr = tmp_sr; }
        }
        return min_t;
    }
    for (int i = 0; i < (int)s->objects.size(); i++) {
        HitInfo tmp_sr;
        cn->prims++;
        float tmp_t = objIntersect(s, s->objects[i], ray, &tmp_sr);
        if (tmp_t < min_t) { min_t = tmp_t; tmp_sr.obj = i; *sr = tmp_sr; }
    }
    return min_t;
}

/* shadowIntersectTest, intersect.h:443-545 */
float shadowIntersectTest(const oracle_scene* s, const Ray& ray, float light_dist, Counters* cn)
{
    if (s->accel == CRT_ACCEL_GRID) {
        return gridShadow(s, ray, light_dist, cn);
    } else if (s->accel == CRT_ACCEL_BVH || s->accel == CRT_ACCEL_BVH4) {
        int n_accel = (int)s->objects.size() - s->num_non_grid;
        float t = TMAX;
        if (s->accel == CRT_ACCEL_BVH) {
            if (n_accel > 0) {
                cn->nodes++;
                if (rayAABB(s->nodes2[0].aabb.min, s->nodes2[0].aabb.max, ray) < TMAX) t = bvh2Shadow(s, 0, ray, cn);
            }
        } else if (n_accel > 0) {
            t = bvh4Shadow(s, 0, ray, light_dist, cn);
        }
        if (t >= light_dist) {
            for (int i = 0; i < s->num_non_grid; i++) {
                cn->prims++;
                t = objShadow(s, s->objects[i], ray);
                if (t < light_dist) return t;
            }
        }
        return t;
    }
    float t = TMAX;
    for (int i = 0; i < (int)s->objects.size(); i++) {
        cn->prims++;
        t = objShadow(s, s->objects[i], ray);
        if (t < light_dist) return t;
    }
    return t;
}

template <class F>
void parallelFor(size_t n, int nthreads, F f)
{
    if (nthreads <= 1 || n < 1024) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    size_t chunk = (n + nthreads - 1) / nthreads;
    for (int k = 0; k < nthreads; k++) {
        size_t b = (size_t)k * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        th.emplace_back([=]() { f(b, e, k); });
    }
    for (auto& t : th) t.join();
}

} // namespace

/* ====================================================================== C API */

extern "C" {

oracle_scene* oracle_scene_create(const crt_scene_input* in, int accel, int sort_mode)
{
    if (!in || in->num_obj < 0) return 0;
    oracle_scene* s = new oracle_scene();
    s->accel = accel;
    s->sort_mode = sort_mode;
    s->tri.assign(in->tri_verts, in->tri_verts + (size_t)in->num_tris * 9);
    if (in->num_shapes > 0) s->shapes.assign(in->shapes, in->shapes + in->num_shapes);
    s->objects.resize(in->num_obj);
    for (int i = 0; i < in->num_obj; i++) {
        s->objects[i].type = in->obj_type ? in->obj_type[i] : (int)CRT_SMOOTH_TRIANGLE;
        s->objects[i].ref = in->obj_type ? in->obj_ref[i] : i;
        s->objects[i].id = i;
    }
    /* mvNonGridObjToStart, scenedata.h:368-382 */
    int j = 0;
    for (int i = 0; i < in->num_obj; ++i) {
        if (!isGridObj(s, s->objects[i])) { std::swap(s->objects[i], s->objects[j]); ++j; }
    }
    s->num_non_grid = j;
    int n_accel = in->num_obj - j;
    /* buildSceneAccel, buildscene.h:698-731 */
    if (accel == CRT_ACCEL_GRID) buildGrid(s);
    else if (accel == CRT_ACCEL_BVH && n_accel > 0) buildBVH2(s, s->objects.data() + j, n_accel, j);
    else if (accel == CRT_ACCEL_BVH4 && n_accel > 0) buildBVH4(s, s->objects.data() + j, n_accel, j);
    return s;
}

/* Adopt an already built scene (final order + flat accelerator), e.g. the product builder's
 * output, so that large scenes are traversed over the very same tree without rebuilding it. */
oracle_scene* oracle_scene_from_desc(const crt_scene_desc* d)
{
    if (!d) return 0;
    oracle_scene* s = new oracle_scene();
    s->accel = d->accel;
    s->num_non_grid = d->num_non_grid_obj;
    s->tri.assign(d->tri_verts, d->tri_verts + (size_t)d->num_tris * 9);
    if (d->num_shapes > 0) s->shapes.assign(d->shapes, d->shapes + d->num_shapes);
    s->objects.resize(d->num_obj);
    for (int i = 0; i < d->num_obj; i++) { s->objects[i].type = d->obj_type[i]; s->objects[i].ref = d->obj_ref[i]; s->objects[i].id = d->obj_id[i]; }
    s->nodes4.resize(d->num_nodes4);
    for (int i = 0; i < d->num_nodes4; i++) {
        memcpy(s->nodes4[i].bbox, d->nodes4[i].bbox, sizeof(float) * 24);
        for (int c = 0; c < 4; c++) { s->nodes4[i].child[c] = d->nodes4[i].child[c]; s->nodes4[i].type[c] = d->nodes4[i].type[c]; }
        s->nodes4[i].axis = d->nodes4[i].axis;
    }
    s->nodes2.resize(d->num_nodes2);
    for (int i = 0; i < d->num_nodes2; i++) {
        ONode2& n = s->nodes2[i];
        for (int k = 0; k < 3; k++) { n.aabb.min[k] = d->nodes2[i].bmin[k]; n.aabb.max[k] = d->nodes2[i].bmax[k]; }
        n.leaf = d->nodes2[i].left < 0;
        n.left = n.leaf ? -1 : d->nodes2[i].left;
        n.right = n.leaf ? -1 : d->nodes2[i].right;
        n.obj = n.leaf ? d->nodes2[i].right : -1;
    }
    if (d->accel == CRT_ACCEL_GRID) {
        s->nx = d->grid.nx; s->ny = d->grid.ny; s->nz = d->grid.nz; s->num_cells = d->grid.num_cells;
        s->wx = d->grid.wx; s->wy = d->grid.wy; s->wz = d->grid.wz;
        for (int k = 0; k < 3; k++) { s->gaabb.min[k] = d->grid.bmin[k]; s->gaabb.max[k] = d->grid.bmax[k]; }
        s->cells.assign(s->num_cells, std::vector<int>());
        for (int c = 0; c < s->num_cells; c++)
            s->cells[c].assign(d->cell_items + d->cell_start[c], d->cell_items + d->cell_start[c+1]);
    }
    return s;
}

void oracle_scene_destroy(oracle_scene* s) { delete s; }

int oracle_scene_order(const oracle_scene* s, int32_t* ids, int32_t* num_non_grid)
{
    if (ids) for (size_t i = 0; i < s->objects.size(); i++) ids[i] = s->objects[i].id;
    if (num_non_grid) *num_non_grid = s->num_non_grid;
    return (int)s->objects.size();
}

int oracle_scene_nodes4(const oracle_scene* s, crt_bvh4_node* out)
{
    if (out) for (size_t i = 0; i < s->nodes4.size(); i++) {
        memset(&out[i], 0, sizeof(crt_bvh4_node));
        memcpy(out[i].bbox, s->nodes4[i].bbox, sizeof(float) * 24);
        for (int c = 0; c < 4; c++) { out[i].child[c] = s->nodes4[i].child[c]; out[i].type[c] = s->nodes4[i].type[c]; }
        out[i].axis = s->nodes4[i].axis;
    }
    return (int)s->nodes4.size();
}

int oracle_scene_nodes2(const oracle_scene* s, crt_bvh2_node* out)
{
    if (out) for (size_t i = 0; i < s->nodes2.size(); i++) {
        const ONode2& n = s->nodes2[i];
        for (int k = 0; k < 3; k++) { out[i].bmin[k] = n.aabb.min[k]; out[i].bmax[k] = n.aabb.max[k]; }
        out[i].left = n.leaf ? -1 : n.left;
        out[i].right = n.leaf ? n.obj : n.right;
    }
    return (int)s->nodes2.size();
}

int oracle_scene_grid(const oracle_scene* s, crt_grid* g, int32_t* cell_start, int32_t* cell_items)
{
    if (g) {
        g->nx = s->nx; g->ny = s->ny; g->nz = s->nz; g->num_cells = s->num_cells;
        g->wx = s->wx; g->wy = s->wy; g->wz = s->wz;
        for (int k = 0; k < 3; k++) { g->bmin[k] = s->gaabb.min[k]; g->bmax[k] = s->gaabb.max[k]; }
    }
    int total = 0;
    for (int c = 0; c < s->num_cells; c++) {
        if (cell_start) cell_start[c] = total;
        for (size_t i = 0; i < s->cells[c].size(); i++) { if (cell_items) cell_items[total] = s->cells[c][i]; total++; }
    }
    if (cell_start) cell_start[s->num_cells] = total;
    return total;
}

void oracle_intersect(const oracle_scene* s, const crt_ray* rays, size_t n, crt_hit* hits,
                      int mode, crt_counters* counters, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> cn(nthreads);
    parallelFor(n, nthreads, [&](size_t b, size_t e, int k) {
        Counters local;
        for (size_t i = b; i < e; i++) {
            Ray ray;
            memcpy(&ray, &rays[i], sizeof(Ray));
            HitInfo h;
            float t = intersectTest(s, ray, &h, mode, &local);
            crt_hit out;
            out.t = t;
            if (t < TMAX) { out.obj = s->objects[h.obj].id; out.part = h.part; out.beta = h.beta; out.gamma = h.gamma; }
            else { out.obj = -1; out.part = -1; out.beta = 0; out.gamma = 0; }
            hits[i] = out;
        }
        cn[k] = local;
    });
    if (counters) {
        counters->rays_closest += n;
        for (auto& c : cn) { counters->nodes_visited += c.nodes; counters->prims_tested += c.prims; }
    }
}

void oracle_shadow(const oracle_scene* s, const crt_ray* rays, const float* light_dist, size_t n,
                   float* t_out, uint8_t* occluded_out, crt_counters* counters, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> cn(nthreads);
    parallelFor(n, nthreads, [&](size_t b, size_t e, int k) {
        Counters local;
        for (size_t i = b; i < e; i++) {
            Ray ray;
            memcpy(&ray, &rays[i], sizeof(Ray));
            float t = shadowIntersectTest(s, ray, light_dist[i], &local);
            if (t_out) t_out[i] = t;
            if (occluded_out) occluded_out[i] = (t < light_dist[i] - K_EPSILON) ? 1 : 0;   /* trace.h:348 */
        }
        cn[k] = local;
    });
    if (counters) {
        counters->rays_shadow += n;
        for (auto& c : cn) { counters->nodes_visited += c.nodes; counters->prims_tested += c.prims; }
    }
}

float oracle_intersect_object(const oracle_scene* s, int final_index, const crt_ray* ray, crt_hit* hit)
{
    Ray r; memcpy(&r, ray, sizeof r);
    HitInfo h;
    float t = objIntersect(s, s->objects[final_index], r, &h);
    if (hit) { hit->t = t; hit->obj = s->objects[final_index].id; hit->part = h.part; hit->beta = h.beta; hit->gamma = h.gamma; }
    return t;
}
float oracle_shadow_object(const oracle_scene* s, int final_index, const crt_ray* ray)
{
    Ray r; memcpy(&r, ray, sizeof r);
    return objShadow(s, s->objects[final_index], r);
}
float oracle_ray_aabb(const float bmin[3], const float bmax[3], const crt_ray* ray)
{
    Ray r; memcpy(&r, ray, sizeof r);
    return rayAABB(bmin, bmax, r);
}

} // extern "C"

/* ---- primary rays: camera.cpp of the reference, restated (test infrastructure) ---------- */
namespace {
inline void cNormalize(float* r, const float* a)                   /* util/vec.h:206-215 */
{
    float length = (float)sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]);
    if (length == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    float inv = 1.0f / length;
    r[0] = a[0] * inv; r[1] = a[1] * inv; r[2] = a[2] * inv;
}
inline void cCross(float* r, const float* a, const float* b)       /* util/vec.h:228-233 */
{
    r[0] = a[1]*b[2] - a[2]*b[1]; r[1] = a[2]*b[0] - a[0]*b[2]; r[2] = a[0]*b[1] - a[1]*b[0];
}
}
extern "C" {
void oracle_camera_init(crt_camera* c, int type, float focal_length, float lens_radius)
{   /* initPinholeCameraDefault / initThinLensCameraDefault, camera.cpp:21-51 (sample table not needed here) */
    for (int k = 0; k < 3; k++) { c->position[k] = 0.0f; c->x_axis[k] = (k == 0); c->y_axis[k] = (k == 1); c->z_axis[k] = (k == 2); }
    c->focal_pt_dist = 0.035f;
    c->focal_point[0] = 0.0f; c->focal_point[1] = 0.0f; c->focal_point[2] = c->focal_pt_dist;
    c->type = 0; c->focal_length = 0.0f; c->lens_radius = 0.0f;
    if (type == 1) { c->type = 1; c->focal_length = focal_length; c->lens_radius = lens_radius; }
}
void oracle_camera_look_at(crt_camera* c, const float* position, const float* look_point, const float* up)
{   /* cameraLookAt, camera.cpp:53-67 */
    float v[3], look_vec[3], x[3], disp[3];
    for (int k = 0; k < 3; k++) v[k] = look_point[k] - position[k];
    cNormalize(look_vec, v);
    for (int k = 0; k < 3; k++) c->z_axis[k] = -look_vec[k];
    cCross(x, c->z_axis, up);
    cNormalize(c->x_axis, x);
    for (int k = 0; k < 3; k++) c->x_axis[k] = -c->x_axis[k];
    cCross(c->y_axis, c->z_axis, c->x_axis);
    for (int k = 0; k < 3; k++) disp[k] = c->z_axis[k] * c->focal_pt_dist;
    for (int k = 0; k < 3; k++) c->focal_point[k] = position[k] + disp[k];
    for (int k = 0; k < 3; k++) c->position[k] = position[k];
}
void oracle_camera_set_focal_dist(crt_camera* c, float d)
{   /* setFocalDist, camera.cpp:70-76 */
    c->focal_pt_dist = d;
    for (int k = 0; k < 3; k++) c->focal_point[k] = c->position[k] + c->z_axis[k] * d;
}
void oracle_film_dimension(crt_film* f, float fov, int w, int h, const crt_camera* c)
{   /* calcFilmDimension, camera.cpp:143-148 */
    f->fov = fov; f->frame_res_width = w; f->frame_res_height = h;
    f->frame_length = 2.0f * (sin(f->fov/2.0f) * c->focal_pt_dist);
    f->frame_height = f->frame_length * (float)(f->frame_res_height)/(float)(f->frame_res_width);
    f->pixel_length = (float)(f->frame_length)/(float)(f->frame_res_width);
}
void oracle_primary_rays(const crt_camera* c, const crt_film* f, unsigned first_pixel, size_t n,
                         const float* image_samples, const float* lens_samples, crt_ray* rays)
{
    for (size_t i = 0; i < n; i++) {
        const int pixel_index = (int)(first_pixel + i);
        const float s0 = image_samples ? image_samples[2*i] : 0.5f, s1 = image_samples ? image_samples[2*i+1] : 0.5f;
        const float l0 = lens_samples ? lens_samples[2*i] : 0.5f, l1 = lens_samples ? lens_samples[2*i+1] : 0.5f;
        float ic[2];                                                /* calcImageCoord, camera.cpp:151-157 */
        ic[0] = -f->frame_length/2 + f->pixel_length * ((float)(pixel_index % f->frame_res_width) + s0);
        ic[1] = f->frame_height/2 - f->pixel_length * ((float)(pixel_index / f->frame_res_width) + s1);
        crt_ray* ray = rays + i;
        if (c->type == 0) {                                         /* calcRayPinhole, camera.cpp:79-91 */
            float t1[3], t2[3], loc[3], fp[3], d[3];
            for (int k = 0; k < 3; k++) t1[k] = c->x_axis[k] * ic[0];
            for (int k = 0; k < 3; k++) t2[k] = c->y_axis[k] * ic[1];
            for (int k = 0; k < 3; k++) loc[k] = t1[k] + t2[k];
            for (int k = 0; k < 3; k++) fp[k] = c->z_axis[k] * c->focal_pt_dist;
            for (int k = 0; k < 3; k++) d[k] = loc[k] - fp[k];
            cNormalize(ray->d, d);
            for (int k = 0; k < 3; k++) ray->o[k] = loc[k] + c->position[k];
        } else {                                                    /* calcRayThinLens, camera.cpp:94-128 */
            float phi = 2.0f * M_PI * l0;                           /* mapSampleToDisk, sampling.cpp:606-612 */
            float radius = sqrtf(l1);
            float disk[2] = { cosf(phi) * radius, sinf(phi) * radius };
            float pv[3] = { ic[0], ic[1], 0.0f };
            float fpp[3] = { pv[0] * (c->focal_length/c->focal_pt_dist), pv[1] * (c->focal_length/c->focal_pt_dist), -c->focal_length };
            float o[3] = { disk[0] * c->lens_radius, disk[1] * c->lens_radius, c->focal_pt_dist };
            float d[3] = { fpp[0] - o[0], fpp[1] - o[1], fpp[2] - o[2] }, dn[3];
            cNormalize(dn, d);
            float a[3], b[3], e[3];
            for (int k = 0; k < 3; k++) { a[k] = c->x_axis[k] * dn[0]; b[k] = c->y_axis[k] * dn[1]; e[k] = c->z_axis[k] * dn[2]; }
            for (int k = 0; k < 3; k++) { a[k] = a[k] + b[k]; ray->d[k] = a[k] + e[k]; }
            for (int k = 0; k < 3; k++) { a[k] = c->x_axis[k] * o[0]; b[k] = c->y_axis[k] * o[1]; e[k] = c->z_axis[k] * o[2]; }
            for (int k = 0; k < 3; k++) { a[k] = a[k] + b[k]; a[k] = a[k] + e[k]; ray->o[k] = a[k] + c->position[k]; }
        }
    }
}
} // extern "C"
