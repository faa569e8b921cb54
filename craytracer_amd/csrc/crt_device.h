/*
 * crt_device.h -- device-side data layout and primitive intersectors (gfx950).
 *
 * Everything here is float32 arithmetic in the reference's operation order; the file is
 * compiled with -ffp-contract=off so no multiply-add is fused (the reference is built for
 * x86-64 SSE2 without FMA, Makefile:12) and with HIP's default correctly rounded f32
 * divide / sqrt.  Comparisons are written as the reference writes them (select on `<` / `>`,
 * never fminf/fmaxf) because NaN and -0.0 operands must take the same branch as on the CPU.
 */
#pragma once
#ifndef CRT_OBJ_STRIDE
#define CRT_OBJ_STRIDE 3                     /* float4 per object record: 48 bytes */
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/crt_gpu.h"

#define CRT_DEV __device__ __forceinline__

/* ---- HBM layout -------------------------------------------------------------------
 * QBVH node, 128 B, 128-B aligned (one L2 line), 8 x float4:
 *   q0..q5 : minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4]      (BVHNode4.bbox, bvh4.h:105)
 *   q6     : child words.  inner: node index (>= 0).  leaf: 0x80000000 | count << 28 | first
 *            (count 0..4 objects starting at objects[first], final order)
 *   q7.x   : axis2 = node.axis & 3 (the only axis the reference's decode keeps, bvh4.h:308-310)
 * Object record, 48 B, 3 x float4, final objects[] order:
 *   r0 = v0.x v0.y v0.z v1.x   r1 = v1.y v1.z v2.x v2.y   r2 = v2.z | id | type | ref
 *   (id = insertion-order index reported to the caller; for non-triangles the vertex words
 *    are zero and ref indexes the shape table)
 * Binary BVH node, 32 B, 2 x float4: (bmin.xyz, left) (bmax.xyz, right); leaf: left = -1,
 *   right = object index.
 * Shape table: crt_shape as is (128 B each).
 */
struct DevScene {
    const float4* nodes4;
    const uint4* cnodes4;        /* 64-byte conservative copy of nodes4 (4 x uint4 per node), or a dummy */
    const uint4* wnodes8;        /* the tree once more, up to eight conservative boxes per 128-byte record (crt_wide8_host.h), or a dummy */
    const float4* objs;          /* CRT_OBJ_STRIDE float4 per object */
    const crt_shape* shapes;
    const float4* nodes2;
    const int* cell_start;
    const int* cell_items;
    const float4* cell_objs;           /* grid: the object records again, in cell order (record k belongs to cell_items[k]), or null */
    const unsigned char* cell_dist;    /* per cell: chessboard distance to the nearest occupied cell (0: occupied), crt_gpu.hip grid_distance_field */
    unsigned long long* counters;   /* [4]: rays_closest, rays_shadow, nodes, prims */
    crt_grid grid;
    int num_obj;
    int nng;                        /* num_non_grid_obj */
    int n_accel;                    /* num_obj - nng */
    int accel;
    const float4* bnodes2w;         /* binary BVH collapsed two levels at a time: 128-byte records of up to four boxes (crt_gpu.hip) */
    const uint4* bnodes2wc;         /* ... the same records once more as 64-byte conservative records (the QBVH's compressed-node layout), or null */
    unsigned root_word_w;           /* its root in child-word form */
    unsigned root_word;             /* QBVH: axis2(root) << 28 | 0;  BVH: child-word form of the root */
    const float4* bnodes2;          /* BVH: 64-byte inner-node records for k_bvh2 (see crt_kernels.h) */
    float root_bmin[3], root_bmax[3];
};

#define CRT_LEAF_BIT 0x80000000u
#define CRT_NODE_MASK 0x01FFFFFFu   /* inner child word: axis2 of the child << 28 | node index (< 2^25) */

struct DRay { float ox, oy, oz, dx, dy, dz; };
struct DHit { float t; int obj; int part; float beta; float gamma; };

CRT_DEV bool dev_is_tri(int t) { return t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE; }

/* min / max exactly as util/util.h:9-17 */
CRT_DEV float rmin(float a, float b) { return (a < b) ? a : b; }
CRT_DEV float rmax(float a, float b) { return (a > b) ? a : b; }
CRT_DEV float rclamp(float x, float lo, float hi) { return (x < lo ? lo : (x > hi ? hi : x)); }

/* calcTriangleIntersect, shapes/triangle.cpp:14-77 */
CRT_DEV float dev_tri(const float4 r0, const float4 r1, const float4 r2, const DRay& ray, float* beta_out, float* gamma_out)
{
    const float v0x = r0.x, v0y = r0.y, v0z = r0.z;
    const float v1x = r0.w, v1y = r1.x, v1z = r1.y;
    const float v2x = r1.z, v2y = r1.w, v2z = r2.x;
    float a = v0x - v1x, b = v0x - v2x, c = ray.dx, d = v0x - ray.ox;
    float e = v0y - v1y, f = v0y - v2y, g = ray.dy, h = v0y - ray.oy;
    float i = v0z - v1z, j = v0z - v2z, k = ray.dz, l = v0z - ray.oz;

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
    if (beta < 0.0f) return CRT_TMAX;
    if (gamma < 0.0f) return CRT_TMAX;
    if (beta + gamma > 1.0f) return CRT_TMAX;
    float e3 = a*p - b*r + d*s;
    float t = e3 * inv_denom;
    if (t < CRT_K_EPSILON) return CRT_TMAX;
    return t;
}

/* rayIntersectAABB, aabb.cpp:89-168 (branch on 1/d >= 0; comparison-select min/max) */
CRT_DEV float dev_ray_aabb(const float* bmin, const float* bmax, const DRay& ray, float* t0_out)
{
    float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
    float a = 1.0f / ray.dx;
    if (a >= 0) { tx_min = (bmin[0] - ray.ox) * a; tx_max = (bmax[0] - ray.ox) * a; }
    else        { tx_min = (bmax[0] - ray.ox) * a; tx_max = (bmin[0] - ray.ox) * a; }
    float b = 1.0f / ray.dy;
    if (b >= 0) { ty_min = (bmin[1] - ray.oy) * b; ty_max = (bmax[1] - ray.oy) * b; }
    else        { ty_min = (bmax[1] - ray.oy) * b; ty_max = (bmin[1] - ray.oy) * b; }
    float c = 1.0f / ray.dz;
    if (c >= 0) { tz_min = (bmin[2] - ray.oz) * c; tz_max = (bmax[2] - ray.oz) * c; }
    else        { tz_min = (bmax[2] - ray.oz) * c; tz_max = (bmin[2] - ray.oz) * c; }
    float t0, t1;
    if (tx_min > ty_min) t0 = tx_min; else t0 = ty_min;
    if (tz_min > t0) t0 = tz_min;
    if (tx_max < ty_max) t1 = tx_max; else t1 = ty_max;
    if (tz_max < t1) t1 = tz_max;
    *t0_out = t0;
    if (t0 < t1 && t1 > CRT_K_EPSILON) {
        if (t0 > CRT_K_EPSILON) return t0;
        return t1;
    }
    return CRT_TMAX;
}

/* the same with the three reciprocals taken once per ray (they do not depend on the box) */
CRT_DEV float dev_ray_aabb_inv(const float* bmin, const float* bmax, const DRay& ray, float a, float b, float c, float* t0_out)
{
    float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
    if (a >= 0) { tx_min = (bmin[0] - ray.ox) * a; tx_max = (bmax[0] - ray.ox) * a; }
    else        { tx_min = (bmax[0] - ray.ox) * a; tx_max = (bmin[0] - ray.ox) * a; }
    if (b >= 0) { ty_min = (bmin[1] - ray.oy) * b; ty_max = (bmax[1] - ray.oy) * b; }
    else        { ty_min = (bmax[1] - ray.oy) * b; ty_max = (bmin[1] - ray.oy) * b; }
    if (c >= 0) { tz_min = (bmin[2] - ray.oz) * c; tz_max = (bmax[2] - ray.oz) * c; }
    else        { tz_min = (bmax[2] - ray.oz) * c; tz_max = (bmin[2] - ray.oz) * c; }
    float t0, t1;
    if (tx_min > ty_min) t0 = tx_min; else t0 = ty_min;
    if (tz_min > t0) t0 = tz_min;
    if (tx_max < ty_max) t1 = tx_max; else t1 = ty_max;
    if (tz_max < t1) t1 = tz_max;
    *t0_out = t0;
    if (t0 < t1 && t1 > CRT_K_EPSILON) {
        if (t0 > CRT_K_EPSILON) return t0;
        return t1;
    }
    return CRT_TMAX;
}

CRT_DEV float dot3f(float ax, float ay, float az, float bx, float by, float bz) { return ax*bx + ay*by + az*bz; }

/* ---- analytic shapes (closest = false selects the shadow variants) ---- */

/* rayIntersectSphere / shadowRayIntersectSphere, shapes/sphere.cpp:33-134 */
CRT_DEV float dev_sphere(const float* p, const DRay& ray, bool shadow)
{
    float cx = p[0], cy = p[1], cz = p[2], radius = p[3], min_theta = p[4], max_theta = p[5], sphi = p[6];
    float a = dot3f(ray.dx, ray.dy, ray.dz, ray.dx, ray.dy, ray.dz);
    float ocx = ray.ox - cx, ocy = ray.oy - cy, ocz = ray.oz - cz;
    float tx = ocx * 2, ty = ocy * 2, tz = ocz * 2;
    float b = dot3f(tx, ty, tz, ray.dx, ray.dy, ray.dz);
    float c = dot3f(ocx, ocy, ocz, ocx, ocy, ocz) - radius*radius;
    float disc = b*b - 4*a*c;
    if (disc < 0.0f) return CRT_TMAX;
    float e = sqrtf(disc);
    float denom = 2*a;
    float t = (-b - e)/denom;
    if (t > CRT_K_EPSILON) {
        float hx = ray.ox + ray.dx * t, hy = ray.oy + ray.dy * t, hz = ray.oz + ray.dz * t;
        float phi = shadow ? atan2f(hx - cx, hz - cz) : atan2f(hz - cz, hx - cx);   /* sphere.cpp:58 vs :110 */
        float theta = acosf((hy - cy) / radius);
        if (fabsf(phi) <= sphi && theta >= min_theta && theta <= max_theta) return t;
    }
    t = (-b + e)/denom;
    if (t > CRT_K_EPSILON) {
        float hx = ray.ox + ray.dx * t, hy = ray.oy + ray.dy * t, hz = ray.oz + ray.dz * t;
        float phi = atan2f(hx - cx, hz - cz);
        float theta = acosf((hy - cy) / radius);
        if (fabsf(phi) <= sphi && theta >= min_theta && theta <= max_theta) return t;
    }
    return CRT_TMAX;
}

/* rayIntersectPlane, shapes/plane.cpp:4-31 */
CRT_DEV float dev_plane(const float* p, const DRay& ray)
{
    float dx = p[0] - ray.ox, dy = p[1] - ray.oy, dz = p[2] - ray.oz;
    float t = dot3f(dx, dy, dz, p[3], p[4], p[5]) / dot3f(ray.dx, ray.dy, ray.dz, p[3], p[4], p[5]);
    if (t > CRT_K_EPSILON) return t;
    return CRT_TMAX;
}

/* rayIntersectRect (shapes/rect.cpp:3-52) and shadowRayIntersectRect (:54-83) */
CRT_DEV float dev_rect(const float* p, const DRay& ray, bool shadow)
{
    float dx = p[0] - ray.ox, dy = p[1] - ray.oy, dz = p[2] - ray.oz;
    float nx = p[9], ny = p[10], nz = p[11];
    if (shadow || dot3f(dx, dy, dz, nx, ny, nz) < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    float t = dot3f(dx, dy, dz, nx, ny, nz) / dot3f(ray.dx, ray.dy, ray.dz, nx, ny, nz);
    if (t > CRT_K_EPSILON) {
        float px = ray.ox + ray.dx * t, py = ray.oy + ray.dy * t, pz = ray.oz + ray.dz * t;
        dx = px - p[0]; dy = py - p[1]; dz = pz - p[2];
        if (!shadow) {
            float u = dot3f(dx, dy, dz, p[3], p[4], p[5]) / dot3f(p[3], p[4], p[5], p[3], p[4], p[5]);
            if (u >= 0.0f && u <= 1.0f) {
                float v = dot3f(dx, dy, dz, p[6], p[7], p[8]) / dot3f(p[6], p[7], p[8], p[6], p[7], p[8]);
                if (v >= 0.0f && v <= 1.0f) return t;
            }
        } else {
            float dp = dot3f(dx, dy, dz, p[3], p[4], p[5]);
            if (dp >= 0 && dp <= dot3f(p[3], p[4], p[5], p[3], p[4], p[5])) {
                dp = dot3f(dx, dy, dz, p[6], p[7], p[8]);
                if (dp >= 0 && dp <= dot3f(p[6], p[7], p[8], p[6], p[7], p[8])) return t;
            }
        }
    }
    return CRT_TMAX;
}

/* rayIntersectDisk, shapes/disk.cpp:3-47 */
CRT_DEV float dev_disk(const float* p, const DRay& ray)
{
    float dx = p[0] - ray.ox, dy = p[1] - ray.oy, dz = p[2] - ray.oz;
    float t = dot3f(dx, dy, dz, p[3], p[4], p[5]) / dot3f(ray.dx, ray.dy, ray.dz, p[3], p[4], p[5]);
    if (t > CRT_K_EPSILON) {
        float hx = ray.ox + ray.dx * t, hy = ray.oy + ray.dy * t, hz = ray.oz + ray.dz * t;
        dx = hx - p[0]; dy = hy - p[1]; dz = hz - p[2];
        if (sqrtf(dot3f(dx, dy, dz, dx, dy, dz)) <= p[6]) return t;
    }
    return CRT_TMAX;
}

/* rayIntersectGenericOpenCylinder, shapes/generic.cpp:34-125 */
CRT_DEV float dev_cylinder(const float* p, const DRay& ray)
{
    float half_height = p[0], radius = p[1], cphi = p[2];
    float a = ray.dx*ray.dx + ray.dz*ray.dz;
    float b = 2*(ray.ox*ray.dx + ray.oz*ray.dz);
    float c = ray.ox*ray.ox + ray.oz*ray.oz - radius*radius;
    float disc = b*b - 4*a*c;
    if (disc < 0.0f) return CRT_TMAX;
    float e = sqrtf(disc);
    float denom = 2*a;
    float t = (-b - e)/denom;
    if (t > CRT_K_EPSILON) {
        float hx = ray.ox + ray.dx * t, hy = ray.oy + ray.dy * t, hz = ray.oz + ray.dz * t;
        if (fabsf(hy) <= half_height) {
            float phi = atan2f(hx, hz);
            if (fabsf(phi) <= cphi) return t;
        }
    }
    t = (-b + e)/denom;
    if (t > CRT_K_EPSILON) {
        float hx = ray.ox + ray.dx * t, hy = ray.oy + ray.dy * t, hz = ray.oz + ray.dz * t;
        if (fabsf(hy) <= half_height) {
            float phi = atan2f(hx, hz);
            if (fabsf(phi) <= cphi) return t;
        }
    }
    return CRT_TMAX;
}

/* Schwarze's solvers as the reference carries them, util/math.cpp:147-341 (double) */
#define CRT_EQN_EPS 1e-90
#define CRT_ISZERO(x) ((x) > -CRT_EQN_EPS && (x) < CRT_EQN_EPS)
__device__ __noinline__ int dev_solve_quadric(const double c[3], double s[2])
{
    double p = c[1] / (2 * c[2]);
    double q = c[0] / c[2];
    double D = p * p - q;
    if (CRT_ISZERO(D)) { s[0] = -p; return 1; }
    else if (D > 0) { double sqrt_D = sqrt(D); s[0] = sqrt_D - p; s[1] = -sqrt_D - p; return 2; }
    return 0;
}
__device__ __noinline__ int dev_solve_cubic(const double c[4], double s[3])
{
    const double PI_D = 3.14159265358979323846264338327950288;
    int num;
    double A = c[2] / c[3], B = c[1] / c[3], C = c[0] / c[3];
    double sq_A = A * A;
    double p = 1.0f/3 * (- 1.0f/3 * sq_A + B);
    double q = 1.0f/2 * (2.0f/27 * A * sq_A - 1.0f/3 * A * B + C);
    double cb_p = p * p * p;
    double D = q * q + cb_p;
    if (CRT_ISZERO(D)) {
        if (CRT_ISZERO(q)) { s[0] = 0; num = 1; }
        else { double u = cbrt(-q); s[0] = 2 * u; s[1] = -u; num = 2; }
    } else if (D < 0) {
        double phi = 1.0f/3 * acos(-q / sqrt(-cb_p));
        double t = 2.0f * sqrt(-p);
        s[0] =   t * cos(phi);
        s[1] = - t * cos(phi + PI_D / 3);
        s[2] = - t * cos(phi - PI_D / 3);
        num = 3;
    } else {
        double sqrt_D = sqrt(D);
        double u = cbrt(sqrt_D - q);
        double v = - cbrt(sqrt_D + q);
        s[0] = u + v;
        num = 1;
    }
    double sub = 1.0f/3 * A;
    for (int i = 0; i < num; ++i) s[i] -= sub;
    return num;
}
__device__ __noinline__ int dev_solve_quartic(const double c[5], double s[4])
{
    double coeffs[4];
    double z, u, v, sub;
    int num;
    double A = c[3] / c[4], B = c[2] / c[4], C = c[1] / c[4], D = c[0] / c[4];
    double sq_A = A * A;
    double p = - 3.0f/8 * sq_A + B;
    double q = 1.0f/8 * sq_A * A - 1.0f/2 * A * B + C;
    double r = - 3.0f/256*sq_A*sq_A + 1.0f/16*sq_A*B - 1.0f/4*A*C + D;
    if (CRT_ISZERO(r)) {
        coeffs[0] = q; coeffs[1] = p; coeffs[2] = 0; coeffs[3] = 1;
        num = dev_solve_cubic(coeffs, s);
        s[num++] = 0;
    } else {
        coeffs[0] = 1.0f/2 * r * p - 1.0f/8 * q * q;
        coeffs[1] = - r;
        coeffs[2] = - 1.0f/2 * p;
        coeffs[3] = 1;
        (void)dev_solve_cubic(coeffs, s);
        z = s[0];
        u = z * z - r;
        v = 2 * z - p;
        if (CRT_ISZERO(u)) u = 0; else if (u > 0) u = sqrt(u); else return 0;
        if (CRT_ISZERO(v)) v = 0; else if (v > 0) v = sqrt(v); else return 0;
        coeffs[0] = z - u;
        coeffs[1] = q < 0 ? -v : v;
        coeffs[2] = 1;
        num = dev_solve_quadric(coeffs, s);
        coeffs[0] = z + u;
        coeffs[1] = q < 0 ? v : -v;
        coeffs[2] = 1;
        num += dev_solve_quadric(coeffs, s + num);
    }
    sub = 1.0f/4 * A;
    for (int i = 0; i < num; ++i) s[i] -= sub;
    return num;
}

/* rayIntersectGenericTorus (generic.cpp:156-221, double coefficients, returns t even when the
 * phi window rejects) and shadowRayIntersectGenericTorus (:223-278, float coefficients). */
__device__ __noinline__ float dev_torus(const float* p, const DRay& ray, bool shadow)
{
    float swept_radius = p[0], tube_radius = p[1], tphi = p[2];
    double coeffs[5];
    double roots[4];
    if (!shadow) {
        double x1 = ray.ox, y1 = ray.oy, z1 = ray.oz;
        double d1 = ray.dx, d2 = ray.dy, d3 = ray.dz;
        double sum_d_sqrd = d1*d1 + d2*d2 + d3*d3;
        double e = x1*x1 + y1*y1 + z1*z1 - swept_radius*swept_radius - tube_radius*tube_radius;
        double f = x1*d1 + y1*d2 + z1*d3;
        double four_a_sqrd = 4.0f * swept_radius * swept_radius;
        coeffs[0] = e*e - four_a_sqrd*(tube_radius * tube_radius - y1*y1);
        coeffs[1] = 4.0f*f*e + 2.0f*four_a_sqrd*y1*d2;
        coeffs[2] = 2.0f*sum_d_sqrd*e + 4.0f*f*f + four_a_sqrd*d2*d2;
        coeffs[3] = 4.0f*sum_d_sqrd*f;
        coeffs[4] = sum_d_sqrd*sum_d_sqrd;
    } else {
        float x1 = ray.ox, y1 = ray.oy, z1 = ray.oz;
        float d1 = ray.dx, d2 = ray.dy, d3 = ray.dz;
        float sum_d_sqrd = d1*d1 + d2*d2 + d3*d3;
        float e = x1*x1 + y1*y1 + z1*z1 - swept_radius*swept_radius - tube_radius*tube_radius;
        float f = x1*d1 + y1*d2 + z1*d3;
        float four_a_sqrd = 4.0f * swept_radius * swept_radius;
        coeffs[0] = e*e - four_a_sqrd*(tube_radius * tube_radius - y1*y1);
        coeffs[1] = 4.0f*f*e + 2.0f*four_a_sqrd*y1*d2;
        coeffs[2] = 2.0f*sum_d_sqrd*e + 4.0f*f*f + four_a_sqrd*d2*d2;
        coeffs[3] = 4.0f*sum_d_sqrd*f;
        coeffs[4] = sum_d_sqrd*sum_d_sqrd;
    }
    int num_real_roots = dev_solve_quartic(coeffs, roots);
    bool intersected = false;
    float t = CRT_TMAX;
    if (num_real_roots == 0) return CRT_TMAX;
    for (int i = 0; i < num_real_roots; i++) {
        if (roots[i] > CRT_K_EPSILON) {
            intersected = true;
            if (roots[i] < t) t = (float)roots[i];
        }
    }
    if (!intersected) return CRT_TMAX;
    if (!shadow) return t;
    float hx = ray.ox + ray.dx * t, hz = ray.oz + ray.dz * t;
    float phi = atan2f(hx, hz);
    if (fabsf(phi) <= tphi) return t;
    return CRT_TMAX;
}

/* transformRay, util/math.cpp:41-51: 4-term dot products with w = 1 (origin) / 0 (direction) */
CRT_DEV DRay dev_transform_ray(const float* m, const DRay& r)
{
    DRay o;
    o.ox = m[0]*r.ox + m[1]*r.oy + m[2]*r.oz + m[3]*1.0f;
    o.oy = m[4]*r.ox + m[5]*r.oy + m[6]*r.oz + m[7]*1.0f;
    o.oz = m[8]*r.ox + m[9]*r.oy + m[10]*r.oz + m[11]*1.0f;
    o.dx = m[0]*r.dx + m[1]*r.dy + m[2]*r.dz + m[3]*0.0f;
    o.dy = m[4]*r.dx + m[5]*r.dy + m[6]*r.dz + m[7]*0.0f;
    o.dz = m[8]*r.dx + m[9]*r.dy + m[10]*r.dz + m[11]*0.0f;
    return o;
}

/* One non-instanced, non-compound shape: the per-type switch of rayIntersectObject /
 * shadowRayIntersectObject (shapes/shapes.cpp:4-93). */
__device__ __noinline__ float dev_simple_shape(const crt_shape* sh, const DRay& ray, bool shadow)
{
    const float* p = sh->p;
    float t0;
    switch (sh->type) {
    case CRT_SPHERE: return dev_sphere(p, ray, shadow);
    case CRT_PLANE: return dev_plane(p, ray);
    case CRT_RECTANGLE: return dev_rect(p, ray, shadow);
    case CRT_AABOX:                                  /* generic.cpp:311-493 */
        if (shadow && !(p[6] != 0.0f)) return CRT_TMAX;
        return dev_ray_aabb(p, p + 3, ray, &t0);
    case CRT_GENERICOPENCYLINDER: return dev_cylinder(p, ray);
    case CRT_DISK: return dev_disk(p, ray);
    case CRT_GENERICTORUS: return dev_torus(p, ray, shadow);
    default: return CRT_TMAX;
    }
}

/* Compound (instanced.cpp:155-273): box reject, then strict-min (closest) or first hit (shadow) */
__device__ __noinline__ float dev_compound(const crt_shape* shapes, const crt_shape* co, const DRay& ray, bool shadow, int* part)
{
    float t0;
    if (dev_ray_aabb(co->p, co->p + 3, ray, &t0) == CRT_TMAX) return CRT_TMAX;
    if (!shadow) {
        float min_t = CRT_TMAX;
        for (int i = 0; i < co->num_parts; i++) {
            float tmp_t = dev_simple_shape(&shapes[co->first_part + i], ray, false);
            if (tmp_t < min_t) { min_t = tmp_t; *part = i; }
        }
        return min_t;
    }
    float t = CRT_TMAX;
    for (int i = 0; i < co->num_parts; i++) {
        t = dev_simple_shape(&shapes[co->first_part + i], ray, true);
        if (t < CRT_TMAX) return t;
    }
    return t;
}

/* A shape-table object: simple | compound | instanced{simple | compound}
 * (rayIntersectInstanced, instanced.cpp:48-153; deeper nesting is rejected at scene creation). */
__device__ __noinline__ float dev_shape(const crt_shape* shapes, int si, const DRay& ray, bool shadow, int* part)
{
    const crt_shape* sh = &shapes[si];
    *part = 0;
    if (sh->type == CRT_INSTANCED) {
        DRay r2 = dev_transform_ray(sh->inv_xform, ray);
        const crt_shape* inner = &shapes[sh->inner];
        if (inner->type == CRT_COMPOUND) return dev_compound(shapes, inner, r2, shadow, part);
        return dev_simple_shape(inner, r2, shadow);
    }
    if (sh->type == CRT_COMPOUND) return dev_compound(shapes, sh, ray, shadow, part);
    return dev_simple_shape(sh, ray, shadow);
}

/* rayIntersectObject on final-order object `oi` (shapes/shapes.cpp:4-48).
 * TRI_ONLY kernels never take the shape branch. */
/* KIND: which object types the scene holds, known on the host -- 0 any, 1 (= true) triangles only, 2 triangles and rectangles
 * (CRT_KIND_TRI_RECT: config C1's cornell box; a kernel of that kind never references the analytic shapes, whose f64 quartic
 * and libm calls cost every other instance its registers) */
#define CRT_KIND_TRI_RECT 2
template <int KIND>
CRT_DEV float dev_object_rec(const DevScene& sc, const float4* rec, const DRay& ray, bool shadow, DHit* h)
{
    float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    h->obj = __float_as_int(r2.y);
    h->part = 0;
    if (KIND == 1 || dev_is_tri(__float_as_int(r2.z)))
        return dev_tri(r0, r1, r2, ray, &h->beta, &h->gamma);
    h->beta = 0.0f; h->gamma = 0.0f;
    /* rectangles in line: cornell_box.txt (config C1) is rectangles and triangles, and the general path is two out-of-line calls */
    if (KIND == CRT_KIND_TRI_RECT || __float_as_int(r2.z) == CRT_RECTANGLE) return dev_rect(sc.shapes[__float_as_int(r2.w)].p, ray, shadow);
    return dev_shape(sc.shapes, __float_as_int(r2.w), ray, shadow, &h->part);
}
template <int TRI_ONLY>
CRT_DEV float dev_object(const DevScene& sc, int oi, const DRay& ray, bool shadow, DHit* h)
{
    return dev_object_rec<TRI_ONLY>(sc, sc.objs + (size_t)oi * CRT_OBJ_STRIDE, ray, shadow, h);
}
