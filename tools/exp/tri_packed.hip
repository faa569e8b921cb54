/* Experiment (round 2): calcTriangleIntersect with packed-f32 instructions against the scalar form, bit for bit.
 * Build and run on an MI355X:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DHOST_MAIN tri_packed.hip -o tri_packed && ./tri_packed
 * Result: 0 of 12 582 912 output words differ (181 695 hits, 102 864 exact-zero barycentrics); 15 VALU instructions fewer per test; the walk kernels run no faster with it, so the library keeps the scalar form. */
#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
struct DRay { float ox, oy, oz, dx, dy, dz; };
#define CRT_TMAX 1.0e30f
#define CRT_K_EPSILON 1.0e-4f
#define CRT_DEV __device__ __forceinline__
/* one packed-f32 instruction: lane 0 = f(a[sel], b[sel]), lane 1 likewise; each lane rounds as the scalar instruction does */
#define PK_OP(name, ins, mods) \
    CRT_DEV v2f name(v2f a, v2f b) { v2f r; asm(ins " %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b)); return r; }
PK_OP(pk_sub_bcast_lo, "v_pk_add_f32", "op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]")     /* (a.x - b.x, a.x - b.y) */
PK_OP(pk_sub_bcast_hi, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]")   /* (a.y - b.x, a.y - b.y) */
PK_OP(pk_mul_bcast_lo, "v_pk_mul_f32", "op_sel_hi:[1,0]")                               /* (a.x * b.x, a.y * b.x) */
PK_OP(pk_mul_bcast_hi, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")                  /* (a.x * b.y, a.y * b.y) */
PK_OP(pk_mul,          "v_pk_mul_f32", "")                                              /* (a.x * b.x, a.y * b.y) */
PK_OP(pk_mul_swap,     "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]")                  /* (a.x * b.y, a.y * b.x) */
PK_OP(pk_sub,          "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]")                     /* (a.x - b.x, a.y - b.y) */
PK_OP(pk_sub_rsub,     "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[1,0]")                     /* (a.x - b.x, b.y - a.y) */
PK_OP(pk_hsub_ladd,    "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1]")     /* (a.y - b.y, a.x + b.x) */
PK_OP(pk_subhi_addlo,  "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")     /* (a.x - b.y, a.y + b.x) */
/* calcTriangleIntersect, shapes/triangle.cpp:14-77, every product, sum and difference in the reference's operand order, two at a
 * time.  Record: r0 = (v1x v2x v1y v2y)  r1 = (v1z v2z v0x v0y)  r2.x = v0z */
CRT_DEV float dev_tri_pk(const float4 r0, const float4 r1, const float4 r2, const DRay& ray, float* beta_out, float* gamma_out)
{
    const v2f v12x = { r0.x, r0.y }, v12y = { r0.z, r0.w }, v12z = { r1.x, r1.y }, v0xy = { r1.z, r1.w }, v0z_ = { r2.x, r2.y };
    const v2f ab = pk_sub_bcast_lo(v0xy, v12x);       /* a = v0x - v1x, b = v0x - v2x */
    const v2f ef = pk_sub_bcast_hi(v0xy, v12y);       /* e = v0y - v1y, f = v0y - v2y */
    const v2f ij = pk_sub_bcast_lo(v0z_, v12z);       /* i = v0z - v1z, j = v0z - v2z */
    const float c = ray.dx, g = ray.dy, k = ray.dz;
    const float d = r1.z - ray.ox, h = r1.w - ray.oy, l = r2.x - ray.oz;
    const v2f gk = { g, k }, hl = { h, l }, cd = { c, d };
    const v2f efk = pk_mul_bcast_hi(ef, gk);          /* e*k, f*k */
    const v2f gij = pk_mul_bcast_lo(ij, gk);          /* i*g, j*g  (= g*i, g*j) */
    const v2f qm = pk_sub_rsub(gij, efk);             /* q = g*i - e*k, m = f*k - g*j */
    const v2f efl = pk_mul_bcast_hi(ef, hl);          /* e*l, f*l */
    const v2f hij = pk_mul_bcast_lo(ij, hl);          /* i*h, j*h  (= h*i, h*j) */
    const v2f rp = pk_sub(efl, hij);                  /* r = e*l - h*i, p = f*l - h*j */
    const v2f sji = pk_mul_swap(ef, ij);              /* e*j, f*i */
    const float s = sji.x - sji.y;
    const float n = h*k - g*l;
    const v2f amq = pk_mul_swap(ab, qm);              /* a*m, b*q */
    const float inv_denom = 1.0f / (amq.x + amq.y + c*s);
    const v2f nn = { n, n };
    const v2f abn = pk_mul_bcast_lo(ab, nn);          /* a*n, b*n */
    const v2f abpr = pk_mul_swap(ab, rp);             /* a*p, b*r */
    const v2f dqm = pk_mul_bcast_hi(qm, cd);          /* q*d, m*d  (= d*q, d*m) */
    const v2f crp = pk_mul_bcast_lo(rp, cd);          /* r*c, p*c  (= c*r, c*p) */
    const v2f e12a = pk_hsub_ladd(dqm, abn);          /* d*m - b*n,  d*q + a*n (= a*n + d*q) */
    const v2f e12 = pk_subhi_addlo(e12a, crp);        /* e1 = d*m - b*n - c*p,  e2 = a*n + d*q + c*r */
    const v2f ii = { inv_denom, inv_denom };
    const v2f bg = pk_mul_bcast_lo(e12, ii);
    *beta_out = bg.x; *gamma_out = bg.y;
    if (bg.x < 0.0f) return CRT_TMAX;
    if (bg.y < 0.0f) return CRT_TMAX;
    if (bg.x + bg.y > 1.0f) return CRT_TMAX;
    const float e3 = abpr.x - abpr.y + d*s;
    const float t = e3 * inv_denom;
    if (t < CRT_K_EPSILON) return CRT_TMAX;
    return t;
}
/* the scalar form, same record layout, for comparison */
CRT_DEV float dev_tri_sc(const float4 r0, const float4 r1, const float4 r2, const DRay& ray, float* beta_out, float* gamma_out)
{
    const float v1x = r0.x, v2x = r0.y, v1y = r0.z, v2y = r0.w, v1z = r1.x, v2z = r1.y, v0x = r1.z, v0y = r1.w, v0z = r2.x;
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
    *beta_out = beta; *gamma_out = gamma;
    if (beta < 0.0f) return CRT_TMAX;
    if (gamma < 0.0f) return CRT_TMAX;
    if (beta + gamma > 1.0f) return CRT_TMAX;
    float e3 = a*p - b*r + d*s;
    float t = e3 * inv_denom;
    if (t < CRT_K_EPSILON) return CRT_TMAX;
    return t;
}
template <bool PK>
__global__ void k(const float4* recs, const DRay* rays, float* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DRay ray = rays[i];
    float4 r0 = recs[3*i], r1 = recs[3*i+1], r2 = recs[3*i+2];
    float b, g;
    float t = PK ? dev_tri_pk(r0, r1, r2, ray, &b, &g) : dev_tri_sc(r0, r1, r2, ray, &b, &g);
    out[3*i] = t; out[3*i+1] = b; out[3*i+2] = g;
}
template __global__ void k<true>(const float4*, const DRay*, float*, int);
template __global__ void k<false>(const float4*, const DRay*, float*, int);
#ifdef HOST_MAIN
#include <vector>
#include <cstdio>
#include <cstring>
#include <random>
int main()
{
    const int n = 1 << 22;
    std::vector<float> recs((size_t)n * 12), rays((size_t)n * 6);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (int i = 0; i < n; i++) {
        float v[9];
        const bool grid = (i & 3) == 0;       /* a quarter on a coarse lattice: exact zeros, edges, ties */
        for (int c = 0; c < 9; c++) v[c] = grid ? (float)((int)(U(rng) * 4.0f)) * 0.5f : U(rng);
        float* r = &recs[(size_t)i * 12];
        r[0] = v[3]; r[1] = v[6]; r[2] = v[4]; r[3] = v[7]; r[4] = v[5]; r[5] = v[8]; r[6] = v[0]; r[7] = v[1]; r[8] = v[2];
        r[9] = r[10] = r[11] = 0.0f;
        float* y = &rays[(size_t)i * 6];
        for (int c = 0; c < 6; c++) y[c] = grid ? (float)((int)(U(rng) * 4.0f)) * 0.5f : U(rng);
        if ((i & 15) == 1) y[3] = 0.0f;
        if ((i & 15) == 2) y[4] = -0.0f;
    }
    float4* d_recs; DRay* d_rays; float* d_out[2];
    hipMalloc(&d_recs, recs.size() * 4); hipMalloc(&d_rays, rays.size() * 4);
    hipMalloc(&d_out[0], (size_t)n * 12); hipMalloc(&d_out[1], (size_t)n * 12);
    hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice);
    k<true><<<n / 256, 256>>>(d_recs, d_rays, d_out[0], n);
    k<false><<<n / 256, 256>>>(d_recs, d_rays, d_out[1], n);
    std::vector<float> a((size_t)n * 3), b((size_t)n * 3);
    hipMemcpy(a.data(), d_out[0], a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_out[1], b.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0, hits = 0, zeros = 0;
    for (size_t i = 0; i < a.size(); i++) {
        if (memcmp(&a[i], &b[i], 4) != 0) { if (diff < 5) printf("diff at %zu: %a vs %a\n", i, a[i], b[i]); diff++; }
        if (i % 3 == 0 && b[i] < 1e29f) hits++;
        if (i % 3 != 0 && b[i] == 0.0f) zeros++;
    }
    printf("packed vs scalar: %zu of %zu words differ; %zu hits, %zu exact-zero barycentrics\n", diff, a.size(), hits, zeros);
    return diff != 0;
}
#endif
