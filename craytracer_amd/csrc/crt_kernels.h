/*
 * crt_kernels.h -- traversal kernels (gfx950, wave64).
 *
 * Execution model shared by every kernel here: a persistent grid (a few 256-thread
 * workgroups per CU); each 64-lane wavefront pulls 64 consecutive rays at a time from an
 * HBM-resident queue head (one agent-scope atomic per wave, broadcast with readfirstlane),
 * one ray per lane, and loops until the queue is drained.  Per-lane traversal stacks live in
 * LDS, laid out [entry][thread] so that a wave's access to one entry touches 64 consecutive
 * dwords (bank = lane mod 32: conflict-free); entries beyond CRT_LDS_DEPTH spill to a per-scene
 * HBM overflow area laid out the same way ([entry][global thread]).
 */
#pragma once
#include "crt_device.h"

/* The ray records and the hit records are touched once per launch: they are read and written with the non-temporal hint, so that
 * 740 MB of them per 2^24 rays pass through the L2s and the Infinity Cache without displacing the tree.  Closest hit 7.98 -> 7.81 ms
 * (+2 %), any hit unchanged.  The same hint on the ray order, the light distances and the any-hit results measured -1 % on the
 * any-hit walk, and on the binning kernels' scattered writes 0.32 -> 0.88 ms (they lose their write combining): rays and hits only. */
#ifndef CRT_NT_IO
#define CRT_NT_IO 1
#endif
#ifndef CRT_BLOCK
#define CRT_BLOCK      256         /* threads per workgroup of the walk kernels (a multiple of 64 that divides 256) */
#endif
/* LDS part of the per-lane stack.  8-byte entries (closest-hit walks): 14 deep = 28 KB per workgroup, so that FIVE workgroups
 * are resident per CU -- at 16 deep (32 KB) only four fit in the 160 KB, and the fifth wave per SIMD is worth 8 % (1 813 ->
 * 1 954 Mrays/s; 12 deep: 1 915, six waves at 80 VGPRs and 12 / 13 deep: 1 844 / 1 854).  4-byte entries (any-hit walks): 16 deep. */
#ifndef CRT_LDS_DEPTH
#define CRT_LDS_DEPTH  14          /* (13 in the tail-splitting instance of the headline kernel, whose slots take 4 KB: 30 KB, still five per CU) */
#endif
#ifndef CRT_LDS_DEPTH32
#define CRT_LDS_DEPTH32 16
#endif
#define CRT_OVF_DEPTH  48
#define CRT_STACK_MAX  (CRT_LDS_DEPTH + CRT_OVF_DEPTH)     /* 62: what crt_scene_create checks tree depths against */
/* stack entries a QBVH walk can need per tree level: 3 pending siblings, + 1 frame marker in the
 * reference-control-flow (EXACT) walk */
#define CRT_STACK_PER_LEVEL_PRUNED 3
#define CRT_STACK_PER_LEVEL_EXACT  4
#define CRT_FRAME_MARK 0xFFFFFFFFu   /* not a leaf word: count field 7 */

struct LaunchArgs {
    const crt_ray* rays;
    const float* light_dist;      /* shadow only */
    crt_hit* hits;                /* closest only */
    float* t_out;                 /* shadow only, may be null */
    uint8_t* occ_out;             /* shadow only, may be null */
    unsigned n;
    unsigned* queue_head;         /* zeroed before launch */
    unsigned* ovf_word;           /* [CRT_OVF_DEPTH][grid threads] */
    float* ovf_key;               /* [CRT_OVF_DEPTH][grid threads] */
    unsigned grid_threads;
    const unsigned* perm;         /* queue slot -> ray index (crt_binning.h), or null: slot == ray index */
    const unsigned* n_dev;        /* counted launches: the ray count lives on the device (n is then its upper bound), or null */
    const unsigned* q_range;      /* slice launches (crt_intersect_batch_dev_slice): {first, end} queue slot of this launch's share of the
                                     binned batch, written by k_bin_scan; or null: the whole queue */
    unsigned q_first;             /* first queue slot of the launch (0 unless q_range, or a slice by ray index) */
    unsigned* fb_count;           /* k_bvh8_closest_tri: rays it leaves to the 4-wide walk of the same launch (crt_wide8.h): how many, */
    unsigned* fb_list;            /* ... and their indices */
};
/* first statement of every walk kernel: the launch's ray count and the part of the queue it walks, [q_first, n) */
CRT_DEV void resolve_count(LaunchArgs& la)
{
    if (la.n_dev) { const unsigned m = *la.n_dev; la.n = m < la.n ? m : la.n; }
    if (la.q_range) {
        const unsigned lo = la.q_range[0], hi = la.q_range[1];
        la.n = hi < la.n ? hi : la.n;
        la.q_first = lo < la.n ? lo : la.n;
    }
}

/* ---- per-lane stack ----------------------------------------------------------------
 * Entry = (child word, key) packed in 8 bytes -> one ds_write_b64 / ds_read_b64.  The LDS part is
 * addressed through the __shared__ array itself so that every access is a DS instruction; the
 * overflow part is kept out of line (a pointer select between the two would turn every pop into
 * a flat load that waits on both memory counters). */
typedef uint2 StackEntry;

__device__ __noinline__ void ovf_push(unsigned* ovf_word, float* ovf_key, size_t o, unsigned w, float key)
{
    ovf_word[o] = w; ovf_key[o] = key;
}
__device__ __noinline__ StackEntry ovf_pop(const unsigned* ovf_word, const float* ovf_key, size_t o)
{
    StackEntry e; e.x = ovf_word[o]; e.y = __float_as_uint(ovf_key[o]);
    return e;
}

template <int DEPTH>
struct LaneStackT {
    StackEntry (*lds)[CRT_BLOCK];   /* the workgroup's __shared__ [DEPTH][CRT_BLOCK] array */
    unsigned* ovf_word;           /* &ovf_word[0][gtid] */
    float* ovf_key;
    unsigned ovf_stride;
    int sp;
    CRT_DEV void push(unsigned w, float key)
    {
        if (__builtin_expect(sp < DEPTH, 1)) {
            StackEntry e; e.x = w; e.y = __float_as_uint(key);
            lds[sp][threadIdx.x] = e;
        } else if (sp < CRT_STACK_MAX) {
            ovf_push(ovf_word, ovf_key, (size_t)(sp - DEPTH) * ovf_stride, w, key);
        }
        /* deeper than CRT_STACK_MAX cannot happen: crt_scene_create rejects such trees */
        sp++;
    }
    CRT_DEV void pop(unsigned* w, float* key)
    {
        sp--;
        StackEntry e;
        if (__builtin_expect(sp < DEPTH, 1)) e = lds[sp][threadIdx.x];
        else e = ovf_pop(ovf_word, ovf_key, (size_t)(sp - DEPTH) * ovf_stride);
        *w = e.x; *key = __uint_as_float(e.y);
    }
};
typedef LaneStackT<CRT_LDS_DEPTH> LaneStack;

/* words only (any-hit walks need no key): 4-byte entries, half the LDS, more resident waves */
__device__ __noinline__ void ovf_push32(unsigned* ovf_word, size_t o, unsigned w) { ovf_word[o] = w; }
__device__ __noinline__ unsigned ovf_pop32(const unsigned* ovf_word, size_t o) { return ovf_word[o]; }
struct LaneStack32 {
    unsigned (*lds)[CRT_BLOCK];
    unsigned* ovf_word;
    unsigned ovf_stride;
    int sp;
    CRT_DEV void push(unsigned w)
    {
        if (__builtin_expect(sp < CRT_LDS_DEPTH32, 1)) lds[sp][threadIdx.x] = w;
        else if (sp < CRT_STACK_MAX) ovf_push32(ovf_word, (size_t)(sp - CRT_LDS_DEPTH32) * ovf_stride, w);
        sp++;
    }
    CRT_DEV unsigned pop()
    {
        sp--;
        if (__builtin_expect(sp < CRT_LDS_DEPTH32, 1)) return lds[sp][threadIdx.x];
        return ovf_pop32(ovf_word, (size_t)(sp - CRT_LDS_DEPTH32) * ovf_stride);
    }
};

CRT_DEV DRay load_ray(const crt_ray* rays, unsigned i)
{
    typedef float ld2f __attribute__((ext_vector_type(2)));
    const ld2f* p = reinterpret_cast<const ld2f*>(rays + i);        /* 24-B records, 8-B aligned */
#if CRT_NT_IO
    const ld2f a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1), c = __builtin_nontemporal_load(p + 2);
#else
    const ld2f a = p[0], b = p[1], c = p[2];
#endif
    DRay r; r.ox = a.x; r.oy = a.y; r.oz = b.x; r.dx = b.y; r.dy = c.x; r.dz = c.y;
    return r;
}
CRT_DEV void store_hit(crt_hit* hits, unsigned i, float t, const DHit& h)
{
    crt_hit o;
    o.t = t;
    if (t < CRT_TMAX) { o.obj = h.obj; o.part = h.part; o.beta = h.beta; o.gamma = h.gamma; }
    else { o.obj = -1; o.part = -1; o.beta = 0.0f; o.gamma = 0.0f; }
#if CRT_NT_IO
    typedef float st4f __attribute__((ext_vector_type(4)));
    const st4f v = { o.t, __int_as_float(o.obj), __int_as_float(o.part), o.beta };
    __builtin_nontemporal_store(v, reinterpret_cast<st4f*>(hits + i));          /* 20-B records, 4-B aligned: global_store_dwordx4 takes that */
    __builtin_nontemporal_store(o.gamma, reinterpret_cast<float*>(hits + i) + 4);
#else
    hits[i] = o;
#endif
}

/* Wave-level fetch of the next 64 queue slots. */
CRT_DEV unsigned next_batch(const LaunchArgs& la)
{
    unsigned base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(la.queue_head, 64u);
    return la.q_first + (unsigned)__builtin_amdgcn_readfirstlane((int)base);
}

template <bool COUNT>
CRT_DEV void flush_counters(const DevScene& sc, unsigned rays, unsigned nodes, unsigned prims, bool shadow)
{
    if (COUNT) {
        atomicAdd(&sc.counters[shadow ? 1 : 0], (unsigned long long)rays);
        atomicAdd(&sc.counters[2], (unsigned long long)nodes);
        atomicAdd(&sc.counters[3], (unsigned long long)prims);
    }
}

/* rayIntersectAABB4 (accelerator/bvh4.h:23-97) for child c of a node held in registers.
 * SSE max/min semantics: a > b ? a : b (second operand when unordered).  Returns the t the
 * reference stores (entry t0 if t0 > eps, else exit t1, TMAX on miss); *lt = (t0 < eps). */
CRT_DEV float box4(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                   const DRay& r, float ia, float ib, float ic, bool sx, bool sy, bool sz, bool* lt)
{
    float tx_min = ((sx ? mnx : mxx) - r.ox) * ia, tx_max = ((sx ? mxx : mnx) - r.ox) * ia;
    float ty_min = ((sy ? mny : mxy) - r.oy) * ib, ty_max = ((sy ? mxy : mny) - r.oy) * ib;
    float tz_min = ((sz ? mnz : mxz) - r.oz) * ic, tz_max = ((sz ? mxz : mnz) - r.oz) * ic;
    float m0 = tx_min > ty_min ? tx_min : ty_min;
    float t0 = m0 > tz_min ? m0 : tz_min;
    float m1 = tx_max < ty_max ? tx_max : ty_max;
    float t1 = m1 < tz_max ? m1 : tz_max;
    bool hit = (t0 < t1) && (t1 > CRT_K_EPSILON);
    *lt = (t0 < CRT_K_EPSILON);
    float t = (t0 > CRT_K_EPSILON) ? t0 : t1;
    return hit ? t : CRT_TMAX;
}

/* The same test with the near / far planes already chosen (per-ray load offsets), returning the
 * stack key directly: TMAX = not hit; -inf = hit with the origin inside / behind the box (the
 * reference's `less_than && bbox_t < TMAX`, bvh4.h:358: always visited); else the box t. */
CRT_DEV float box4_key_tail(float t0, float t1)
{
    const bool hit = (t0 < t1) && (t1 > CRT_K_EPSILON);
    const float t = (t0 > CRT_K_EPSILON) ? t0 : t1;
    const float b = hit ? t : CRT_TMAX;
    return ((t0 < CRT_K_EPSILON) && (b < CRT_TMAX)) ? -__builtin_huge_valf() : b;
}
/* SSE semantics spelled out: max(a, b) = a > b ? a : b, min(a, b) = a < b ? a : b (second operand
 * when unordered) -- needed when a product can be NaN (zero direction component with the origin on
 * a slab plane, bvh4.h:43-75). */
CRT_DEV float box4_key_sse(float nx, float ny, float nz, float fx, float fy, float fz,
                           const DRay& r, float ia, float ib, float ic)
{
    const float tx_min = (nx - r.ox) * ia, tx_max = (fx - r.ox) * ia;
    const float ty_min = (ny - r.oy) * ib, ty_max = (fy - r.oy) * ib;
    const float tz_min = (nz - r.oz) * ic, tz_max = (fz - r.oz) * ic;
    const float m0 = tx_min > ty_min ? tx_min : ty_min;
    const float t0 = m0 > tz_min ? m0 : tz_min;
    const float m1 = tx_max < ty_max ? tx_max : ty_max;
    const float t1 = m1 < tz_max ? m1 : tz_max;
    return box4_key_tail(t0, t1);
}
/* No operand can be NaN (see `clean`): the chains select the same values as v_max3 / v_min3
 * (a zero may come out with the other sign; it is only compared afterwards, never returned). */
CRT_DEV float box4_key_fast(float nx, float ny, float nz, float fx, float fy, float fz,
                            const DRay& r, float ia, float ib, float ic)
{
    const float tx_min = (nx - r.ox) * ia, tx_max = (fx - r.ox) * ia;
    const float ty_min = (ny - r.oy) * ib, ty_max = (fy - r.oy) * ib;
    const float tz_min = (nz - r.oz) * ic, tz_max = (fz - r.oz) * ic;
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(tx_min, ty_min), tz_min);
    const float t1 = __builtin_fminf(__builtin_fminf(tx_max, ty_max), tz_max);
    return box4_key_tail(t0, t1);
}

/* two boxes at a time so that the subtract / multiply pairs become v_pk_add_f32 / v_pk_mul_f32
 * (each lane of a packed op rounds like the scalar op; contraction is off) */
typedef float v2f __attribute__((ext_vector_type(2)));
CRT_DEV void box4_keys_fast2(v2f nx, v2f ny, v2f nz, v2f fx, v2f fy, v2f fz,
                             const DRay& r, float ia, float ib, float ic, float* ka, float* kb)
{
    const v2f ox = {r.ox, r.ox}, oy = {r.oy, r.oy}, oz = {r.oz, r.oz};
    const v2f ia2 = {ia, ia}, ib2 = {ib, ib}, ic2 = {ic, ic};
    const v2f tx_min = (nx - ox) * ia2, tx_max = (fx - ox) * ia2;
    const v2f ty_min = (ny - oy) * ib2, ty_max = (fy - oy) * ib2;
    const v2f tz_min = (nz - oz) * ic2, tz_max = (fz - oz) * ic2;
    *ka = box4_key_tail(__builtin_fmaxf(__builtin_fmaxf(tx_min.x, ty_min.x), tz_min.x),
                        __builtin_fminf(__builtin_fminf(tx_max.x, ty_max.x), tz_max.x));
    *kb = box4_key_tail(__builtin_fmaxf(__builtin_fmaxf(tx_min.y, ty_min.y), tz_min.y),
                        __builtin_fminf(__builtin_fminf(tx_max.y, ty_max.y), tz_max.y));
}

/* ---- compressed nodes ---------------------------------------------------------------------------
 * A second, 64-byte copy of every QBVH node (built on the host in crt_scene_create, checked in double):
 *   c0 = origin.x origin.y origin.z scale.x     c1 = scale.y scale.z lo_x[4] lo_y[4]
 *   c2 = lo_z[4] hi_x[4] hi_y[4] hi_z[4]        c3 = the four child words (as in the full node)
 * plane = origin + byte * scale, bytes rounded outward, so each compressed box CONTAINS the node's box.  A visit reads
 * 4 x 16 bytes per lane instead of 7 -- the walk is bound by how many lane-loads the L1 takes per clock,
 * not by bytes.  The slab distances come out of one fma per plane, moved outward by a margin that covers every rounding
 * on the way (8 ulp of |a| + 255 |b|; analysis in DESIGN.md), so t0 <= the reference's t0 and t1 >= the reference's t1:
 * the walk visits a SUPERSET of the reference's nodes in the reference's order.  What it finds is the same: a ray that
 * hits a primitive passes through the primitive's padded box (K_FLAT_AABB = 1e-3 against roundings of 1e-6) and so
 * through every box above it in the reference's own test; a subtree visited only because its box was widened holds
 * no hit.  Used for the best-t-pruned closest-hit walk and the any-hit walk of waves whose rays are all `clean`
 * (no inf / NaN in the slab arithmetic); everything else reads the full nodes.
 * Key = t0 (conservative): "origin inside the box" needs no special value, t0 < eps <= any hit's t. */
CRT_DEV void cnode_axis(float origin, float scale, float o, float inv, float* a_near, float* a_far, float* b)
{
    const float a = (origin - o) * inv;
    *b = scale * inv;
    const float m = __builtin_fmaf(255.0f, fabsf(*b), fabsf(a)) * 4.76837158203125e-07f;      /* 2^-21 */
    *a_near = a - m;
    *a_far = a + m;
}
CRT_DEV float cnode_key(float tnx, float tny, float tnz, float tfx, float tfy, float tfz)
{
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
    const float t1 = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
    return ((t0 < t1) && (t1 > CRT_K_EPSILON)) ? t0 : CRT_TMAX;
}
#define CRT_UB(w, c) ((float)(((w) >> (8 * (c))) & 0xFFu))       /* v_cvt_f32_ubyteN */
/* keys of the four children of compressed node `nb` (byte offset); sx / sy / sz: direction component >= 0 */
CRT_DEV void cnode_keys_regs(const uint4 c0, const uint4 c1, const uint4 c2, const DRay& r, float ia, float ib, float ic,
                             bool sx, bool sy, bool sz, float* k0, float* k1, float* k2, float* k3);
CRT_DEV void cnode_keys(const uint4* cnodes, unsigned nb, const DRay& r, float ia, float ib, float ic, bool sx, bool sy, bool sz,
                        float* k0, float* k1, float* k2, float* k3, uint4* words)
{
    const char* cb = reinterpret_cast<const char*>(cnodes);
    const uint4 c0 = *reinterpret_cast<const uint4*>(cb + nb);
    const uint4 c1 = *reinterpret_cast<const uint4*>(cb + (nb + 16u));
    const uint4 c2 = *reinterpret_cast<const uint4*>(cb + (nb + 32u));
    *words = *reinterpret_cast<const uint4*>(cb + (nb + 48u));
    cnode_keys_regs(c0, c1, c2, r, ia, ib, ic, sx, sy, sz, k0, k1, k2, k3);
}
/* the same from a node already in registers */
CRT_DEV void cnode_keys_regs(const uint4 c0, const uint4 c1, const uint4 c2, const DRay& r, float ia, float ib, float ic,
                             bool sx, bool sy, bool sz, float* k0, float* k1, float* k2, float* k3)
{
    float anx, afx, bx, any_, afy, by, anz, afz, bz;
    cnode_axis(__uint_as_float(c0.x), __uint_as_float(c0.w), r.ox, ia, &anx, &afx, &bx);
    cnode_axis(__uint_as_float(c0.y), __uint_as_float(c1.x), r.oy, ib, &any_, &afy, &by);
    cnode_axis(__uint_as_float(c0.z), __uint_as_float(c1.y), r.oz, ic, &anz, &afz, &bz);
    const unsigned qnx = sx ? c1.z : c2.y, qfx = sx ? c2.y : c1.z;
    const unsigned qny = sy ? c1.w : c2.z, qfy = sy ? c2.z : c1.w;
    const unsigned qnz = sz ? c2.x : c2.w, qfz = sz ? c2.w : c2.x;
    const v2f bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
    const v2f anx2 = {anx, anx}, afx2 = {afx, afx}, any2 = {any_, any_}, afy2 = {afy, afy}, anz2 = {anz, anz}, afz2 = {afz, afz};
    {
        const v2f tnx = __builtin_elementwise_fma(v2f{CRT_UB(qnx, 0), CRT_UB(qnx, 1)}, bx2, anx2);
        const v2f tfx = __builtin_elementwise_fma(v2f{CRT_UB(qfx, 0), CRT_UB(qfx, 1)}, bx2, afx2);
        const v2f tny = __builtin_elementwise_fma(v2f{CRT_UB(qny, 0), CRT_UB(qny, 1)}, by2, any2);
        const v2f tfy = __builtin_elementwise_fma(v2f{CRT_UB(qfy, 0), CRT_UB(qfy, 1)}, by2, afy2);
        const v2f tnz = __builtin_elementwise_fma(v2f{CRT_UB(qnz, 0), CRT_UB(qnz, 1)}, bz2, anz2);
        const v2f tfz = __builtin_elementwise_fma(v2f{CRT_UB(qfz, 0), CRT_UB(qfz, 1)}, bz2, afz2);
        *k0 = cnode_key(tnx.x, tny.x, tnz.x, tfx.x, tfy.x, tfz.x);
        *k1 = cnode_key(tnx.y, tny.y, tnz.y, tfx.y, tfy.y, tfz.y);
    }
    {
        const v2f tnx = __builtin_elementwise_fma(v2f{CRT_UB(qnx, 2), CRT_UB(qnx, 3)}, bx2, anx2);
        const v2f tfx = __builtin_elementwise_fma(v2f{CRT_UB(qfx, 2), CRT_UB(qfx, 3)}, bx2, afx2);
        const v2f tny = __builtin_elementwise_fma(v2f{CRT_UB(qny, 2), CRT_UB(qny, 3)}, by2, any2);
        const v2f tfy = __builtin_elementwise_fma(v2f{CRT_UB(qfy, 2), CRT_UB(qfy, 3)}, by2, afy2);
        const v2f tnz = __builtin_elementwise_fma(v2f{CRT_UB(qnz, 2), CRT_UB(qnz, 3)}, bz2, anz2);
        const v2f tfz = __builtin_elementwise_fma(v2f{CRT_UB(qfz, 2), CRT_UB(qfz, 3)}, bz2, afz2);
        *k2 = cnode_key(tnx.x, tny.x, tnz.x, tfx.x, tfy.x, tfz.x);
        *k3 = cnode_key(tnx.y, tny.y, tnz.y, tfx.y, tfy.y, tfz.y);
    }
}

/* Linear pass over objects[first, last) -- the non-grid loops of intersectTest
 * (intersect.h:387-395,:403-411) and the NONE branch (:416-432): strict `<`. */
template <int TRI_ONLY, bool COUNT>
CRT_DEV void linear_closest(const DevScene& sc, int first, int last, const DRay& ray, float* best_t, DHit* best, unsigned* prims)
{
    for (int i = first; i < last; i++) {
        DHit h;
        float t = dev_object<TRI_ONLY>(sc, i, ray, false, &h);
        if (COUNT) (*prims)++;
        if (t < *best_t) { *best_t = t; *best = h; }
    }
}
/* shadow flavour (intersect.h:460-467,:476-483,:490-497): t = last tested, return on t < bound */
template <int TRI_ONLY, bool COUNT>
CRT_DEV float linear_shadow(const DevScene& sc, int first, int last, const DRay& ray, float bound, float t, unsigned* prims)
{
    for (int i = first; i < last; i++) {
        DHit h;
        t = dev_object<TRI_ONLY>(sc, i, ray, true, &h);
        if (COUNT) (*prims)++;
        if (t < bound) return t;
    }
    return t;
}

/* ---- lane refill -------------------------------------------------------------------
 * Incoherent rays visit very different numbers of nodes (the tail is several times the mean),
 * so a wave that waits for the slowest of 64 rays runs with most lanes switched off (measured:
 * 12 % active lanes per VALU instruction, profiles/r01_pmc_baseline.txt).  Instead, lanes whose
 * ray has finished are refilled from the queue while the others keep walking: once at least
 * CRT_REFILL_MIN lanes are idle, each idle lane takes the next ray of the wave's feed (below) given
 * by its rank among the idle lanes (ballot + mbcnt). */
#ifndef CRT_REFILL_MIN
#define CRT_REFILL_MIN 8          /* closest-hit walks: 6..10 measure the same, 2 and 16 are 2 % slower */
#endif
#ifndef CRT_REFILL_MIN_SHADOW
#define CRT_REFILL_MIN_SHADOW 32  /* any-hit walks are short: 28..36 measure the same, 16 and 48 are 6..9 % slower */
#endif
#ifndef CRT_WALK_WAVES_N
#define CRT_WALK_WAVES_N 0        /* n > 0: register budget of the QBVH walk kernels for n waves per SIMD */
#endif
#if CRT_WALK_WAVES_N
#define CRT_WALK_WAVES __attribute__((amdgpu_waves_per_eu(CRT_WALK_WAVES_N, CRT_WALK_WAVES_N)))
#else
#define CRT_WALK_WAVES
#endif
#ifndef CRT_FUSE_LEAF_SHADOW
#define CRT_FUSE_LEAF_SHADOW 1    /* the same in the any-hit walk: +3.6 % */
#endif
#ifndef CRT_PROBE_NO_FALLBACK
#define CRT_PROBE_NO_FALLBACK 0   /* register-pressure probe only: results are wrong for rays that are not `clean` */
#endif
#ifndef CRT_PENDING_LEAF_SHADOW
#define CRT_PENDING_LEAF_SHADOW 1
#endif
#ifndef CRT_DIAG
#define CRT_DIAG 0                /* diagnostic build: per-iteration lane statistics of the closest-hit walk into counters[8..] */
#endif
#ifndef CRT_TIMELINE
#define CRT_TIMELINE 0            /* diagnostic build: when each wave of the closest-hit walk started, ran dry and ended, into counters[32..] */
#endif
/* ordering of the tail-splitting slots (LDS, touched by the lanes of one wave only) */
#define CRT_WG_FENCE()    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#ifndef CRT_STEAL_BOTTOM
#define CRT_STEAL_BOTTOM 0
#endif
#ifndef CRT_TAIL_SPLIT
#define CRT_TAIL_SPLIT 0          /* default of the body's SPLIT parameter; the host picks k_bvh4_closest_tri<COUNT, true> per launch for small batches */
#endif
#ifndef CRT_PENDING_LEAF
#define CRT_PENDING_LEAF 1
#endif
#ifndef CRT_SPLIT_MIN_IDLE
#define CRT_SPLIT_MIN_IDLE 8      /* tail splitting: idle lanes a hand-over round waits for */
#endif

CRT_DEV unsigned lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
/* ---- ray feed: lane refill without the stalls ---------------------------------------
 * Claiming rays at the moment lanes go idle costs the whole wave two dependent trips to HBM (the
 * atomic on the queue head, then perm -> ray record), which shows once rays are short: with that
 * scheme the any-hit walk went from 1 330 to 1 826 Mrays/s merely by refilling half as often.  The
 * feed takes both trips
 * off the critical path: a wave claims 64 queue slots at a time one claim AHEAD, loads the claimed
 * rays into registers (lane l holds slot l) as soon as the previous claim becomes current, and
 * hands them to idle lanes through ds_bpermute -- by the time a lane asks, the data is there. */
template <bool WITH_LD>
struct RayFeed {
    unsigned pend;                    /* lane 0: result of the claim issued ahead */
    unsigned res_pos, res_cnt;        /* wave-uniform: next unread slot / slots in the current claim */
    bool empty;                       /* wave-uniform: the queue has nothing more for this wave */
    unsigned pre_ri; DRay pre_ray; float pre_ld;
    CRT_DEV void claim(const LaunchArgs& la)
    {
        if ((threadIdx.x & 63) == 0) pend = atomicAdd(la.queue_head, 64u);
    }
    CRT_DEV unsigned claimed_base(const LaunchArgs& la) { return la.q_first + (unsigned)__builtin_amdgcn_readfirstlane((int)pend); }
    CRT_DEV unsigned claim_limit(const LaunchArgs& la, unsigned base) { return la.n - base; }
    CRT_DEV void swap_in(const LaunchArgs& la)
    {
        const unsigned base = claimed_base(la);
        res_pos = 0;
        if (base >= la.n) { res_cnt = 0; empty = true; return; }
        const unsigned lim = claim_limit(la, base);
        res_cnt = lim < 64u ? lim : 64u;
        const unsigned lane = threadIdx.x & 63;
        if (lane < res_cnt) {
            const unsigned idx = base + lane;
            pre_ri = la.perm ? la.perm[idx] : idx;
            pre_ray = load_ray(la.rays, pre_ri);
            if (WITH_LD) pre_ld = la.light_dist[pre_ri];
        }
        claim(la);
    }
    CRT_DEV void start(const LaunchArgs& la)
    {
        pend = 0; res_pos = res_cnt = 0; empty = false;
        pre_ri = 0; pre_ld = 0.0f;
        pre_ray.ox = pre_ray.oy = pre_ray.oz = 0.0f; pre_ray.dx = pre_ray.dy = pre_ray.dz = 1.0f;
        claim(la);
        swap_in(la);
    }
    /* idle: ballot of lanes without a ray.  Returns true for the lanes that received one. */
    CRT_DEV bool take(const LaunchArgs& la, unsigned long long idle, bool lane_idle, unsigned* ri, DRay* ray, float* ld)
    {
        const unsigned cnt = (unsigned)__popcll(idle);
        const unsigned avail = res_cnt - res_pos;
        const unsigned n_take = cnt < avail ? cnt : avail;
        const unsigned rank = lane_rank(idle);
        const unsigned src = (res_pos + rank) & 63u;
        const int a = (int)(src << 2);
        const unsigned v_ri = (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)pre_ri);
        DRay v;
        v.ox = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ray.ox)));
        v.oy = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ray.oy)));
        v.oz = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ray.oz)));
        v.dx = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ray.dx)));
        v.dy = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ray.dy)));
        v.dz = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ray.dz)));
        float v_ld = 0.0f;
        if (WITH_LD) v_ld = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(pre_ld)));
        const bool got = lane_idle && rank < n_take;
        if (got) { *ri = v_ri; *ray = v; if (WITH_LD) *ld = v_ld; }
        res_pos += n_take;
        if (res_pos == res_cnt) swap_in(la);          /* the next claim's rays load while the wave walks */
        return got;
    }
};

/* =====================================================================================
 * QBVH closest hit, best-t pruned, reference child order.
 * BVH4IntersectTest (accelerator/bvh4.h:299-389) + the BVH4 branch of intersectTest
 * (intersect.h:397-412).  Children are visited in the reference's order and with its test
 *     box_t < min_t || (less_than && box_t < TMAX)                            (bvh4.h:358)
 * where min_t is the ray's nearest hit so far instead of the per-node local of bvh4.h:354;
 * the test is applied when a child is pushed and again when it is popped, which is the same
 * as applying it when the recursive version reaches the child.
 * ===================================================================================== */
/* EXACT = the reference's own control flow on the same stack machine.  In BVH4IntersectTest the
 * threshold a child is tested against is min_t LOCAL to the node visit (bvh4.h:354-371): the
 * smallest t returned by the children visited before it.  A subtree's return value is the smallest t
 * of all primitives tested inside it, so "local min_t of node V" = smallest t found since V was
 * entered.  The walk keeps that number for the innermost open node in cur_min; entering an inner
 * node pushes a frame marker holding the parent's value and restarts cur_min at TMAX, and popping
 * the marker (all of the node's children done) folds the node's result into the parent's:
 * `if (t < min_t) min_t = t` (bvh4.h:366-370).  The hit record that survives this chain of strict
 * comparisons is the first-found global minimum, which best_t / best track as in the pruned walk. */
template <bool TRI_ONLY, bool COUNT, bool EXACT, bool COMPRESSED, bool SPLIT = (CRT_TAIL_SPLIT != 0)>
CRT_DEV void bvh4_closest_body(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    /* PENDING: the conservative-node walk keeps the leaf it is testing in a register of its own (`pl`) and walks on while
     * the leaf's remaining objects are tested, one per iteration, next to the node steps (see the loop below). */
    constexpr bool PENDING = COMPRESSED && !EXACT && TRI_ONLY && CRT_PENDING_LEAF;
    constexpr bool STEAL = PENDING && SPLIT;
    /* the tail-splitting instance gives one LDS stack entry (2 KB) per lane for its 4 KB of slots: 30 KB, still five per CU */
    constexpr int LDS_D = STEAL ? CRT_LDS_DEPTH - 1 : CRT_LDS_DEPTH;
    __shared__ StackEntry s_stack[LDS_D][CRT_BLOCK];
    /* tail splitting: one slot per lane, {(t bits, object) of the best hit of the ray's pieces; pieces pending | tie << 32} */
    __shared__ unsigned long long s_slot[STEAL ? 2 : 1][STEAL ? CRT_BLOCK : 1];
    const unsigned gtid = blockIdx.x * CRT_BLOCK + threadIdx.x;
    LaneStackT<LDS_D> st;
    st.lds = s_stack;
    st.ovf_word = la.ovf_word + gtid;
    st.ovf_key = la.ovf_key + gtid;
    st.ovf_stride = la.grid_threads;
    st.sp = 0;
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;

    bool alive = false;
    unsigned ri = 0, cur = 0;
    unsigned pl = 0;              /* PENDING: leaf word whose objects are being tested (count field 0 = none) */
    /* tail splitting (see the block at the top of the loop): bottom of this lane's stack (entries below were given away), the group
     * this lane's piece belongs to, and the final-order index of the best object found (what the pieces' results are merged on) */
    /* bit 0: piece of a split ray; bit 1: never split this ray (re-walk after a tie); bits 8..13: slot.  In a piece the best
     * object's final-order index rides in best.part (0 for every triangle, and written as 0); best.obj < 0 with best_t < TMAX
     * means the bound came from another piece. */
    unsigned grp = 0;
#if CRT_STEAL_BOTTOM
    #define SBASE ((int)(grp >> 16))          /* bits 16..: bottom of this lane's stack (entries below were given away) */
#else
    #define SBASE 0
#endif
    DRay ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = ray.dy = ray.dz = 1.0f;
    float ia = 1.0f, ib = 1.0f, ic = 1.0f;
    bool px = true, walking = false;
    /* byte offsets, inside a node, of the plane rows the ray enters / leaves through: the
     * `d >= 0 ? min : max` selects of bvh4.h:43-75 done once per ray instead of once per box */
    unsigned onx = 0, ofx = 48, ony = 16, ofy = 64, onz = 32, ofz = 80;
    unsigned pbits = 7;           /* bit a = (d[a] > 0) */
    bool clean = true;            /* no product of the slab test can be NaN for this ray */
    bool wave_clean = true;       /* wave-uniform: every ray loaded so far into this wave was clean */
    float best_t = CRT_TMAX;
    float cur_min = CRT_TMAX;     /* EXACT: min_t of the innermost open node visit */
    DHit best; best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;

    /* The children whose keys pass are pushed in reverse visiting order and the first becomes `cur`; false = none passed. */
    auto order_push = [&](int ax2, float k0, float k1, float k2, float k3, unsigned w0, unsigned w1, unsigned w2, unsigned w3)
                      __attribute__((always_inline)) -> bool {
        const bool p2 = ((pbits >> ax2) & 1u) != 0u;                      /* bvh4.h:323,:334: d[axis2] > 0 */
        /* visiting sequence: px ? (0, 1, C, D) : (C, D, 1, 0) with (C, D) = p2 ? (2, 3) : (3, 2) */
        const float kC = p2 ? k2 : k3, kD = p2 ? k3 : k2;
        const unsigned wC = p2 ? w2 : w3, wD = p2 ? w3 : w2;
        const float s0k = px ? k0 : kC, s1k = px ? k1 : kD, s2k = px ? kC : k1, s3k = px ? kD : k0;
        const unsigned s0w = px ? w0 : wC, s1w = px ? w1 : wD, s2w = px ? wC : w1, s3w = px ? wD : w0;
        const float thr = EXACT ? cur_min : best_t;                       /* EXACT: TMAX here, the visit has just opened */
        if (s3k < thr) st.push(s3w, s3k);
        if (s2k < thr) st.push(s2w, s2k);
        if (s1k < thr) st.push(s1w, s1k);
        if (s0k < thr) { cur = s0w; return true; }
        return false;
    };
    /* One node visit of the lane's `cur`: the four keys, then order_push.  Returns false when no child passes (the lane has to pop). */
    auto node_step = [&]() __attribute__((always_inline)) -> bool {
        /* inner child word = (axis2 of that node) << 28 | node index (< 2^25: 32-bit byte offsets) */
        const int ax2 = (int)((cur >> 28) & 3u);
        float k0, k1, k2, k3;
        unsigned w0, w1, w2, w3;
        if (COUNT) c_nodes++;
        if (COMPRESSED && wave_clean) {
            uint4 wds;
            cnode_keys(sc.cnodes4, (cur & CRT_NODE_MASK) << 6, ray, ia, ib, ic, onx == 0u, ony == 16u, onz == 32u, &k0, &k1, &k2, &k3, &wds);
            w0 = wds.x; w1 = wds.y; w2 = wds.z; w3 = wds.w;
            /* keep the child words of the one 16-byte load: without this the compiler merges them with the full-node
             * branch's words as a select of ADDRESSES and loads the four words again, one dword each, after the keys */
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
        } else {
            const unsigned nb = (cur & CRT_NODE_MASK) << 7;
            const char* nbase = reinterpret_cast<const char*>(sc.nodes4);
            const float4 nx = *reinterpret_cast<const float4*>(nbase + (nb + onx));
            const float4 fx = *reinterpret_cast<const float4*>(nbase + (nb + ofx));
            const float4 ny = *reinterpret_cast<const float4*>(nbase + (nb + ony));
            const float4 fy = *reinterpret_cast<const float4*>(nbase + (nb + ofy));
            const float4 nz = *reinterpret_cast<const float4*>(nbase + (nb + onz));
            const float4 fz = *reinterpret_cast<const float4*>(nbase + (nb + ofz));
            const float4 q6 = *reinterpret_cast<const float4*>(nbase + (nb + 96u));
            if (EXACT) { st.push(CRT_FRAME_MARK, cur_min); cur_min = CRT_TMAX; }      /* a new node visit opens: bvh4.h:354 */
            if (wave_clean && !COMPRESSED) {
                box4_keys_fast2(v2f{nx.x, nx.y}, v2f{ny.x, ny.y}, v2f{nz.x, nz.y}, v2f{fx.x, fx.y}, v2f{fy.x, fy.y}, v2f{fz.x, fz.y},
                                ray, ia, ib, ic, &k0, &k1);
                box4_keys_fast2(v2f{nx.z, nx.w}, v2f{ny.z, ny.w}, v2f{nz.z, nz.w}, v2f{fx.z, fx.w}, v2f{fy.z, fy.w}, v2f{fz.z, fz.w},
                                ray, ia, ib, ic, &k2, &k3);
            } else {
                k0 = box4_key_sse(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, ray, ia, ib, ic);
                k1 = box4_key_sse(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, ray, ia, ib, ic);
                k2 = box4_key_sse(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, ray, ia, ib, ic);
                k3 = box4_key_sse(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, ray, ia, ib, ic);
            }
            w0 = __float_as_uint(q6.x); w1 = __float_as_uint(q6.y);
            w2 = __float_as_uint(q6.z); w3 = __float_as_uint(q6.w);
        }
        return order_push(ax2, k0, k1, k2, k3, w0, w1, w2, w3);
    };
    /* the object at objects[first] against the ray: strict `<`, array order (bvh4.h:374-383) */
    auto object_step = [&](int first) __attribute__((always_inline)) {
        DHit h;
        const float t = dev_object<TRI_ONLY>(sc, first, ray, false, &h);
        if (COUNT) c_prims++;
        if (t < best_t || (STEAL && (grp & 1u) && best.obj < 0 && t == best_t && t < CRT_TMAX)) {
            best_t = t; best = h;
            if (STEAL) {
                best.part = first;
                if (grp & 1u) {                                    /* a piece of a split ray: the other pieces prune with this hit from now on */
                    const unsigned si = (threadIdx.x & ~63u) + ((grp >> 8) & 63u);
                    const unsigned long long old = atomicMin(&s_slot[0][si], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)first);
                    if ((unsigned)(old >> 32) == __float_as_uint(t) && (unsigned)old != (unsigned)first) atomicOr(&s_slot[1][si], 1ull << 32);
                }
            }
        }
        if (EXACT && t < cur_min) cur_min = t;                             /* bvh4.h:378-382, the open node's min_t */
    };
    /* pops until an entry survives the cull against the nearest hit so far; false = the stack is empty */
    auto pop_next = [&]() __attribute__((always_inline)) -> bool {
        while (st.sp > SBASE) {
            unsigned w; float key;
            st.pop(&w, &key);
            if (EXACT && w == CRT_FRAME_MARK) {                            /* the node visit closes: bvh4.h:366-370 */
                cur_min = cur_min < key ? cur_min : key;
                continue;
            }
            if (key < (EXACT ? cur_min : best_t)) { cur = w; return true; }
        }
        return false;
    };

    /* what a lane derives from a ray it has just been given, and the start of the walk at the root */
    auto start_ray = [&]() __attribute__((always_inline)) {
        cur_min = CRT_TMAX;
        ia = 1.0f / ray.dx; ib = 1.0f / ray.dy; ic = 1.0f / ray.dz;           /* bvh4.h:26-28 */
        const bool sx = ray.dx >= 0.0f, sy = ray.dy >= 0.0f, sz = ray.dz >= 0.0f;   /* bvh4.h:43,:54,:65 */
        onx = sx ? 0u : 48u;  ofx = 48u - onx;
        ony = sy ? 16u : 64u; ofy = 80u - ony;
        onz = sz ? 32u : 80u; ofz = 112u - onz;
        px = ray.dx > 0.0f;                                                    /* bvh4.h:312 (axis0 = axis1 = 0) */
        pbits = (px ? 1u : 0u) | (ray.dy > 0.0f ? 2u : 0u) | (ray.dz > 0.0f ? 4u : 0u);
        /* finite non-zero 1/d and a finite origin: every (plane - o) * (1/d) is then a number
         * (possibly +-inf), never NaN, and max3 / min3 equal the SSE max / min chains */
        clean = fabsf(ia) < CRT_TMAX && fabsf(ib) < CRT_TMAX && fabsf(ic) < CRT_TMAX &&
                fabsf(ray.ox) < 1.0e30f && fabsf(ray.oy) < 1.0e30f && fabsf(ray.oz) < 1.0e30f &&
                ia != 0.0f && ib != 0.0f && ic != 0.0f;
        /* compressed nodes: (origin - o) * (1/d) and 255 * scale * (1/d) must stay far from overflow (boxes are < 1e30) */
        if (COMPRESSED) clean = clean && fabsf(ia) < 1.0e6f && fabsf(ib) < 1.0e6f && fabsf(ic) < 1.0e6f;
        best_t = CRT_TMAX;
        best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;
        st.sp = 0; grp = 0u;
        cur = sc.root_word;
        pl = 0u;
        walking = sc.n_accel > 0;
        alive = true;
    };

    RayFeed<false> feed;
    feed.start(la);
#if CRT_DIAG
    unsigned d_it = 0, d_alive = 0, d_node = 0, d_obj = 0, d_pop = 0, d_stall = 0, d_nodeit = 0, d_objit = 0, d_walk = 0;
    unsigned d_ray_it = 0;        /* iterations the lane's current ray has been in flight */
    unsigned d_tail_it = 0, d_tail_alive = 0, d_steals = 0;
    unsigned d_uniform = 0, d_distinct = 0;      /* node-step iterations whose lanes all visit ONE node; distinct nodes per node-step iteration, summed */
#endif
#if CRT_TIMELINE
    /* wall clock (100 MHz, one counter for the whole device) when this wave started, when its queue ran dry, when it ended */
    const unsigned long long d_t0 = wall_clock64();
    unsigned long long d_tdry = 0ull;
#endif
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !feed.empty && __popcll(idle) >= CRT_REFILL_MIN) {
            float unused_ld;
            if (feed.take(la, idle, !alive, &ri, &ray, &unused_ld)) start_ray();
            if (__ballot(alive && !clean) != 0ull) wave_clean = false;
        }
        /* ---- tail splitting ----------------------------------------------------------------------------------------------
         * Once the queue has nothing more for this wave, its lanes run dry one by one while the longest rays walk on alone
         * (per wave 137 iterations on average and up to 476 on the 10.5 M-triangle scene, with 12 lanes alive).  A pending
         * stack entry is an independent
         * subtree, so in that phase an idle lane takes the TOP entry (the subtree the walk would enter next: the one most likely to be
         * needed -- the bottom ones are mostly culled once a near hit is found) of a lane that still has some, with a copy of the
         * ray and of its best t as a pruning bound.  The pieces of a ray form a group with one slot in memory: every hit a piece
         * finds goes into it with a 64-bit atomic min on (t bits, final-order object index), every piece reads the slot's t once
         * per iteration and prunes with it, and the piece that finishes last re-runs the triangle test on the winner and
         * writes the hit record.  The
         * nearest t is independent of the order in which subtrees are walked; WHICH of several exactly-equal-t objects wins
         * is not, so when two pieces report the same t for different objects the ray is walked again from the root, by one
         * lane, unsplit (ties are a handful per 2^24 rays). */
        if (STEAL && feed.empty) {
            /* givers: at least two pending entries, all in LDS, the top one an inner node (a leaf is not worth a hand-over) */
#if CRT_STEAL_BOTTOM
            bool can_give = alive && !(grp & 2u) && st.sp > SBASE + 1 && SBASE < LDS_D;
            if (can_give) can_give = !(s_stack[SBASE][threadIdx.x].x & CRT_LEAF_BIT);
#else
            bool can_give = alive && !(grp & 2u) && st.sp > 1 && st.sp <= LDS_D;
            if (can_give) can_give = !(s_stack[st.sp - 1][threadIdx.x].x & CRT_LEAF_BIT);
#endif
            const unsigned long long give_m = __ballot(can_give), idle_m = __ballot(!alive);
            /* a hand-over round costs the whole wave about as much as an iteration of the walk (ballots, a rank search, a
             * dozen lane permutes, the slot's atomics), so rounds are held until CRT_SPLIT_MIN_IDLE lanes are idle */
            if (give_m != 0ull && __popcll(idle_m) >= CRT_SPLIT_MIN_IDLE) {
                const unsigned n_give = (unsigned)__popcll(give_m), n_idle = (unsigned)__popcll(idle_m);
                const unsigned n_pair = n_give < n_idle ? n_give : n_idle;
                const unsigned lane = threadIdx.x & 63u;
                const unsigned wave0 = threadIdx.x & ~63u;          /* this wave's 64 slots */
                const bool giving = can_give && lane_rank(give_m) < n_pair;
                unsigned e_w = 0u, e_k = 0u;
                if (giving) {
#if CRT_STEAL_BOTTOM
                    const StackEntry e = s_stack[SBASE][threadIdx.x];
                    grp += 1u << 16;
#else
                    st.sp--;
                    const StackEntry e = s_stack[st.sp][threadIdx.x];
#endif
                    e_w = e.x; e_k = e.y;
                    if (!(grp & 1u)) {                               /* first split of this ray: open its group in this lane's slot */
                        grp |= 1u | (lane << 8);
                        const unsigned long long key0 = best.obj >= 0 ? (((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned)best.part) : ~0ull;
                        s_slot[0][wave0 + lane] = key0;
                        s_slot[1][wave0 + lane] = 1ull;              /* pieces pending: this one */
                        CRT_WG_FENCE();
                    }
                    atomicAdd(&s_slot[1][wave0 + ((grp >> 8) & 63u)], 1ull);
                    CRT_WG_FENCE();
                }
                /* the k-th idle lane takes from the k-th giving lane */
                unsigned k = lane_rank(idle_m), src = 0u;
                {
                    unsigned long long m = give_m;
                    for (unsigned sh = 32u; sh >= 1u; sh >>= 1) {
                        const unsigned c = (unsigned)__popcll(m & ((1ull << sh) - 1ull));
                        if (k >= c) { k -= c; m >>= sh; src += sh; }
                    }
                }
                const int a = (int)((src & 63u) << 2);
                const bool taking = !alive && lane_rank(idle_m) < n_pair;
                /* every value moves straight into the taking lane's own register (givers are alive lanes, takers idle ones: no
                 * lane is both), so the copies do not pile up in registers */
                #define CRT_TAKE_U(x) { const unsigned t_ = (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)(x)); if (taking) (x) = t_; }
                #define CRT_TAKE_F(x) { const float t_ = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(x))); if (taking) (x) = t_; }
                CRT_TAKE_U(ri)
                CRT_TAKE_F(ray.ox) CRT_TAKE_F(ray.oy) CRT_TAKE_F(ray.oz) CRT_TAKE_F(ray.dx) CRT_TAKE_F(ray.dy) CRT_TAKE_F(ray.dz)
                const float v_bt = __int_as_float(__builtin_amdgcn_ds_bpermute(a, __float_as_int(best_t)));
                const unsigned v_grp = (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)grp);
                /* what start_ray derives from the ray comes along too (three correctly rounded divisions otherwise) */
                CRT_TAKE_F(ia) CRT_TAKE_F(ib) CRT_TAKE_F(ic)
                const unsigned v_flags = (unsigned)__builtin_amdgcn_ds_bpermute(a, (int)(pbits | (clean ? 8u : 0u)));
                CRT_TAKE_U(e_w) CRT_TAKE_U(e_k)
                #undef CRT_TAKE_U
                #undef CRT_TAKE_F
#if CRT_DIAG
                if (COUNT) d_steals += n_pair;
#endif
                if (taking) {
                    {
                        const bool sx = ray.dx >= 0.0f, sy = ray.dy >= 0.0f, sz = ray.dz >= 0.0f;
                        onx = sx ? 0u : 48u;  ofx = 48u - onx;
                        ony = sy ? 16u : 64u; ofy = 80u - ony;
                        onz = sz ? 32u : 80u; ofz = 112u - onz;
                        pbits = v_flags & 7u; px = (pbits & 1u) != 0u; clean = (v_flags & 8u) != 0u;
                        cur_min = CRT_TMAX;
                        best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;
                        st.sp = 0; pl = 0u;
                        alive = true;
                    }
                    grp = v_grp & 0x3FFFu;
                    best_t = v_bt;                                   /* a bound to prune with; `best` stays empty until this piece finds a hit */
                    cur = e_w;
                    walking = __uint_as_float(e_k) < v_bt;
                }
                if (__ballot(alive && !clean) != 0ull) wave_clean = false;
            }
        }
        if (__ballot(alive) == 0ull) break;           /* nothing in flight: the queue is drained */
#if CRT_TIMELINE
        if (COUNT && feed.empty && d_tdry == 0ull) d_tdry = wall_clock64();
#endif
#if CRT_DIAG
        if (COUNT && feed.empty) { d_tail_it++; d_tail_alive += (unsigned)__popcll(__ballot(alive)); }
        if (COUNT && alive) d_ray_it++;
        if (COUNT) {
            const bool inner = alive && walking && !(cur & CRT_LEAF_BIT);
            const bool leafcur = alive && walking && (cur & CRT_LEAF_BIT);
            const bool plbusy = ((pl >> 28) & 7u) != 0u;
            const unsigned n_in = (unsigned)__popcll(__ballot(inner));
            const unsigned n_ob = (unsigned)__popcll(__ballot(alive && (PENDING ? plbusy : leafcur)));
            d_it++; d_alive += (unsigned)__popcll(__ballot(alive)); d_node += n_in; d_obj += n_ob;
            d_stall += (unsigned)__popcll(__ballot(leafcur && plbusy));
            d_walk += (unsigned)__popcll(__ballot(alive && walking));
            d_nodeit += n_in ? 1u : 0u; d_objit += n_ob ? 1u : 0u;
            {
                unsigned long long left = __ballot(inner);
                unsigned distinct = 0;
                while (left) {
                    const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)cur, (int)__builtin_ctzll(left));
                    left &= ~__ballot(inner && cur == v);
                    distinct++;
                }
                d_distinct += distinct; d_uniform += distinct == 1u ? 1u : 0u;
            }
        }
#endif
        if (alive) {
            if (PENDING) {
                /* Each iteration a lane may take a node step AND test one object.  A leaf reached by the walk moves into `pl`
                 * (when `pl` is free) and the walk pops on at once; `pl` gives one object per iteration, in array order.
                 * Objects are therefore tested in exactly the order of the plain walk -- leaves enter `pl` one at a time, in
                 * walk order, and a second leaf waits in `cur` until the first is finished -- so the nearest hit AND which of
                 * several equal-t hits is kept are unchanged; only the boxes tested while `pl` still holds objects are culled
                 * against a nearest hit that is up to three objects stale, i.e. a few more children get pushed (they are culled
                 * again, against the current value, when popped). */
                bool need_pop = false;
                unsigned long long group_key = ~0ull;
                if (STEAL && (grp & 1u))                             /* what the ray's other pieces have found so far (read past the L1) */
                    group_key = *(volatile unsigned long long*)&s_slot[0][(threadIdx.x & ~63u) + ((grp >> 8) & 63u)];
                if (walking) {
                    if (!(cur & CRT_LEAF_BIT)) need_pop = !node_step();
                    if ((cur & CRT_LEAF_BIT) && !need_pop && ((pl >> 28) & 7u) == 0u) { pl = cur; need_pop = true; }
                }
                if (STEAL && (grp & 1u) && __uint_as_float((unsigned)(group_key >> 32)) < best_t && group_key != ~0ull) {
                    best_t = __uint_as_float((unsigned)(group_key >> 32));
                    best.obj = -1;                                   /* a bound, not this piece's hit */
                }
                const unsigned pcount = (pl >> 28) & 7u;
                if (pcount > 0u) {
                    const unsigned first = pl & 0x0FFFFFFFu;
                    object_step((int)first);
                    pl = pcount > 1u ? (CRT_LEAF_BIT | ((pcount - 1u) << 28) | (first + 1u)) : 0u;
                }
                if (need_pop) walking = pop_next();
                if (!walking && ((pl >> 28) & 7u) == 0u) {
                    alive = false;
                    if (STEAL && (grp & 1u)) {
                        /* a piece of a split ray: merge, and if it is the last one, settle the ray */
                        const unsigned si = (threadIdx.x & ~63u) + ((grp >> 8) & 63u);
                        CRT_WG_FENCE();                              /* this piece's hits went into the slot as they were found */
                        const unsigned long long before = atomicAdd(&s_slot[1][si], ~0ull);     /* pending - 1 */
                        if ((unsigned)before == 1u) {
                            CRT_WG_FENCE();
                            const unsigned long long key = atomicMin(&s_slot[0][si], ~0ull);
                            const bool tie = (atomicOr(&s_slot[1][si], 0ull) >> 32) != 0ull;
                            if (tie) {                               /* equal t from two pieces: which object wins is a matter of walk order */
                                start_ray();
                                grp = 2u;
                            } else {
                                DHit h; h.obj = -1; h.part = -1; h.beta = 0.0f; h.gamma = 0.0f;
                                float t = CRT_TMAX;
                                if (key != ~0ull) t = dev_object<TRI_ONLY>(sc, (int)(unsigned)key, ray, false, &h);
                                store_hit(la.hits, ri, t, h);
                                if (COUNT) c_rays++;
                            }
                        }
                    } else {
                        if (STEAL) best.part = 0;
                        linear_closest<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, &best_t, &best, &c_prims);
                        store_hit(la.hits, ri, best_t, best);
                        if (COUNT) c_rays++;
                    }
#if CRT_DIAG
                    if (COUNT) {                                   /* histogram of iterations per ray: bucket = floor(log2), and the maximum */
                        atomicAdd(&sc.counters[20 + (31 - __clz((int)(d_ray_it | 1u)) < 11 ? 31 - __clz((int)(d_ray_it | 1u)) : 11)], 1ull);
                        atomicMax(&sc.counters[19], (unsigned long long)d_ray_it);
                        d_ray_it = 0;
                    }
#endif
                }
            } else {
                if (walking) {
                    bool need_pop = true;
                    if (!(cur & CRT_LEAF_BIT)) need_pop = !node_step();
                    /* a lane whose node step went straight into a leaf tests that leaf's first object in the same iteration;
                     * one object of a leaf per iteration: lanes in leaves of different sizes stay in step with the others */
                    if (cur & CRT_LEAF_BIT) {
                        const int count = (int)((cur >> 28) & 7u);
                        const int first = (int)(cur & 0x0FFFFFFFu);
                        need_pop = true;
                        if (count > 0) object_step(first);
                        if (count > 1) { cur = CRT_LEAF_BIT | ((unsigned)(count - 1) << 28) | (unsigned)(first + 1); need_pop = false; }
                    }
                    if (need_pop) walking = pop_next();
                }
                if (!walking) {
                    linear_closest<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, &best_t, &best, &c_prims);
                    store_hit(la.hits, ri, best_t, best);
                    if (COUNT) c_rays++;
                    alive = false;
                }
            }
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, false);
#if CRT_DIAG
    if (COUNT && (threadIdx.x & 63) == 0) {
        const unsigned v[9] = { d_it, d_alive, d_node, d_obj, d_pop, d_stall, d_nodeit, d_objit, d_walk };
        for (int k = 0; k < 9; k++) atomicAdd(&sc.counters[8 + k], (unsigned long long)v[k]);
        atomicAdd(&sc.counters[4], (unsigned long long)d_tail_it);
        atomicAdd(&sc.counters[5], (unsigned long long)d_tail_alive);
        atomicAdd(&sc.counters[6], (unsigned long long)d_steals);
        atomicMax(&sc.counters[7], (unsigned long long)d_tail_it);
        atomicAdd(&sc.counters[17], (unsigned long long)d_uniform);
        atomicAdd(&sc.counters[18], (unsigned long long)d_distinct);
    }
#endif
#if CRT_TIMELINE
    if (COUNT && (threadIdx.x & 63) == 0) {
        const unsigned long long d_t1 = wall_clock64();
        if (d_tdry == 0ull) d_tdry = d_t1;
        atomicMax(&sc.counters[32], ~d_t0);   atomicMax(&sc.counters[33], d_t0);       /* first / last wave start */
        atomicMax(&sc.counters[34], ~d_tdry); atomicMax(&sc.counters[35], d_tdry);     /* first / last queue-dry */
        atomicMax(&sc.counters[36], ~d_t1);   atomicMax(&sc.counters[37], d_t1);       /* first / last wave end */
        atomicAdd(&sc.counters[38], d_t1 - d_tdry);                                     /* summed tail time of the waves */
        atomicAdd(&sc.counters[39], 1ull);
    }
#endif
}

template <bool TRI_ONLY, bool COUNT, bool EXACT = false>
__global__ __launch_bounds__(CRT_BLOCK) CRT_WALK_WAVES void k_bvh4_closest(DevScene sc, LaunchArgs la)
{
    bvh4_closest_body<TRI_ONLY, COUNT, EXACT, false>(sc, la);
}
/* The headline kernel: triangle-only scene, pruned walk over the conservative nodes, pending leaf, tail splitting.  A kernel of
 * its own so that its register budget (five waves per SIMD: 96 VGPRs; the tail-splitting block would otherwise take it to 104
 * and the fifth wave is worth 8 %) does not bind the other instantiations of the body. */
template <bool COUNT, bool SPLIT = false>
__global__ __launch_bounds__(CRT_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_bvh4_closest_tri(DevScene sc, LaunchArgs la)
{
    bvh4_closest_body<true, COUNT, false, true, SPLIT>(sc, la);
}

/* =====================================================================================
 * QBVH any hit: BVH4ShadowIntersectTest (bvh4.h:392-478) + BVH4 branch of shadowIntersectTest
 * (intersect.h:470-485).  Same order; bound is the constant light_dist; first t < light_dist
 * in traversal order ends the walk (so the reported t is the reference's, quirk (5)).
 * ===================================================================================== */
template <bool TRI_ONLY, bool COUNT, bool COMPRESSED = false>
__global__ __launch_bounds__(CRT_BLOCK) CRT_WALK_WAVES void k_bvh4_shadow(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    __shared__ unsigned s_stack[CRT_LDS_DEPTH32][CRT_BLOCK];
    const unsigned gtid = blockIdx.x * CRT_BLOCK + threadIdx.x;
    LaneStack32 st;
    st.lds = s_stack;
    st.ovf_word = la.ovf_word + gtid;
    st.ovf_stride = la.grid_threads;
    st.sp = 0;
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;

    bool alive = false;
    unsigned ri = 0, cur = 0;
    unsigned pl = 0;              /* CRT_PENDING_LEAF_SHADOW: leaf word whose objects are being tested (count field 0 = none) */
    DRay ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = ray.dy = ray.dz = 1.0f;
    float ia = 1.0f, ib = 1.0f, ic = 1.0f, ld = 0.0f, result = CRT_TMAX;
    bool px = true, walking = false;
    unsigned onx = 0, ofx = 48, ony = 16, ofy = 64, onz = 32, ofz = 80, pbits = 7;   /* as in k_bvh4_closest */
    bool clean = true, wave_clean = true;
    const float ninf = -__builtin_huge_valf();

    RayFeed<true> feed;
    feed.start(la);
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !feed.empty && __popcll(idle) >= CRT_REFILL_MIN_SHADOW) {
            if (feed.take(la, idle, !alive, &ri, &ray, &ld)) {
                ia = 1.0f / ray.dx; ib = 1.0f / ray.dy; ic = 1.0f / ray.dz;
                const bool sx = ray.dx >= 0.0f, sy = ray.dy >= 0.0f, sz = ray.dz >= 0.0f;
                onx = sx ? 0u : 48u;  ofx = 48u - onx;
                ony = sy ? 16u : 64u; ofy = 80u - ony;
                onz = sz ? 32u : 80u; ofz = 112u - onz;
                px = ray.dx > 0.0f;
                pbits = (px ? 1u : 0u) | (ray.dy > 0.0f ? 2u : 0u) | (ray.dz > 0.0f ? 4u : 0u);
                clean = fabsf(ia) < CRT_TMAX && fabsf(ib) < CRT_TMAX && fabsf(ic) < CRT_TMAX &&
                        fabsf(ray.ox) < 1.0e30f && fabsf(ray.oy) < 1.0e30f && fabsf(ray.oz) < 1.0e30f &&
                        ia != 0.0f && ib != 0.0f && ic != 0.0f;
                if (COMPRESSED) clean = clean && fabsf(ia) < 1.0e6f && fabsf(ib) < 1.0e6f && fabsf(ic) < 1.0e6f;
                result = CRT_TMAX;
                st.sp = 0;
                cur = sc.root_word;
                pl = 0u;
                walking = sc.n_accel > 0;
                alive = true;
            }
            if (__ballot(alive && !clean) != 0ull) wave_clean = false;
        }
        if (__ballot(alive) == 0ull) break;
        if (alive) {
            constexpr bool PEND = TRI_ONLY && COMPRESSED && CRT_PENDING_LEAF_SHADOW;
            bool need_pop = true, found = false;
            if (walking) {
                if (PEND && (cur & CRT_LEAF_BIT)) need_pop = false;       /* a leaf that has been waiting for `pl` to become free */
                if (!(cur & CRT_LEAF_BIT)) {
                    const int ax2 = (int)((cur >> 28) & 3u);
                    if (COUNT) c_nodes++;
                    float k0, k1, k2, k3;
                    unsigned w0, w1, w2, w3;
                    bool v0, v1, v2, v3;
                    if (COMPRESSED && wave_clean) {
                        /* conservative entry distances (t0 <= the reference's): visited iff the reference's test could pass,
                         * bvh4.h:450 with box_t = t0 or "origin inside" (t0 < eps) */
                        uint4 wds;
                        cnode_keys(sc.cnodes4, (cur & CRT_NODE_MASK) << 6, ray, ia, ib, ic, onx == 0u, ony == 16u, onz == 32u, &k0, &k1, &k2, &k3, &wds);
                        w0 = wds.x; w1 = wds.y; w2 = wds.z; w3 = wds.w;
                        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
                        const float bound = ld > CRT_K_EPSILON ? ld : CRT_K_EPSILON;
                        v0 = k0 < bound; v1 = k1 < bound; v2 = k2 < bound; v3 = k3 < bound;
                    } else {
                    const unsigned nb = (cur & CRT_NODE_MASK) << 7;
                    const char* nbase = reinterpret_cast<const char*>(sc.nodes4);
                    const float4 nx = *reinterpret_cast<const float4*>(nbase + (nb + onx));
                    const float4 fx = *reinterpret_cast<const float4*>(nbase + (nb + ofx));
                    const float4 ny = *reinterpret_cast<const float4*>(nbase + (nb + ony));
                    const float4 fy = *reinterpret_cast<const float4*>(nbase + (nb + ofy));
                    const float4 nz = *reinterpret_cast<const float4*>(nbase + (nb + onz));
                    const float4 fz = *reinterpret_cast<const float4*>(nbase + (nb + ofz));
                    const float4 q6 = *reinterpret_cast<const float4*>(nbase + (nb + 96u));
                    if (wave_clean && !COMPRESSED) {
                        box4_keys_fast2(v2f{nx.x, nx.y}, v2f{ny.x, ny.y}, v2f{nz.x, nz.y}, v2f{fx.x, fx.y}, v2f{fy.x, fy.y}, v2f{fz.x, fz.y},
                                        ray, ia, ib, ic, &k0, &k1);
                        box4_keys_fast2(v2f{nx.z, nx.w}, v2f{ny.z, ny.w}, v2f{nz.z, nz.w}, v2f{fx.z, fx.w}, v2f{fy.z, fy.w}, v2f{fz.z, fz.w},
                                        ray, ia, ib, ic, &k2, &k3);
                    } else {
                        k0 = box4_key_sse(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, ray, ia, ib, ic);
                        k1 = box4_key_sse(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, ray, ia, ib, ic);
                        k2 = box4_key_sse(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, ray, ia, ib, ic);
                        k3 = box4_key_sse(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, ray, ia, ib, ic);
                    }
                    /* bvh4.h:450: bbox_t < light_dist || (less_than && bbox_t < TMAX); key = -inf encodes the latter */
                    v0 = k0 < ld || k0 == ninf; v1 = k1 < ld || k1 == ninf;
                    v2 = k2 < ld || k2 == ninf; v3 = k3 < ld || k3 == ninf;
                    w0 = __float_as_uint(q6.x); w1 = __float_as_uint(q6.y);
                    w2 = __float_as_uint(q6.z); w3 = __float_as_uint(q6.w);
                    }
                    const bool p2 = ((pbits >> ax2) & 1u) != 0u;
                    const bool vC = p2 ? v2 : v3, vD = p2 ? v3 : v2;
                    const unsigned wC = p2 ? w2 : w3, wD = p2 ? w3 : w2;
                    const bool s0v = px ? v0 : vC, s1v = px ? v1 : vD, s2v = px ? vC : v1, s3v = px ? vD : v0;
                    const unsigned s0w = px ? w0 : wC, s1w = px ? w1 : wD, s2w = px ? wC : w1, s3w = px ? wD : w0;
                    if (s3v) st.push(s3w);
                    if (s2v) st.push(s2w);
                    if (s1v) st.push(s1w);
                    if (s0v) { cur = s0w; need_pop = false; }
                }
                if (PEND) {
                    /* as in the closest-hit walk: the leaf being tested sits in a register of its own (`pl`) and gives one object
                     * per iteration while the walk goes on.  Leaves enter `pl` one at a time, in walk order (a second one waits in
                     * `cur`), so objects are tested in the reference's order and the first t < light_dist found is the reference's. */
                    if (!need_pop && (cur & CRT_LEAF_BIT) && ((pl >> 28) & 7u) == 0u) { pl = cur; need_pop = true; }
                } else
#if CRT_FUSE_LEAF_SHADOW
                if (cur & CRT_LEAF_BIT) {
                    need_pop = true;
#else
                if (!!(cur & CRT_LEAF_BIT) == need_pop) {                  /* (a leaf that the node step has not just entered) */
#endif
                    /* one object of the leaf per step, in array order (bvh4.h:464-472) */
                    const int count = (int)((cur >> 28) & 7u);
                    const int first = (int)(cur & 0x0FFFFFFFu);
                    if (count > 0) {
                        DHit h;
                        const float t = dev_object<TRI_ONLY>(sc, first, ray, true, &h);
                        if (COUNT) c_prims++;
                        if (t < ld) { result = t; walking = false; need_pop = false; }
                    }
                    if (walking && count > 1) { cur = CRT_LEAF_BIT | ((unsigned)(count - 1) << 28) | (unsigned)(first + 1); need_pop = false; }
                }
            }
            if (PEND) {
                const unsigned pcount = (pl >> 28) & 7u;
                if (pcount > 0u) {
                    const unsigned first = pl & 0x0FFFFFFFu;
                    DHit h;
                    const float t = dev_object<TRI_ONLY>(sc, (int)first, ray, true, &h);
                    if (COUNT) c_prims++;
                    pl = pcount > 1u ? (CRT_LEAF_BIT | ((pcount - 1u) << 28) | (first + 1u)) : 0u;
                    if (t < ld) { result = t; found = true; walking = false; pl = 0u; }
                }
            }
            if (walking && need_pop) {
                if (st.sp > 0) cur = st.pop();
                else walking = false;
            }
            if (!walking && (!PEND || found || ((pl >> 28) & 7u) == 0u)) {
                if (result >= ld)                                                      /* intersect.h:474 */
                    result = linear_shadow<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, ld, result, &c_prims);
                if (la.t_out) la.t_out[ri] = result;
                if (la.occ_out) la.occ_out[ri] = (result < ld - CRT_K_EPSILON) ? 1 : 0;   /* trace.h:348 */
                if (COUNT) c_rays++;
                alive = false;
            }
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, true);
}

/* =====================================================================================
 * No accelerator: the NONE branches of intersectTest / shadowIntersectTest
 * (intersect.h:413-433, :486-498).
 * ===================================================================================== */
template <bool TRI_ONLY, bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK) void k_none_closest(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    const unsigned lane = threadIdx.x & 63;
    unsigned c_rays = 0, c_prims = 0;
    for (;;) {
        const unsigned base = next_batch(la);
        if (base >= la.n) break;
        const unsigned ri = base + lane;
        if (ri < la.n) {
            const DRay ray = load_ray(la.rays, ri);
            float best_t = CRT_TMAX;
            DHit best; best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;
            linear_closest<TRI_ONLY, COUNT>(sc, 0, sc.num_obj, ray, &best_t, &best, &c_prims);
            store_hit(la.hits, ri, best_t, best);
            if (COUNT) c_rays++;
        }
    }
    flush_counters<COUNT>(sc, c_rays, 0, c_prims, false);
}
template <bool TRI_ONLY, bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK) void k_none_shadow(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    const unsigned lane = threadIdx.x & 63;
    unsigned c_rays = 0, c_prims = 0;
    for (;;) {
        const unsigned base = next_batch(la);
        if (base >= la.n) break;
        const unsigned ri = base + lane;
        if (ri < la.n) {
            const DRay ray = load_ray(la.rays, ri);
            const float ld = la.light_dist[ri];
            const float t = linear_shadow<TRI_ONLY, COUNT>(sc, 0, sc.num_obj, ray, ld, CRT_TMAX, &c_prims);
            if (la.t_out) la.t_out[ri] = t;
            if (la.occ_out) la.occ_out[ri] = (t < ld - CRT_K_EPSILON) ? 1 : 0;
            if (COUNT) c_rays++;
        }
    }
    flush_counters<COUNT>(sc, c_rays, 0, c_prims, true);
}

/* =====================================================================================
 * Binary BVH, best-t pruned.  BVHIntersectTest (intersect.h:310-342): left then right, the
 * right subtree wins equal t (t1 < t2 ? left : right) -> a later hit replaces on `<=`.
 * A child box is skipped when the reference would skip it (rayIntersectAABB == TMAX) or when
 * its entry t0 (> eps) lies strictly beyond the nearest hit.  SHADOW = true gives
 * BVHShadowIntersectTest (intersect.h:344-362): min t over the tree, light_dist unused.
 * ===================================================================================== */
/* Device layout (crt_gpu.hip): one 64-byte record per INNER node holding both children's boxes and
 * child words, so a visit is one 64-byte read and leaves are never loaded:
 *   q0 = lmin.xyz lmax.x   q1 = lmax.yz rmin.xy   q2 = rmin.z rmax.xyz   q3 = left word, right word
 *   word = inner-record index, or CRT_LEAF_BIT | object index (final order).
 * Lanes are refilled from the queue as in k_bvh4_closest. */
/* EXACT = BVHIntersectTest / BVHShadowIntersectTest as written (intersect.h:310-362): every
 * box-overlapped subtree is walked, nothing is culled by t; `t1 < t2 ? left : right` at every level
 * makes the LAST of equal hits in left-first order win, which is the `<=` below. */
/* ORDERED (pruned mode only, and only when the host has checked that left-first depth-first order meets the leaves in
 * increasing object index — the reference's in-place builder gives that): the nearer child is walked first, so best t
 * tightens early and far subtrees fall to the pruning test.  The reference's answer is a function of the SET of
 * reachable hits, not of the walk: least t, and among equal t the last in left-first order = the largest object index;
 * the update rule below says exactly that, whatever the order the hits are met in.  A subtree is still skipped only when its
 * entry t lies strictly beyond best t, so no hit of equal t is ever lost. */
/* WIDE (with ORDERED): rays without NaN in their slab arithmetic walk the tree two levels at a time through the collapsed
 * records (sc.bnodes2w: four boxes per visit, the hit ones sorted by entry t, nearest first); the others -- a zero or
 * non-finite direction component, a non-finite origin -- walk the two-box records as before, lane by lane. */
/* WCOMP (with WIDE): the collapsed records are read from their 64-byte conservative copy (sc.bnodes2wc: the QBVH's compressed-node
 * layout, 4 x 16 bytes per visit instead of 7) -- every box of it contains the record's box, so a superset of the reference's
 * subtrees is walked, with entry distances that are never too large; the answer does not depend on the order of the walk (ORDERED
 * above), so it is the same.  Rays whose fma-form slab arithmetic could overflow (|1/d| >= 1e6) walk the two-box records. */
template <bool TRI_ONLY, bool COUNT, bool SHADOW, bool EXACT = false, bool ORDERED = false, bool WIDE = false, bool WCOMP = false>
__global__ __launch_bounds__(CRT_BLOCK) void k_bvh2(DevScene sc, LaunchArgs la)
{
    static_assert(!WCOMP || (WIDE && ORDERED && !EXACT), "the conservative records belong to the collapsed, ordered walk");
    resolve_count(la);
    __shared__ StackEntry s_stack[CRT_LDS_DEPTH][CRT_BLOCK];
    const unsigned gtid = blockIdx.x * CRT_BLOCK + threadIdx.x;
    LaneStack st;
    st.lds = s_stack;
    st.ovf_word = la.ovf_word + gtid;
    st.ovf_key = la.ovf_key + gtid;
    st.ovf_stride = la.grid_threads;
    st.sp = 0;
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;
    const float ninf = -__builtin_huge_valf();

    bool alive = false, walking = false;
    bool wide = false;                  /* WIDE: this lane's ray walks the collapsed records */
    unsigned ri = 0, cur = 0;
    DRay ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = ray.dy = ray.dz = 1.0f;
    float ia = 1.0f, ib = 1.0f, ic = 1.0f;
    float best_t = CRT_TMAX;
    unsigned best_idx = 0;
    DHit best; best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;

    RayFeed<false> feed;
    feed.start(la);
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !feed.empty && __popcll(idle) >= CRT_REFILL_MIN) {
            float unused_ld;
            if (feed.take(la, idle, !alive, &ri, &ray, &unused_ld)) {
                ia = 1.0f / ray.dx; ib = 1.0f / ray.dy; ic = 1.0f / ray.dz;           /* aabb.cpp:94,:106,:118 */
                best_t = CRT_TMAX;
                best_idx = 0;
                best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;
                st.sp = 0;
                cur = sc.root_word;
                if (WIDE) {
                    wide = fabsf(ia) < CRT_TMAX && fabsf(ib) < CRT_TMAX && fabsf(ic) < CRT_TMAX && ia != 0.0f && ib != 0.0f && ic != 0.0f &&
                           fabsf(ray.ox) < 1.0e30f && fabsf(ray.oy) < 1.0e30f && fabsf(ray.oz) < 1.0e30f;
                    if (WCOMP) wide = wide && fabsf(ia) < 1.0e6f && fabsf(ib) < 1.0e6f && fabsf(ic) < 1.0e6f;
                    if (wide) cur = sc.root_word_w;
                }
                walking = false;
                if (sc.n_accel > 0) {                                                  /* root box, intersect.h:381,:454 */
                    float t0;
                    if (COUNT) c_nodes++;
                    walking = dev_ray_aabb_inv(sc.root_bmin, sc.root_bmax, ray, ia, ib, ic, &t0) < CRT_TMAX;
                }
                alive = true;
            }
        }
        if (__ballot(alive) == 0ull) break;
        if (alive) {
            if (walking) {
                bool need_pop = true;
                /* WCOMP: a node step whose nearest box is a leaf tests that leaf's object in the same iteration (the lane would do
                 * nothing else in the next one): the leaf test below runs after the node step instead of in place of it */
                const bool leaf_first = (cur & CRT_LEAF_BIT) != 0u;
                if (!leaf_first && WCOMP && wide) {
                    if (COUNT) c_nodes += 4;
                    float key[4];
                    uint4 wds;
                    cnode_keys(sc.bnodes2wc, cur << 6, ray, ia, ib, ic, ray.dx >= 0.0f, ray.dy >= 0.0f, ray.dz >= 0.0f, &key[0], &key[1], &key[2], &key[3], &wds);
                    unsigned wd[4] = { wds.x, wds.y, wds.z, wds.w };
                    asm volatile("" : "+v"(wd[0]), "+v"(wd[1]), "+v"(wd[2]), "+v"(wd[3]));      /* the words of the one 16-byte load (see node_step) */
                    const float pinf = __builtin_huge_valf();
#pragma unroll
                    for (int c = 0; c < 4; c++) key[c] = (key[c] < CRT_TMAX && !(key[c] > best_t)) ? key[c] : pinf;   /* cnode_key: TMAX = missed */
#define CRT_CSWAP(a, b) { if (key[b] < key[a]) { const float tk = key[a]; key[a] = key[b]; key[b] = tk; const unsigned tw = wd[a]; wd[a] = wd[b]; wd[b] = tw; } }
                    CRT_CSWAP(0, 1) CRT_CSWAP(2, 3) CRT_CSWAP(0, 2) CRT_CSWAP(1, 3) CRT_CSWAP(1, 2)
#undef CRT_CSWAP
                    if (key[3] < pinf) st.push(wd[3], key[3]);
                    if (key[2] < pinf) st.push(wd[2], key[2]);
                    if (key[1] < pinf) st.push(wd[1], key[1]);
                    if (key[0] < pinf) { cur = wd[0]; need_pop = false; }
                } else if (leaf_first) {
                } else if (WIDE && wide) {
                    const float4* nd = sc.bnodes2w + (size_t)cur * 8;
                    const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5], q6 = nd[6];
                    if (COUNT) c_nodes += 4;
                    const float mnx[4] = { q0.x, q0.y, q0.z, q0.w }, mny[4] = { q1.x, q1.y, q1.z, q1.w }, mnz[4] = { q2.x, q2.y, q2.z, q2.w };
                    const float mxx[4] = { q3.x, q3.y, q3.z, q3.w }, mxy[4] = { q4.x, q4.y, q4.z, q4.w }, mxz[4] = { q5.x, q5.y, q5.z, q5.w };
                    unsigned wd[4] = { __float_as_uint(q6.x), __float_as_uint(q6.y), __float_as_uint(q6.z), __float_as_uint(q6.w) };
                    float key[4];
                    const float pinf = __builtin_huge_valf();
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float lo[3] = { mnx[c], mny[c], mnz[c] }, hi[3] = { mxx[c], mxy[c], mxz[c] };
                        float t0;
                        const float bt = dev_ray_aabb_inv(lo, hi, ray, ia, ib, ic, &t0);
                        const float kc = t0 > CRT_K_EPSILON ? t0 : ninf;
                        key[c] = (bt < CRT_TMAX && !(kc > best_t)) ? kc : pinf;           /* +inf: not to be visited */
                    }
#define CRT_CSWAP(a, b) { if (key[b] < key[a]) { const float tk = key[a]; key[a] = key[b]; key[b] = tk; const unsigned tw = wd[a]; wd[a] = wd[b]; wd[b] = tw; } }
                    CRT_CSWAP(0, 1) CRT_CSWAP(2, 3) CRT_CSWAP(0, 2) CRT_CSWAP(1, 3) CRT_CSWAP(1, 2)
#undef CRT_CSWAP
                    if (key[3] < pinf) st.push(wd[3], key[3]);
                    if (key[2] < pinf) st.push(wd[2], key[2]);
                    if (key[1] < pinf) st.push(wd[1], key[1]);
                    if (key[0] < pinf) { cur = wd[0]; need_pop = false; }
                } else {
                    const float4* nd = sc.bnodes2 + (size_t)cur * 4;
                    const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
                    if (COUNT) c_nodes += 2;
                    const float lmin[3] = { q0.x, q0.y, q0.z }, lmax[3] = { q0.w, q1.x, q1.y };
                    const float rmn[3] = { q1.z, q1.w, q2.x }, rmx[3] = { q2.y, q2.z, q2.w };
                    float lt0, rt0;
                    const float lbt = dev_ray_aabb_inv(lmin, lmax, ray, ia, ib, ic, &lt0);
                    const float rbt = dev_ray_aabb_inv(rmn, rmx, ray, ia, ib, ic, &rt0);
                    const float lkey = lt0 > CRT_K_EPSILON ? lt0 : ninf;
                    const float rkey = rt0 > CRT_K_EPSILON ? rt0 : ninf;
                    const bool lv = lbt < CRT_TMAX && (EXACT || !(lkey > best_t));
                    const bool rv = rbt < CRT_TMAX && (EXACT || !(rkey > best_t));
                    const unsigned lw = __float_as_uint(q3.x), rw = __float_as_uint(q3.y);
                    if (ORDERED && lv && rv) {
                        const bool right_first = rkey < lkey;
                        st.push(right_first ? lw : rw, right_first ? lkey : rkey);
                        cur = right_first ? rw : lw; need_pop = false;
                    } else if (lv) {
                        if (rv) st.push(rw, rkey);
                        cur = lw; need_pop = false;
                    } else if (rv) { cur = rw; need_pop = false; }
                }
                if (leaf_first || (WCOMP && wide && !need_pop && (cur & CRT_LEAF_BIT))) {
                    DHit h;
                    const float t = dev_object<TRI_ONLY>(sc, (int)(cur & ~CRT_LEAF_BIT), ray, SHADOW, &h);
                    if (COUNT) c_prims++;
                    if (SHADOW) { if (t < best_t) best_t = t; }
                    else if (ORDERED) {
                        const unsigned idx = cur & ~CRT_LEAF_BIT;
                        if (t < CRT_TMAX && (t < best_t || (t == best_t && idx > best_idx))) { best_t = t; best = h; best_idx = idx; }
                    }
                    else if (t <= best_t && t < CRT_TMAX) { best_t = t; best = h; }    /* right / later wins ties */
                    need_pop = true;
                }
                if (need_pop) {
                    walking = false;
                    while (st.sp > 0) {
                        unsigned w; float key;
                        st.pop(&w, &key);
                        if (EXACT || !(key > best_t)) { cur = w; walking = true; break; }
                    }
                }
            }
            if (!walking) {
                if (!SHADOW) {
                    linear_closest<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, &best_t, &best, &c_prims);
                    store_hit(la.hits, ri, best_t, best);
                } else {
                    const float ld = la.light_dist[ri];
                    float t = best_t;
                    if (t >= ld) t = linear_shadow<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, ld, t, &c_prims);   /* intersect.h:458-468 */
                    if (la.t_out) la.t_out[ri] = t;
                    if (la.occ_out) la.occ_out[ri] = (t < ld - CRT_K_EPSILON) ? 1 : 0;
                }
                if (COUNT) c_rays++;
                alive = false;
            }
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, SHADOW);
}

/* =====================================================================================
 * Uniform grid 3D-DDA: gridIntersectTest (intersect.h:61-194) / gridShadowIntersectTest
 * (:196-308).  Divisions, the `min(...) == obj_t` accept test and the stepping comparisons are
 * kept as written, so rays with a zero direction component follow the reference's inf/NaN flow.
 * ===================================================================================== */
struct GridWalk {
    int ix, iy, iz, sx, sy, sz, ex, ey, ez;
    float tx, ty, tz, dtx, dty, dtz;
    bool hits;
};
CRT_DEV void grid_setup(const DevScene& sc, const DRay& ray, GridWalk* w)
{
    const crt_grid& g = sc.grid;
    w->hits = false; w->ix = w->iy = w->iz = 0;
    float px = ray.ox, py = ray.oy, pz = ray.oz;
    const bool inside = !(px < g.bmin[0] || px > g.bmax[0]) && !(py < g.bmin[1] || py > g.bmax[1]) &&
                        !(pz < g.bmin[2] || pz > g.bmax[2]);                          /* aabb.cpp:171-181 */
    if (inside) w->hits = true;
    else {
        float t0e;
        const float t0 = dev_ray_aabb(g.bmin, g.bmax, ray, &t0e);
        if (t0 < CRT_TMAX) { px = ray.ox + ray.dx * t0; py = ray.oy + ray.dy * t0; pz = ray.oz + ray.dz * t0; w->hits = true; }
    }
    if (!w->hits) return;
    w->ix = (int)rclamp((px - g.bmin[0]) / g.wx * g.nx, 0.0f, (float)g.nx - 1);
    w->iy = (int)rclamp((py - g.bmin[1]) / g.wy * g.ny, 0.0f, (float)g.ny - 1);
    w->iz = (int)rclamp((pz - g.bmin[2]) / g.wz * g.nz, 0.0f, (float)g.nz - 1);
    const bool gx = ray.dx > 0, gy = ray.dy > 0, gz = ray.dz > 0;
    const float tx_min = ((gx ? g.bmin[0] : g.bmax[0]) - ray.ox) / ray.dx;
    const float ty_min = ((gy ? g.bmin[1] : g.bmax[1]) - ray.oy) / ray.dy;
    const float tz_min = ((gz ? g.bmin[2] : g.bmax[2]) - ray.oz) / ray.dz;
    const float tx_max = ((gx ? g.bmax[0] : g.bmin[0]) - ray.ox) / ray.dx;
    const float ty_max = ((gy ? g.bmax[1] : g.bmin[1]) - ray.oy) / ray.dy;
    const float tz_max = ((gz ? g.bmax[2] : g.bmin[2]) - ray.oz) / ray.dz;
    w->dtx = (tx_max - tx_min) / g.nx;
    w->dty = (ty_max - ty_min) / g.ny;
    w->dtz = (tz_max - tz_min) / g.nz;
    w->sx = gx ? 1 : -1; w->sy = gy ? 1 : -1; w->sz = gz ? 1 : -1;
    w->ex = gx ? g.nx : -1; w->ey = gy ? g.ny : -1; w->ez = gz ? g.nz : -1;
    const int tix = gx ? w->ix : g.nx - w->ix - 1;
    const int tiy = gy ? w->iy : g.ny - w->iy - 1;
    const int tiz = gz ? w->iz : g.nz - w->iz - 1;
    w->tx = tx_min + w->dtx * (tix + 1);
    w->ty = ty_min + w->dty * (tiy + 1);
    w->tz = tz_min + w->dtz * (tiz + 1);
}
CRT_DEV void grid_step(GridWalk* w)
{
    if (w->tx < w->ty && w->tx < w->tz) { w->tx += w->dtx; w->ix += w->sx; }
    else if (w->ty < w->tz) { w->ty += w->dty; w->iy += w->sy; }
    else { w->tz += w->dtz; w->iz += w->sz; }
}

template <int TRI_ONLY, bool COUNT, bool SHADOW>
__global__ __launch_bounds__(CRT_BLOCK) void k_grid(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    const unsigned lane = threadIdx.x & 63;
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;
    for (;;) {
        const unsigned base = next_batch(la);
        if (base >= la.n) break;
        const unsigned slot = base + lane;
        if (slot < la.n) {
            const unsigned ri = la.perm ? la.perm[slot] : slot;                       /* binned batches: neighbours in space walk together */
            const DRay ray = load_ray(la.rays, ri);
            const float ld = SHADOW ? la.light_dist[ri] : 0.0f;
            const float bound = ld - CRT_K_EPSILON;                                    /* intersect.h:276 */
            GridWalk w;
            grid_setup(sc, ray, &w);
            float t = CRT_TMAX;
            DHit best; best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;
            bool done = false;
            if (w.hits) {
                while (w.ix != w.ex && w.iy != w.ey && w.iz != w.ez) {
                    const int cell = w.iy * sc.grid.nz * sc.grid.nx + w.iz * sc.grid.nx + w.ix;
                    /* an empty cell only steps (its obj_t is TMAX, intersect.h:152 never accepts): the distance field says how many
                     * cells from here are empty whatever the direction, and those are stepped through without a fetch */
                    unsigned skip = sc.cell_dist[cell];
                    if (skip != 0u) {
                        do {
                            if (COUNT) c_nodes++;
                            grid_step(&w);
                        } while (--skip != 0u && w.ix != w.ex && w.iy != w.ey && w.iz != w.ez);
                        continue;
                    }
                    const int cb = sc.cell_start[cell], ce = sc.cell_start[cell + 1];
                    if (COUNT) c_nodes++;
                    if (!SHADOW) {
                        float obj_t = CRT_TMAX;
                        DHit obj_h = best;
                        for (int k = cb; k < ce; k++) {
                            DHit h;
                            const float tt = sc.cell_objs ? dev_object_rec<TRI_ONLY>(sc, sc.cell_objs + (size_t)k * CRT_OBJ_STRIDE, ray, false, &h)
                                                          : dev_object<TRI_ONLY>(sc, sc.cell_items[k], ray, false, &h);
                            if (COUNT) c_prims++;
                            if (tt < obj_t) { obj_t = tt; obj_h = h; }
                        }
                        if (rmin(obj_t, rmin(w.tx, rmin(w.ty, w.tz))) == obj_t) {       /* intersect.h:152 */
                            if (obj_t < CRT_TMAX) { best = obj_h; t = obj_t; break; }
                        }
                    } else {
                        for (int k = cb; k < ce; k++) {
                            DHit h;
                            /* through the item index: an any-hit ray tests five times the objects of a closest-hit ray, many of them
                             * again in the next cell -- one record per object stays in the caches, a copy per cell does not
                             * (654 against 688 Mrays/s) */
                            const float tt = dev_object<TRI_ONLY>(sc, sc.cell_items[k], ray, true, &h);
                            if (COUNT) c_prims++;
                            if (tt < bound) { t = tt; done = true; break; }
                        }
                        if (done) break;
                    }
                    grid_step(&w);
                }
            }
            if (!SHADOW) {
                float min_t = CRT_TMAX;
                DHit mh = best;
                linear_closest<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, &min_t, &mh, &c_prims);   /* intersect.h:177-192 */
                if (min_t < t) { best = mh; t = min_t; }
                store_hit(la.hits, ri, t, best);
            } else {
                if (!done) t = linear_shadow<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, bound, CRT_TMAX, &c_prims);  /* :298-307 */
                if (la.t_out) la.t_out[ri] = t;
                if (la.occ_out) la.occ_out[ri] = (t < ld - CRT_K_EPSILON) ? 1 : 0;
            }
            if (COUNT) c_rays++;
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, SHADOW);
}

/* The same walk with lane refill (round 2; closest hit, triangle-only scenes).  The plain kernel above keeps 64 rays per wave until
 * the longest is done; its counters say 7 % of the lanes are active per VALU instruction -- the rays' walks differ in length by orders of magnitude and each step
 * of a walk is short.  Here a lane holds one ray and does ONE thing per wave iteration:
 *   - at a cell it has not looked at: range check, the distance byte; an empty cell -> up to CRT_GRID_SKIP_STEPS DDA steps (what
 *     is left of the distance is carried over), an occupied one -> its item range;
 *   - inside an occupied cell: one object test; after the cell's last object the accept test of intersect.h:152 and a DDA step;
 *   - finished: the non-grid objects, the store, and the lane is idle until the feed hands it the next ray.
 * Cells, objects and their order per ray are exactly those of the loop above. */
#ifndef CRT_GRID_SKIP_STEPS
#define CRT_GRID_SKIP_STEPS 4
#endif
#ifndef CRT_GRID_REFILL_MIN
#define CRT_GRID_REFILL_MIN 16
#endif
template <int TRI_ONLY, bool COUNT, bool SHADOW>
__global__ __launch_bounds__(CRT_BLOCK) void k_grid_refill(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;
    bool alive = false;
    unsigned ri = 0;
    DRay ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = ray.dy = ray.dz = 1.0f;
    float ld = 0.0f, bound = 0.0f;
    GridWalk w; w.hits = false; w.ix = w.iy = w.iz = 0; w.sx = w.sy = w.sz = 1; w.ex = w.ey = w.ez = 0;
    w.tx = w.ty = w.tz = 0.0f; w.dtx = w.dty = w.dtz = 0.0f;
    int k = 0, ce = 0;                  /* objects of the current cell still to test: cell_items[k .. ce) */
    int item = 0;                       /* cell_items[k], fetched one iteration ahead so that an object step waits for one fetch, not two */
    unsigned skip_left = 0;             /* empty cells ahead that need no look */
    float obj_t = CRT_TMAX;             /* nearest hit among the current cell's objects (closest) / the accepted t */
    DHit obj_h; obj_h.obj = -1; obj_h.part = -1; obj_h.beta = 0.0f; obj_h.gamma = 0.0f;
    bool found = false;                 /* the grid walk ended with a hit (closest: accepted; shadow: below the bound) */

    RayFeed<SHADOW> feed;
    feed.start(la);
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !feed.empty && __popcll(idle) >= CRT_GRID_REFILL_MIN) {
            if (feed.take(la, idle, !alive, &ri, &ray, &ld)) {
                bound = ld - CRT_K_EPSILON;                                            /* intersect.h:276 */
                grid_setup(sc, ray, &w);
                k = ce = 0; skip_left = 0;
                obj_t = CRT_TMAX; obj_h.obj = -1; obj_h.part = -1; obj_h.beta = 0.0f; obj_h.gamma = 0.0f;
                found = false;
                alive = true;
            }
        }
        if (__ballot(alive) == 0ull) break;
        if (alive) {
            bool finish = !w.hits;
            if (!finish) {
                if (k < ce) {
                    DHit h;
                    float tt;
                    if (sc.cell_objs) {                                                /* the cell's records lie side by side: no index to fetch */
                        tt = dev_object_rec<TRI_ONLY>(sc, sc.cell_objs + (size_t)k * CRT_OBJ_STRIDE, ray, SHADOW, &h);
                        k++;
                    } else {
                        const int cur_item = item;
                        k++;
                        if (k < ce) item = sc.cell_items[k];
                        tt = dev_object<TRI_ONLY>(sc, cur_item, ray, SHADOW, &h);
                    }
                    if (COUNT) c_prims++;
                    if (!SHADOW) {
                        if (tt < obj_t) { obj_t = tt; obj_h = h; }
                        if (k == ce) {
                            if (rmin(obj_t, rmin(w.tx, rmin(w.ty, w.tz))) == obj_t && obj_t < CRT_TMAX) { found = true; finish = true; }   /* intersect.h:152 */
                            else { obj_t = CRT_TMAX; obj_h.obj = -1; obj_h.part = -1; obj_h.beta = 0.0f; obj_h.gamma = 0.0f; grid_step(&w); }
                        }
                    } else {
                        if (tt < bound) { obj_t = tt; found = true; finish = true; }
                        else if (k == ce) grid_step(&w);
                    }
                } else if (!(w.ix != w.ex && w.iy != w.ey && w.iz != w.ez)) {
                    finish = true;
                } else {
                    if (skip_left == 0u) {
                        const int cell = w.iy * sc.grid.nz * sc.grid.nx + w.iz * sc.grid.nx + w.ix;
                        skip_left = sc.cell_dist[cell];
                        if (skip_left == 0u) {
                            k = sc.cell_start[cell]; ce = sc.cell_start[cell + 1];
                            if (!sc.cell_objs) item = sc.cell_items[k];
                            if (COUNT) c_nodes++;
                        }
                    }
                    if (skip_left != 0u) {
                        unsigned steps = skip_left < (unsigned)CRT_GRID_SKIP_STEPS ? skip_left : (unsigned)CRT_GRID_SKIP_STEPS;
                        do {
                            if (COUNT) c_nodes++;
                            grid_step(&w);
                            skip_left--;
                        } while (--steps != 0u && w.ix != w.ex && w.iy != w.ey && w.iz != w.ez);
                    }
                }
            }
            if (finish) {
                if (!SHADOW) {
                    float t = found ? obj_t : CRT_TMAX;
                    DHit best = obj_h;
                    if (!found) { best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f; }
                    float min_t = CRT_TMAX;
                    DHit mh = best;
                    linear_closest<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, &min_t, &mh, &c_prims);   /* intersect.h:177-192 */
                    if (min_t < t) { best = mh; t = min_t; }
                    store_hit(la.hits, ri, t, best);
                } else {
                    float t = found ? obj_t : CRT_TMAX;
                    if (!found) t = linear_shadow<TRI_ONLY, COUNT>(sc, 0, sc.nng, ray, bound, CRT_TMAX, &c_prims);  /* :298-307 */
                    if (la.t_out) la.t_out[ri] = t;
                    if (la.occ_out) la.occ_out[ri] = (t < ld - CRT_K_EPSILON) ? 1 : 0;
                }
                if (COUNT) c_rays++;
                alive = false;
            }
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, SHADOW);
}
