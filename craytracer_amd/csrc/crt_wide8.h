/*
 * crt_wide8.h -- closest hit over the 8-wide conservative copy of the QBVH (triangle-only scenes, pruned walk).
 *
 * What BVH4IntersectTest (accelerator/bvh4.h:299-389) computes is the nearest hit of the ray among the tree's objects, and,
 * among objects with exactly that t, the one its own visiting order meets first.  This walk finds the nearest hit through a
 * shallower tree (crt_wide8_host.h: each node has absorbed some of its inner children, up to eight boxes per visit), near
 * to far by the octant of the ray direction, culling by the nearest hit so far -- and leaves every case in which WHICH hit
 * is reported could depend on the visiting order to the 4-wide walk in the reference's order:
 *   - two different objects at exactly the nearest t.  Children are culled with `key <= best t` (not `<`), keys are
 *     conservative entry distances (never beyond the t of a hit inside the box), so every object that attains the final
 *     nearest t is tested while best t >= that t: the first becomes the best, each further one compares equal -- seen;
 *   - a ray whose slab arithmetic can produce NaN or overflow (not `clean`);
 *   - a ray whose pending children do not fit the lane's stack.
 * Such rays go onto a list (LaunchArgs.fb_list / fb_count) that the same launch hands to k_bvh4_closest_tri, so every
 * hit record is the one the reference-order walk writes.
 */
#pragma once
#include "crt_kernels.h"

#ifndef CRT_W8_LDS_DEPTH
#define CRT_W8_LDS_DEPTH 14        /* 8-byte entries: 28 KB per 256-thread workgroup, five per CU */
#endif
#ifndef CRT_W8_WAVES
#define CRT_W8_WAVES 5
#endif

/* The lane stack of crt_kernels.h with one difference: a push beyond CRT_STACK_MAX is remembered (the tree's depth is not
 * validated against 7 pending siblings per level; such a ray is walked again by the 4-wide kernel). */
struct LaneStackW8 {
    StackEntry (*lds)[CRT_BLOCK];
    unsigned* ovf_word;
    float* ovf_key;
    unsigned ovf_stride;
    int sp;
    bool failed;
    CRT_DEV void push(unsigned w, float key)
    {
        if (__builtin_expect(sp < CRT_W8_LDS_DEPTH, 1)) {
            StackEntry e; e.x = w; e.y = __float_as_uint(key);
            lds[sp][threadIdx.x] = e;
            sp++;
        } else if (sp < CRT_W8_LDS_DEPTH + CRT_OVF_DEPTH) {
            ovf_push(ovf_word, ovf_key, (size_t)(sp - CRT_W8_LDS_DEPTH) * ovf_stride, w, key);
            sp++;
        } else failed = true;
    }
    CRT_DEV void pop(unsigned* w, float* key)
    {
        sp--;
        StackEntry e;
        if (__builtin_expect(sp < CRT_W8_LDS_DEPTH, 1)) e = lds[sp][threadIdx.x];
        else e = ovf_pop(ovf_word, ovf_key, (size_t)(sp - CRT_W8_LDS_DEPTH) * ovf_stride);
        *w = e.x; *key = __uint_as_float(e.y);
    }
};

/* conservative entry distance of a box, +inf when the ray misses it (t0 <= the reference's t0, t1 >= the reference's t1) */
CRT_DEV float w8_key(float tnx, float tny, float tnz, float tfx, float tfy, float tfz)
{
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
    const float t1 = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
    return ((t0 < t1) && (t1 > CRT_K_EPSILON)) ? t0 : __builtin_huge_valf();
}
/* the four keys of one half of a node: one dword per plane kind, byte j = the child at visiting position j of the half */
CRT_DEV void w8_keys4(unsigned qnx, unsigned qny, unsigned qnz, unsigned qfx, unsigned qfy, unsigned qfz,
                      v2f anx2, v2f any2, v2f anz2, v2f afx2, v2f afy2, v2f afz2, v2f bx2, v2f by2, v2f bz2,
                      float* k0, float* k1, float* k2, float* k3)
{
    {
        const v2f tnx = __builtin_elementwise_fma(v2f{CRT_UB(qnx, 0), CRT_UB(qnx, 1)}, bx2, anx2);
        const v2f tfx = __builtin_elementwise_fma(v2f{CRT_UB(qfx, 0), CRT_UB(qfx, 1)}, bx2, afx2);
        const v2f tny = __builtin_elementwise_fma(v2f{CRT_UB(qny, 0), CRT_UB(qny, 1)}, by2, any2);
        const v2f tfy = __builtin_elementwise_fma(v2f{CRT_UB(qfy, 0), CRT_UB(qfy, 1)}, by2, afy2);
        const v2f tnz = __builtin_elementwise_fma(v2f{CRT_UB(qnz, 0), CRT_UB(qnz, 1)}, bz2, anz2);
        const v2f tfz = __builtin_elementwise_fma(v2f{CRT_UB(qfz, 0), CRT_UB(qfz, 1)}, bz2, afz2);
        *k0 = w8_key(tnx.x, tny.x, tnz.x, tfx.x, tfy.x, tfz.x);
        *k1 = w8_key(tnx.y, tny.y, tnz.y, tfx.y, tfy.y, tfz.y);
    }
    {
        const v2f tnx = __builtin_elementwise_fma(v2f{CRT_UB(qnx, 2), CRT_UB(qnx, 3)}, bx2, anx2);
        const v2f tfx = __builtin_elementwise_fma(v2f{CRT_UB(qfx, 2), CRT_UB(qfx, 3)}, bx2, afx2);
        const v2f tny = __builtin_elementwise_fma(v2f{CRT_UB(qny, 2), CRT_UB(qny, 3)}, by2, any2);
        const v2f tfy = __builtin_elementwise_fma(v2f{CRT_UB(qfy, 2), CRT_UB(qfy, 3)}, by2, afy2);
        const v2f tnz = __builtin_elementwise_fma(v2f{CRT_UB(qnz, 2), CRT_UB(qnz, 3)}, bz2, anz2);
        const v2f tfz = __builtin_elementwise_fma(v2f{CRT_UB(qfz, 2), CRT_UB(qfz, 3)}, bz2, afz2);
        *k2 = w8_key(tnx.x, tny.x, tnz.x, tfx.x, tfy.x, tfz.x);
        *k3 = w8_key(tnx.y, tny.y, tnz.y, tfx.y, tfy.y, tfz.y);
    }
}
/* position j of a row of four words takes word j ^ (b0 + 2 b1) */
CRT_DEV void w8_xor4(const uint4 r, bool b0, bool b1, unsigned* w0, unsigned* w1, unsigned* w2, unsigned* w3)
{
    const unsigned a0 = b0 ? r.y : r.x, a1 = b0 ? r.x : r.y, a2 = b0 ? r.w : r.z, a3 = b0 ? r.z : r.w;
    *w0 = b1 ? a2 : a0; *w1 = b1 ? a3 : a1; *w2 = b1 ? a0 : a2; *w3 = b1 ? a1 : a3;
}

template <bool COUNT>
CRT_DEV void bvh8_closest_body(DevScene sc, LaunchArgs la)
{
    resolve_count(la);
    __shared__ StackEntry s_stack[CRT_W8_LDS_DEPTH][CRT_BLOCK];
    const unsigned gtid = blockIdx.x * CRT_BLOCK + threadIdx.x;
    LaneStackW8 st;
    st.lds = s_stack;
    st.ovf_word = la.ovf_word + gtid;
    st.ovf_key = la.ovf_key + gtid;
    st.ovf_stride = la.grid_threads;
    st.sp = 0; st.failed = false;
    unsigned c_rays = 0, c_nodes = 0, c_prims = 0;

    bool alive = false, walking = false, tie = false;
    unsigned ri = 0, cur = 0, pl = 0;
    DRay ray; ray.ox = ray.oy = ray.oz = 0.0f; ray.dx = ray.dy = ray.dz = 1.0f;
    float ia = 1.0f, ib = 1.0f, ic = 1.0f;
    bool sx = true, sy = true, sz = true;          /* direction component >= 0: the near plane of the axis is the low one */
    unsigned selA = 0x03020100u, selB = 0x07060504u;   /* v_perm selectors: byte j of a half <- byte (j ^ m) of the eight */
    unsigned wrow_near = 64u, wrow_far = 80u;      /* byte offsets of the word rows of visiting positions 0..3 / 4..7 */
    bool m0 = false, m1 = false;                    /* bits 0 and 1 of m = 7 ^ octant */
    float best_t = CRT_TMAX;
    DHit best; best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;

    auto to_fallback = [&]() __attribute__((always_inline)) {
        const unsigned k = atomicAdd(la.fb_count, 1u);
        la.fb_list[k] = ri;
    };
    /* One visit of the inner node `cur`: eight keys in visiting order, the passing children pushed far to near, the nearest
     * position (if it passes) becomes `cur`.  False: the lane has to pop. */
    auto node_step = [&]() __attribute__((always_inline)) -> bool {
        if (COUNT) c_nodes++;
        const unsigned nb = cur << 7;
        const char* base = reinterpret_cast<const char*>(sc.wnodes8);
        const uint4 q0 = *reinterpret_cast<const uint4*>(base + nb);
        const uint4 q1 = *reinterpret_cast<const uint4*>(base + (nb + 16u));
        const uint4 q2 = *reinterpret_cast<const uint4*>(base + (nb + 32u));
        const uint4 q3 = *reinterpret_cast<const uint4*>(base + (nb + 48u));
        const uint4 wn = *reinterpret_cast<const uint4*>(base + (nb + wrow_near));
        const uint4 wf = *reinterpret_cast<const uint4*>(base + (nb + wrow_far));
        float anx, afx, bx, any_, afy, by, anz, afz, bz;
        cnode_axis(__uint_as_float(q0.x), __uint_as_float((q0.w & 0xFFu) << 23), ray.ox, ia, &anx, &afx, &bx);
        cnode_axis(__uint_as_float(q0.y), __uint_as_float(((q0.w >> 8) & 0xFFu) << 23), ray.oy, ib, &any_, &afy, &by);
        cnode_axis(__uint_as_float(q0.z), __uint_as_float(((q0.w >> 16) & 0xFFu) << 23), ray.oz, ic, &anz, &afz, &bz);
        /* near / far plane bytes per axis (two dwords = eight slots each), then into visiting order */
        const unsigned nx0 = sx ? q1.x : q2.z, nx1 = sx ? q1.y : q2.w, fx0 = sx ? q2.z : q1.x, fx1 = sx ? q2.w : q1.y;
        const unsigned ny0 = sy ? q1.z : q3.x, ny1 = sy ? q1.w : q3.y, fy0 = sy ? q3.x : q1.z, fy1 = sy ? q3.y : q1.w;
        const unsigned nz0 = sz ? q2.x : q3.z, nz1 = sz ? q2.y : q3.w, fz0 = sz ? q3.z : q2.x, fz1 = sz ? q3.w : q2.y;
        const v2f bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
        const v2f anx2 = {anx, anx}, afx2 = {afx, afx}, any2 = {any_, any_}, afy2 = {afy, afy}, anz2 = {anz, anz}, afz2 = {afz, afz};
        {   /* visiting positions 7..4 */
            float k4, k5, k6, k7;
            w8_keys4(__builtin_amdgcn_perm(nx1, nx0, selB), __builtin_amdgcn_perm(ny1, ny0, selB), __builtin_amdgcn_perm(nz1, nz0, selB),
                     __builtin_amdgcn_perm(fx1, fx0, selB), __builtin_amdgcn_perm(fy1, fy0, selB), __builtin_amdgcn_perm(fz1, fz0, selB),
                     anx2, any2, anz2, afx2, afy2, afz2, bx2, by2, bz2, &k4, &k5, &k6, &k7);
            unsigned w4, w5, w6, w7;
            w8_xor4(wf, m0, m1, &w4, &w5, &w6, &w7);
            if (k7 <= best_t) st.push(w7, k7);
            if (k6 <= best_t) st.push(w6, k6);
            if (k5 <= best_t) st.push(w5, k5);
            if (k4 <= best_t) st.push(w4, k4);
        }
        float k0, k1, k2, k3;
        w8_keys4(__builtin_amdgcn_perm(nx1, nx0, selA), __builtin_amdgcn_perm(ny1, ny0, selA), __builtin_amdgcn_perm(nz1, nz0, selA),
                 __builtin_amdgcn_perm(fx1, fx0, selA), __builtin_amdgcn_perm(fy1, fy0, selA), __builtin_amdgcn_perm(fz1, fz0, selA),
                 anx2, any2, anz2, afx2, afy2, afz2, bx2, by2, bz2, &k0, &k1, &k2, &k3);
        unsigned w0, w1, w2, w3;
        w8_xor4(wn, m0, m1, &w0, &w1, &w2, &w3);
        if (k3 <= best_t) st.push(w3, k3);
        if (k2 <= best_t) st.push(w2, k2);
        if (k1 <= best_t) st.push(w1, k1);
        if (k0 <= best_t) { cur = w0; return true; }
        return false;
    };
    auto object_step = [&](int first) __attribute__((always_inline)) {
        DHit h;
        const float t = dev_object<true>(sc, first, ray, false, &h);
        if (COUNT) c_prims++;
        if (t < best_t) { best_t = t; best = h; tie = false; }
        else if (t == best_t && t < CRT_TMAX && h.obj != best.obj) tie = true;      /* which of the two is reported is the reference walk's to say */
    };
    auto pop_next = [&]() __attribute__((always_inline)) -> bool {
        while (st.sp > 0) {
            unsigned w; float key;
            st.pop(&w, &key);
            if (key <= best_t) { cur = w; return true; }
        }
        return false;
    };
    /* true: the ray walks here; false: it is one for the 4-wide walk (see the head of the file) */
    auto start_ray = [&]() __attribute__((always_inline)) -> bool {
        ia = 1.0f / ray.dx; ib = 1.0f / ray.dy; ic = 1.0f / ray.dz;
        sx = ray.dx >= 0.0f; sy = ray.dy >= 0.0f; sz = ray.dz >= 0.0f;
        const unsigned m = (sx ? 0u : 1u) | (sy ? 0u : 2u) | (sz ? 0u : 4u);          /* 7 ^ octant */
        selA = 0x03020100u ^ (m * 0x01010101u);
        selB = selA ^ 0x04040404u;
        wrow_near = (m & 4u) ? 80u : 64u;
        wrow_far = 144u - wrow_near;
        m0 = (m & 1u) != 0u; m1 = (m & 2u) != 0u;
        const bool clean = fabsf(ia) < 1.0e6f && fabsf(ib) < 1.0e6f && fabsf(ic) < 1.0e6f &&
                           fabsf(ray.ox) < 1.0e30f && fabsf(ray.oy) < 1.0e30f && fabsf(ray.oz) < 1.0e30f &&
                           ia != 0.0f && ib != 0.0f && ic != 0.0f;
        best_t = CRT_TMAX;
        best.obj = -1; best.part = -1; best.beta = 0.0f; best.gamma = 0.0f;
        st.sp = 0; st.failed = false; tie = false;
        cur = 0u;                                  /* record 0 is the root */
        pl = 0u;
        walking = sc.n_accel > 0;
        return clean;
    };

    RayFeed<false> feed;
    feed.start(la);
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !feed.empty && __popcll(idle) >= CRT_REFILL_MIN) {
            float unused_ld;
            if (feed.take(la, idle, !alive, &ri, &ray, &unused_ld)) {
                if (start_ray()) alive = true;
                else to_fallback();
            }
        }
        if (__ballot(alive) == 0ull) {
            if (feed.empty) break;                 /* nothing in flight and nothing left to claim */
            continue;
        }
        if (alive) {
            bool need_pop = false;
            if (walking) {
                if (!(cur & CRT_LEAF_BIT)) need_pop = !node_step();
                if ((cur & CRT_LEAF_BIT) && !need_pop && ((pl >> 28) & 7u) == 0u) { pl = cur; need_pop = true; }
            }
            const unsigned pcount = (pl >> 28) & 7u;
            if (pcount > 0u) {
                const unsigned first = pl & 0x0FFFFFFFu;
                object_step((int)first);
                pl = pcount > 1u ? (CRT_LEAF_BIT | ((pcount - 1u) << 28) | (first + 1u)) : 0u;
            }
            if (need_pop) walking = pop_next();
            if (st.failed) {
                to_fallback();
                alive = false;
            } else if (!walking && ((pl >> 28) & 7u) == 0u) {
                alive = false;
                if (tie) to_fallback();
                else {
                    linear_closest<true, COUNT>(sc, 0, sc.nng, ray, &best_t, &best, &c_prims);
                    store_hit(la.hits, ri, best_t, best);
                    if (COUNT) c_rays++;
                }
            }
        }
    }
    flush_counters<COUNT>(sc, c_rays, c_nodes, c_prims, false);
}

template <bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK) __attribute__((amdgpu_waves_per_eu(CRT_W8_WAVES, CRT_W8_WAVES))) void k_bvh8_closest_tri(DevScene sc, LaunchArgs la)
{
    bvh8_closest_body<COUNT>(sc, la);
}
