/*
 * crt_wide8_host.h -- host side of the 8-wide conservative QBVH copy (crt_wide8.h walks it).
 *
 * The reference's QBVH (accelerator/bvh4.h:100-110) has four children per node, and a ray's walk through it is a chain of
 * dependent node fetches (35 per ray on the 10.5 M-triangle scene).  For triangle-only scenes crt_scene_create keeps a
 * second copy of the tree with up to eight children per node, cut out of the same binary hierarchy (the 4-wide tree is a
 * binary tree taken two levels at a time) so as to minimise the expected number of node visits -- the dynamic programme of
 * Ylitie, Karras and Laine 2017 with the leaves fixed -- so the chain is about half as long.  Leaves are untouched: a leaf child is still a
 * run of <= 4 objects of the final objects[] order, with the reference's box (the union of getObjectAABB, bvh4.h:268-296).
 *
 * Record, 128 bytes (32 words), 96 used:
 *   w0..2   frame origin (float), the minimum corner of the children's boxes
 *   w3      three biased exponents ex | ey << 8 | ez << 16: scale = 2^(e - 127) per axis
 *   w4..15  one byte per child slot and plane, rounded OUTWARD: lo_x[8] lo_y[8] lo_z[8] hi_x[8] hi_y[8] hi_z[8];
 *           plane = origin + byte * scale; an unused slot holds lo = 255, hi = 0
 *   w16..23 child words by slot: record index of an inner child, CRT_LEAF_BIT | count << 28 | first for a leaf child,
 *           CRT_LEAF_BIT alone (count 0) for an unused slot
 * Slots: a child sits in the slot whose sign pattern (bit a set = on the + side of the node's centre along axis a) matches
 * where it lies, so that visiting the slots in the order j ^ (7 ^ octant of the ray direction), j = 0..7, is near to far
 * (the scheme of Ylitie, Karras, Laine 2017; greedy assignment).  The walk does not depend on it for its result.
 */
#pragma once
#include <vector>
#include <math.h>
#include <string.h>
#include <stdint.h>
#include "../../include/crt_gpu.h"

#define CRT_W8_WORDS 32
#ifndef CRT_W8_LEAF_BIT
#define CRT_W8_LEAF_BIT 0x80000000u
#endif

namespace wide8 {

/* The 4-wide tree is a binary tree taken two levels at a time (bvh4.h:160-200: one cut, then one cut per half).  Binary node
 * id = 3 * (nodes4 index) + 0 for the whole 4-node (children: its two halves), + 1 + h for half h (children: the node's
 * children 2h and 2h + 1).  A half with one live child IS that child, an empty half is nothing (Ref resolves that). */
struct Ref { int kind; int id; int node; int c; };        /* kind 0: nothing, 1: leaf = child c of `node`, 2: inner binary node `id` */
struct Item { int inner; int ref; int count; float lo[3], hi[3]; };     /* inner: binary node id `ref`; else objects [ref, ref + count) */

inline bool live_child(const crt_bvh4_node& nd, int c)
{
    if (!(nd.type[c] == -1 || nd.type[c] > 0)) return false;
    for (int k = 0; k < 3; k++) if (!(nd.bbox[4 * k + c] <= nd.bbox[12 + 4 * k + c])) return false;   /* empty subtree (inverted box) or NaN: never hit */
    return true;
}
inline Ref resolve_full(const crt_scene_desc* d, int node);
inline Ref resolve_child(const crt_scene_desc* d, int node, int c)
{
    const crt_bvh4_node& nd = d->nodes4[node];
    if (!live_child(nd, c)) return Ref{ 0, -1, -1, -1 };
    if (nd.type[c] == -1) return resolve_full(d, nd.child[c]);
    return Ref{ 1, -1, node, c };
}
inline Ref resolve_half(const crt_scene_desc* d, int node, int h)
{
    const Ref a = resolve_child(d, node, 2 * h), b = resolve_child(d, node, 2 * h + 1);
    if (a.kind && b.kind) return Ref{ 2, 3 * node + 1 + h, node, -1 };
    return a.kind ? a : b;
}
inline Ref resolve_full(const crt_scene_desc* d, int node)
{
    const Ref a = resolve_half(d, node, 0), b = resolve_half(d, node, 1);
    if (a.kind && b.kind) return Ref{ 2, 3 * node, node, -1 };
    return a.kind ? a : b;
}
inline void bin_children(const crt_scene_desc* d, int id, Ref* l, Ref* r)
{
    const int node = id / 3, part = id % 3;
    if (part == 0) { *l = resolve_half(d, node, 0); *r = resolve_half(d, node, 1); }
    else { *l = resolve_child(d, node, 2 * (part - 1)); *r = resolve_child(d, node, 2 * (part - 1) + 1); }
}
/* box of the children [c0, c1] of a 4-node that are live */
inline void span_box(const crt_bvh4_node& nd, int c0, int c1, float* lo, float* hi)
{
    bool any = false;
    for (int c = c0; c <= c1; c++) if (live_child(nd, c)) for (int k = 0; k < 3; k++) {
        const float mn = nd.bbox[4 * k + c], mx = nd.bbox[12 + 4 * k + c];
        if (!any || mn < lo[k]) lo[k] = mn;
        if (!any || mx > hi[k]) hi[k] = mx;
        if (k == 2) any = true;
    }
    if (!any) for (int k = 0; k < 3; k++) { lo[k] = 0.0f; hi[k] = 0.0f; }
}
inline void bin_box(const crt_scene_desc* d, int id, float* lo, float* hi)
{
    const int node = id / 3, part = id % 3;
    if (part == 0) span_box(d->nodes4[node], 0, 3, lo, hi); else span_box(d->nodes4[node], 2 * (part - 1), 2 * (part - 1) + 1, lo, hi);
}
inline Item make_item(const crt_scene_desc* d, const Ref& r)
{
    Item it; memset(&it, 0, sizeof it);
    if (r.kind == 1) {
        const crt_bvh4_node& nd = d->nodes4[r.node];
        it.inner = 0; it.ref = nd.child[r.c]; it.count = nd.type[r.c];
        for (int k = 0; k < 3; k++) { it.lo[k] = nd.bbox[4 * k + r.c]; it.hi[k] = nd.bbox[12 + 4 * k + r.c]; }
    } else { it.inner = 1; it.ref = r.id; it.count = 0; bin_box(d, r.id, it.lo, it.hi); }
    return it;
}

/* out: CRT_W8_WORDS words per record, record 0 = the root.  False: no copy for this tree (a non-finite box).
 * Which binary nodes become records: cost(n, i) = the least expected number of record visits below n when n's subtree may
 * hang into its parent record as at most i children -- as one child (n is then a record of its own: its box area, the
 * probability of a visit, plus the best way of spending that record's max_width slots on n's two children) or with n
 * dissolved and the i slots shared between its two children.  A leaf costs nothing and takes one slot. */
inline bool collapse(const crt_scene_desc* d, std::vector<uint32_t>& out, int max_width = 8)
{
    out.clear();
    const int n4 = d->num_nodes4;
    if (n4 < 1) return false;
    if (max_width < 2) max_width = 2;
    if (max_width > 8) max_width = 8;
    const int W = max_width;
    /* cost[id][i - 1], i = 1..W.  split[id][0]: slots of n's OWN record given to its left child; split[id][i - 1], i >= 2: slots
     * given to the left child when n is dissolved into exactly i slots; eff[id][i - 1]: how many of i offered slots n takes
     * (1 = it stays one child, a record of its own) */
    std::vector<float> cost((size_t)n4 * 3 * 8, 0.0f);
    std::vector<unsigned char> split((size_t)n4 * 3 * 8, 0), eff((size_t)n4 * 3 * 8, 1);
    auto cost_of = [&](const Ref& r, int i) -> float { return r.kind == 2 ? cost[(size_t)r.id * 8 + (size_t)(i - 1)] : 0.0f; };
    auto solve = [&](int id) {
        Ref l, r;
        bin_children(d, id, &l, &r);
        if (!(l.kind && r.kind)) return;                       /* not a binary node of its own (Ref resolves through it) */
        float lo[3], hi[3];
        bin_box(d, id, lo, hi);
        float area = (hi[0] - lo[0]) * (hi[1] - lo[1]) + (hi[1] - lo[1]) * (hi[2] - lo[2]) + (hi[2] - lo[2]) * (hi[0] - lo[0]);
        if (!(area >= 0.0f) || !(area < 1.0e30f)) area = 0.0f;
        float* c = &cost[(size_t)id * 8];
        unsigned char* sp = &split[(size_t)id * 8];
        float dis[9]; int dk[9];
        for (int i = 2; i <= W; i++) {
            dis[i] = 3.0e38f; dk[i] = 1;
            for (int k = 1; k < i; k++) { const float v = cost_of(l, k) + cost_of(r, i - k); if (v < dis[i]) { dis[i] = v; dk[i] = k; } }
        }
        unsigned char* ef = &eff[(size_t)id * 8];
        c[0] = area + dis[W]; sp[0] = (unsigned char)dk[W]; ef[0] = 1;
        for (int i = 2; i <= W; i++) {
            sp[i - 1] = (unsigned char)dk[i];
            if (dis[i] < c[i - 2]) { c[i - 1] = dis[i]; ef[i - 1] = (unsigned char)i; }
            else { c[i - 1] = c[i - 2]; ef[i - 1] = ef[i - 2]; }
        }
    };
    for (int node = n4 - 1; node >= 0; node--) { solve(3 * node + 1); solve(3 * node + 2); solve(3 * node); }     /* children have larger indices */

    std::vector<int> roots;                    /* binary node id of each record, in creation order (breadth first) */
    {
        const Ref top = resolve_full(d, 0);
        if (top.kind != 2) {
            /* a tree of one leaf (or none): one record holding it */
            roots.push_back(-1);
        } else roots.push_back(top.id);
    }
    for (size_t r = 0; r < roots.size(); r++) {
        if (roots.size() > (size_t)0x01FFFFFF) return false;
        Item items[8];
        int n = 0;
        /* spend `i` slots on the subtree of `ref` as the costs decided */
        struct Todo { Ref ref; int i; };
        Todo stack[32]; int sp = 0;
        if (roots[r] < 0) {
            const Ref top = resolve_full(d, 0);
            if (top.kind == 1) items[n++] = make_item(d, top);
        } else {
            Ref l, rr;
            bin_children(d, roots[r], &l, &rr);
            const int k = split[(size_t)roots[r] * 8];
            stack[sp++] = Todo{ rr, W - k };
            stack[sp++] = Todo{ l, k };
        }
        while (sp > 0) {
            const Todo t = stack[--sp];
            if (t.ref.kind == 0) continue;
            if (t.ref.kind == 1 || t.i <= 1) { if (n < 8) items[n++] = make_item(d, t.ref); else return false; continue; }
            const int e = eff[(size_t)t.ref.id * 8 + (size_t)(t.i - 1)];
            if (e <= 1) { if (n < 8) items[n++] = make_item(d, t.ref); else return false; continue; }      /* a record of its own */
            const int k = split[(size_t)t.ref.id * 8 + (size_t)(e - 1)];
            Ref l, rr;
            bin_children(d, t.ref.id, &l, &rr);
            if (sp + 2 > 32) return false;
            stack[sp++] = Todo{ rr, e - k };
            stack[sp++] = Todo{ l, k };
        }
        /* frame */
        float lo[3] = { 0.0f, 0.0f, 0.0f }, hi[3] = { 0.0f, 0.0f, 0.0f };
        for (int k = 0; k < 3; k++) {
            for (int i = 0; i < n; i++) {
                if (!(fabsf(items[i].lo[k]) < 1.0e30f) || !(fabsf(items[i].hi[k]) < 1.0e30f)) return false;
                if (i == 0 || items[i].lo[k] < lo[k]) lo[k] = items[i].lo[k];
                if (i == 0 || items[i].hi[k] > hi[k]) hi[k] = items[i].hi[k];
            }
        }
        int eb[3]; double scale[3];
        for (int k = 0; k < 3; k++) {
            const double ext = (double)hi[k] - (double)lo[k];
            int e = -126;
            if (ext > 0.0) { int x; (void)frexp(ext / 255.0, &x); e = x; }           /* ext / 255 = f 2^x, f in [0.5, 1): 2^x >= ext / 255 */
            if (e < -126) e = -126;
            while (e < 127 && (double)lo[k] + 255.0 * ldexp(1.0, e) < (double)hi[k]) e++;
            if ((double)lo[k] + 255.0 * ldexp(1.0, e) < (double)hi[k]) return false;
            eb[k] = e + 127; scale[k] = ldexp(1.0, e);
        }
        /* slots: greedy assignment by how far a child lies along each slot's sign pattern */
        int slot_of[8], item_at[8];
        for (int s = 0; s < 8; s++) item_at[s] = -1;
        for (int i = 0; i < n; i++) slot_of[i] = -1;
        {
            double fit[8][8];
            for (int i = 0; i < n; i++) for (int s = 0; s < 8; s++) {
                double c = 0.0;
                for (int k = 0; k < 3; k++) {
                    const double cen = 0.5 * ((double)items[i].lo[k] + (double)items[i].hi[k]) - 0.5 * ((double)lo[k] + (double)hi[k]);
                    c += ((s >> k) & 1) ? cen : -cen;
                }
                fit[i][s] = c;
            }
            for (int round = 0; round < n; round++) {
                int bi = -1, bs = -1; double bc = -1.0e300;
                for (int i = 0; i < n; i++) if (slot_of[i] < 0) for (int s = 0; s < 8; s++) if (item_at[s] < 0 && fit[i][s] > bc) { bc = fit[i][s]; bi = i; bs = s; }
                slot_of[bi] = bs; item_at[bs] = bi;
            }
        }
        out.resize((r + 1) * CRT_W8_WORDS, 0u);
        uint32_t* w = &out[r * CRT_W8_WORDS];
        memcpy(&w[0], &lo[0], 4); memcpy(&w[1], &lo[1], 4); memcpy(&w[2], &lo[2], 4);
        w[3] = (uint32_t)eb[0] | ((uint32_t)eb[1] << 8) | ((uint32_t)eb[2] << 16);
        unsigned char* planes = reinterpret_cast<unsigned char*>(&w[4]);       /* [6][8] */
        for (int s = 0; s < 8; s++) {
            const int i = item_at[s];
            for (int k = 0; k < 3; k++) {
                int qlo = 255, qhi = 0;
                if (i >= 0) {
                    const double mn = items[i].lo[k], mx = items[i].hi[k], o = lo[k], sc = scale[k];
                    qlo = (int)floor((mn - o) / sc); if (qlo > 255) qlo = 255; if (qlo < 0) qlo = 0;
                    while (qlo > 0 && o + qlo * sc > mn) qlo--;
                    qhi = (int)ceil((mx - o) / sc); if (qhi < 0) qhi = 0; if (qhi > 255) qhi = 255;
                    while (qhi < 255 && o + qhi * sc < mx) qhi++;
                    if (o + qlo * sc > mn || o + qhi * sc < mx) return false;     /* cannot happen (2^e covers the extent); checked anyway */
                }
                planes[k * 8 + s] = (unsigned char)qlo;
                planes[(3 + k) * 8 + s] = (unsigned char)qhi;
            }
            uint32_t word = CRT_W8_LEAF_BIT;                                       /* unused slot: a leaf of no objects */
            if (i >= 0) {
                if (items[i].inner) { word = (uint32_t)roots.size(); roots.push_back(items[i].ref); }
                else word = CRT_W8_LEAF_BIT | ((uint32_t)items[i].count << 28) | (uint32_t)items[i].ref;
            }
            w[16 + s] = word;
        }
    }
    return true;
}

} // namespace wide8
