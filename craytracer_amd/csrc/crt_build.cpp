/*
 * crt_build.cpp -- host-side scene ordering + accelerator construction.
 *
 * Produces the flat crt_scene_desc that crt_scene_create uploads.  The reference's host
 * does this work in mvNonGridObjToStart (scene/scenedata.h:368-382) and buildSceneAccel
 * (buildscene.h:698-731) -> UniformGrid_create (accelerator/uniformgrid.h:54-154),
 * BVH_build (accelerator/bvh.h:117-154), BVH4_build (accelerator/bvh4.h:127-297).
 * The split rules, box arithmetic and cell assignment follow those functions so that the
 * trees (and therefore traversal order and tie-breaking) are the reference's; the data
 * structures are index-based arrays instead of malloc'd pointer nodes, per-object boxes are
 * computed once, and sibling subtrees are built on separate threads (node numbering is
 * pre-order and depends only on range sizes, so it is known before a subtree is built).
 *
 * Builder order: every range is sorted with a STABLE sort on the float centroid.  The
 * reference does the same for ranges <= 30 objects (insertion sort, bvh.h:99-111) and uses a
 * rand()-pivot quicksort above that (bvh.h:56-83), whose order among EQUAL centroids depends
 * on the libc rand() stream; with distinct centroids both give the same unique order.
 */
#include "../../include/crt_gpu.h"

#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <thread>
#include <new>

namespace {

struct Box { float mn[3]; float mx[3]; };

inline bool is_tri(int t) { return t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE; }

struct Builder {
    const crt_scene_input* in;
    int n;                              /* num_obj */
    std::vector<int32_t> type, ref, id; /* current objects[] order */
    std::vector<Box> box;               /* per object, same order */
    int nng;                            /* num_non_grid_obj */
    std::vector<crt_bvh4_node> nodes4;
    std::vector<crt_bvh2_node> nodes2;
    crt_grid grid;
    std::vector<int32_t> cell_start, cell_items;
    int max_threads;
};
/* four-way tree levels whose children are built by threads of their own: 16 threads, 4, or none */
static inline int fork_levels(int max_threads) { return max_threads >= 16 ? 2 : (max_threads >= 4 ? 1 : 0); }

/* ---- per-object boxes: getObjectAABB, shapes/shapes.cpp:141-314 ---- */

void shape_box(const crt_scene_input* in, int si, Box* b)
{
    const crt_shape& sh = in->shapes[si];
    const float* p = sh.p;
    switch (sh.type) {
    case CRT_SPHERE:
        for (int i = 0; i < 3; i++) { b->mn[i] = p[i] - p[3]; b->mx[i] = p[i] + p[3]; }
        break;
    case CRT_RECTANGLE: {
        float c[4][3];
        for (int i = 0; i < 3; i++) {
            c[0][i] = p[i];
            c[1][i] = c[0][i] + p[3+i];
            c[2][i] = c[0][i] + p[6+i];
            float wh = p[3+i] + p[6+i];
            c[3][i] = c[0][i] + wh;
        }
        for (int i = 0; i < 3; i++) {
            float lo = c[0][i], hi = c[0][i];
            for (int k = 1; k < 4; k++) { if (c[k][i] < lo) lo = c[k][i]; if (c[k][i] > hi) hi = c[k][i]; }
            b->mn[i] = lo - CRT_K_EPSILON;
            b->mx[i] = hi + CRT_K_EPSILON;
        }
    } break;
    case CRT_AABOX: {
        const float gap = 0.00001f;
        for (int i = 0; i < 3; i++) { b->mn[i] = p[i] - gap; b->mx[i] = p[3+i] + gap; }
    } break;
    case CRT_GENERICOPENCYLINDER:
        b->mn[0] = -p[1]; b->mn[1] = -p[0]; b->mn[2] = -p[1];
        b->mx[0] =  p[1]; b->mx[1] =  p[0]; b->mx[2] =  p[1];
        break;
    case CRT_DISK:
        for (int i = 0; i < 3; i++) {
            float lo = p[i] - p[6], hi = p[i] + p[6];
            b->mn[i] = lo - CRT_K_FLAT_AABB;
            b->mx[i] = hi + CRT_K_FLAT_AABB;
        }
        break;
    case CRT_GENERICTORUS: {
        float outer = p[0] + p[1];
        b->mn[0] = -outer; b->mn[1] = -p[1]; b->mn[2] = -outer;
        b->mx[0] =  outer; b->mx[1] =  p[1]; b->mx[2] =  outer;
    } break;
    case CRT_COMPOUND:
        for (int i = 0; i < 3; i++) { b->mn[i] = p[i]; b->mx[i] = p[3+i]; }
        break;
    case CRT_INSTANCED: {
        /* The reference maps the inner box with inv_transform's own 3x3 block and
         * -inv_transform[i][3] (affine_inverse + transposed indexing in updateAABB,
         * shapes.cpp:119-139,:284-304) -- right only for pure translations; kept, because the
         * tree the reference traverses is built from these boxes. */
        Box inner;
        shape_box(in, sh.inner, &inner);
        const float* m = sh.inv_xform;
        for (int i = 0; i < 3; i++) {
            float lo = -m[4*i+3], hi = -m[4*i+3];
            for (int j = 0; j < 3; j++) {
                float e = m[4*i+j] * inner.mn[j];
                float f = m[4*i+j] * inner.mx[j];
                if (e < f) { lo += e; hi += f; } else { lo += f; hi += e; }
            }
            b->mn[i] = lo; b->mx[i] = hi;
        }
    } break;
    default:   /* PLANE has no box; never reaches a builder */
        for (int i = 0; i < 3; i++) { b->mn[i] = 0; b->mx[i] = 0; }
        break;
    }
}

void object_box(const crt_scene_input* in, int type, int ref, Box* b)
{
    if (is_tri(type)) {
        const float* v = in->tri_verts + (size_t)ref * 9;
        for (int i = 0; i < 3; i++) {
            float lo = v[i], hi = v[i];
            if (v[3+i] < lo) lo = v[3+i];
            if (v[3+i] > hi) hi = v[3+i];
            if (v[6+i] < lo) lo = v[6+i];
            if (v[6+i] > hi) hi = v[6+i];
            b->mn[i] = lo - CRT_K_FLAT_AABB;
            b->mx[i] = hi + CRT_K_FLAT_AABB;
        }
        return;
    }
    shape_box(in, ref, b);
}

/* isGridObjType, shapes.cpp:95-117 */
bool grid_shape(const crt_scene_input* in, int si)
{
    const crt_shape& sh = in->shapes[si];
    switch (sh.type) {
    case CRT_SPHERE: case CRT_RECTANGLE: case CRT_AABOX: case CRT_GENERICOPENCYLINDER:
    case CRT_DISK: case CRT_GENERICTORUS: return true;
    case CRT_INSTANCED: return grid_shape(in, sh.inner);
    default: return false;
    }
}

/* calcBoundingVolume, bvh.h:28-39, over precomputed boxes */
Box range_box(const Builder& B, int first, int count)
{
    Box r;
    for (int i = 0; i < 3; i++) { r.mn[i] = FLT_MAX; r.mx[i] = -CRT_HUGEVALUE; }
    for (int k = first; k < first + count; k++) {
        const Box& a = B.box[k];
        for (int i = 0; i < 3; i++) {
            if (a.mn[i] < r.mn[i]) r.mn[i] = a.mn[i];
            if (a.mx[i] > r.mx[i]) r.mx[i] = a.mx[i];
        }
    }
    return r;
}

inline int longest_axis(const Box& b)
{
    float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    if (x > y && x > z) return 0;
    if (y > z) return 1;
    return 2;
}

/* partitionObjects, bvh.h:85-115: stable ascending order of (min+max)*0.5f on `axis`. */
void sort_range(Builder& B, int first, int count, int axis)
{
    if (count < 2) return;
    struct Key { float c; int32_t k; };
    std::vector<Key> keys(count);
    for (int i = 0; i < count; i++) {
        const Box& b = B.box[first + i];
        keys[i].c = (b.mn[axis] + b.mx[axis]) * 0.5f;
        keys[i].k = i;
    }
    std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.c < b.c; });
    std::vector<int32_t> t(count), r(count), d(count);
    std::vector<Box> bx(count);
    for (int i = 0; i < count; i++) {
        int s = first + keys[i].k;
        t[i] = B.type[s]; r[i] = B.ref[s]; d[i] = B.id[s]; bx[i] = B.box[s];
    }
    for (int i = 0; i < count; i++) {
        B.type[first + i] = t[i]; B.ref[first + i] = r[i]; B.id[first + i] = d[i]; B.box[first + i] = bx[i];
    }
}

/* number of QBVH nodes a range of `count` objects produces (depends on count only) */
struct NodeCountCache {
    std::vector<std::pair<int, long long> > memo;
    long long get(int count)
    {
        if (count <= 4) return 0;
        for (size_t i = 0; i < memo.size(); i++) if (memo[i].first == count) return memo[i].second;
        int mid = count / 2, left = mid / 2, right = mid + (count - mid) / 2;
        long long total = 1 + get(left) + get(mid - left) + get(right - mid) + get(count - right);
        memo.push_back(std::make_pair(count, total));
        return total;
    }
};

/* BVH4_build, bvh4.h:127-297.  Builds the subtree for objects [first, first+count) into
 * nodes4[self ...] (pre-order). */
void build4(Builder& B, NodeCountCache& ncc, int first, int count, int self, int depth)
{
    Box all = range_box(B, first, count);
    sort_range(B, first, count, longest_axis(all));
    int mid = count / 2;

    Box lo = range_box(B, first, mid);
    sort_range(B, first, mid, longest_axis(lo));
    int left = mid / 2;

    /* third split axis from objects[left .. left + (count - mid)) -- the reference's range (bvh4.h:180) */
    Box third = range_box(B, first + left, count - mid);
    int axis_c = longest_axis(third);
    sort_range(B, first + mid, count - mid, axis_c);
    int right = mid + (count - mid) / 2;

    int start[5] = { 0, left, mid, right, count };
    crt_bvh4_node nd;
    memset(&nd, 0, sizeof nd);
    nd.axis = (uint8_t)axis_c;                       /* node->axis ends up as the last assignment (bvh4.h:194-195) */
    int child_self[4];
    int next = self + 1;
    for (int c = 0; c < 4; c++) {
        int cnt = start[c+1] - start[c];
        Box b = range_box(B, first + start[c], cnt);
        nd.bbox[c]      = b.mn[0]; nd.bbox[4 + c]  = b.mn[1]; nd.bbox[8 + c]  = b.mn[2];
        nd.bbox[12 + c] = b.mx[0]; nd.bbox[16 + c] = b.mx[1]; nd.bbox[20 + c] = b.mx[2];
        if (cnt > 4) {
            nd.type[c] = -1;
            nd.child[c] = next;
            child_self[c] = next;
            next += (int)ncc.get(cnt);
        } else {
            nd.type[c] = (int8_t)cnt;
            nd.child[c] = first + start[c];
            child_self[c] = -1;
        }
    }
    B.nodes4[self] = nd;

    bool fork = (depth < fork_levels(B.max_threads) && count > 65536);
    if (fork) {
        std::vector<std::thread> th;
        std::vector<NodeCountCache> caches(4, ncc);
        for (int c = 0; c < 4; c++) if (child_self[c] >= 0) {
            int f = first + start[c], cnt = start[c+1] - start[c], cs = child_self[c];
            NodeCountCache* cc = &caches[c];
            th.emplace_back([&B, cc, f, cnt, cs, depth]() { build4(B, *cc, f, cnt, cs, depth + 1); });
        }
        for (auto& t : th) t.join();
    } else {
        for (int c = 0; c < 4; c++) if (child_self[c] >= 0)
            build4(B, ncc, first + start[c], start[c+1] - start[c], child_self[c], depth + 1);
    }
}


/* ---- SAH QBVH (CRT_BUILD_SAH) ------------------------------------------------------------------
 * Same node format, same leaf rule (a child with <= 4 objects is a leaf) and the same traversal as the
 * reference's tree; only the choice of the three split planes of a node differs: binned surface-area
 * heuristic over object centroids (16 bins per axis, all three axes) instead of the object median on
 * the longest axis (bvh4.h:143-195).  Hits do not depend on the tree for objects that lie inside their
 * boxes, so results are those of the reference's tree except for which of two exactly equal-t hits is
 * found first.  The reference's builder balances by count, so a few scene-sized triangles (walls, a
 * ground quad) bloat the boxes of every level they pass through; SAH puts them in their own subtrees.
 * Depth is bounded for the traversal stack: near the limit the builder falls back to median splits. */
/* 16 bins: measured against 4 / 8 / 12 / 20 / 24 / 32 / 64 / 128 on the 10.5 M-triangle courtyard; more planes lower the
 * greedy cost of each split without making the tree better */
#ifndef CRT_SAH_BINS
#define CRT_SAH_BINS 16
#endif
const int SAH_BINS = CRT_SAH_BINS;
#ifndef CRT_SAH_LEAF
#define CRT_SAH_LEAF 4                              /* objects per leaf of the SAH tree (the reference's rule: 4) */
#endif
/* levels: the pruned walk's stack holds 3 entries per level (3 * 20 + 1 <= 62), the reference-control-flow walk -- which
 * crt_scene_create picks when an instanced, compound or object-space shape sits under the accelerator -- 4 (4 * 15 + 1 <= 62) */
const int SAH_MAX_DEPTH_PRUNED = 20, SAH_MAX_DEPTH_EXACT = 15;

inline float box_area(const Box& b)
{
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dx * dz);
}
inline void box_grow(Box& a, const Box& b)
{
    for (int k = 0; k < 3; k++) { if (b.mn[k] < a.mn[k]) a.mn[k] = b.mn[k]; if (b.mx[k] > a.mx[k]) a.mx[k] = b.mx[k]; }
}
inline Box box_empty()
{
    Box b; for (int k = 0; k < 3; k++) { b.mn[k] = INFINITY; b.mx[k] = -INFINITY; } return b;
}
inline int levels_needed(int count)                 /* levels a median-split QBVH needs for `count` objects */
{
    int levels = 1;
    long long cap = 4 * CRT_SAH_LEAF;               /* one node: 4 children of <= CRT_SAH_LEAF objects */
    while (cap < count) { cap *= 4; levels++; }
    return levels;
}

/* Reorders objects [first, first+count) so that the left part comes first; returns its size (0 < size < count),
 * or 0 when no useful SAH plane exists.  *axis_out = split axis. */
int sah_partition(Builder& B, int first, int count, int* axis_out)
{
    Box cb = box_empty();
    for (int i = 0; i < count; i++) {
        const Box& b = B.box[first + i];
        for (int k = 0; k < 3; k++) {
            const float c = (b.mn[k] + b.mx[k]) * 0.5f;
            if (c < cb.mn[k]) cb.mn[k] = c;
            if (c > cb.mx[k]) cb.mx[k] = c;
        }
    }
    float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; axis++) {
        const float ext = cb.mx[axis] - cb.mn[axis];
        if (!(ext > 0.0f) || !(ext < INFINITY)) continue;
        const float scale = (float)SAH_BINS / ext;
        int cnt[SAH_BINS]; Box bb[SAH_BINS];
        for (int k = 0; k < SAH_BINS; k++) { cnt[k] = 0; bb[k] = box_empty(); }
        for (int i = 0; i < count; i++) {
            const Box& b = B.box[first + i];
            int k = (int)(((b.mn[axis] + b.mx[axis]) * 0.5f - cb.mn[axis]) * scale);
            if (k < 0) k = 0; if (k >= SAH_BINS) k = SAH_BINS - 1;
            cnt[k]++; box_grow(bb[k], b);
        }
        float right_area[SAH_BINS]; int right_cnt[SAH_BINS];
        Box acc = box_empty(); int n = 0;
        for (int k = SAH_BINS - 1; k > 0; k--) { box_grow(acc, bb[k]); n += cnt[k]; right_area[k] = box_area(acc); right_cnt[k] = n; }
        acc = box_empty(); n = 0;
        for (int k = 0; k < SAH_BINS - 1; k++) {
            box_grow(acc, bb[k]); n += cnt[k];
            if (n == 0 || right_cnt[k + 1] == 0) continue;
            const float cost = box_area(acc) * (float)n + right_area[k + 1] * (float)right_cnt[k + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
        }
    }
    if (best_axis < 0) return 0;
    /* stable partition by bin (keeps the builder deterministic) */
    const float scale = (float)SAH_BINS / (cb.mx[best_axis] - cb.mn[best_axis]);
    std::vector<int32_t> t(count), r(count), d(count);
    std::vector<Box> bx(count);
    int nl = 0;
    int w = 0;
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < count; i++) {
            const Box& b = B.box[first + i];
            int k = (int)(((b.mn[best_axis] + b.mx[best_axis]) * 0.5f - cb.mn[best_axis]) * scale);
            if (k < 0) k = 0; if (k >= SAH_BINS) k = SAH_BINS - 1;
            if ((k <= best_bin) == (pass == 0)) {
                const int s = first + i;
                t[w] = B.type[s]; r[w] = B.ref[s]; d[w] = B.id[s]; bx[w] = B.box[s];
                w++;
                if (pass == 0) nl++;
            }
        }
    for (int i = 0; i < count; i++) { B.type[first + i] = t[i]; B.ref[first + i] = r[i]; B.id[first + i] = d[i]; B.box[first + i] = bx[i]; }
    *axis_out = best_axis;
    return nl;
}

/* one binary split of [first, first+count): SAH when allowed and useful, else the reference's median split */
int split_range(Builder& B, int first, int count, bool allow_sah, int* axis_out)
{
    if (count < 2) { *axis_out = 0; return count; }
    if (allow_sah) {
        const int nl = sah_partition(B, first, count, axis_out);
        if (nl > 0 && nl < count) return nl;
    }
    const int axis = longest_axis(range_box(B, first, count));
    sort_range(B, first, count, axis);
    *axis_out = axis;
    return count / 2;
}

/* Appends the subtree of objects [first, first+count) to `out` (indices relative to out's start); returns its root. */
int build4_sah(Builder& B, std::vector<crt_bvh4_node>& out, int first, int count, int depth, int max_depth)
{
    const int self = (int)out.size();
    out.push_back(crt_bvh4_node());
    /* SAH only while a median-split subtree for any part would still fit below the depth limit */
    const bool allow_sah = depth + 1 + levels_needed(count) <= max_depth;
    int ax0 = 0, ax1 = 0, ax2 = 0;
    const int mid = split_range(B, first, count, allow_sah, &ax0);
    const int left = split_range(B, first, mid, allow_sah, &ax1);
    const int right = mid + split_range(B, first + mid, count - mid, allow_sah, &ax2);
    const int start[5] = { 0, left, mid, right, count };
    crt_bvh4_node nd;
    memset(&nd, 0, sizeof nd);
    nd.axis = (uint8_t)ax2;
    int child_cnt[4];
    for (int c = 0; c < 4; c++) {
        const int cnt = start[c + 1] - start[c];
        child_cnt[c] = cnt;
        if (cnt > 0) {
            const Box b = range_box(B, first + start[c], cnt);
            nd.bbox[c] = b.mn[0]; nd.bbox[4 + c] = b.mn[1]; nd.bbox[8 + c] = b.mn[2];
            nd.bbox[12 + c] = b.mx[0]; nd.bbox[16 + c] = b.mx[1]; nd.bbox[20 + c] = b.mx[2];
        } else {                                     /* empty child: a box no ray enters */
            nd.bbox[c] = nd.bbox[4 + c] = nd.bbox[8 + c] = 1.0f;
            nd.bbox[12 + c] = nd.bbox[16 + c] = nd.bbox[20 + c] = -1.0f;
        }
        if (cnt > CRT_SAH_LEAF) { nd.type[c] = -1; nd.child[c] = -1; }
        else { nd.type[c] = (int8_t)cnt; nd.child[c] = first + start[c]; }
    }
    const bool fork = depth < fork_levels(B.max_threads) && count > 65536;
    if (fork) {
        std::vector<std::vector<crt_bvh4_node> > sub(4);
        std::vector<std::thread> th;
        for (int c = 0; c < 4; c++) if (child_cnt[c] > CRT_SAH_LEAF) {
            std::vector<crt_bvh4_node>* dst = &sub[c];
            const int f = first + start[c], cnt = child_cnt[c];
            th.emplace_back([&B, dst, f, cnt, depth, max_depth]() { build4_sah(B, *dst, f, cnt, depth + 1, max_depth); });
        }
        for (auto& t : th) t.join();
        for (int c = 0; c < 4; c++) if (child_cnt[c] > CRT_SAH_LEAF) {
            const int base = (int)out.size();
            nd.child[c] = base;
            for (crt_bvh4_node n : sub[c]) {
                for (int k = 0; k < 4; k++) if (n.type[k] == -1) n.child[k] += base;
                out.push_back(n);
            }
        }
    } else {
        for (int c = 0; c < 4; c++) if (child_cnt[c] > CRT_SAH_LEAF) {
            std::vector<crt_bvh4_node> sub;
            build4_sah(B, sub, first + start[c], child_cnt[c], depth + 1, max_depth);
            const int base = (int)out.size();
            nd.child[c] = base;
            for (crt_bvh4_node n : sub) {
                for (int k = 0; k < 4; k++) if (n.type[k] == -1) n.child[k] += base;
                out.push_back(n);
            }
        }
    }
    out[self] = nd;
    return self;
}

/* BVH_build, bvh.h:117-154: node box = union of the range, leaf at 1 object, median split. */
void build2(Builder& B, int first, int count, int self, int depth)
{
    Box all = range_box(B, first, count);
    crt_bvh2_node nd;
    for (int i = 0; i < 3; i++) { nd.bmin[i] = all.mn[i]; nd.bmax[i] = all.mx[i]; }
    if (count <= 1) {
        nd.left = -1; nd.right = first;
        B.nodes2[self] = nd;
        return;
    }
    sort_range(B, first, count, longest_axis(all));
    int k = count / 2;
    nd.left = self + 1;
    nd.right = self + 1 + (2 * k - 1);
    B.nodes2[self] = nd;
    if (depth < 3 && B.max_threads >= 8 && count > 65536) {
        std::thread t([&B, first, k, self, depth]() { build2(B, first, k, self + 1, depth + 1); });
        build2(B, first + k, count - k, nd.right, depth + 1);
        t.join();
    } else {
        build2(B, first, k, self + 1, depth + 1);
        build2(B, first + k, count - k, nd.right, depth + 1);
    }
}

/* ---- uniform grid: UniformGrid_create, uniformgrid.h:54-154 ---- */

inline float fmin2(float a, float b) { return a < b ? a : b; }
inline float fmax2(float a, float b) { return a > b ? a : b; }
inline float fmin3(float a, float b, float c) { return fmin2(a, fmin2(b, c)); }
inline float fmax3(float a, float b, float c) { return fmax2(a, fmax2(b, c)); }
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float dot(const float* a, const float* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }

/* triangle / box overlap: testTriangleAABB, shapes.cpp:434-565 (9 edge-cross axes, 3 box axes,
 * triangle plane via testAABBPlane :416-431).  Operation order kept: the grid's cell lists, and
 * with them which rays find which triangle first, depend on its rounding. */
bool tri_overlaps_cell(const float* t0, const float* t1, const float* t2, const float* cmin, const float* cmax)
{
    float half[3], ctr[3];
    for (int i = 0; i < 3; i++) { float span = cmax[i] - cmin[i]; half[i] = span * 0.5f; ctr[i] = half[i] + cmin[i]; }
    float v[3][3];
    for (int i = 0; i < 3; i++) { v[0][i] = t0[i] - ctr[i]; v[1][i] = t1[i] - ctr[i]; v[2][i] = t2[i] - ctr[i]; }
    float edge[3][3];
    for (int i = 0; i < 3; i++) { edge[0][i] = v[1][i] - v[0][i]; edge[1][i] = v[2][i] - v[1][i]; edge[2][i] = v[0][i] - v[2][i]; }
    for (int ax = 0; ax < 3; ax++) {
        float u[3] = { ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f };
        for (int ed = 0; ed < 3; ed++) {
            const float* e = edge[ed];
            float a[3] = { u[1]*e[2] - u[2]*e[1], u[2]*e[0] - u[0]*e[2], u[0]*e[1] - u[1]*e[0] };
            float p0 = dot(v[0], a), p1 = dot(v[1], a), p2 = dot(v[2], a);
            float r = half[0] * fabsf(a[0]) + half[1] * fabsf(a[1]) + half[2] * fabsf(a[2]);
            float sep = (ax == 0 && ed == 0) ? fmax2(-fmax2(p0, p2), fmin2(p0, p2))     /* shapes.cpp:505 */
                                             : fmax2(-fmax3(p0, p1, p2), fmin3(p0, p1, p2));
            if (sep > r) return false;
        }
    }
    for (int i = 0; i < 3; i++)
        if (fmax3(v[0][i], v[1][i], v[2][i]) < -half[i] || fmin3(v[0][i], v[1][i], v[2][i]) > half[i]) return false;
    float nrm[3] = { edge[0][1]*edge[1][2] - edge[0][2]*edge[1][1],
                     edge[0][2]*edge[1][0] - edge[0][0]*edge[1][2],
                     edge[0][0]*edge[1][1] - edge[0][1]*edge[1][0] };
    float len = sqrtf(dot(nrm, nrm));
    if (len == 0) { nrm[0] = nrm[1] = nrm[2] = 0.0f; }
    else { float inv = 1.0f / len; nrm[0] *= inv; nrm[1] *= inv; nrm[2] *= inv; }
    float plane_d = dot(nrm, t0);
    /* testAABBPlane recomputes centre as min + half */
    float c2[3];
    for (int i = 0; i < 3; i++) c2[i] = cmin[i] + half[i];
    float r = half[0] * fabsf(nrm[0]) + half[1] * fabsf(nrm[1]) + half[2] * fabsf(nrm[2]);
    float s = dot(nrm, c2) - plane_d;
    return fabsf(s) <= r;
}

/* (int)clamp(v, 0, n - 1) of uniformgrid.h:100-105, with a NaN coordinate (which the reference would
 * convert to int, undefined) sent to cell 0 so that no index can leave the grid */
static inline int cell_of(float v, int n)
{
    if (!(v > 0.0f)) return 0;
    if (v >= (float)n - 1) return n - 1;
    return (int)v;
}

/* Returns CRT_E_INVALID when the reference's resolution formula has no usable value: a zero-volume
 * or non-finite bounding box makes s zero / NaN and the reference then converts inf or NaN to int
 * (uniformgrid.h:73-80, undefined behaviour there); CRT_E_TOOBIG beyond 2^30 cells. */
int build_grid(Builder& B)
{
    const int multiplier = 3;                           /* buildscene.h:705 */
    crt_grid& g = B.grid;
    Box r;
    for (int i = 0; i < 3; i++) { r.mn[i] = FLT_MAX; r.mx[i] = -CRT_HUGEVALUE; }
    for (int k = B.nng; k < B.n; k++)
        for (int i = 0; i < 3; i++) {
            if (B.box[k].mn[i] < r.mn[i]) r.mn[i] = B.box[k].mn[i];
            if (B.box[k].mx[i] > r.mx[i]) r.mx[i] = B.box[k].mx[i];
        }
    for (int i = 0; i < 3; i++) { g.bmin[i] = r.mn[i]; g.bmax[i] = r.mx[i]; }
    g.wx = r.mx[0] - r.mn[0]; g.wy = r.mx[1] - r.mn[1]; g.wz = r.mx[2] - r.mn[2];
    if (B.n - B.nng == 0) {                             /* nothing to put in a grid: one empty cell inside the
                                                           inverted box, which no ray enters */
        g.nx = g.ny = g.nz = 1; g.num_cells = 1;
        B.cell_start.assign(2, 0);
        B.cell_items.clear();
        return CRT_OK;
    }
    float s = powf(g.wx * g.wy * g.wz / B.n, 0.33333f);
    const float fx = g.wx * multiplier / s + 1, fy = g.wy * multiplier / s + 1, fz = g.wz * multiplier / s + 1;
    const float lim = 1048576.0f;
    if (!(fx >= 1.0f && fx < lim && fy >= 1.0f && fy < lim && fz >= 1.0f && fz < lim)) return CRT_E_INVALID;
    g.nx = (int)fx;
    g.ny = (int)fy;
    g.nz = (int)fz;
    if ((long long)g.nx * g.ny * g.nz > (1ll << 30)) return CRT_E_TOOBIG;
    g.num_cells = g.nx * g.ny * g.nz;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const float ux = g.wx / nx, uy = g.wy / ny, uz = g.wz / nz;

    /* two passes (count, fill) over the same deterministic enumeration -> CSR with items in
     * object-array order inside each cell, as repeated IntVector_push gives */
    std::vector<int32_t> count((size_t)g.num_cells + 1, 0);
    std::vector<int32_t> pairs_cell, pairs_obj;
    for (int k = B.nng; k < B.n; k++) {
        const Box& a = B.box[k];
        int lo[3], hi[3];
        const float w[3] = { g.wx, g.wy, g.wz };
        const int nn[3] = { nx, ny, nz };
        for (int i = 0; i < 3; i++) {
            lo[i] = cell_of(((a.mn[i] - r.mn[i]) / w[i]) * nn[i], nn[i]);
            hi[i] = cell_of(((a.mx[i] - r.mn[i]) / w[i]) * nn[i], nn[i]);
        }
        bool tri = is_tri(B.type[k]);
        const float* v = tri ? B.in->tri_verts + (size_t)B.ref[k] * 9 : 0;
        for (int j = lo[1]; j <= hi[1]; ++j)
            for (int kk = lo[2]; kk <= hi[2]; ++kk)
                for (int p = lo[0]; p <= hi[0]; ++p) {
                    if (tri) {
                        float disp[3] = { p * ux, j * uy, kk * uz };
                        float cmin[3] = { r.mn[0] + disp[0], r.mn[1] + disp[1], r.mn[2] + disp[2] };
                        float cmax[3] = { cmin[0] + ux, cmin[1] + uy, cmin[2] + uz };
                        if (!tri_overlaps_cell(v, v + 3, v + 6, cmin, cmax)) continue;
                    }
                    int cell = j*nx*nz + kk*nx + p;                 /* uniformgrid.h:122 */
                    pairs_cell.push_back(cell);
                    pairs_obj.push_back(k);
                    count[cell]++;
                }
    }
    B.cell_start.assign((size_t)g.num_cells + 1, 0);
    for (int c = 0; c < g.num_cells; c++) B.cell_start[c+1] = B.cell_start[c] + count[c];
    B.cell_items.assign(pairs_obj.size(), 0);
    std::vector<int32_t> fill(B.cell_start.begin(), B.cell_start.end() - 1);
    for (size_t i = 0; i < pairs_obj.size(); i++) B.cell_items[fill[pairs_cell[i]]++] = pairs_obj[i];
    return CRT_OK;
}

} // namespace

struct crt_built_scene {
    Builder B;
    std::vector<float> tri;
    std::vector<crt_shape> shapes;
    crt_scene_desc desc;
};

extern "C" {

int crt_build_scene(const crt_scene_input* in, int accel, crt_built_scene** out)
{
    return crt_build_scene_ex(in, accel, 0, out);
}

int crt_build_scene_ex(const crt_scene_input* in, int accel, int flags, crt_built_scene** out)
{
    if (flags & ~CRT_BUILD_SAH) return CRT_E_INVALID;
    if ((flags & CRT_BUILD_SAH) && accel != CRT_ACCEL_BVH4) return CRT_E_UNSUPPORTED;
    if (!in || !out || in->num_obj < 0 || in->num_tris < 0 || in->num_shapes < 0) return CRT_E_INVALID;
    if (accel < CRT_ACCEL_NONE || accel > CRT_ACCEL_BVH4) return CRT_E_INVALID;
    if (in->num_obj > 0 && in->obj_type && !in->obj_ref) return CRT_E_INVALID;
    if (!in->obj_type && in->num_obj != in->num_tris) return CRT_E_INVALID;
    if (in->num_tris > 0 && !in->tri_verts) return CRT_E_INVALID;
    if (in->num_shapes > 0 && !in->shapes) return CRT_E_INVALID;
    if (in->num_obj >= (1 << 28)) return CRT_E_TOOBIG;          /* leaf words carry 28-bit object indices */
    for (int i = 0; i < in->num_shapes; i++) {
        const crt_shape& sh = in->shapes[i];
        if (sh.type < 0 || sh.type > CRT_COMPOUND || sh.type == CRT_TRIANGLE) return CRT_E_INVALID;
        if (sh.type == CRT_INSTANCED && (sh.inner < 0 || sh.inner >= in->num_shapes || sh.inner == i)) return CRT_E_INVALID;
        /* nesting the scene grammar cannot produce and crt_scene_create refuses (an instance of an instance, a compound inside a
         * compound): refused here too, before shape_box / grid_shape recurse through it -- a two-shape cycle would never return */
        if (sh.type == CRT_INSTANCED && in->shapes[sh.inner].type == CRT_INSTANCED) return CRT_E_UNSUPPORTED;
        if (sh.type == CRT_COMPOUND && (sh.first_part < 0 || sh.num_parts < 0 || sh.first_part + sh.num_parts > in->num_shapes))
            return CRT_E_INVALID;
        if (sh.type == CRT_COMPOUND)
            for (int k = 0; k < sh.num_parts; k++) {
                const int pt = in->shapes[sh.first_part + k].type;
                if (pt == CRT_COMPOUND || pt == CRT_INSTANCED) return CRT_E_UNSUPPORTED;
            }
    }
    crt_built_scene* bs = new (std::nothrow) crt_built_scene();
    if (!bs) return CRT_E_NOMEM;
    try {
        Builder& B = bs->B;
        B.in = in;
        B.n = in->num_obj;
        B.max_threads = (int)std::thread::hardware_concurrency();
        if (B.max_threads > 16) B.max_threads = 16;
        /* one process per GPU: a launcher that starts k ranks on a node gives each its share of the host cores
         * (bench.py sets CRT_BUILD_THREADS = cores / LOCAL_WORLD_SIZE); the tree does not depend on the thread count */
        if (const char* e = getenv("CRT_BUILD_THREADS")) { const int v = atoi(e); if (v >= 1 && v < B.max_threads) B.max_threads = v; }
        if (B.max_threads < 1) B.max_threads = 1;
        B.type.resize(B.n); B.ref.resize(B.n); B.id.resize(B.n); B.box.resize(B.n);
        std::vector<char> gridable(B.n);
        for (int i = 0; i < B.n; i++) {
            int t = in->obj_type ? in->obj_type[i] : (int)CRT_SMOOTH_TRIANGLE;
            int r = in->obj_type ? in->obj_ref[i] : i;
            if (is_tri(t)) { if (r < 0 || r >= in->num_tris) { delete bs; return CRT_E_INVALID; } }
            else { if (t < 0 || t > CRT_SMOOTH_TRIANGLE || r < 0 || r >= in->num_shapes || in->shapes[r].type != t) { delete bs; return CRT_E_INVALID; } }
            B.type[i] = t; B.ref[i] = r; B.id[i] = i;
            gridable[i] = is_tri(t) ? 1 : (grid_shape(in, r) ? 1 : 0);
        }
        /* mvNonGridObjToStart, scenedata.h:368-382: swap each unbounded object down to slot j */
        int j = 0;
        for (int i = 0; i < B.n; ++i) if (!gridable[i]) {
            std::swap(B.type[i], B.type[j]); std::swap(B.ref[i], B.ref[j]); std::swap(B.id[i], B.id[j]);
            std::swap(gridable[i], gridable[j]);
            ++j;
        }
        B.nng = j;
        for (int i = 0; i < B.n; i++) object_box(in, B.type[i], B.ref[i], &B.box[i]);
        memset(&B.grid, 0, sizeof B.grid);
        int na = B.n - B.nng;
        if (accel == CRT_ACCEL_BVH4 && na > 0 && (flags & CRT_BUILD_SAH)) {
            bool bounded = true;                       /* crt_scene_create's rule for the pruned walk (crt_gpu.hip: accel_bounded) */
            for (int i = B.nng; i < B.n; i++) {
                const int t = B.type[i];
                if (!(is_tri(t) || t == CRT_SPHERE || t == CRT_RECTANGLE || t == CRT_DISK)) bounded = false;
            }
            build4_sah(B, B.nodes4, B.nng, na, 0, bounded ? SAH_MAX_DEPTH_PRUNED : SAH_MAX_DEPTH_EXACT);
        } else if (accel == CRT_ACCEL_BVH4 && na > 0) {
            NodeCountCache ncc;
            long long total = na > 4 ? ncc.get(na) : 1;
            B.nodes4.resize((size_t)total);
            build4(B, ncc, B.nng, na, 0, 0);
        } else if (accel == CRT_ACCEL_BVH && na > 0) {
            B.nodes2.resize((size_t)2 * na - 1);
            build2(B, B.nng, na, 0, 0);
        } else if (accel == CRT_ACCEL_GRID) {
            const int rc = build_grid(B);
            if (rc != CRT_OK) { delete bs; return rc; }
        }
        bs->tri.assign(in->tri_verts, in->tri_verts + (size_t)in->num_tris * 9);
        if (in->num_shapes) bs->shapes.assign(in->shapes, in->shapes + in->num_shapes);
        std::vector<Box>().swap(B.box);
        B.in = 0;
        crt_scene_desc& d = bs->desc;
        memset(&d, 0, sizeof d);
        d.accel = accel; d.num_obj = B.n; d.num_non_grid_obj = B.nng;
        d.obj_type = B.type.data(); d.obj_ref = B.ref.data(); d.obj_id = B.id.data();
        d.tri_verts = bs->tri.data(); d.num_tris = in->num_tris;
        d.shapes = bs->shapes.data(); d.num_shapes = in->num_shapes;
        d.nodes4 = B.nodes4.data(); d.num_nodes4 = (int)B.nodes4.size();
        d.nodes2 = B.nodes2.data(); d.num_nodes2 = (int)B.nodes2.size();
        d.grid = B.grid;
        d.cell_start = B.cell_start.data(); d.cell_items = B.cell_items.data();
    } catch (const std::bad_alloc&) {
        delete bs;
        return CRT_E_NOMEM;
    } catch (const std::exception&) {                   /* nothing may cross the C ABI */
        delete bs;
        return CRT_E_INVALID;
    }
    *out = bs;
    return CRT_OK;
}

const crt_scene_desc* crt_built_scene_desc(const crt_built_scene* b) { return b ? &b->desc : 0; }
void crt_built_scene_destroy(crt_built_scene* b) { delete b; }

} // extern "C"
