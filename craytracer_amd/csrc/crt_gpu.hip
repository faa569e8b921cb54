/*
 * crt_gpu.hip -- the C ABI of include/crt_gpu.h over the kernels in crt_kernels.h.
 *
 * Replaces, for the intersect step only, the reference's per-pixel worker loop
 * (parallel.h + main.cpp:88-149) that calls intersectTest / shadowIntersectTest
 * (intersect.h:365,:443) one ray at a time: the host hands over a batch of rays and gets
 * a batch of hit records back.
 */
#include "crt_kernels.h"
#include "crt_wide8.h"
#include "crt_wide8_host.h"
#include "crt_binning.h"
#include "crt_camera.h"
#include "crt_internal.h"

#include <string.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <new>
#include <mutex>
#include <atomic>

namespace {

thread_local char g_hip_err[256] = "";

bool hip_ok(hipError_t e, const char* what)
{
    if (e == hipSuccess) return true;
    snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
    return false;
}
#define HIPCHK(call) do { if (!hip_ok((call), #call)) return CRT_E_HIP; } while (0)

struct DeviceGuard {
    int prev; bool ok;
    explicit DeviceGuard(int dev) : prev(-1), ok(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hip_ok(hipSetDevice(dev), "hipSetDevice");
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

} // namespace

/* Device memory a launch needs beyond the caller's rays and results. */
struct Scratch {
    unsigned* queue_head;     /* 256 B, word 0 used; zeroed on the launch stream before every launch */
    unsigned* ovf_word; float* ovf_key; unsigned ovf_threads;   /* per-lane stack overflow, [CRT_OVF_DEPTH][ovf_threads] */
    unsigned* bin_hist; unsigned* bin_cursor;                   /* ray binning: CRT_BIN_CELLS counters each */
    unsigned* perm; size_t perm_cap;                            /* ray order, one word per ray */
    unsigned short* cells;                                      /* binning cell of every ray (written by the histogram pass, read by the
                                                                   scatter pass instead of the 24-byte ray records), or null */
    unsigned* fb_list;                                          /* perm_cap words: rays the 8-wide walk leaves to the 4-wide one (crt_wide8.h), or null */
};
/* A scene owns CRT_SLOTS sets of launch scratch, used in turn: launch k waits (on the device) only for launch k - CRT_SLOTS, the
 * last one that used ITS set, so two launches enqueued on two streams overlap -- the drain tail of one (its longest rays walking
 * on alone, about 0.4 ms whatever the batch size) runs beside the start of the next.  On one stream nothing changes. */
#ifndef CRT_SLOTS
#define CRT_SLOTS 2
#endif

/* Events of one launch.  A scene keeps a ring of them so that each caller can ask for the duration of
 * ITS launch (crt_scene_last_kernel_ms) while other threads keep launching through the same scene. */
#define CRT_TIMING_RING 16
struct LaunchTiming { hipEvent_t ev_bin, ev_start, ev_stop; unsigned long long seq; bool binned; };

#define CRT_OCC_CACHE 32
struct OccEntry { const void* fn; int per_cu; };

struct crt_scene {
    int device;
    unsigned long long uid;   /* distinguishes scenes that reuse an address (thread-local launch tokens) */
    DevScene dev;
    bool tri_only;            /* every object is a triangle type */
    bool tri_rect_only;       /* every object is a triangle type or a RECTANGLE (and not all are triangles) */
    bool accel_tri_only;      /* every object below the accelerator is a triangle type */
    bool accel_bounded;       /* ... or another shape whose box bounds it: sphere, rectangle, disk (shapes.cpp:146-263) */
    int depth4, depth2;       /* levels of the QBVH / BVH */
    int mode;                 /* requested crt_traverse_mode */
    bool count;
    bool compress;            /* walk the 64-byte conservative copy of the QBVH nodes where that is allowed (see crt_scene_set_node_compression) */
    bool cnodes_valid;        /* the copy exists (every node box was finite) */
    bool wide8_valid;         /* the 8-wide copy exists (crt_wide8_host.h) */
    bool wide8;               /* ... and closest-hit launches walk it (crt_scene_set_wide_nodes) */
    unsigned split_max_rays;  /* closest-hit batches up to this many rays take the tail-splitting walk */
    unsigned wide8_min_rays;  /* closest-hit batches of more rays than this take the 8-wide walk (CRT_WIDE8_MIN) */
    bool bvh2_ordered;        /* binary BVH: left-first depth-first order meets the leaves in increasing object index, so the pruned
                                 walk may take the nearer child first and settle equal t by index (k_bvh2 ORDERED) */
    int num_cus;
    /* device allocations */
    void* d_nodes4; void* d_cnodes4; void* d_wnodes8; void* d_objs; void* d_shapes; void* d_nodes2; void* d_bnodes2; void* d_bnodes2w; void* d_bnodes2wc; void* d_cell_start; void* d_cell_items; void* d_cell_dist; void* d_cell_objs;
    unsigned long long* d_counters;
    Scratch own[CRT_SLOTS];   /* allocated by crt_scene_create (perm / cells may grow); launch k uses own[k % CRT_SLOTS] */
    unsigned long long slot_last[CRT_SLOTS];   /* launch number (seq) of the last launch through each set, 0 = none yet */
    unsigned long long own_launches;           /* launches through the own sets so far */
    Scratch na;               /* views into this GPU's copy of the file-scope `scratch` array (crt_intersect_batch_dev_noalloc) */
    LaunchTiming ring[CRT_TIMING_RING]; int ring_valid;   /* events created so far */
    unsigned long long seq;   /* launches so far; launch k uses ring[k % CRT_TIMING_RING] */
    OccEntry occ[CRT_OCC_CACHE]; int n_occ;               /* resident blocks per CU of each kernel, queried once */
    /* A launch uses one of the scene's scratch sets (queue head, ray order, stack overflow) and waits, on the device,
     * for the launch that used that set before it, even when it is enqueued on another stream (see CRT_SLOTS).
     * `mu` guards the launch sequence and the events against concurrent callers. */
    std::mutex mu;
    /* host-buffer entry points: two chunk-sized device buffers and two streams, created on first use */
    void* d_chunk_in[2]; void* d_chunk_aux[2]; void* d_chunk_out[2]; void* d_chunk_out2[2];
    size_t chunk_cap;         /* rays per chunk buffer */
    /* ray binning (crt_binning.h): triangle-only QBVH scenes, batches of at least CRT_BIN_MIN_RAYS */
    unsigned root_word2;      /* BVH: root in k_bvh2's child-word form */
    unsigned root_word2w;     /* ... and in the collapsed (four-box) records */
    bool bvh2_wide;           /* the collapsed records exist and the ordered walk may use them */
    bool bvh2_wcomp;          /* ... and so does their 64-byte conservative copy (walked when `compress` is on; CRT_BVH2_WCOMP=0: never) */
    bool binning; BinGrid bin_grid; bool bin_grid_valid;
    hipStream_t chunk_stream[2]; bool chunk_streams_valid;
};

#ifndef CRT_SPLIT_MAX_RAYS_DEFAULT
#define CRT_SPLIT_MAX_RAYS_DEFAULT (1u << 23)     /* +18 % at 2^20 rays, +12 % at 2^21, +7 % at 2^22, +3 % at 2^23, -1 % at 2^24 (CRT_TAIL_SPLIT_MAX overrides).
                                                     A stall of four bench processes sharing one GPU, first seen with this on, reproduces
                                                     without this library at all (tools/exp/four_procs.py, mode sel+copy: torch selects,
                                                     randperm and copies only) */
#endif
#ifndef CRT_NODE_ORDER_DEFAULT
#define CRT_NODE_ORDER_DEFAULT 1
#endif

namespace {

int resolved_mode(const crt_scene* s)
{
    if (s->mode == CRT_TRAVERSE_AUTO) return s->accel_bounded ? CRT_TRAVERSE_PRUNED : CRT_TRAVERSE_EXACT;
    return s->mode;
}

/* levels below (and including) node `root` of the flat QBVH / BVH */
int depth4(const crt_scene_desc* d)
{
    if (d->num_nodes4 == 0) return 0;
    std::vector<std::pair<int, int> > stack;
    stack.push_back(std::make_pair(0, 1));
    int best = 0;
    while (!stack.empty()) {
        std::pair<int, int> e = stack.back(); stack.pop_back();
        if (e.second > best) best = e.second;
        const crt_bvh4_node& n = d->nodes4[e.first];
        for (int c = 0; c < 4; c++) if (n.type[c] == -1) stack.push_back(std::make_pair(n.child[c], e.second + 1));
    }
    return best;
}
int depth2(const crt_scene_desc* d)
{
    if (d->num_nodes2 == 0) return 0;
    std::vector<std::pair<int, int> > stack;
    stack.push_back(std::make_pair(0, 1));
    int best = 0;
    while (!stack.empty()) {
        std::pair<int, int> e = stack.back(); stack.pop_back();
        if (e.second > best) best = e.second;
        const crt_bvh2_node& n = d->nodes2[e.first];
        if (n.left >= 0) { stack.push_back(std::make_pair(n.left, e.second + 1)); stack.push_back(std::make_pair(n.right, e.second + 1)); }
    }
    return best;
}


/* Order of the QBVH nodes in device memory (results do not depend on it: child words are renumbered).  The host hands the nodes
 * over in pre-order, where a node's first inner child is its neighbour and the other three lie a whole subtree away.
 *   policy 0: as given;
 *   policy 1: sibling clusters — the inner children of a node lie side by side (<= 256 B), clusters in depth-first order;
 *   policy 2: the same, a cluster of two or more starting on a 128-byte line (the gaps are dead records);
 *   policy >= 4: treelets of `policy` nodes grown from their root by largest box area first (the subtree tops a ray is most
 *                likely to visit together share lines), treelets in depth-first order.
 * order[new] = old (-1: a dead record), newid[old] = new. */
void node_layout(const crt_scene_desc* d, int policy, std::vector<int>& order, std::vector<int>& newid)
{
    const int n = d->num_nodes4;
    order.clear(); newid.assign((size_t)n, -1);
    if (n == 0) return;
    if (policy <= 0) { order.resize((size_t)n); for (int i = 0; i < n; i++) { order[i] = i; newid[i] = i; } return; }
    auto place = [&](int node) { newid[node] = (int)order.size(); order.push_back(node); };
    if (policy < 4) {
        std::vector<int> stack;
        place(0); stack.push_back(0);
        while (!stack.empty()) {
            const int node = stack.back(); stack.pop_back();
            const crt_bvh4_node& nd = d->nodes4[node];
            int kids[4], nk = 0;
            for (int c = 0; c < 4; c++) if (nd.type[c] == -1 && newid[nd.child[c]] < 0) {
                bool dup = false;
                for (int k = 0; k < nk; k++) dup = dup || kids[k] == nd.child[c];
                if (!dup) kids[nk++] = nd.child[c];
            }
            if (policy == 2 && nk >= 2 && (order.size() & 1)) order.push_back(-1);
            for (int k = 0; k < nk; k++) place(kids[k]);
            for (int k = nk - 1; k >= 0; k--) stack.push_back(kids[k]);
        }
    } else {
        struct Cand { float area; int node; bool operator<(const Cand& o) const { return area < o.area; } };
        auto child_area = [&](const crt_bvh4_node& nd, int c) -> float {
            const float ex = nd.bbox[12 + c] - nd.bbox[c], ey = nd.bbox[16 + c] - nd.bbox[4 + c], ez = nd.bbox[20 + c] - nd.bbox[8 + c];
            const float a = ex * ey + ey * ez + ez * ex;
            return a == a ? a : 0.0f;
        };
        std::vector<int> roots;                      /* treelet roots still to be laid out, depth-first */
        roots.push_back(0);
        std::vector<Cand> heap;
        while (!roots.empty()) {
            const int root = roots.back(); roots.pop_back();
            if (newid[root] >= 0) continue;
            heap.clear();
            heap.push_back(Cand{ 3.0e38f, root });
            int placed = 0;
            while (!heap.empty() && placed < policy) {
                std::pop_heap(heap.begin(), heap.end());
                const Cand c = heap.back(); heap.pop_back();
                if (newid[c.node] >= 0) continue;
                place(c.node); placed++;
                const crt_bvh4_node& nd = d->nodes4[c.node];
                for (int k = 0; k < 4; k++) if (nd.type[k] == -1 && newid[nd.child[k]] < 0) {
                    heap.push_back(Cand{ child_area(nd, k), nd.child[k] });
                    std::push_heap(heap.begin(), heap.end());
                }
            }
            /* what is left on the frontier roots its own treelet; the largest is laid out next */
            std::sort(heap.begin(), heap.end());
            for (size_t k = 0; k < heap.size(); k++) roots.push_back(heap[k].node);
        }
    }
    for (int i = 0; i < n; i++) if (newid[i] < 0) place(i);          /* not reachable from the root: kept, at the end */
}

/* Uniform grid: per cell the chessboard (Chebyshev) distance to the nearest cell that holds an object, 0 for a cell that holds
 * one, capped at 255.  A 3D-DDA step changes one cell coordinate by one, so from a cell at distance d the next d - 1 cells of ANY
 * ray are empty: k_grid steps through them without reading cell_start.  Two-pass chamfer over a volume padded by one cell (exact
 * for this metric). */
void grid_distance_field(const crt_scene_desc* d, std::vector<unsigned char>& out)
{
    const int nx = d->grid.nx, ny = d->grid.ny, nz = d->grid.nz;
    const size_t px = (size_t)nx + 2, pz = (size_t)nz + 2, py = (size_t)ny + 2;
    std::vector<unsigned char> v(px * py * pz, (unsigned char)255);
    auto at = [&](int ix, int iy, int iz) -> size_t { return ((size_t)(iy + 1) * pz + (size_t)(iz + 1)) * px + (size_t)(ix + 1); };
    for (int iy = 0; iy < ny; iy++) for (int iz = 0; iz < nz; iz++) for (int ix = 0; ix < nx; ix++) {
        const size_t c = ((size_t)iy * nz + iz) * nx + ix;
        if (d->cell_start[c + 1] > d->cell_start[c]) v[at(ix, iy, iz)] = 0;
    }
    ptrdiff_t before[13]; int nb = 0;             /* the 13 neighbours that precede a cell in scan order (y, then z, then x) */
    for (int dy = -1; dy <= 1; dy++) for (int dz = -1; dz <= 1; dz++) for (int dx = -1; dx <= 1; dx++) {
        const ptrdiff_t off = ((ptrdiff_t)dy * (ptrdiff_t)pz + dz) * (ptrdiff_t)px + dx;
        if (off < 0) before[nb++] = off;
    }
    for (int iy = 0; iy < ny; iy++) for (int iz = 0; iz < nz; iz++) {
        unsigned char* row = &v[at(0, iy, iz)];
        for (int ix = 0; ix < nx; ix++) {
            unsigned m = row[ix];
            if (m == 0) continue;
            for (int k = 0; k < 13; k++) { const unsigned q = (unsigned)row[ix + before[k]] + 1u; if (q < m) m = q; }
            row[ix] = (unsigned char)m;
        }
    }
    for (int iy = ny - 1; iy >= 0; iy--) for (int iz = nz - 1; iz >= 0; iz--) {
        unsigned char* row = &v[at(0, iy, iz)];
        for (int ix = nx - 1; ix >= 0; ix--) {
            unsigned m = row[ix];
            if (m == 0) continue;
            for (int k = 0; k < 13; k++) { const unsigned q = (unsigned)row[ix - before[k]] + 1u; if (q < m) m = q; }
            row[ix] = (unsigned char)m;
        }
    }
    out.resize((size_t)nx * ny * nz);
    for (int iy = 0; iy < ny; iy++) for (int iz = 0; iz < nz; iz++)
        memcpy(&out[((size_t)iy * nz + iz) * nx], &v[at(0, iy, iz)], (size_t)nx);
}

/* Binary BVH: does a left-first depth-first walk meet the leaves in strictly increasing object index?  (BVH_build's in-place
 * partition gives that.)  Then "the last of equal hits in left-first order" (intersect.h:339, t1 < t2 ? left : right) is "the
 * largest object index", whatever order a walk meets them in. */
bool bvh2_leaves_in_index_order(const crt_scene_desc* desc)
{
    if (desc->num_nodes2 < 1) return false;
    std::vector<int> stack;
    stack.push_back(0);
    int last = -1;
    while (!stack.empty()) {
        const int node = stack.back(); stack.pop_back();
        const crt_bvh2_node& c = desc->nodes2[node];
        if (c.left < 0) { if (!(c.right > last)) return false; last = c.right; }
        else { stack.push_back(c.right); stack.push_back(c.left); }
    }
    return true;
}

/* Binary BVH, two levels at a time: a 128-byte record per binary inner node of every other level, holding the boxes of its (up
 * to four) grandchildren -- a child that is a leaf takes one slot itself:
 *   rows 0..5 = minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4], row 6 = four child words (record index, CRT_LEAF_BIT | object,
 *   or 0xFFFFFFFF with an inverted box for an unused slot), row 7 unused.  Record 0 is the root's.
 * The intermediate child's own box is then never tested; that changes nothing for a ray without NaN in its slab arithmetic,
 * because a box that contains another passes the reference's test whenever the inner one does (subtraction and multiplication
 * by one factor are monotone), and the containment is checked here.  Half the dependent steps per ray.  False: the tree does
 * not qualify (a root that is a leaf, a child box sticking out of its parent's, a tree too deep for the walk's stack). */
bool bvh2_collapse(const crt_scene_desc* desc, int depth, std::vector<float4>& b2w)
{
    b2w.clear();
    if (desc->num_nodes2 < 1 || desc->nodes2[0].left < 0 || 3 * ((depth + 1) / 2) + 1 > CRT_STACK_MAX) return false;
    std::vector<int> rec_of((size_t)desc->num_nodes2, -1), rec_nodes;
    rec_of[0] = 0; rec_nodes.push_back(0);
    auto inside = [&](int c, int p) -> bool {
        const crt_bvh2_node& a = desc->nodes2[c]; const crt_bvh2_node& b = desc->nodes2[p];
        for (int k = 0; k < 3; k++) if (!(a.bmin[k] >= b.bmin[k] && a.bmax[k] <= b.bmax[k])) return false;
        return true;
    };
    for (size_t r = 0; r < rec_nodes.size(); r++) {
        const crt_bvh2_node& nd = desc->nodes2[rec_nodes[r]];
        int slots[4], ns = 0;
        const int kids[2] = { nd.left, nd.right };
        for (int k = 0; k < 2; k++) {
            const crt_bvh2_node& c = desc->nodes2[kids[k]];
            if (c.left < 0) slots[ns++] = kids[k];
            else {
                if (!inside(c.left, kids[k]) || !inside(c.right, kids[k])) { b2w.clear(); return false; }
                slots[ns++] = c.left; slots[ns++] = c.right;
            }
        }
        b2w.resize((r + 1) * 8);
        float bx[6][4]; unsigned wd[4];
        for (int c = 0; c < 4; c++) {
            for (int k = 0; k < 3; k++) { bx[k][c] = CRT_TMAX; bx[3 + k][c] = -CRT_TMAX; }     /* empty slot: never hit */
            wd[c] = 0xFFFFFFFFu;
            if (c >= ns) continue;
            const crt_bvh2_node& sn = desc->nodes2[slots[c]];
            for (int k = 0; k < 3; k++) { bx[k][c] = sn.bmin[k]; bx[3 + k][c] = sn.bmax[k]; }
            if (sn.left < 0) wd[c] = CRT_LEAF_BIT | (unsigned)sn.right;
            else {
                if (rec_of[slots[c]] < 0) { rec_of[slots[c]] = (int)rec_nodes.size(); rec_nodes.push_back(slots[c]); }
                wd[c] = (unsigned)rec_of[slots[c]];
            }
        }
        float4* q = &b2w[r * 8];
        for (int k = 0; k < 6; k++) q[k] = make_float4(bx[k][0], bx[k][1], bx[k][2], bx[k][3]);
        float wf[4]; memcpy(wf, wd, 16);
        q[6] = make_float4(wf[0], wf[1], wf[2], wf[3]);
        q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    return true;
}

/* The collapsed records once more in the QBVH's 64-byte conservative layout (see the cnodes above crt_scene_create's copy of them):
 * per record a frame (origin, scale per axis) and per slot and plane one byte, rounded OUTWARD and checked in double; an empty
 * slot keeps an inverted box.  False (and nothing made) when a box is not finite. */
bool bvh2_compress(const std::vector<float4>& b2w, std::vector<uint32_t>& out)
{
    out.clear();
    const size_t nrec = b2w.size() / 8;
    if (nrec == 0 || nrec >= ((size_t)1 << 26)) return false;       /* record index << 6 is a 32-bit byte offset */
    out.assign(nrec * 16, 0u);
    for (size_t r = 0; r < nrec; r++) {
        const float4* q = &b2w[r * 8];
        float bx[6][4];
        for (int k = 0; k < 6; k++) { bx[k][0] = q[k].x; bx[k][1] = q[k].y; bx[k][2] = q[k].z; bx[k][3] = q[k].w; }
        unsigned wd[4]; { float wf[4] = { q[6].x, q[6].y, q[6].z, q[6].w }; memcpy(wd, wf, 16); }
        float lo[3] = { CRT_TMAX, CRT_TMAX, CRT_TMAX }, hi[3] = { -CRT_TMAX, -CRT_TMAX, -CRT_TMAX };
        bool valid[4], any = false;
        for (int c = 0; c < 4; c++) {
            valid[c] = wd[c] != 0xFFFFFFFFu;
            if (!valid[c]) continue;
            for (int k = 0; k < 3; k++) {
                const float mn = bx[k][c], mx = bx[3 + k][c];
                if (!(fabsf(mn) < 1.0e30f) || !(fabsf(mx) < 1.0e30f)) { out.clear(); return false; }
                if (mn > mx) valid[c] = false;                       /* an inverted box is never hit: keep it that way */
            }
            if (!valid[c]) continue;
            for (int k = 0; k < 3; k++) { if (bx[k][c] < lo[k]) lo[k] = bx[k][c]; if (bx[3 + k][c] > hi[k]) hi[k] = bx[3 + k][c]; }
            any = true;
        }
        if (!any) { lo[0] = lo[1] = lo[2] = 0.0f; hi[0] = hi[1] = hi[2] = 0.0f; }
        float scale[3];
        for (int k = 0; k < 3; k++) {
            float sc = (float)(((double)hi[k] - (double)lo[k]) / 255.0);
            while ((double)lo[k] + 255.0 * (double)sc < (double)hi[k]) sc = nextafterf(sc, CRT_TMAX);
            scale[k] = sc;
        }
        uint32_t* o = &out[r * 16];
        memcpy(&o[0], &lo[0], 4); memcpy(&o[1], &lo[1], 4); memcpy(&o[2], &lo[2], 4);
        memcpy(&o[3], &scale[0], 4); memcpy(&o[4], &scale[1], 4); memcpy(&o[5], &scale[2], 4);
        for (int k = 0; k < 3; k++) {
            uint32_t lo_word = 0, hi_word = 0;
            for (int c = 0; c < 4; c++) {
                int qlo = 255, qhi = 0;                              /* empty slot: an inverted compressed box */
                if (valid[c]) {
                    const double mn = bx[k][c], mx = bx[3 + k][c], og = lo[k], sc = scale[k];
                    if (sc > 0.0) {
                        qlo = (int)floor((mn - og) / sc); if (qlo > 255) qlo = 255; if (qlo < 0) qlo = 0;
                        while (qlo > 0 && og + qlo * sc > mn) qlo--;
                        qhi = (int)ceil((mx - og) / sc); if (qhi < 0) qhi = 0; if (qhi > 255) qhi = 255;
                        while (qhi < 255 && og + qhi * sc < mx) qhi++;
                        if (og + qlo * sc > mn || og + qhi * sc < mx) { out.clear(); return false; }
                    } else { qlo = 0; qhi = 0; }
                }
                lo_word |= (uint32_t)qlo << (8 * c);
                hi_word |= (uint32_t)qhi << (8 * c);
            }
            o[6 + k] = lo_word; o[9 + k] = hi_word;
        }
        for (int c = 0; c < 4; c++) o[12 + c] = wd[c];
    }
    return true;
}

int validate_desc(const crt_scene_desc* d, int* d4, int* d2)
{
    if (!d || d->num_obj < 0 || d->num_non_grid_obj < 0 || d->num_non_grid_obj > d->num_obj) return CRT_E_INVALID;
    if (d->accel < CRT_ACCEL_NONE || d->accel > CRT_ACCEL_BVH4) return CRT_E_INVALID;
    if (d->num_obj >= (1 << 28)) return CRT_E_TOOBIG;
    if (d->num_obj > 0 && (!d->obj_type || !d->obj_ref || !d->obj_id)) return CRT_E_INVALID;
    {   /* obj_id is what crt_hit.obj reports and what callers index their per-object tables with: a permutation of 0..num_obj-1 */
        std::vector<bool> seen((size_t)d->num_obj, false);
        for (int i = 0; i < d->num_obj; i++) {
            const int id = d->obj_id[i];
            if (id < 0 || id >= d->num_obj || seen[(size_t)id]) return CRT_E_INVALID;
            seen[(size_t)id] = true;
        }
    }
    for (int i = 0; i < d->num_obj; i++) {
        int t = d->obj_type[i], r = d->obj_ref[i];
        if (t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE) {
            if (r < 0 || r >= d->num_tris || !d->tri_verts) return CRT_E_INVALID;
        } else {
            if (t < 0 || t > CRT_COMPOUND || r < 0 || r >= d->num_shapes || !d->shapes || d->shapes[r].type != t) return CRT_E_INVALID;
        }
    }
    /* shape nesting the kernels handle: simple | compound{simple} | instanced{simple | compound{simple}} */
    for (int i = 0; i < d->num_shapes; i++) {
        const crt_shape& sh = d->shapes[i];
        if (sh.type == CRT_INSTANCED) {
            if (sh.inner < 0 || sh.inner >= d->num_shapes) return CRT_E_INVALID;
            if (d->shapes[sh.inner].type == CRT_INSTANCED) return CRT_E_UNSUPPORTED;
        } else if (sh.type == CRT_COMPOUND) {
            if (sh.first_part < 0 || sh.num_parts < 0 || sh.first_part + sh.num_parts > d->num_shapes) return CRT_E_INVALID;
            for (int k = 0; k < sh.num_parts; k++) {
                int pt = d->shapes[sh.first_part + k].type;
                if (pt == CRT_INSTANCED || pt == CRT_COMPOUND) return CRT_E_UNSUPPORTED;
            }
        } else if (sh.type < 0 || sh.type > CRT_COMPOUND || sh.type == CRT_TRIANGLE) return CRT_E_INVALID;
    }
    int na = d->num_obj - d->num_non_grid_obj;
    *d4 = 0; *d2 = 0;
    if (d->accel == CRT_ACCEL_BVH4 && na > 0) {
        if (!d->nodes4 || d->num_nodes4 < 1) return CRT_E_INVALID;
        if (d->num_nodes4 > (int)CRT_NODE_MASK) return CRT_E_TOOBIG;       /* 25-bit node index, 32-bit byte offsets */
        for (int i = 0; i < d->num_nodes4; i++) for (int c = 0; c < 4; c++) {
            const crt_bvh4_node& n = d->nodes4[i];
            if (n.type[c] == -1) { if (n.child[c] <= i || n.child[c] >= d->num_nodes4) return CRT_E_INVALID; }
            else if (n.type[c] < 0 || n.type[c] > 4 || n.child[c] < 0 || n.child[c] + n.type[c] > d->num_obj) return CRT_E_INVALID;
        }
        *d4 = depth4(d);
        if (CRT_STACK_PER_LEVEL_PRUNED * *d4 + 1 > CRT_STACK_MAX) return CRT_E_TOOBIG;   /* per-lane stack: <= 3 pending siblings per level */
    }
    if (d->accel == CRT_ACCEL_BVH && na > 0) {
        if (!d->nodes2 || d->num_nodes2 < 1) return CRT_E_INVALID;
        for (int i = 0; i < d->num_nodes2; i++) {
            const crt_bvh2_node& n = d->nodes2[i];
            if (n.left < 0) { if (n.right < 0 || n.right >= d->num_obj) return CRT_E_INVALID; }
            else if (n.left <= i || n.left >= d->num_nodes2 || n.right <= i || n.right >= d->num_nodes2) return CRT_E_INVALID;
        }
        *d2 = depth2(d);
        if (*d2 + 1 > CRT_STACK_MAX) return CRT_E_TOOBIG;
    }
    if (d->accel == CRT_ACCEL_GRID) {
        const crt_grid& g = d->grid;
        if (g.nx < 1 || g.ny < 1 || g.nz < 1 || (long long)g.nx * g.ny * g.nz != g.num_cells) return CRT_E_INVALID;
        if (!d->cell_start) return CRT_E_INVALID;
        if (d->cell_start[0] != 0) return CRT_E_INVALID;
        for (int c = 0; c < g.num_cells; c++) if (d->cell_start[c+1] < d->cell_start[c]) return CRT_E_INVALID;
        int total = d->cell_start[g.num_cells];
        if (total > 0 && !d->cell_items) return CRT_E_INVALID;
        for (int k = 0; k < total; k++) if (d->cell_items[k] < 0 || d->cell_items[k] >= d->num_obj) return CRT_E_INVALID;
    }
    return CRT_OK;
}

template <class T>
int upload(void** dst, const std::vector<T>& v)
{
    *dst = 0;
    size_t bytes = v.size() * sizeof(T);
    if (bytes == 0) bytes = 256;                                   /* keep pointers non-null */
    if (!hip_ok(hipMalloc(dst, bytes), "hipMalloc")) return CRT_E_NOMEM;
    if (!v.empty() && !hip_ok(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D")) return CRT_E_HIP;
    return CRT_OK;
}

void free_scene(crt_scene* s)
{
    if (!s) return;
    void* ptrs[] = { s->d_nodes4, s->d_cnodes4, s->d_wnodes8, s->d_objs, s->d_shapes, s->d_nodes2, s->d_cell_start, s->d_cell_items, s->d_cell_dist, s->d_cell_objs,
                     s->d_bnodes2, s->d_bnodes2w, s->d_bnodes2wc, s->d_counters };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int k = 0; k < CRT_SLOTS; k++) {
        void* q[] = { s->own[k].queue_head, s->own[k].ovf_word, s->own[k].ovf_key, s->own[k].bin_hist, s->own[k].bin_cursor, s->own[k].perm, s->own[k].cells, s->own[k].fb_list };
        for (void* p : q) if (p) (void)hipFree(p);
    }
    for (int k = 0; k < 2; k++) {
        void* q[] = { s->d_chunk_in[k], s->d_chunk_aux[k], s->d_chunk_out[k], s->d_chunk_out2[k] };
        for (void* p : q) if (p) (void)hipFree(p);
        if (s->chunk_streams_valid) (void)hipStreamDestroy(s->chunk_stream[k]);
    }
    for (int k = 0; k < s->ring_valid; k++) {
        (void)hipEventDestroy(s->ring[k].ev_bin); (void)hipEventDestroy(s->ring[k].ev_start); (void)hipEventDestroy(s->ring[k].ev_stop);
    }
    delete s;
}

/* ---- file-scope scratch of the allocation-free entry point ----------------------------
 * crt_intersect_batch_dev_noalloc may not allocate, so the device memory its kernels need beyond the
 * caller's rays and hits lives in one file-scope device array.  Every GPU the library runs on holds its
 * own copy at its own address until the process ends; the array's NAME is a host-side stand-in, not a
 * device address, so host code only ever uses the pointer hipGetSymbolAddress returned for the device
 * (resolved once per device in crt_scene_create, with that device current).
 *   [0, 256)                       queue head
 *   [256, +CRT_BIN_CELLS * 4)      binning histogram     (zeroed with the queue head before every launch)
 *   [.., +CRT_BIN_CELLS * 4)       binning cursors
 *   [.., +CRT_NA_MAX_RAYS * 4)     ray order (perm)
 *   [.., +2 x CRT_NA_THREADS * CRT_OVF_DEPTH * 4)   per-lane stack overflow (words, keys)
 * = 64 MiB + 33 024 B. */
#define CRT_NA_MAX_RAYS  CRT_NOALLOC_MAX_RAYS         /* include/crt_gpu.h: 2^22 */
#define CRT_NA_THREADS   (512u * 256u)                    /* 512 resident 256-thread blocks: 2 per CU on 256 CUs */
#define CRT_NA_OFF_HIST  256u
#define CRT_NA_OFF_CURS  (CRT_NA_OFF_HIST + CRT_BIN_CELLS * 4u)
#define CRT_NA_OFF_PERM  (CRT_NA_OFF_CURS + CRT_BIN_CELLS * 4u)
#define CRT_NA_OFF_OVFW  (CRT_NA_OFF_PERM + CRT_NA_MAX_RAYS * 4u)
#define CRT_NA_OFF_OVFK  (CRT_NA_OFF_OVFW + CRT_NA_THREADS * CRT_OVF_DEPTH * 4u)
#define CRT_NA_BYTES     (CRT_NA_OFF_OVFK + CRT_NA_THREADS * CRT_OVF_DEPTH * 4u)
static_assert(CRT_NA_BYTES == CRT_NOALLOC_SCRATCH_BYTES, "include/crt_gpu.h states the size of the scratch array");
static __device__ __align__(256) unsigned char scratch[CRT_NA_BYTES];

#define CRT_MAX_DEVICES 64
std::mutex g_na_mu;
unsigned char* g_na_ptr[CRT_MAX_DEVICES];                  /* per device: address of ITS copy of `scratch` */
/* Every no-allocation launch on a GPU shares that GPU's array, whichever scene it comes through, so such launches are ordered
 * per DEVICE: each waits (on the device) for the event the one before it recorded.  One library-owned event per device, created
 * with the address above at scene creation -- the entry point itself still creates nothing.  g_na_mu is held from the wait to
 * the record, so two threads launching through two scenes cannot interleave their enqueues. */
hipEvent_t g_na_ev[CRT_MAX_DEVICES];
bool g_na_ev_recorded[CRT_MAX_DEVICES];
std::atomic<unsigned long long> g_scene_uid(0);

/* call with `device` current */
int resolve_noalloc_scratch(int device, Scratch* out)
{
    if (device < 0 || device >= CRT_MAX_DEVICES) return CRT_E_TOOBIG;
    std::lock_guard<std::mutex> lock(g_na_mu);
    if (!g_na_ptr[device]) {
        void* ptr = 0;
        if (!hip_ok(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)") || !ptr) return CRT_E_HIP;
        hipEvent_t ev;
        if (!hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) return CRT_E_HIP;
        g_na_ev[device] = ev;
        g_na_ev_recorded[device] = false;
        g_na_ptr[device] = (unsigned char*)ptr;
    }
    unsigned char* base = g_na_ptr[device];
    out->queue_head = (unsigned*)base;
    out->bin_hist = (unsigned*)(base + CRT_NA_OFF_HIST);
    out->bin_cursor = (unsigned*)(base + CRT_NA_OFF_CURS);
    out->perm = (unsigned*)(base + CRT_NA_OFF_PERM);
    out->perm_cap = CRT_NA_MAX_RAYS;
    out->cells = 0;                                    /* the scatter pass reads the ray records again */
    out->fb_list = 0;                                  /* these launches take the 4-wide walk */
    out->ovf_word = (unsigned*)(base + CRT_NA_OFF_OVFW);
    out->ovf_key = (float*)(base + CRT_NA_OFF_OVFK);
    out->ovf_threads = CRT_NA_THREADS;
    return CRT_OK;
}

/* the calling thread's most recent launch: which scene, which launch number */
thread_local unsigned long long tl_scene_uid = 0, tl_seq = 0;

/* ---- launch ------------------------------------------------------------------------- */

const unsigned CRT_BIN_MIN_RAYS = 1u << 18;
#define CRT_QUEUE_RANGE_WORD 32                        /* words 32..35 of the 256-byte queue-head line: a slice launch's range[] (BinSlice) */
const size_t CRT_PERM_PRESIZE = (size_t)1 << 24;      /* 64 MB: ray order for batches up to 2^24 rays, allocated with the scene */
/* The walks saturate at 4 resident blocks (waves per SIMD) per CU: 691 / 936 / 1 090 / 1 092 Mrays/s at 2 / 3 / 4 / 5
 * (closest hit, courtyard-10M); the any-hit walk, whose 16 KB stack would allow 7, runs 1 904 at 5 against 1 869 at 7. */
#ifndef CRT_MAX_BLOCKS
#define CRT_MAX_BLOCKS 5
#endif
const int CRT_MAX_BLOCKS_PER_CU = CRT_MAX_BLOCKS;
#ifndef CRT_GRID_BLOCKS
#define CRT_GRID_BLOCKS 8          /* the grid walk has no LDS stack and waits on dependent fetches: all the waves its registers allow */
#endif

/* How a launch is made.  noalloc: the launch uses the scene's `na` scratch (file-scope array) and never allocates or frees;
 * otherwise the scene's own scratch, whose ray-order buffer grows when a batch is larger than any before
 * (crt_scene_reserve_rays sizes it ahead).  slice_count > 0: only slice `slice` of the batch is walked (see BinSlice). */
typedef void (*WalkKernel)(DevScene, LaunchArgs);
struct LaunchOpts {
    bool use_binning; bool noalloc; int max_blocks; unsigned slice, slice_count;
    bool no_bin;              /* this launch is not binned whatever the scene's setting (the renderer's queues are in path order) */
    WalkKernel followup;      /* k_bvh8_closest_tri launches: the 4-wide walk that takes the rays the 8-wide one leaves (crt_wide8.h), or null */
    LaunchOpts(bool bin = false, bool na = false, int mb = CRT_MAX_BLOCKS_PER_CU) : use_binning(bin), noalloc(na), max_blocks(mb), slice(0), slice_count(0), no_bin(false), followup(0) {}
};
#define CRT_FB_COUNT_WORD 8        /* words of the 256-byte queue-head line (zeroed before every launch): the fallback list's length, */
#define CRT_FB_HEAD_WORD  16       /* ... and the queue head of the launch that walks it */

int grow_perm(Scratch& sx, size_t n, hipStream_t stream)
{
    if (n <= sx.perm_cap) return CRT_OK;
    HIPCHK(hipStreamSynchronize(stream));     /* grows only; the previous launch has been waited for */
    const bool had_fb = sx.fb_list != 0;
    if (sx.perm) { (void)hipFree(sx.perm); sx.perm = 0; sx.perm_cap = 0; }
    if (sx.cells) { (void)hipFree(sx.cells); sx.cells = 0; }
    if (sx.fb_list) { (void)hipFree(sx.fb_list); sx.fb_list = 0; }
    if (!hip_ok(hipMalloc((void**)&sx.perm, n * sizeof(unsigned)), "hipMalloc")) return CRT_E_NOMEM;
    if (!hip_ok(hipMalloc((void**)&sx.cells, n * sizeof(unsigned short)), "hipMalloc")) { (void)hipFree(sx.perm); sx.perm = 0; return CRT_E_NOMEM; }
    if (had_fb && !hip_ok(hipMalloc((void**)&sx.fb_list, n * sizeof(unsigned)), "hipMalloc")) {
        (void)hipFree(sx.perm); sx.perm = 0; (void)hipFree(sx.cells); sx.cells = 0; return CRT_E_NOMEM;
    }
    sx.perm_cap = n;
    return CRT_OK;
}

template <class K>
int launch_locked(crt_scene* s, K kernel, LaunchArgs la, hipStream_t stream, const LaunchOpts& o)
{
    const bool noalloc = o.noalloc;
    const int slot = (int)(s->own_launches % CRT_SLOTS);
    Scratch& sx = noalloc ? s->na : s->own[slot];
    int per_cu = 0;
    for (int k = 0; k < s->n_occ; k++) if (s->occ[k].fn == (const void*)kernel) per_cu = s->occ[k].per_cu;
    if (per_cu == 0) {                                  /* once per kernel and scene, not once per launch */
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, CRT_BLOCK, 0));
        if (per_cu < 1) per_cu = 1;
        if (s->n_occ < CRT_OCC_CACHE) { s->occ[s->n_occ].fn = (const void*)kernel; s->occ[s->n_occ].per_cu = per_cu; s->n_occ++; }
    }
    if (per_cu > o.max_blocks * (256 / CRT_BLOCK)) per_cu = o.max_blocks * (256 / CRT_BLOCK);      /* max_blocks counts 256-thread blocks */
    unsigned blocks = (unsigned)(s->num_cus * per_cu);
    const bool bin = o.use_binning && !o.no_bin && s->binning && s->bin_grid_valid && la.n >= CRT_BIN_MIN_RAYS;
    /* a slice of a batch that is not binned is a slice by ray index */
    unsigned walk_n = la.n;
    la.q_first = 0; la.q_range = 0;
    if (o.slice_count > 0 && !bin) {
        la.q_first = (unsigned)((unsigned long long)la.n * o.slice / o.slice_count);
        la.n = (unsigned)((unsigned long long)la.n * (o.slice + 1u) / o.slice_count);
        walk_n = la.n - la.q_first;
    } else if (o.slice_count > 0) walk_n = la.n / o.slice_count + 1u;                 /* about that many; the kernels read the range */
    unsigned need = (walk_n + CRT_BLOCK - 1) / CRT_BLOCK;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    if (blocks * CRT_BLOCK > sx.ovf_threads) blocks = sx.ovf_threads / CRT_BLOCK;   /* overflow area is sized for this many */
    la.queue_head = sx.queue_head;
    la.ovf_word = sx.ovf_word;
    la.ovf_key = sx.ovf_key;
    la.grid_threads = sx.ovf_threads;
    LaunchTiming& lt = s->ring[s->seq % CRT_TIMING_RING];
    /* the launch that used this scratch before (CRT_SLOTS launches ago: its ring entry is still its own) */
    if (!noalloc && s->slot_last[slot] > 0) HIPCHK(hipStreamWaitEvent(stream, s->ring[(s->slot_last[slot] - 1) % CRT_TIMING_RING].ev_stop, 0));
    if (noalloc && g_na_ev_recorded[s->device]) HIPCHK(hipStreamWaitEvent(stream, g_na_ev[s->device], 0));    /* see g_na_ev */
    /* what must start at zero: the queue head, and (the binning kernels leave it zeroed, but a launch that
     * failed half way may not have) the histogram behind it in the file-scope array */
    if (noalloc) HIPCHK(hipMemsetAsync(sx.queue_head, 0, CRT_NA_OFF_CURS, stream));
    else HIPCHK(hipMemsetAsync(sx.queue_head, 0, 256, stream));
    la.perm = 0;
    WalkKernel followup = o.followup;
    if (followup && (noalloc || !sx.fb_list)) return CRT_E_INVALID;      /* dispatch_closest does not ask for it then */
    if (followup && la.n > sx.perm_cap) {                                 /* the fallback list is sized with the ray order */
        const int rc = grow_perm(sx, la.n, stream);
        if (rc != CRT_OK) return rc;
    }
    la.fb_count = sx.queue_head + CRT_FB_COUNT_WORD;
    la.fb_list = sx.fb_list;
    if (bin) {
        if (la.n > sx.perm_cap) {
            if (noalloc) return CRT_E_TOOBIG;
            const int rc = grow_perm(sx, la.n, stream);
            if (rc != CRT_OK) return rc;
        }
        HIPCHK(hipEventRecord(lt.ev_bin, stream));
        const unsigned bb = (la.n + CRT_BIN_CHUNK - 1) / CRT_BIN_CHUNK;
        BinSlice sl; sl.k = o.slice; sl.count = o.slice_count;
        unsigned* range = o.slice_count > 0 ? sx.queue_head + CRT_QUEUE_RANGE_WORD : (unsigned*)0;
        hipLaunchKernelGGL(k_bin_hist, dim3(bb), dim3(CRT_BIN_BLOCK), 0, stream, la.rays, la.n, la.n_dev, s->bin_grid, sx.bin_hist, sx.cells);
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(CRT_BIN_BLOCK), 0, stream, sx.bin_hist, sx.bin_cursor, sl, range);
        hipLaunchKernelGGL(k_bin_scatter, dim3(bb), dim3(CRT_BIN_BLOCK), 0, stream, la.rays, la.n, la.n_dev, s->bin_grid, sx.bin_cursor, sx.perm,
                           (const unsigned*)range, (const unsigned short*)sx.cells);
        HIPCHK(hipGetLastError());
        la.perm = sx.perm;
        la.q_range = range;
    }
    HIPCHK(hipEventRecord(lt.ev_start, stream));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(CRT_BLOCK), 0, stream, s->dev, la);
    HIPCHK(hipGetLastError());
    if (followup) {
        /* the rays the 8-wide walk put on its list (exact-t ties, rays with NaN-capable slab arithmetic, stack overflows: a
         * handful per 2^24), through the 4-wide walk in the reference's order: a counted launch whose ray order is the list */
        LaunchArgs fa = la;
        fa.perm = sx.fb_list; fa.n_dev = la.fb_count; fa.q_range = 0; fa.q_first = 0;
        fa.queue_head = sx.queue_head + CRT_FB_HEAD_WORD;
        unsigned fblocks = (unsigned)s->num_cus * 2u;
        if (fblocks > blocks) fblocks = blocks;
        hipLaunchKernelGGL(followup, dim3(fblocks), dim3(CRT_BLOCK), 0, stream, s->dev, fa);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(lt.ev_stop, stream));
    if (noalloc) { HIPCHK(hipEventRecord(g_na_ev[s->device], stream)); g_na_ev_recorded[s->device] = true; }
    lt.binned = bin;
    lt.seq = ++s->seq;
    if (!noalloc) { s->slot_last[slot] = lt.seq; s->own_launches++; }
    tl_scene_uid = s->uid; tl_seq = lt.seq;
    return CRT_OK;
}

template <class K>
int launch(crt_scene* s, K kernel, const LaunchArgs& la, hipStream_t stream, const LaunchOpts& o)
{
    std::lock_guard<std::mutex> lock(s->mu);
    if (!o.noalloc) return launch_locked(s, kernel, la, stream, o);
    std::lock_guard<std::mutex> lock_dev(g_na_mu);            /* lock order: scene, then device */
    return launch_locked(s, kernel, la, stream, o);
}

template <bool TRI_ONLY, bool COUNT>
int dispatch_closest(crt_scene* s, const LaunchArgs& la, hipStream_t st, LaunchOpts o = LaunchOpts())
{
    const bool exact = resolved_mode(s) == CRT_TRAVERSE_EXACT;
    const bool na = o.noalloc;
    o.use_binning = true; o.max_blocks = CRT_MAX_BLOCKS_PER_CU;
    switch (s->dev.accel) {
    case CRT_ACCEL_BVH4: if (TRI_ONLY && !exact && s->compress && s->cnodes_valid && s->dev.num_obj > 0) {
                             /* small batches: the walk whose idle lanes take pending subtrees from busy ones once the queue is dry
                              * (crt_kernels.h, "tail splitting").  For a counted launch n is an upper bound, for a slice the share is
                              * about n / slices.  Not for the no-allocation entry point (its overflow area is sized for the 14-entry
                              * LDS stack), and not when objects wait outside the tree: a split ray is settled without the non-grid
                              * loop of intersect.h:403-411. */
                             const unsigned share = o.slice_count > 0 ? la.n / o.slice_count : la.n;
                             /* the 8-wide copy of the tree (crt_wide8.h), with the 4-wide walk behind it for the rays it leaves */
                             if (!na && s->wide8 && s->wide8_valid && share > s->wide8_min_rays) {
                                 o.followup = k_bvh4_closest_tri<COUNT>;
                                 return launch(s, k_bvh8_closest_tri<COUNT>, la, st, o);
                             }
                             if (!na && s->dev.nng == 0 && share <= s->split_max_rays) return launch(s, k_bvh4_closest_tri<COUNT, true>, la, st, o);
                             return launch(s, k_bvh4_closest_tri<COUNT>, la, st, o);
                         }
                         return exact ? launch(s, k_bvh4_closest<TRI_ONLY, COUNT, true>, la, st, o)
                                      : launch(s, k_bvh4_closest<TRI_ONLY, COUNT, false>, la, st, o);
    case CRT_ACCEL_BVH:  /* the conservative records: triangle-only scenes, as for the QBVH -- a triangle's box is padded (K_FLAT_AABB) far beyond any
                          * rounding, so a ray that hits it passes the reference's own box tests; a flat RECTANGLE's box is not, the reference
                          * misses such hits, and a wider box would find them */
                         if (TRI_ONLY && !exact && s->bvh2_wide && s->bvh2_wcomp && s->compress) return launch(s, k_bvh2<TRI_ONLY, COUNT, false, false, true, true, true>, la, st, o);
                         if (!exact && s->bvh2_wide) return launch(s, k_bvh2<TRI_ONLY, COUNT, false, false, true, true>, la, st, o);
                         if (!exact && s->bvh2_ordered) return launch(s, k_bvh2<TRI_ONLY, COUNT, false, false, true>, la, st, o);
                         return exact ? launch(s, k_bvh2<TRI_ONLY, COUNT, false, true>, la, st, o)
                                      : launch(s, k_bvh2<TRI_ONLY, COUNT, false, false>, la, st, o);
    case CRT_ACCEL_GRID: /* lane refill pays for incoherent closest-hit rays through triangles (735 -> 1 229 Mrays/s); the any-hit walk
                            (694 plain, 637 with refill) and the all-shapes instance (C1: 7.1 against 8.7 ms) keep the plain loop */
                         o.max_blocks = CRT_GRID_BLOCKS;
                         if constexpr (TRI_ONLY) return launch(s, k_grid_refill<1, COUNT, false>, la, st, o);
                         else {
                             /* triangles and rectangles only (config C1, the cornell box): an instance of the plain walk that never
                              * references the analytic shapes -- 56-60 VGPRs and no scratch against 137 + 256 bytes of spills, seven
                              * waves per SIMD against three.  C1 6.46 -> 5.90 ms; the lane-refill form measured 6.21 (a scene of 29
                              * objects in 1 000 cells is all arithmetic, and coherent rays keep a wave's lanes in step) */
                             if (s->tri_rect_only) return launch(s, k_grid<CRT_KIND_TRI_RECT, COUNT, false>, la, st, o);
                             return launch(s, k_grid<0, COUNT, false>, la, st, o);
                         }
    default:             o.use_binning = false;
                         return launch(s, k_none_closest<TRI_ONLY, COUNT>, la, st, o);
    }
}
template <bool TRI_ONLY, bool COUNT>
int dispatch_shadow(crt_scene* s, const LaunchArgs& la, hipStream_t st, unsigned slice = 0, unsigned slice_count = 0, bool no_bin = false)
{
    const bool exact = resolved_mode(s) == CRT_TRAVERSE_EXACT;
    LaunchOpts o(true);
    o.slice = slice; o.slice_count = slice_count; o.no_bin = no_bin;
    switch (s->dev.accel) {
    case CRT_ACCEL_BVH4: if (TRI_ONLY && s->compress && s->cnodes_valid) return launch(s, k_bvh4_shadow<TRI_ONLY, COUNT, true>, la, st, o);
                         return launch(s, k_bvh4_shadow<TRI_ONLY, COUNT>, la, st, o);
    case CRT_ACCEL_BVH:  if (TRI_ONLY && !exact && s->bvh2_wide && s->bvh2_wcomp && s->compress) return launch(s, k_bvh2<TRI_ONLY, COUNT, true, false, true, true, true>, la, st, o);
                         if (!exact && s->bvh2_wide) return launch(s, k_bvh2<TRI_ONLY, COUNT, true, false, true, true>, la, st, o);
                         if (!exact && s->bvh2_ordered) return launch(s, k_bvh2<TRI_ONLY, COUNT, true, false, true>, la, st, o);
                         return exact ? launch(s, k_bvh2<TRI_ONLY, COUNT, true, true>, la, st, o)
                                      : launch(s, k_bvh2<TRI_ONLY, COUNT, true, false>, la, st, o);
    case CRT_ACCEL_GRID: o.max_blocks = CRT_GRID_BLOCKS;
                         if constexpr (!TRI_ONLY) { if (s->tri_rect_only) return launch(s, k_grid<CRT_KIND_TRI_RECT, COUNT, true>, la, st, o); }
                         return launch(s, k_grid<TRI_ONLY ? 1 : 0, COUNT, true>, la, st, o);
    default:             o.use_binning = false;
                         return launch(s, k_none_shadow<TRI_ONLY, COUNT>, la, st, o);
    }
}

} // namespace

extern "C" {

const char* crt_strerror(int code)
{
    switch (code) {
    case CRT_OK: return "ok";
    case CRT_E_INVALID: return "invalid argument or inconsistent scene description";
    case CRT_E_NOMEM: return "out of host or device memory";
    case CRT_E_HIP: return "HIP runtime call failed (see crt_last_hip_error)";
    case CRT_E_UNSUPPORTED: return "unsupported request";
    case CRT_E_TOOBIG: return "size beyond a documented limit";
    case CRT_E_NODEVICE: return "no usable gfx950 device";
    default: return "unknown error code";
    }
}
const char* crt_last_hip_error(void) { return g_hip_err; }

int crt_device_count(int* count)
{
    if (!count) return CRT_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return CRT_OK;
}

int crt_scene_create(int device, const crt_scene_desc* desc, crt_scene** out)
{
    if (!out) return CRT_E_INVALID;
    *out = 0;
    int d4 = 0, d2 = 0;
    int rc;
    try { rc = validate_desc(desc, &d4, &d2); } catch (const std::bad_alloc&) { return CRT_E_NOMEM; }
    if (rc != CRT_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return CRT_E_NODEVICE;
    if (device < 0 || device >= ndev) return CRT_E_INVALID;
    DeviceGuard guard(device);
    if (!guard.ok) return CRT_E_HIP;

    crt_scene* s = new (std::nothrow) crt_scene();
    if (!s) return CRT_E_NOMEM;
    s->device = device;                         /* value-initialised above: every other field is zero */
    s->mode = CRT_TRAVERSE_AUTO;
    const int n = desc->num_obj, nng = desc->num_non_grid_obj;

    try {
        /* object records (48 B) in final order */
        std::vector<float4> objs((size_t)n * CRT_OBJ_STRIDE);
        bool all_tri = true, accel_tri = true, accel_bounded = true;
        for (int i = 0; i < n; i++) {
            const int t = desc->obj_type[i], r = desc->obj_ref[i];
            const bool tri = (t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE);
            float v[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            if (tri) memcpy(v, desc->tri_verts + (size_t)r * 9, sizeof v);
            else {
                all_tri = false;
                if (i >= nng) {
                    accel_tri = false;
                    if (!(t == CRT_SPHERE || t == CRT_RECTANGLE || t == CRT_DISK)) accel_bounded = false;
                }
            }
            float idf, tf, rf;
            memcpy(&idf, &desc->obj_id[i], 4); memcpy(&tf, &t, 4); memcpy(&rf, &r, 4);
            objs[(size_t)i*CRT_OBJ_STRIDE + 0] = make_float4(v[0], v[1], v[2], v[3]);
            objs[(size_t)i*CRT_OBJ_STRIDE + 1] = make_float4(v[4], v[5], v[6], v[7]);
            objs[(size_t)i*CRT_OBJ_STRIDE + 2] = make_float4(v[8], idf, tf, rf);
        }
        s->tri_only = all_tri;
        {
            bool tr = !all_tri;
            for (int i = 0; i < n && tr; i++) {
                const int t = desc->obj_type[i];
                if (!(t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE || t == CRT_RECTANGLE)) tr = false;
            }
            const char* e = getenv("CRT_GRID_TRI_RECT");                  /* A/B knob: 0 = the all-shapes instance */
            s->tri_rect_only = tr && !(e && atoi(e) == 0);
        }
        s->accel_tri_only = accel_tri;
        s->accel_bounded = accel_bounded;
        s->depth4 = d4; s->depth2 = d2;
        if ((desc->accel == CRT_ACCEL_BVH4 || desc->accel == CRT_ACCEL_BVH) && n - nng > 0) {   /* box of the triangles under the accelerator */
            float lo[3] = { CRT_TMAX, CRT_TMAX, CRT_TMAX }, hi[3] = { -CRT_TMAX, -CRT_TMAX, -CRT_TMAX };
            bool any = false;
            for (int i = nng; i < n; i++) {
                const int t = desc->obj_type[i];
                if (!(t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE)) continue;
                const float* v = desc->tri_verts + (size_t)desc->obj_ref[i] * 9;
                for (int k = 0; k < 9; k++) {
                    const float x = v[k];
                    if (x == x && x > -1.0e30f && x < 1.0e30f) { if (x < lo[k % 3]) lo[k % 3] = x; if (x > hi[k % 3]) hi[k % 3] = x; any = true; }
                }
            }
            if (any) {
                for (int a = 0; a < 3; a++) {
                    const float ext = hi[a] - lo[a];
                    s->bin_grid.lo[a] = lo[a];
                    s->bin_grid.scale[a] = ext > 0.0f ? (float)(1 << CRT_BIN_BITS) / ext : 0.0f;
                }
                s->bin_grid_valid = true;
            }
        }
        if (desc->accel == CRT_ACCEL_GRID && desc->grid.num_cells > 0) {       /* the grid's own box */
            bool ok = true;
            for (int a = 0; a < 3; a++) {
                const float lo = desc->grid.bmin[a], ext = desc->grid.bmax[a] - desc->grid.bmin[a];
                if (!(fabsf(lo) < 1.0e30f) || !(ext < 1.0e30f)) ok = false;
                s->bin_grid.lo[a] = lo;
                s->bin_grid.scale[a] = ext > 0.0f ? (float)(1 << CRT_BIN_BITS) / ext : 0.0f;
            }
            s->bin_grid_valid = ok;
        }
        s->binning = true;
        /* the reference-control-flow QBVH walk adds one frame marker per tree level to the stack */
        if (desc->accel == CRT_ACCEL_BVH4 && !accel_bounded && CRT_STACK_PER_LEVEL_EXACT * d4 + 1 > CRT_STACK_MAX) { free_scene(s); return CRT_E_TOOBIG; }

        std::vector<int> order, newid;
        if (desc->accel == CRT_ACCEL_BVH4) {
            const char* e = getenv("CRT_NODE_ORDER");
            node_layout(desc, e ? atoi(e) : CRT_NODE_ORDER_DEFAULT, order, newid);
            if (order.size() > (size_t)CRT_NODE_MASK) { free_scene(s); return CRT_E_TOOBIG; }
        }
        const int num_slots = (int)order.size();
        auto child_word = [&](const crt_bvh4_node& nd, int c) -> unsigned {      /* an inner word carries the child's own axis2: known at pop, no load */
            return nd.type[c] == -1 ? (((unsigned)(desc->nodes4[nd.child[c]].axis & 0x03) << 28) | (unsigned)newid[nd.child[c]])
                                    : (CRT_LEAF_BIT | ((unsigned)nd.type[c] << 28) | (unsigned)nd.child[c]);
        };
        std::vector<float4> n4;
        if (desc->accel == CRT_ACCEL_BVH4) {
            n4.assign((size_t)num_slots * 8, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            for (int i = 0; i < num_slots; i++) {
                if (order[i] < 0) continue;
                const crt_bvh4_node& nd = desc->nodes4[order[i]];
                float4* q = &n4[(size_t)i * 8];
                for (int k = 0; k < 6; k++) q[k] = make_float4(nd.bbox[4*k], nd.bbox[4*k+1], nd.bbox[4*k+2], nd.bbox[4*k+3]);
                unsigned w[4];
                for (int c = 0; c < 4; c++) w[c] = child_word(nd, c);
                float wf[4]; memcpy(wf, w, 16);
                q[6] = make_float4(wf[0], wf[1], wf[2], wf[3]);
                int ax2 = nd.axis & 0x03;                          /* bvh4.h:308 */
                float af; memcpy(&af, &ax2, 4);
                q[7] = make_float4(af, 0.0f, 0.0f, 0.0f);
            }
        }
        /* 64-byte conservative copy of the nodes (crt_kernels.h "compressed nodes"): per node a frame (origin, scale per axis) and per
         * child and plane one byte, rounded OUTWARD and checked in double, so that every compressed box contains the node's box */
        std::vector<uint32_t> cn4;
        bool cn_ok = desc->accel == CRT_ACCEL_BVH4 && desc->num_nodes4 > 0;
        if (cn_ok) {
            cn4.assign((size_t)num_slots * 16, 0u);
            for (int i = 0; i < num_slots && cn_ok; i++) {
                if (order[i] < 0) continue;
                const crt_bvh4_node& nd = desc->nodes4[order[i]];
                float lo[3] = { CRT_TMAX, CRT_TMAX, CRT_TMAX }, hi[3] = { -CRT_TMAX, -CRT_TMAX, -CRT_TMAX };
                bool valid[4], any = false;
                for (int c = 0; c < 4; c++) {
                    valid[c] = nd.type[c] == -1 || nd.type[c] > 0;
                    if (!valid[c]) continue;
                    for (int k = 0; k < 3; k++) {
                        const float mn = nd.bbox[4 * k + c], mx = nd.bbox[12 + 4 * k + c];
                        if (!(fabsf(mn) < 1.0e30f) || !(fabsf(mx) < 1.0e30f)) cn_ok = false;     /* NaN / inf / huge: no compressed copy for this scene */
                        if (mn > mx) valid[c] = false;                                            /* an inverted box is never hit: keep it that way */
                        if (mn < lo[k]) lo[k] = mn;
                        if (mx > hi[k]) hi[k] = mx;
                    }
                    any = any || valid[c];
                }
                if (!cn_ok) break;
                if (!any) { lo[0] = lo[1] = lo[2] = 0.0f; hi[0] = hi[1] = hi[2] = 0.0f; }
                float scale[3];
                for (int k = 0; k < 3; k++) {
                    if (hi[k] < lo[k]) { lo[k] = 0.0f; hi[k] = 0.0f; }
                    float sc = (float)(((double)hi[k] - (double)lo[k]) / 255.0);
                    while ((double)lo[k] + 255.0 * (double)sc < (double)hi[k]) sc = nextafterf(sc, CRT_TMAX);
                    scale[k] = sc;
                }
                uint32_t* q = &cn4[(size_t)i * 16];
                memcpy(&q[0], &lo[0], 4); memcpy(&q[1], &lo[1], 4); memcpy(&q[2], &lo[2], 4);
                memcpy(&q[3], &scale[0], 4); memcpy(&q[4], &scale[1], 4); memcpy(&q[5], &scale[2], 4);
                for (int k = 0; k < 3; k++) {
                    uint32_t lo_word = 0, hi_word = 0;
                    for (int c = 0; c < 4; c++) {
                        int qlo = 255, qhi = 0;                        /* empty / inverted child: an inverted compressed box */
                        if (valid[c]) {
                            const double mn = nd.bbox[4 * k + c], mx = nd.bbox[12 + 4 * k + c], o = lo[k], sc = scale[k];
                            if (sc > 0.0) {
                                qlo = (int)floor((mn - o) / sc); if (qlo > 255) qlo = 255; if (qlo < 0) qlo = 0;
                                while (qlo > 0 && o + qlo * sc > mn) qlo--;
                                qhi = (int)ceil((mx - o) / sc); if (qhi < 0) qhi = 0; if (qhi > 255) qhi = 255;
                                while (qhi < 255 && o + qhi * sc < mx) qhi++;
                                if (o + qlo * sc > mn || o + qhi * sc < mx) cn_ok = false;      /* cannot happen (scale was rounded up); checked anyway */
                            } else { qlo = 0; qhi = 0; }
                        }
                        lo_word |= (uint32_t)qlo << (8 * c);
                        hi_word |= (uint32_t)qhi << (8 * c);
                    }
                    q[6 + k] = lo_word; q[9 + k] = hi_word;
                }
                for (int c = 0; c < 4; c++) q[12 + c] = child_word(nd, c);
            }
            if (!cn_ok) cn4.clear();
        }
        s->cnodes_valid = cn_ok;
        std::vector<uint32_t> w8;
        {
            /* measured slower than the 4-wide walk ("8-wide nodes"), so the copy is made only on request */
            const char* e = getenv("CRT_WIDE8");
            const char* ew = getenv("CRT_WIDE8_WIDTH");
            s->wide8_valid = cn_ok && accel_tri && n - nng > 0 && e && atoi(e) != 0 && wide8::collapse(desc, w8, ew ? atoi(ew) : 8);
            if (!s->wide8_valid) w8.clear();
            s->wide8 = s->wide8_valid;
            e = getenv("CRT_WIDE8_MIN");
            s->wide8_min_rays = e ? (unsigned)strtoul(e, 0, 10) : 0u;
        }
        { const char* e = getenv("CRT_TAIL_SPLIT_MAX"); s->split_max_rays = e ? (unsigned)strtoul(e, 0, 10) : CRT_SPLIT_MAX_RAYS_DEFAULT; }
        s->compress = true;

        std::vector<float4> n2;
        if (desc->accel == CRT_ACCEL_BVH) {
            n2.resize((size_t)desc->num_nodes2 * 2);
            for (int i = 0; i < desc->num_nodes2; i++) {
                const crt_bvh2_node& nd = desc->nodes2[i];
                float lf, rf; memcpy(&lf, &nd.left, 4); memcpy(&rf, &nd.right, 4);
                n2[(size_t)i*2]     = make_float4(nd.bmin[0], nd.bmin[1], nd.bmin[2], lf);
                n2[(size_t)i*2 + 1] = make_float4(nd.bmax[0], nd.bmax[1], nd.bmax[2], rf);
            }
        }
        /* k_bvh2's layout: inner nodes only, renumbered in index order, each with both child boxes */
        std::vector<float4> b2, b2w;
        std::vector<uint32_t> b2wc;
        if (desc->accel == CRT_ACCEL_BVH && desc->num_nodes2 > 0) {
            std::vector<int> inner_id((size_t)desc->num_nodes2, -1);
            int n_inner = 0;
            for (int i = 0; i < desc->num_nodes2; i++) if (desc->nodes2[i].left >= 0) inner_id[i] = n_inner++;
            auto word = [&](int node) -> unsigned {
                const crt_bvh2_node& c = desc->nodes2[node];
                return c.left < 0 ? (CRT_LEAF_BIT | (unsigned)c.right) : (unsigned)inner_id[node];
            };
            b2.resize((size_t)n_inner * 4);
            for (int i = 0; i < desc->num_nodes2; i++) {
                const crt_bvh2_node& nd = desc->nodes2[i];
                if (nd.left < 0) continue;
                const crt_bvh2_node& l = desc->nodes2[nd.left];
                const crt_bvh2_node& r = desc->nodes2[nd.right];
                const unsigned lw = word(nd.left), rw = word(nd.right);
                float lf, rf; memcpy(&lf, &lw, 4); memcpy(&rf, &rw, 4);
                float4* q = &b2[(size_t)inner_id[i] * 4];
                q[0] = make_float4(l.bmin[0], l.bmin[1], l.bmin[2], l.bmax[0]);
                q[1] = make_float4(l.bmax[1], l.bmax[2], r.bmin[0], r.bmin[1]);
                q[2] = make_float4(r.bmin[2], r.bmax[0], r.bmax[1], r.bmax[2]);
                q[3] = make_float4(lf, rf, 0.0f, 0.0f);
            }
            s->root_word2 = word(0);
            {
                const char* e = getenv("CRT_BVH2_ORDERED");
                s->bvh2_ordered = bvh2_leaves_in_index_order(desc) && !(e && atoi(e) == 0);
                e = getenv("CRT_BVH2_WIDE");
                s->bvh2_wide = s->bvh2_ordered && !(e && atoi(e) == 0) && bvh2_collapse(desc, d2, b2w);
                if (!s->bvh2_wide) b2w.clear();
                e = getenv("CRT_BVH2_WCOMP");
                s->bvh2_wcomp = s->bvh2_wide && !(e && atoi(e) == 0) && bvh2_compress(b2w, b2wc);
                s->root_word2w = 0u;
            }
            for (int a = 0; a < 3; a++) { s->dev.root_bmin[a] = desc->nodes2[0].bmin[a]; s->dev.root_bmax[a] = desc->nodes2[0].bmax[a]; }
        }
        std::vector<crt_shape> shapes;
        if (desc->num_shapes > 0) shapes.assign(desc->shapes, desc->shapes + desc->num_shapes);
        std::vector<int32_t> cs, ci;
        std::vector<unsigned char> cd;
        std::vector<float4> co;
        if (desc->accel == CRT_ACCEL_GRID) {
            cs.assign(desc->cell_start, desc->cell_start + desc->grid.num_cells + 1);
            if (cs.back() > 0) ci.assign(desc->cell_items, desc->cell_items + cs.back());
            grid_distance_field(desc, cd);
            /* the object records once more, in cell order: a cell's objects lie side by side and the walk needs no item index
             * (48 bytes per cell reference; left out above 16 GB, the walk then goes through cell_items) */
            {
                const char* e2 = getenv("CRT_GRID_CELL_RECORDS");
                if (!(e2 && atoi(e2) == 0) && (double)ci.size() * CRT_OBJ_STRIDE * sizeof(float4) <= 16.0e9) {
                    co.resize(ci.size() * CRT_OBJ_STRIDE);
                    for (size_t k = 0; k < ci.size(); k++)
                        for (int q = 0; q < CRT_OBJ_STRIDE; q++) co[k * CRT_OBJ_STRIDE + q] = objs[(size_t)ci[k] * CRT_OBJ_STRIDE + q];
                }
            }
            const char* e = getenv("CRT_GRID_SKIP");                      /* A/B knob: 0 = no cell is skipped, as the reference walks */
            if (e && atoi(e) == 0) for (size_t c = 0; c < cd.size(); c++) if (cd[c] > 1) cd[c] = 1;
        }
        if ((rc = upload(&s->d_objs, objs)) != CRT_OK || (rc = upload(&s->d_nodes4, n4)) != CRT_OK || (rc = upload(&s->d_cnodes4, cn4)) != CRT_OK || (rc = upload(&s->d_wnodes8, w8)) != CRT_OK ||
            (rc = upload(&s->d_nodes2, n2)) != CRT_OK || (rc = upload(&s->d_bnodes2, b2)) != CRT_OK || (rc = upload(&s->d_bnodes2w, b2w)) != CRT_OK || (rc = upload(&s->d_bnodes2wc, b2wc)) != CRT_OK || (rc = upload(&s->d_shapes, shapes)) != CRT_OK ||
            (rc = upload(&s->d_cell_start, cs)) != CRT_OK || (rc = upload(&s->d_cell_items, ci)) != CRT_OK ||
            (rc = upload(&s->d_cell_dist, cd)) != CRT_OK || (!co.empty() && (rc = upload(&s->d_cell_objs, co)) != CRT_OK)) {
            free_scene(s);
            return rc;
        }
    } catch (const std::bad_alloc&) {
        free_scene(s);
        return CRT_E_NOMEM;
    }

    int cus = 0;
    if (!hip_ok(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device), "hipDeviceGetAttribute")) { free_scene(s); return CRT_E_HIP; }
    s->num_cus = cus > 0 ? cus : 1;
    for (int k = 0; k < CRT_SLOTS; k++) {
        Scratch& sx = s->own[k];
        sx.ovf_threads = (unsigned)s->num_cus * 8u * 256u;            /* at most 8 resident 256-thread blocks (2 048 threads) per CU */
        size_t ovf_bytes = (size_t)sx.ovf_threads * (CRT_OVF_DEPTH + 1) * sizeof(unsigned);    /* + 1: the tail-splitting walk keeps one entry fewer in LDS */
        if (!hip_ok(hipMalloc((void**)&sx.ovf_word, ovf_bytes), "hipMalloc") ||
            !hip_ok(hipMalloc((void**)&sx.ovf_key, ovf_bytes), "hipMalloc") ||
            !hip_ok(hipMalloc((void**)&sx.queue_head, 256), "hipMalloc") ||
            !hip_ok(hipMalloc((void**)&sx.bin_hist, CRT_BIN_CELLS * sizeof(unsigned)), "hipMalloc") ||
            !hip_ok(hipMalloc((void**)&sx.bin_cursor, CRT_BIN_CELLS * sizeof(unsigned)), "hipMalloc")) { free_scene(s); return CRT_E_NOMEM; }
        if (s->bin_grid_valid) {
            /* ray order sized ahead for the batches the binning is meant for, so that the launch path neither allocates
             * nor synchronises; a larger batch still grows it (launch(); crt_scene_reserve_rays does it ahead of time) */
            sx.perm_cap = CRT_PERM_PRESIZE;
            if (!hip_ok(hipMalloc((void**)&sx.perm, sx.perm_cap * sizeof(unsigned)), "hipMalloc") ||
                !hip_ok(hipMalloc((void**)&sx.cells, sx.perm_cap * sizeof(unsigned short)), "hipMalloc") ||
                (s->wide8_valid && !hip_ok(hipMalloc((void**)&sx.fb_list, sx.perm_cap * sizeof(unsigned)), "hipMalloc"))) { free_scene(s); return CRT_E_NOMEM; }
        }
        if (!hip_ok(hipMemset(sx.bin_hist, 0, CRT_BIN_CELLS * sizeof(unsigned)), "hipMemset")) { free_scene(s); return CRT_E_HIP; }
    }
    for (int k = 0; k < CRT_SLOTS; k++) if (!s->own[k].fb_list) s->wide8_valid = false;       /* no list, no 8-wide walk */
    if (!hip_ok(hipMalloc((void**)&s->d_counters, 512), "hipMalloc")) { free_scene(s); return CRT_E_NOMEM; }     /* 4 counters + the diagnostic build's */
    if (!hip_ok(hipMemset(s->d_counters, 0, 512), "hipMemset")) { free_scene(s); return CRT_E_HIP; }
    for (int k = 0; k < CRT_TIMING_RING; k++) {
        LaunchTiming& lt = s->ring[k];
        if (!hip_ok(hipEventCreate(&lt.ev_bin), "hipEventCreate")) { free_scene(s); return CRT_E_HIP; }
        if (!hip_ok(hipEventCreate(&lt.ev_start), "hipEventCreate")) { (void)hipEventDestroy(lt.ev_bin); free_scene(s); return CRT_E_HIP; }
        if (!hip_ok(hipEventCreate(&lt.ev_stop), "hipEventCreate")) { (void)hipEventDestroy(lt.ev_bin); (void)hipEventDestroy(lt.ev_start); free_scene(s); return CRT_E_HIP; }
        s->ring_valid = k + 1;
    }
    /* this GPU's copy of the file-scope scratch array (crt_intersect_batch_dev_noalloc), device `device` is current */
    if ((rc = resolve_noalloc_scratch(device, &s->na)) != CRT_OK) { free_scene(s); return rc; }
    s->uid = ++g_scene_uid;

    DevScene& dv = s->dev;
    dv.nodes4 = (const float4*)s->d_nodes4;
    dv.cnodes4 = (const uint4*)s->d_cnodes4;
    dv.wnodes8 = (const uint4*)s->d_wnodes8;
    dv.objs = (const float4*)s->d_objs;
    dv.shapes = (const crt_shape*)s->d_shapes;
    dv.nodes2 = (const float4*)s->d_nodes2;
    dv.cell_start = (const int*)s->d_cell_start;
    dv.cell_dist = (const unsigned char*)s->d_cell_dist;
    dv.cell_objs = (const float4*)s->d_cell_objs;
    dv.cell_items = (const int*)s->d_cell_items;
    dv.counters = s->d_counters;
    dv.grid = desc->grid;
    dv.num_obj = n; dv.nng = nng; dv.n_accel = n - nng; dv.accel = desc->accel;
    dv.root_word = (desc->accel == CRT_ACCEL_BVH4 && desc->num_nodes4 > 0) ? ((unsigned)(desc->nodes4[0].axis & 0x03) << 28) : 0u;
    if (desc->accel == CRT_ACCEL_BVH) dv.root_word = s->root_word2;
    dv.bnodes2w = (const float4*)s->d_bnodes2w;
    dv.bnodes2wc = (const uint4*)s->d_bnodes2wc;
    dv.root_word_w = s->root_word2w;
    dv.bnodes2 = (const float4*)s->d_bnodes2;
    *out = s;
    return CRT_OK;
}

void crt_scene_destroy(crt_scene* s)
{
    if (!s) return;
    DeviceGuard guard(s->device);
    (void)hipDeviceSynchronize();
    free_scene(s);
}

int crt_scene_device(const crt_scene* s, int* device)
{
    if (!s || !device) return CRT_E_INVALID;
    *device = s->device;
    return CRT_OK;
}

int crt_scene_num_objects(const crt_scene* s, int* num_obj)
{
    if (!s || !num_obj) return CRT_E_INVALID;
    *num_obj = s->dev.num_obj;
    return CRT_OK;
}

int crt_scene_set_traverse_mode(crt_scene* s, int mode)
{
    if (!s || mode < CRT_TRAVERSE_AUTO || mode > CRT_TRAVERSE_EXACT) return CRT_E_INVALID;
    if (mode == CRT_TRAVERSE_EXACT && s->dev.accel == CRT_ACCEL_BVH4 && CRT_STACK_PER_LEVEL_EXACT * s->depth4 + 1 > CRT_STACK_MAX)
        return CRT_E_TOOBIG;                             /* one frame marker per level on top of the pending siblings */
    s->mode = mode;
    return CRT_OK;
}
int crt_host_grid_distance_field(const crt_scene_desc* desc, uint8_t* out)
{
    if (!desc || !out || desc->accel != CRT_ACCEL_GRID) return CRT_E_INVALID;
    int d4 = 0, d2 = 0;
    try {
        const int rc = validate_desc(desc, &d4, &d2);
        if (rc != CRT_OK) return rc;
        std::vector<unsigned char> cd;
        grid_distance_field(desc, cd);
        if (!cd.empty()) memcpy(out, cd.data(), cd.size());
    } catch (const std::bad_alloc&) { return CRT_E_NOMEM; }
    return CRT_OK;
}
int crt_host_bvh2_wide_records(const crt_scene_desc* desc, float* records, int32_t* num_records, int32_t* leaves_in_index_order)
{
    if (!desc || !num_records || !leaves_in_index_order || desc->accel != CRT_ACCEL_BVH) return CRT_E_INVALID;
    int d4 = 0, d2 = 0;
    try {
        const int rc = validate_desc(desc, &d4, &d2);
        if (rc != CRT_OK) return rc;
        *leaves_in_index_order = bvh2_leaves_in_index_order(desc) ? 1 : 0;
        std::vector<float4> b2w;
        const bool ok = *leaves_in_index_order && bvh2_collapse(desc, d2, b2w);
        *num_records = ok ? (int32_t)(b2w.size() / 8) : 0;
        if (ok && records) memcpy(records, b2w.data(), b2w.size() * sizeof(float4));
    } catch (const std::bad_alloc&) { return CRT_E_NOMEM; }
    return CRT_OK;
}
int crt_host_qbvh_node_order(const crt_scene_desc* desc, int policy, int32_t* order, int32_t* num_slots)
{
    if (!desc || !order || !num_slots || desc->accel != CRT_ACCEL_BVH4) return CRT_E_INVALID;
    int d4 = 0, d2 = 0;
    try {
        const int rc = validate_desc(desc, &d4, &d2);
        if (rc != CRT_OK) return rc;
        std::vector<int> ord, newid;
        node_layout(desc, policy, ord, newid);
        if (ord.size() > 2 * (size_t)desc->num_nodes4) return CRT_E_TOOBIG;       /* cannot happen: at most one dead record per cluster */
        for (size_t i = 0; i < ord.size(); i++) order[i] = ord[i];
        *num_slots = (int32_t)ord.size();
    } catch (const std::bad_alloc&) { return CRT_E_NOMEM; }
    return CRT_OK;
}
int crt_scene_set_node_compression(crt_scene* s, int on)
{
    if (!s) return CRT_E_INVALID;
    std::lock_guard<std::mutex> lock(s->mu);
    s->compress = on != 0;
    return CRT_OK;
}
int crt_scene_set_wide_nodes(crt_scene* s, int on)
{
    if (!s) return CRT_E_INVALID;
    std::lock_guard<std::mutex> lock(s->mu);
    if (on && !s->wide8_valid) return CRT_E_UNSUPPORTED;          /* the scene was created without the copy */
    s->wide8 = on != 0;
    return CRT_OK;
}
int crt_host_qbvh_wide8_records(const crt_scene_desc* desc, uint32_t* records, int32_t* num_records)
{
    if (!desc || !num_records || desc->accel != CRT_ACCEL_BVH4) return CRT_E_INVALID;
    int d4 = 0, d2 = 0;
    try {
        const int rc = validate_desc(desc, &d4, &d2);
        if (rc != CRT_OK) return rc;
        std::vector<uint32_t> w8;
        const bool ok = desc->num_nodes4 > 0 && wide8::collapse(desc, w8);
        *num_records = ok ? (int32_t)(w8.size() / CRT_W8_WORDS) : 0;
        if (ok && records) memcpy(records, w8.data(), w8.size() * sizeof(uint32_t));
    } catch (const std::bad_alloc&) { return CRT_E_NOMEM; }
    return CRT_OK;
}
int crt_scene_set_ray_binning(crt_scene* s, int on)
{
    if (!s) return CRT_E_INVALID;
    std::lock_guard<std::mutex> lock(s->mu);
    s->binning = on != 0;
    return CRT_OK;
}
int crt_scene_get_ray_binning(crt_scene* s, int* on)
{
    if (!s || !on) return CRT_E_INVALID;
    std::lock_guard<std::mutex> lock(s->mu);
    *on = s->binning ? 1 : 0;
    return CRT_OK;
}
int crt_scene_get_traverse_mode(const crt_scene* s, int* resolved)
{
    if (!s || !resolved) return CRT_E_INVALID;
    *resolved = resolved_mode(s);
    return CRT_OK;
}

static int intersect_dev(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, void* stream, bool noalloc, const uint32_t* d_n = 0,
                         unsigned slice = 0, unsigned slice_count = 0, bool no_bin = false)
{
    if (!s || (n > 0 && (!d_rays || !d_hits))) return CRT_E_INVALID;
    if (n == 0) return CRT_OK;
    if (n > 0xF0000000u) return CRT_E_TOOBIG;                        /* 32-bit queue head; waves claim 64 slots at a time past the end */
    if (noalloc && n > CRT_NA_MAX_RAYS) return CRT_E_TOOBIG;         /* the file-scope scratch holds the order of this many rays */
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    la.rays = d_rays; la.hits = d_hits; la.n = (unsigned)n; la.n_dev = d_n;
    hipStream_t st = (hipStream_t)stream;
    LaunchOpts o(true, noalloc);
    o.slice = slice; o.slice_count = slice_count; o.no_bin = no_bin;
    if (s->tri_only) return s->count ? dispatch_closest<true, true>(s, la, st, o) : dispatch_closest<true, false>(s, la, st, o);
    return s->count ? dispatch_closest<false, true>(s, la, st, o) : dispatch_closest<false, false>(s, la, st, o);
}

int crt_intersect_batch_dev(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, void* stream)
{
    return intersect_dev(s, d_rays, n, d_hits, stream, false);
}

/* The same kernels on the file-scope scratch array: no hipMalloc / hipFree / allocating call on this path. */
int crt_intersect_batch_dev_noalloc(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, void* stream)
{
    return intersect_dev(s, d_rays, n, d_hits, stream, true);
}

/* One share of a batch that k GPUs (or k launches) trace together: slice `slice` of `num_slices`.  With ray binning the whole
 * batch is binned and the share is a run of whole Morton cells holding about n / num_slices rays (BinSlice, crt_binning.h): each
 * share touches its own piece of the scene with the batch's full ray density per cell.  Without binning (no accelerator, binning
 * switched off, fewer than 2^18 rays) the share is the index range [n k / K, n (k + 1) / K).  Either way the shares of the K
 * slices tile the batch exactly, and each hit lands at its ray's own index in d_hits; records of other shares are not touched. */
int crt_intersect_batch_dev_slice(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, uint32_t slice, uint32_t num_slices, void* stream)
{
    if (num_slices < 1 || slice >= num_slices || num_slices > 65536u) return CRT_E_INVALID;
    return intersect_dev(s, d_rays, n, d_hits, stream, false, 0, slice, num_slices);
}

/* Sizes the scene's launch scratch for batches of up to max_rays rays ahead of time, so that no later launch allocates or
 * synchronises (the ray-order buffer is created for 2^24 rays and otherwise grows, with a stream synchronise, inside the first
 * launch of a larger batch). */
int crt_scene_reserve_rays(crt_scene* s, size_t max_rays)
{
    if (!s) return CRT_E_INVALID;
    if (max_rays > 0xF0000000u) return CRT_E_TOOBIG;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->bin_grid_valid) return CRT_OK;                       /* this scene's launches use no ray order */
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < CRT_SLOTS; k++) { const int rc = grow_perm(s->own[k], max_rays, 0); if (rc != CRT_OK) return rc; }
    return CRT_OK;
}

/* Counted launches: the number of rays is read from device memory by the kernels themselves (*d_n, at most n_max), so a
 * wavefront driver can chain extend -> shade -> shadow launches without the host ever learning the queue sizes. */
int crt_intersect_batch_dev_counted(crt_scene* s, const crt_ray* d_rays, const uint32_t* d_n, size_t n_max, crt_hit* d_hits, void* stream)
{
    if (!d_n) return CRT_E_INVALID;
    return intersect_dev(s, d_rays, n_max, d_hits, stream, false, d_n);
}

static int shadow_dev(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, size_t n,
                      float* d_t_out, uint8_t* d_occluded_out, void* stream, const uint32_t* d_n, unsigned slice = 0, unsigned slice_count = 0,
                      bool no_bin = false)
{
    if (!s || (n > 0 && (!d_rays || !d_light_dist))) return CRT_E_INVALID;
    if (n == 0) return CRT_OK;
    if (n > 0xF0000000u) return CRT_E_TOOBIG;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    LaunchArgs la;
    memset(&la, 0, sizeof la);
    la.rays = d_rays; la.light_dist = d_light_dist; la.t_out = d_t_out; la.occ_out = d_occluded_out; la.n = (unsigned)n; la.n_dev = d_n;
    hipStream_t st = (hipStream_t)stream;
    if (s->tri_only) return s->count ? dispatch_shadow<true, true>(s, la, st, slice, slice_count, no_bin) : dispatch_shadow<true, false>(s, la, st, slice, slice_count, no_bin);
    return s->count ? dispatch_shadow<false, true>(s, la, st, slice, slice_count, no_bin) : dispatch_shadow<false, false>(s, la, st, slice, slice_count, no_bin);
}
int crt_shadow_batch_dev_slice(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, size_t n,
                               float* d_t_out, uint8_t* d_occluded_out, uint32_t slice, uint32_t num_slices, void* stream)
{
    if (num_slices < 1 || slice >= num_slices || num_slices > 65536u) return CRT_E_INVALID;
    return shadow_dev(s, d_rays, d_light_dist, n, d_t_out, d_occluded_out, stream, 0, slice, num_slices);
}
int crt_shadow_batch_dev(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, size_t n,
                         float* d_t_out, uint8_t* d_occluded_out, void* stream)
{
    return shadow_dev(s, d_rays, d_light_dist, n, d_t_out, d_occluded_out, stream, 0);
}
int crt_shadow_batch_dev_counted(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, const uint32_t* d_n, size_t n_max,
                                 float* d_t_out, uint8_t* d_occluded_out, void* stream)
{
    if (!d_n) return CRT_E_INVALID;
    return shadow_dev(s, d_rays, d_light_dist, n_max, d_t_out, d_occluded_out, stream, d_n);
}

} /* extern "C" */

/* Library-internal (crt_internal.h, not exported): the renderer's counted queries with the binning decided per call instead of
 * through the scene's setting, so that a renderer and another caller of the same scene do not see each other's choice. */
int crt_internal_intersect_counted(crt_scene* s, const crt_ray* d_rays, const uint32_t* d_n, size_t n_max, crt_hit* d_hits, void* stream, bool binned)
{
    if (!d_n) return CRT_E_INVALID;
    return intersect_dev(s, d_rays, n_max, d_hits, stream, false, d_n, 0, 0, !binned);
}
int crt_internal_shadow_counted(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, const uint32_t* d_n, size_t n_max,
                                float* d_t_out, uint8_t* d_occluded_out, void* stream, bool binned)
{
    if (!d_n) return CRT_E_INVALID;
    return shadow_dev(s, d_rays, d_light_dist, n_max, d_t_out, d_occluded_out, stream, d_n, 0, 0, !binned);
}

extern "C" {

/* Host-buffer entry points.  The batch is cut into chunks that alternate between two streams,
 * each doing  H2D(chunk) -> traversal kernel -> D2H(chunk)  in stream order, so the copies of one
 * chunk overlap the kernel of the other (the kernels themselves are ordered by launch()).  With
 * buffers from crt_host_alloc (pinned) the copies are DMA straight from / to the caller's memory;
 * with ordinary malloc'd buffers HIP stages them and the calls are merely correct. */
namespace {

#ifndef CRT_CHUNK_LOG2
#define CRT_CHUNK_LOG2 22
#endif
const size_t CRT_CHUNK_RAYS = (size_t)1 << CRT_CHUNK_LOG2;       /* 2^22 rays: 96 MB in, 80 MB out per chunk (2^20 / 2^21 / 2^23 measured slower) */

int ensure_chunks(crt_scene* s, size_t need, bool shadow)
{
    if (!s->chunk_streams_valid) {
        if (!hip_ok(hipStreamCreateWithFlags(&s->chunk_stream[0], hipStreamNonBlocking), "hipStreamCreate")) return CRT_E_HIP;
        if (!hip_ok(hipStreamCreateWithFlags(&s->chunk_stream[1], hipStreamNonBlocking), "hipStreamCreate")) {
            (void)hipStreamDestroy(s->chunk_stream[0]);
            return CRT_E_HIP;
        }
        s->chunk_streams_valid = true;
    }
    if (need > s->chunk_cap) {
        for (int k = 0; k < 2; k++) {
            void** q[] = { &s->d_chunk_in[k], &s->d_chunk_aux[k], &s->d_chunk_out[k], &s->d_chunk_out2[k] };
            for (void** p : q) { if (*p) (void)hipFree(*p); *p = 0; }
        }
        s->chunk_cap = 0;
        for (int k = 0; k < 2; k++) {
            if (!hip_ok(hipMalloc(&s->d_chunk_in[k], need * sizeof(crt_ray)), "hipMalloc") ||
                !hip_ok(hipMalloc(&s->d_chunk_out[k], need * sizeof(crt_hit)), "hipMalloc") ||      /* hits, or shadow t (4 B) */
                !hip_ok(hipMalloc(&s->d_chunk_aux[k], need * sizeof(float)), "hipMalloc") ||        /* light_dist */
                !hip_ok(hipMalloc(&s->d_chunk_out2[k], need), "hipMalloc")) return CRT_E_NOMEM;     /* occluded */
        }
        s->chunk_cap = need;
    }
    (void)shadow;
    return CRT_OK;
}

std::mutex g_chunk_mu;       /* one host-buffer call at a time per process: they share each scene's chunk buffers */

} // namespace

int crt_host_alloc(size_t bytes, void** out)
{
    if (!out) return CRT_E_INVALID;
    *out = 0;
    if (bytes == 0) bytes = 1;
    if (!hip_ok(hipHostMalloc(out, bytes, hipHostMallocDefault), "hipHostMalloc")) return CRT_E_NOMEM;
    return CRT_OK;
}
void crt_host_free(void* p) { if (p) (void)hipHostFree(p); }

int crt_intersect_batch(crt_scene* s, const crt_ray* rays, size_t n, crt_hit* hits)
{
    if (!s || (n > 0 && (!rays || !hits))) return CRT_E_INVALID;
    if (n == 0) return CRT_OK;
    if (n > 0xF0000000u) return CRT_E_TOOBIG;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    std::lock_guard<std::mutex> lock(g_chunk_mu);
    const size_t chunk = n < CRT_CHUNK_RAYS ? n : CRT_CHUNK_RAYS;
    int rc = ensure_chunks(s, chunk, false);
    int k = 0;
    for (size_t off = 0; rc == CRT_OK && off < n; off += chunk, k ^= 1) {
        const size_t m = n - off < chunk ? n - off : chunk;
        hipStream_t st = s->chunk_stream[k];           /* stream order protects buffer k from its previous use */
        if (!hip_ok(hipMemcpyAsync(s->d_chunk_in[k], rays + off, m * sizeof(crt_ray), hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D")) { rc = CRT_E_HIP; break; }
        rc = crt_intersect_batch_dev(s, (const crt_ray*)s->d_chunk_in[k], m, (crt_hit*)s->d_chunk_out[k], st);
        if (rc != CRT_OK) break;
        if (!hip_ok(hipMemcpyAsync(hits + off, s->d_chunk_out[k], m * sizeof(crt_hit), hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H")) rc = CRT_E_HIP;
    }
    /* always drain both streams before the caller's buffers go out of our hands */
    for (int j = 0; j < 2; j++)
        if (s->chunk_streams_valid && !hip_ok(hipStreamSynchronize(s->chunk_stream[j]), "hipStreamSynchronize") && rc == CRT_OK) rc = CRT_E_HIP;
    return rc;
}

int crt_shadow_batch(crt_scene* s, const crt_ray* rays, const float* light_dist, size_t n,
                     float* t_out, uint8_t* occluded_out)
{
    if (!s || (n > 0 && (!rays || !light_dist))) return CRT_E_INVALID;
    if (n == 0) return CRT_OK;
    if (n > 0xF0000000u) return CRT_E_TOOBIG;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    std::lock_guard<std::mutex> lock(g_chunk_mu);
    const size_t chunk = n < CRT_CHUNK_RAYS ? n : CRT_CHUNK_RAYS;
    int rc = ensure_chunks(s, chunk, true);
    int k = 0;
    for (size_t off = 0; rc == CRT_OK && off < n; off += chunk, k ^= 1) {
        const size_t m = n - off < chunk ? n - off : chunk;
        hipStream_t st = s->chunk_stream[k];
        if (!hip_ok(hipMemcpyAsync(s->d_chunk_in[k], rays + off, m * sizeof(crt_ray), hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D") ||
            !hip_ok(hipMemcpyAsync(s->d_chunk_aux[k], light_dist + off, m * sizeof(float), hipMemcpyHostToDevice, st), "hipMemcpyAsync H2D")) { rc = CRT_E_HIP; break; }
        rc = crt_shadow_batch_dev(s, (const crt_ray*)s->d_chunk_in[k], (const float*)s->d_chunk_aux[k], m,
                                  (float*)s->d_chunk_out[k], (uint8_t*)s->d_chunk_out2[k], st);
        if (rc != CRT_OK) break;
        if (t_out && !hip_ok(hipMemcpyAsync(t_out + off, s->d_chunk_out[k], m * sizeof(float), hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H")) { rc = CRT_E_HIP; break; }
        if (occluded_out && !hip_ok(hipMemcpyAsync(occluded_out + off, s->d_chunk_out2[k], m, hipMemcpyDeviceToHost, st), "hipMemcpyAsync D2H")) { rc = CRT_E_HIP; break; }
    }
    for (int j = 0; j < 2; j++)
        if (s->chunk_streams_valid && !hip_ok(hipStreamSynchronize(s->chunk_stream[j]), "hipStreamSynchronize") && rc == CRT_OK) rc = CRT_E_HIP;
    return rc;
}

/* ---- primary rays ------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_primary_rays(crt_camera cam, crt_film film, unsigned first_pixel, unsigned n,
                                                      const float2* image_samples, const float2* lens_samples, crt_ray* rays)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float2 is = image_samples ? image_samples[i] : make_float2(0.5f, 0.5f);
    const float2 ls = lens_samples ? lens_samples[i] : make_float2(0.5f, 0.5f);
    crt_ray r;
    cam_primary_ray(&r, cam, film, first_pixel + i, is.x, is.y, ls.x, ls.y);
    float2* out = reinterpret_cast<float2*>(rays + i);
    out[0] = make_float2(r.o[0], r.o[1]); out[1] = make_float2(r.o[2], r.d[0]); out[2] = make_float2(r.d[1], r.d[2]);
}

int crt_camera_init(crt_camera* cam, int type, float focal_length, float lens_radius)
{
    if (!cam || type < 0 || type > 1) return CRT_E_INVALID;
    cam_init(cam, type, focal_length, lens_radius);
    return CRT_OK;
}
int crt_camera_look_at(crt_camera* cam, const float position[3], const float look_point[3], const float up[3])
{
    if (!cam || !position || !look_point || !up) return CRT_E_INVALID;
    cam_look_at(cam, position, look_point, up);
    return CRT_OK;
}
int crt_camera_set_focal_dist(crt_camera* cam, float focal_pt_dist)
{
    if (!cam) return CRT_E_INVALID;
    cam_set_focal_dist(cam, focal_pt_dist);
    return CRT_OK;
}
int crt_film_dimension(crt_film* film, float fov, int res_width, int res_height, const crt_camera* cam)
{
    if (!film || !cam || res_width < 1 || res_height < 1) return CRT_E_INVALID;
    cam_film_dimension(film, fov, res_width, res_height, cam);
    return CRT_OK;
}
int crt_primary_rays_dev(int device, const crt_camera* cam, const crt_film* film, uint32_t first_pixel, size_t n,
                         const float* d_image_samples, const float* d_lens_samples, crt_ray* d_rays, void* stream)
{
    if (!cam || !film || (n > 0 && !d_rays) || film->frame_res_width < 1 || cam->type < 0 || cam->type > 1) return CRT_E_INVALID;
    if (n == 0) return CRT_OK;
    if (n > 0xF0000000u || (unsigned long long)first_pixel + n > 0xFFFFFFFFull) return CRT_E_TOOBIG;
    DeviceGuard guard(device);
    if (!guard.ok) return CRT_E_HIP;
    hipLaunchKernelGGL(k_primary_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *cam, *film,
                       (unsigned)first_pixel, (unsigned)n, (const float2*)d_image_samples, (const float2*)d_lens_samples, d_rays);
    HIPCHK(hipGetLastError());
    return CRT_OK;
}

int crt_scene_enable_counters(crt_scene* s, int on)
{
    if (!s) return CRT_E_INVALID;
    s->count = on != 0;
    return CRT_OK;
}
int crt_scene_get_counters(crt_scene* s, crt_counters* out)
{
    if (!s || !out) return CRT_E_INVALID;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    unsigned long long h[4];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h, s->d_counters, sizeof h, hipMemcpyDeviceToHost));
    out->rays_closest = h[0]; out->rays_shadow = h[1]; out->nodes_visited = h[2]; out->prims_tested = h[3];
#if CRT_DIAG
    {
        unsigned long long d[24];
        HIPCHK(hipMemcpy(d, s->d_counters + 8, sizeof d, hipMemcpyDeviceToHost));
        {
            unsigned long long t4[4];
            HIPCHK(hipMemcpy(t4, s->d_counters + 4, sizeof t4, hipMemcpyDeviceToHost));
            fprintf(stderr, "[diag] tail (after the wave's queue ran dry): %llu wave-iterations in all, longest wave %llu, alive lanes per tail iteration %.2f, hand-overs %llu\n",
                    t4[0], t4[3], t4[0] ? (double)t4[1] / t4[0] : 0.0, t4[2]);
        }
        fprintf(stderr, "[diag] iterations per ray: max %llu; rays by floor(log2(iterations)) 0..11+:", d[11]);
        for (int k = 0; k < 12; k++) fprintf(stderr, " %llu", d[12 + k]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[diag] wave-iterations %llu  alive/it %.2f  walking/it %.2f  node lanes/it %.2f  object lanes/it %.2f  stalled/it %.2f  iterations with a node step %.3f  with an object step %.3f\n",
                d[0], (double)d[1] / d[0], (double)d[8] / d[0], (double)d[2] / d[0], (double)d[3] / d[0], (double)d[5] / d[0], (double)d[6] / d[0], (double)d[7] / d[0]);
        fprintf(stderr, "[diag] node-step iterations whose lanes all visit one node: %.4f of them; distinct nodes per node-step iteration %.2f\n",
                d[6] ? (double)d[9] / d[6] : 0.0, d[6] ? (double)d[10] / d[6] : 0.0);
    }
#endif
#if CRT_TIMELINE
    {
            unsigned long long w[8];
            HIPCHK(hipMemcpy(w, s->d_counters + 32, sizeof w, hipMemcpyDeviceToHost));
            const double first = (double)(~w[0]), tick = 1.0e-5;      /* 100 MHz -> ms */
            fprintf(stderr, "[diag] timeline of the LAST counted launch only if counters were reset before it (ms after the first wave started): waves start until %.4f; queue dry for the first wave at %.4f, for the last at %.4f; first wave ends at %.4f, last at %.4f; mean tail of a wave %.4f (%llu waves)\n",
                    ((double)w[1] - first) * tick, ((double)(~w[2]) - first) * tick, ((double)w[3] - first) * tick,
                    ((double)(~w[4]) - first) * tick, ((double)w[5] - first) * tick, w[7] ? (double)w[6] / (double)w[7] * tick : 0.0, w[7]);
        }
#endif
    return CRT_OK;
}
int crt_scene_reset_counters(crt_scene* s)
{
    if (!s) return CRT_E_INVALID;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(s->d_counters, 0, 512));
    return CRT_OK;
}
/* The calling thread's most recent launch through `s`: its slot of the timing ring, if it is still there. */
static int my_timing(crt_scene* s, LaunchTiming** out)
{
    if (tl_scene_uid != s->uid || tl_seq == 0 || tl_seq > s->seq) return CRT_E_INVALID;     /* this thread has not launched through s */
    if (s->seq - tl_seq >= CRT_TIMING_RING) return CRT_E_INVALID;                          /* overwritten by later launches */
    LaunchTiming& lt = s->ring[(tl_seq - 1) % CRT_TIMING_RING];
    if (lt.seq != tl_seq) return CRT_E_INVALID;
    *out = &lt;
    return CRT_OK;
}
int crt_scene_last_binning_ms(crt_scene* s, float* ms)
{
    if (!s || !ms) return CRT_E_INVALID;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    std::lock_guard<std::mutex> lock(s->mu);
    LaunchTiming* lt = 0;
    int rc = my_timing(s, &lt);
    if (rc != CRT_OK) return rc;
    *ms = 0.0f;
    if (!lt->binned) return CRT_OK;
    HIPCHK(hipEventSynchronize(lt->ev_start));
    HIPCHK(hipEventElapsedTime(ms, lt->ev_bin, lt->ev_start));
    return CRT_OK;
}
int crt_scene_last_kernel_ms(crt_scene* s, float* ms)
{
    if (!s || !ms) return CRT_E_INVALID;
    DeviceGuard guard(s->device);
    if (!guard.ok) return CRT_E_HIP;
    std::lock_guard<std::mutex> lock(s->mu);
    LaunchTiming* lt = 0;
    int rc = my_timing(s, &lt);
    if (rc != CRT_OK) return rc;
    HIPCHK(hipEventSynchronize(lt->ev_stop));
    HIPCHK(hipEventElapsedTime(ms, lt->ev_start, lt->ev_stop));
    return CRT_OK;
}

} // extern "C"
