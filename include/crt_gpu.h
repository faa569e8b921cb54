/*
 * crt_gpu.h -- C ABI of the MI355X (gfx950) ray traversal + intersection engine
 * that sits behind craytracer's scene-intersection API.
 *
 * Drop-in boundary.  The reference has no FFI layer: its hot path
 * is two free functions,
 *     float intersectTest(ShadeRec*, const SceneObjects*, const Ray)            intersect.h:365
 *     float shadowIntersectTest(const SceneObjects*, const Ray, float light_dist) intersect.h:443
 * over a `SceneObjects` handle (scene/scenedata.h:20-28) whose accelerator was
 * built by buildSceneAccel (buildscene.h:698-731).  This header is what a
 * maintainer binds instead: a flat scene description goes in once, ray batches
 * go in, hit records come out.  Plain C, plain pointers and sizes; no torch or
 * HIP types appear in any signature (streams are passed as void*).
 *
 * All functions return 0 (CRT_OK) or a negative CRT_E_* code; nothing throws,
 * nothing calls exit().  A miss is data (t == CRT_TMAX, obj == -1), not an error.
 */
#ifndef CRT_GPU_H
#define CRT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants: util/constants.h:45-50 -------------------------------- */
#define CRT_K_EPSILON   0.000007f               /* K_EPSILON */
#define CRT_TMAX        3.402823466e+38F        /* TMAX == FLT_MAX */
#define CRT_K_FLAT_AABB 0.001f                  /* K_FLAT_AABB */
#define CRT_HUGEVALUE   1.0E10f                 /* HUGEVALUE */

/* ---- status codes ------------------------------------------------------ */
enum {
    CRT_OK            =  0,
    CRT_E_INVALID     = -1,   /* bad argument / inconsistent description       */
    CRT_E_NOMEM       = -2,   /* host or device allocation failed              */
    CRT_E_HIP         = -3,   /* a HIP runtime call failed (see crt_last_hip_error) */
    CRT_E_UNSUPPORTED = -4,   /* valid request this build does not implement   */
    CRT_E_TOOBIG      = -5,   /* size beyond a documented limit                */
    CRT_E_NODEVICE    = -6    /* no usable gfx950 device                       */
};

/* ---- enums mirrored from the reference -------------------------------- */
/* ObjectType, shapes/objecttype.h:4-18 (same numeric values). */
enum crt_prim_type {
    CRT_SPHERE = 0, CRT_PLANE = 1, CRT_RECTANGLE = 2, CRT_AABOX = 3, CRT_TRIANGLE = 4,
    CRT_GENERICOPENCYLINDER = 5, CRT_DISK = 6, CRT_GENERICTORUS = 7, CRT_INSTANCED = 8,
    CRT_COMPOUND = 9, CRT_FLAT_TRIANGLE = 10, CRT_SMOOTH_TRIANGLE = 11
};
/* AccelType, scene/scenedata.h:12-18 (same numeric values). */
enum crt_accel { CRT_ACCEL_NONE = 0, CRT_ACCEL_GRID = 1, CRT_ACCEL_BVH = 2, CRT_ACCEL_BVH4 = 3 };

/* Traversal semantics for closest-hit on BVH / BVH4.
 *  CRT_TRAVERSE_PRUNED : best-t pruning against the ray's global nearest hit, in the
 *      reference's child order.  Identical result whenever every primitive's hit lies
 *      inside its (padded) box -- true for triangles, K_FLAT_AABB pad (shapes.cpp:207-211).
 *  CRT_TRAVERSE_EXACT  : the reference's own control flow -- BVH4: min_t local to each
 *      node visit (bvh4.h:354); BVH: both overlapped children, no pruning
 *      (intersect.h:310-342).  Needed when boxes are not bounding (rotated/scaled
 *      instances, shapes.cpp:284-304).
 *  CRT_TRAVERSE_AUTO   : PRUNED when every object below the accelerator is a triangle, sphere,
 *      rectangle or disk -- shapes whose reference boxes bound them (shapes.cpp:146-263) -- else EXACT
 *      (instanced / compound / generic shapes).  Stack limits: QBVH depth <= 21 (PRUNED), <= 15
 *      (EXACT, one frame marker per level); BVH depth <= 63; deeper trees get CRT_E_TOOBIG. */
enum crt_traverse_mode { CRT_TRAVERSE_AUTO = 0, CRT_TRAVERSE_PRUNED = 1, CRT_TRAVERSE_EXACT = 2 };

/* ---- ray and hit records ---------------------------------------------- */
/* Bit-identical to `Ray` (util/ray.h:5-9): origin[3], direction[3]; 24 bytes. */
typedef struct crt_ray { float o[3]; float d[3]; } crt_ray;

/* What intersectTest returns, minus the 352-byte ShadeRec (util/shaderec.h:7-19),
 * which the host rebuilds from these five words.
 *   t     : hit distance, CRT_TMAX on miss (intersect.h:372).
 *   obj   : INSERTION-ORDER index of the hit object (SURVEY 8a "Primitive ID"), -1 on miss.
 *   part  : index inside a CompoundObject (instanced.cpp:164), else 0; -1 on miss.
 *   beta,gamma : triangle barycentrics as calcTriangleIntersect writes them
 *           (triangle.cpp:44-45); 0 for other shapes and on miss. */
typedef struct crt_hit { float t; int32_t obj; int32_t part; float beta; float gamma; } crt_hit;

/* ---- hit record -> ShadeRec, host side ------------------------------------------------
 * ShadeRec (util/shaderec.h:7-19) without its by-value Material / BSDF copies (an index instead).  `written`:
 * which fields the reference's intersector writes for the shape that was hit -- the others are left as passed in,
 * as the reference leaves them holding whatever the record held before. */
enum { CRT_SR_HIT_POINT = 1, CRT_SR_NORMAL = 2, CRT_SR_DPDU = 4, CRT_SR_DPDV = 8, CRT_SR_WO = 16, CRT_SR_UV = 32,
       CRT_SR_BY_CONVENTION = 256 };   /* some field was filled by this library's convention, not by reference code (see below) */
typedef struct crt_shade_rec {
    float    hit_point[3], normal[3], dpdu[3], dpdv[3], wo[3], uv[2];
    int32_t  material;                 /* obj_material[hit.obj] (getObjectMatPtr's material, as an index), -1 when no map is given */
    uint32_t written;                  /* CRT_SR_* ; 0 for a miss */
} crt_shade_rec;

/* ---- analytic shapes --------------------------------------------------- */
/* One non-triangle shape.  p[] holds the geometric fields of the reference struct:
 *   SPHERE   (sphere.h:10-17)   p = center[3], radius, min_theta, max_theta, phi
 *   PLANE    (plane.h:8-14)     p = point[3], normal[3]
 *   RECTANGLE(rect.h:9-16)      p = point[3], width[3], height[3], normal[3]
 *   AABOX    (generic.h AABox)  p = min[3], max[3], shadow (0/1)
 *   GENERICOPENCYLINDER         p = half_height, radius, phi, normal_type (0 OPEN, 1 CONVEX, 2 CONCAVE: generic.h:13-18; shading only)
 *   DISK     (disk.h:9-16)      p = center[3], normal[3], radius
 *   GENERICTORUS                p = swept_radius, tube_radius, phi
 *   COMPOUND (instanced.h:7-13) p = aabb.min[3], aabb.max[3]; parts = shapes[first_part .. first_part+num_parts)
 *   INSTANCED(instanced.h:15-20) inner = index in shapes[] of the wrapped shape;
 *                               inv_xform = row-major inv_transform (mat4) */
typedef struct crt_shape {
    int32_t type;
    int32_t inner;
    int32_t first_part;
    int32_t num_parts;
    float   p[12];
    float   inv_xform[16];
} crt_shape;

/* ---- scene input: objects in INSERTION order --------------------------- */
/* Equivalent of the SceneObjects array after SceneObjects_push_obj calls
 * (scenedata.h:40-48) and before mvNonGridObjToStart / the builders reorder it.
 * obj_type[i] in {TRIANGLE, FLAT_TRIANGLE, SMOOTH_TRIANGLE}: obj_ref[i] indexes
 * tri_verts (9 floats: v0, v1, v2); otherwise obj_ref[i] indexes shapes[].
 * obj_type == NULL means "num_obj triangles, object i = triangle i, type SMOOTH_TRIANGLE". */
typedef struct crt_scene_input {
    int32_t         num_obj;
    const int32_t*  obj_type;
    const int32_t*  obj_ref;
    const float*    tri_verts;
    int32_t         num_tris;
    const crt_shape* shapes;
    int32_t         num_shapes;
} crt_scene_input;

/* ---- built accelerator, flat, host memory ------------------------------ */
/* QBVH node: BVHNode4 (bvh4.h:100-110) with pointers turned into indices.
 *   bbox   : minx[4] miny[4] minz[4] maxx[4] maxy[4] maxz[4]   (bvh4.h:105)
 *   child  : type == -1 -> index into nodes4[]; else index of the leaf's first
 *            object in the FINAL objects[] order (absolute, non-grid prefix included)
 *   type   : -1 inner, else 0..4 leaf object count               (bvh4.h:107)
 *   axis   : the byte BVH4_build leaves in node->axis (third split axis; bvh4.h:194-195) */
typedef struct crt_bvh4_node {
    float   bbox[24];
    int32_t child[4];
    int8_t  type[4];
    uint8_t axis;
    uint8_t pad[11];
} crt_bvh4_node;                                   /* 128 bytes */

/* Binary BVH node: BVHNode (bvh.h:19-26).  Leaf: left == -1, obj = object index. */
typedef struct crt_bvh2_node {
    float   bmin[3];
    float   bmax[3];
    int32_t left;      /* -1 for a leaf, else index of left child  */
    int32_t right;     /* leaf: FINAL-order object index; else right child */
} crt_bvh2_node;                                   /* 32 bytes */

/* UniformGrid (uniformgrid.h:10-17); cells as CSR instead of IntVector. */
typedef struct crt_grid {
    int32_t nx, ny, nz, num_cells;
    float   wx, wy, wz;
    float   bmin[3], bmax[3];
} crt_grid;

/* Scene as intersectTest sees it: FINAL objects[] order + one accelerator. */
typedef struct crt_scene_desc {
    int32_t accel;                 /* crt_accel */
    int32_t num_obj;
    int32_t num_non_grid_obj;      /* objects[0..num_non_grid_obj) are tested linearly */
    const int32_t* obj_type;       /* [num_obj] */
    const int32_t* obj_ref;        /* [num_obj] */
    const int32_t* obj_id;         /* [num_obj] insertion-order id reported in crt_hit.obj */
    const float*   tri_verts;      /* [num_tris*9] */
    int32_t        num_tris;
    const crt_shape* shapes;
    int32_t        num_shapes;
    const crt_bvh4_node* nodes4;   /* BVH4: nodes4[0] is the root */
    int32_t        num_nodes4;
    const crt_bvh2_node* nodes2;   /* BVH: nodes2[0] is the root */
    int32_t        num_nodes2;
    crt_grid       grid;           /* GRID */
    const int32_t* cell_start;     /* [num_cells+1] */
    const int32_t* cell_items;     /* [cell_start[num_cells]] FINAL-order object indices */
} crt_scene_desc;

/* ---- host-side build (the reference's host keeps doing this; provided so the
 *      library is usable stand-alone) ------------------------------------- */
typedef struct crt_built_scene crt_built_scene;     /* owns the arrays a crt_scene_desc points to */

/* mvNonGridObjToStart (scenedata.h:368-382) + buildSceneAccel (buildscene.h:698-731):
 * UniformGrid_create (uniformgrid.h:54-154, multiplier 3) | BVH_build (bvh.h:117-154) |
 * BVH4_build (bvh4.h:127-297).  Sorts are stable (see DESIGN.md "builder order"). */
int  crt_build_scene(const crt_scene_input* in, int accel, crt_built_scene** out);
/* Same with options.  CRT_BUILD_SAH (QBVH only): the three split planes of every node come from a binned
 * surface-area heuristic instead of the reference's object median on the longest axis (bvh4.h:143-195); node format,
 * leaf rule and traversal are unchanged, so hits are those of the reference's tree (up to which of two exactly equal-t
 * hits is met first), with several times fewer node visits when a few large triangles share the scene with fine
 * meshes.  No counterpart in the reference. */
enum { CRT_BUILD_SAH = 1 };
int  crt_build_scene_ex(const crt_scene_input* in, int accel, int flags, crt_built_scene** out);
const crt_scene_desc* crt_built_scene_desc(const crt_built_scene* b);
void crt_built_scene_destroy(crt_built_scene* b);

/* ---- device scene ------------------------------------------------------ */
typedef struct crt_scene crt_scene;                 /* opaque; owns device memory */

int  crt_device_count(int* count);
/* Upload `desc` to HBM of `device` in the kernel layout.  No host pointer is retained. */
int  crt_scene_create(int device, const crt_scene_desc* desc, crt_scene** out);
void crt_scene_destroy(crt_scene* s);
/* crt_traverse_mode for closest-hit; default CRT_TRAVERSE_AUTO. */
int  crt_scene_set_traverse_mode(crt_scene* s, int mode);
int  crt_scene_get_traverse_mode(const crt_scene* s, int* resolved_mode);

/* Page-locked host memory for ray / hit arrays.  No counterpart in the reference (it mallocs);
 * buffers from here let the host-buffer calls below DMA straight from / to the caller's arrays
 * and overlap the copies with the traversal kernels.  Ordinary malloc'd buffers work too, slower. */
int  crt_host_alloc(size_t bytes, void** out);
void crt_host_free(void* p);

/* Ray binning (on by default).  No counterpart in the reference.  For QBVH, binary-BVH and grid scenes and batches of
 * 2^18 rays or more, the library first counting-sorts the ray INDICES by the grid cell of the ray
 * origin and pulls rays from the queue in that order (better L2 reuse); every ray is traced
 * exactly as before and its result is written at its own index, so outputs do not change. */
int  crt_scene_set_ray_binning(crt_scene* s, int on);
int  crt_scene_get_ray_binning(crt_scene* s, int* on);   /* the current setting (crt_renderer_render does not touch it: its own queries are unbinned per launch) */

/* Compressed node copy (experiment switch; see DESIGN.md for its default).  No counterpart in the reference.  For QBVH
 * scenes crt_scene_create also keeps a 64-byte conservative copy of every node (per axis an origin and a scale, per
 * child and plane one byte rounded outward, checked in double); the best-t-pruned closest-hit walk may read that copy
 * instead of the 128-byte nodes.  Every compressed box contains the node's box, so the walk visits a superset of the
 * nodes in the same order and returns the same hit records bit for bit. */
int  crt_scene_set_node_compression(crt_scene* s, int on);

/* 8-wide node copy (experiment switch, off by default: measured 0.83 x the 4-wide walk, DESIGN.md "8-wide nodes").  No
 * counterpart in the reference.  When the environment variable CRT_WIDE8 is 1 at crt_scene_create, a triangle-only QBVH scene
 * keeps a third copy of the tree with up to eight conservative boxes per 128-byte record, cut out of the same binary hierarchy
 * so as to minimise the expected number of node visits, and closest-hit launches (not the no-allocation entry point) walk that
 * copy near to far: 37 % fewer dependent node fetches per ray.  What BVH4IntersectTest returns beyond the nearest t -- WHICH of
 * several objects at exactly that t -- depends on its visiting order, so every ray for which that could matter (two objects
 * compare equal at the nearest t; a ray whose slab arithmetic can produce NaN; a ray whose pending children overflow the lane's
 * stack) is put on a list and walked again, inside the same call, by the 4-wide walk in the reference's order.  Hit records are
 * therefore the same bit for bit; the node / primitive counters count the walk that ran.  on = 0: walk the 4-wide copies;
 * on = 1 without the copy: CRT_E_UNSUPPORTED. */
int  crt_scene_set_wide_nodes(crt_scene* s, int on);

/* Three host-side steps of crt_scene_create, exported so that they can be checked without a GPU (tests/test_host_layout.py).
 * No counterpart in the reference; neither touches a device.
 *  - crt_host_grid_distance_field: for a CRT_ACCEL_GRID description, out[cell] = chessboard (Chebyshev) distance from the cell
 *    to the nearest cell that holds an object (0: holds one), capped at 255; cell = iy*nz*nx + iz*nx + ix as in
 *    gridIntersectTest (intersect.h:94).  The grid walk steps through the next out[cell] - 1 cells without fetching them.
 *  - crt_host_qbvh_node_order: the order the QBVH nodes take in device memory under `policy` (0 as handed over, 1 sibling
 *    clusters, 2 line-aligned sibling clusters, >= 4 treelets of that many nodes).  order[slot] = node index or -1 for a
 *    dead record; `order` must hold 2 * num_nodes4 entries, *num_slots receives how many were written.  Hit records do not
 *    depend on the order. */
int  crt_host_grid_distance_field(const crt_scene_desc* desc, uint8_t* out);
int  crt_host_qbvh_node_order(const crt_scene_desc* desc, int policy, int32_t* order, int32_t* num_slots);
/*  - crt_host_bvh2_wide_records: for a CRT_ACCEL_BVH description, whether a left-first walk meets the leaves in increasing object
 *    index (then the pruned walk may take the nearer child first and settle equal t by index: intersect.h:339 keeps the last of
 *    equal hits in left-first order), and the tree collapsed two levels at a time: 32 floats per record = minx[4] miny[4] minz[4]
 *    maxx[4] maxy[4] maxz[4], four child words (record index, 0x80000000 | object index, 0xFFFFFFFF = unused slot), four unused.
 *    `records` may be null (count only) and otherwise holds at most num_nodes2 records; *num_records = 0 when the tree does
 *    not qualify. */
/*  - crt_host_qbvh_wide8_records: for a CRT_ACCEL_BVH4 description, the 8-wide copy crt_scene_create makes (see
 *    crt_scene_set_wide_nodes): 32 words per record -- frame origin (3 floats), three biased power-of-two exponents packed in one
 *    word, 48 plane bytes lo_x[8] lo_y[8] lo_z[8] hi_x[8] hi_y[8] hi_z[8] (plane = origin + byte * 2^(e - 127), rounded outward),
 *    eight child words (record index | 0x80000000 | count << 28 | first object), eight unused.  `records` may be null (count
 *    only) and otherwise holds at most num_nodes4 records; *num_records = 0 when the tree does not qualify. */
int  crt_host_qbvh_wide8_records(const crt_scene_desc* desc, uint32_t* records, int32_t* num_records);
int  crt_host_bvh2_wide_records(const crt_scene_desc* desc, float* records, int32_t* num_records, int32_t* leaves_in_index_order);

/* n x intersectTest (intersect.h:365-440).  Host buffers: the batch is cut into chunks whose
 * H2D copy, kernel and D2H copy are pipelined on two streams; returns when hits[] is complete. */
int  crt_intersect_batch(crt_scene* s, const crt_ray* rays, size_t n, crt_hit* hits);
/* n x shadowIntersectTest (intersect.h:443-545) + the caller's test
 * `in_shadow = t < light_dist - K_EPSILON` (trace.h:348).  t_out[i] is the value the
 * reference function returns; occluded_out[i] the caller's boolean. Either may be NULL. */
int  crt_shadow_batch(crt_scene* s, const crt_ray* rays, const float* light_dist, size_t n,
                      float* t_out, uint8_t* occluded_out);

/* Same, all pointers DEVICE pointers on the scene's GPU; enqueued on `stream`
 * (a hipStream_t passed as void*, NULL = the null stream); returns without syncing.
 * Threading: any thread may call; launches through one crt_scene share its queue head and
 * stack-overflow scratch, so the library orders them on the device (each waits for the previous
 * one's completion event even across streams).  Use one crt_scene per GPU. */
int  crt_intersect_batch_dev(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, void* stream);
int  crt_shadow_batch_dev(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, size_t n,
                          float* d_t_out, uint8_t* d_occluded_out, void* stream);

/* Counted launches.  No counterpart in the reference (its callers know how many rays they hold): for wavefront drivers whose
 * queue sizes live on the device.  The kernels read the ray count from *d_n (a device word that must be final when the launch
 * reaches it in stream order; values above n_max are clamped to n_max, which also sizes the launch) -- otherwise exactly
 * crt_intersect_batch_dev / crt_shadow_batch_dev.  crt_renderer_render chains its bounces through these, so the host does
 * not wait for the GPU between bounces. */
int  crt_intersect_batch_dev_counted(crt_scene* s, const crt_ray* d_rays, const uint32_t* d_n, size_t n_max, crt_hit* d_hits, void* stream);
int  crt_shadow_batch_dev_counted(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, const uint32_t* d_n, size_t n_max,
                                  float* d_t_out, uint8_t* d_occluded_out, void* stream);

/* One share of a batch traced by several GPUs (or launches) together.  Replaces main.cpp:276-317, where the reference cuts a
 * sample pass into `num_patches` pixel ranges for its worker threads: slice `slice` of `num_slices`.  The whole batch is binned
 * (a counting sort of the ray indices by the cell of the ray origin) and the share is a run of whole cells holding about
 * n / num_slices rays, so each GPU touches its own piece of the scene at the batch's full ray density.  The cut points depend on
 * the cell counts only: K processes that hold the same batch on K GPUs and pass slice = 0 .. K - 1 tile it exactly, each ray
 * traced once.  Hits land at the rays' own indices in d_hits; the records of other shares are not written.  A batch that is
 * not binned (no accelerator, fewer than 2^18 rays, binning switched off) is cut by ray index, [n k / K, n (k + 1) / K). */
int  crt_intersect_batch_dev_slice(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, uint32_t slice, uint32_t num_slices, void* stream);
int  crt_shadow_batch_dev_slice(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, size_t n,
                                float* d_t_out, uint8_t* d_occluded_out, uint32_t slice, uint32_t num_slices, void* stream);

/* Sizes the scene's launch scratch for batches of up to max_rays rays, so that no later launch allocates or synchronises.  (The
 * ray-order buffer of the binning step is created for 2^24 rays; the first launch of a larger batch otherwise grows it, which
 * costs a stream synchronise, a hipFree and a hipMalloc inside that launch.)  No counterpart in the reference. */
int  crt_scene_reserve_rays(crt_scene* s, size_t max_rays);

/* crt_intersect_batch_dev for callers that cannot allocate.  No counterpart in the reference: it exists for
 * callers that may not call hipMalloc -- tight serving loops, harnesses that forbid allocation.  (The launch is not made for HIP
 * graph capture: it records and waits on the library's own events and reads them back for crt_scene_last_kernel_ms.)
 * Same kernels, same results bit for bit; the device memory they need beyond d_rays / d_hits (queue
 * head, ray order of the binning step, per-lane stack overflow) is one file-scope device array of
 * CRT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends, instead of the
 * scene's own buffers.  The call performs no hipMalloc / hipFree / allocating library call.
 *   n > CRT_NOALLOC_MAX_RAYS              -> CRT_E_TOOBIG
 *   resident grid capped at 512 workgroups (the overflow part of the array is sized for that many)
 * All scenes on a GPU share that GPU's array (within a process), so the library orders these calls per DEVICE: each waits, on
 * the device, for the no-allocation launch enqueued before it, whichever scene and stream that one went through (one
 * library-owned event per device, created at scene creation).  Two processes on one GPU hold two arrays. */
#define CRT_NOALLOC_MAX_RAYS      (1u << 22)
#define CRT_NOALLOC_SCRATCH_BYTES (64u * 1024u * 1024u + 33024u)
int  crt_intersect_batch_dev_noalloc(crt_scene* s, const crt_ray* d_rays, size_t n, crt_hit* d_hits, void* stream);

/* ---- primary rays ------------------------------------------- */
/* Camera (camera.h:14-26) without its sample table, and the Film fields the ray set-up reads
 * (camera.h:44-52).  Same field meaning as the reference. */
typedef struct crt_camera {
    int32_t type;                 /* CameraType: 0 Pinhole, 1 ThinLens (camera.h:8-12) */
    float   focal_pt_dist, focal_length, lens_radius;
    float   position[3], x_axis[3], y_axis[3], z_axis[3], focal_point[3];
} crt_camera;
typedef struct crt_film {
    float   fov;
    float   frame_length, frame_height, pixel_length;
    int32_t frame_res_width, frame_res_height;
} crt_film;

/* Host-side set-up, as the reference does it once per scene:
 * initPinholeCameraDefault / initThinLensCameraDefault geometry (camera.cpp:21-51),
 * cameraLookAt (camera.cpp:53-67), setFocalDist (:70-76), calcFilmDimension (:143-148). */
int  crt_camera_init(crt_camera* cam, int type, float focal_length, float lens_radius);
int  crt_camera_look_at(crt_camera* cam, const float position[3], const float look_point[3], const float up[3]);
int  crt_camera_set_focal_dist(crt_camera* cam, float focal_pt_dist);
int  crt_film_dimension(crt_film* film, float fov, int res_width, int res_height, const crt_camera* cam);

/* n x { calcImageCoord (camera.cpp:151-157); calcCameraRay (:130-141) } for pixels
 * first_pixel .. first_pixel + n - 1 (pixel_index as in main.cpp:100-115), on `device`.
 * d_image_samples / d_lens_samples: [n][2] floats on the device (the two Sampler_getSample
 * values of main.cpp:107,:112), or NULL for (0.5, 0.5).  Enqueued on `stream`, no sync. */
int  crt_primary_rays_dev(int device, const crt_camera* cam, const crt_film* film, uint32_t first_pixel, size_t n,
                          const float* d_image_samples, const float* d_lens_samples, crt_ray* d_rays, void* stream);

/* ---- wavefront path tracing ------------------------------------ */
/* pathTrace (trace.h:399-532) + uniformSampleOneLight (:359-397) + estimateDirect (:221-357) +
 * the per-pixel sample loop of threadFunc (main.cpp:88-149), turned from "one path per call" into
 * "one bounce of every path per kernel": generate -> { extend (crt_intersect_batch_dev_counted) -> shade ->
 * shadow (crt_shadow_batch_dev_counted) -> accumulate } x (max_depth + 1) -> add to the colour buffer.
 * What it takes (anything else: CRT_E_UNSUPPORTED at create):
 *   geometry  every object type.  SMOOTH_TRIANGLE needs crt_mesh_shading.  Fields the reference's intersectors never write
 *             but pathTrace reads get the conventions listed at crt_fill_shade_rec (dpdu of triangles, planes, disks, boxes,
 *             cylinders, tori; the back of a disk; a torus hit outside its phi window);
 *   materials every MatType: MATTE (Lambertian, or Oren-Nayar when sigma != 0, materials.cpp:116-128), MIRROR
 *             (specular reflection, :129-136), TRANSPARENT (thin specular transmission, :137-141), PLASTIC (Fresnel-blend
 *             diffuse + specular lobes, :142-171), GLASS (rough dielectric, :172-178), METAL (:179-186), EMISSIVE; colours
 *             >= 0 and PLASTIC ks <= 1 (outside those bounds the reference paints paths BLUE / RED, trace.h:452-453,:478-479);
 *   textures  Matte.diffuse / Plastic.diffuse on FLAT_TRIANGLE / SMOOTH_TRIANGLE objects (interpolated texcoords) and on SPHERE /
 *             RECTANGLE objects (the uv their intersectors write, sphere.cpp:14-19, rect.cpp:28-37) (crt_texture below); no normal maps;
 *   lights    area lights on RECTANGLE / SPHERE / DISK objects with an EMISSIVE material
 *             (initAreaLights, buildscene.h:567-608) and one EnvLight, constant or a lat-long map, behind EnvLight.transform;
 *             EMISSIVE triangles of any kind emit when a path finds them and are never sampled (the reference's MeshLights
 *             carry no power, buildscene.h:879-917, and estimateDirect has no case for them, trace.h:311-313).
 * Random numbers: the reference draws from per-thread sample tables (Sampler_getSample) and from libc
 * rand() (trace.h:371,:516), both dependent on thread scheduling.  Here every number is
 * crt_rng(seed, pixel, sample, dimension) (definition in DESIGN.md "N1"), so an image is a pure
 * function of (scene, camera, seed, sample range).  Images therefore agree with the reference's in
 * distribution, not sample for sample. */
enum crt_mat_type { CRT_MAT_MATTE = 1, CRT_MAT_MIRROR = 2, CRT_MAT_TRANSPARENT = 4, CRT_MAT_EMISSIVE = 8,
                    CRT_MAT_PLASTIC = 16, CRT_MAT_GLASS = 32, CRT_MAT_METAL = 64 };                                /* MatType, materials.h:8-18 */
/* The lobes computeScatteringFunc (materials.cpp:111-188) builds from each, textures aside:
 *   MATTE        color = Matte.color, sigma: one Lambertian lobe, or one Oren-Nayar lobe when sigma != 0
 *   MIRROR       color = Mirror.color: one SpecularReflection lobe
 *   TRANSPARENT  ior_in, ior_out (the reference's `thin` SpecularTransmission lobe, reflection.cpp:243-269, ignores cf_in / cf_out)
 *   EMISSIVE     color, intensity (Emissive_s)
 *   PLASTIC      color = Plastic.kd, ks, uroughness = Plastic.roughness: FresnelBlendDiffuse + FresnelBlendSpecular (Beckmann, alpha = roughness)
 *   GLASS        ior_in, ior_out, uroughness, vroughness (Glass_s): one MicrofacetFresnel lobe, white, Beckmann,
 *                alpha = BeckmannRoughnessToAlpha(roughness) (microfacet.h:25-31)
 *   METAL        color = Metal.eta, k = Metal.k, uroughness, vroughness (Metal_s; procMetalType's table, materials.cpp:5-20, is the
 *                caller's): one MicrofacetReflection lobe with the conductor Fresnel term, Beckmann, alpha = roughness */
typedef struct crt_material { int32_t type; float color[3]; float intensity; float sigma; float ior_in; float ior_out;
                              float ks[3]; float k[3]; float uroughness; float vroughness; } crt_material;

/* Shading data of mesh triangles, per entry of crt_scene_input.tri_verts (optional).
 *   FLAT_TRIANGLE   normal = flat_normal[tri] (FlatTriangle.normal: normal_mat * mesh face normal, normalised;
 *                   buildscene.h:278-284).  Without it the normal is calcTriangleNormal of the baked vertices.
 *   SMOOTH_TRIANGLE vertex_normals / vertex_tangents [tri][9]: mesh->normals / mesh->tangents of the three
 *                   vertices, model space; normal_mats[normal_mat_index[tri]]: SmoothTriangle.normal_mat, row-major
 *                   (rayIntersectSmoothTriangle, triangle.cpp:218-258).  Required for SMOOTH_TRIANGLE objects. */
typedef struct crt_mesh_shading {
    const float*   flat_normal;
    const float*   vertex_normals;
    const float*   vertex_tangents;
    const int32_t* normal_mat_index;
    const float*   normal_mats;
    int32_t        num_normal_mats;
    const float*   vertex_uv;            /* [tri][6]: mesh->texcoords of the three vertices (interpTexcoord, triangle.cpp:170-183), or NULL
                                            (a mesh without texcoords: the reference interpolates nothing, triangle.cpp:191-194) */
} crt_mesh_shading;

/* Texture (texture.h:5-13): width x height texels of `comp` channels, rows top to bottom as stb_image / readRgba1 leave
 * them; data = unsigned char [h][w][comp], or float [h][w][comp] when is_float (the EXR environment maps,
 * buildscene.h:479-483).  getTexColor (texture.cpp:27-86) reads the NEAREST texel (its mirrored wrap and its rounding as
 * written) and divides by 255 -- float texels too. */
typedef struct crt_texture { int32_t width, height, comp, is_float; const void* data; } crt_texture;

typedef struct crt_render_desc {
    const crt_scene_input* objects;      /* the insertion-order arrays the scene was built from */
    const int32_t*      obj_material;    /* [num_obj] index into materials[] (getObjectMatPtr, shapes.cpp) */
    const crt_material* materials;
    int32_t             num_materials;
    int32_t             has_env_light;   /* sl->env_light != NULL (lights.h EnvLight, type CONSTANT); sampled as a light only when env_intensity > 0 (buildscene.h:525-533) */
    float               env_color[3];
    float               env_intensity;
    float               env_world_radius;/* EnvLight.world_radius as buildscene.h:836-872 computed it */
    int32_t             max_depth;       /* params.max_depth (config.txt) */
    uint32_t            samples_in_flight;/* samples per pixel traced concurrently; 0 = library picks (~2^25 paths, 150 bytes each) */
    uint64_t            seed;
    const crt_mesh_shading* mesh_shading;/* NULL: no FLAT normals given, SMOOTH_TRIANGLE objects refused */
    /* textures (all optional; zero-filled = none) */
    const crt_texture*  textures;
    int32_t             num_textures;
    const int32_t*      material_diffuse_tex; /* [num_materials] index into textures[] or -1: Matte.diffuse / Plastic.diffuse
                                                 (materials.cpp:51-60,:146-149); honoured on FLAT_TRIANGLE / SMOOTH_TRIANGLE objects with
                                                 mesh_shading->vertex_uv and on SPHERE / RECTANGLE objects; any other kind of object with
                                                 a textured material is refused */
    int32_t             env_map;         /* 1 + index into textures[] of EnvLight.env_map (type TEXTURE, lights.cpp:238-247); 0 = type CONSTANT */
    float               env_transform[9];/* EnvLight.transform, row-major (buildscene.h:515: mat3_rotate_y(-0.76) for a map); all zero = identity */
    /* the part of the frame this renderer renders: image rows [row_begin, row_end), 0 / 0 = all.  The colour buffer then holds
     * those rows only; pixel indices (camera rays, random numbers) stay those of the whole frame, so bands rendered apart are
     * bit for bit the rows of the whole frame (tile sharding: one band per GPU, all-gather). */
    int32_t             row_begin, row_end;
} crt_render_desc;

typedef struct crt_render_stats {
    uint64_t samples;                    /* samples per pixel accumulated so far */
    uint64_t rays_closest, rays_shadow;  /* rays handed to crt_intersect_batch_dev / crt_shadow_batch_dev */
    double   ms_total;                   /* wall time spent inside crt_renderer_render */
    double   ms_extend, ms_shadow;       /* traversal-kernel time of the two query kinds (HIP events) */
} crt_render_stats;

typedef struct crt_renderer crt_renderer;            /* opaque; owns device memory on the scene's GPU */

/* GPU the scene lives on; number of objects it was built from (hit ids are < this). */
int  crt_scene_device(const crt_scene* s, int* device);
int  crt_scene_num_objects(const crt_scene* s, int* num_obj);
/* The scene must have been built from desc->objects (hit ids index obj_material; a different object
 * count is refused with CRT_E_INVALID).  The renderer
 * keeps a pointer to `scene` (destroy the renderer first); nothing else host-side is retained. */
int  crt_renderer_create(crt_scene* scene, const crt_render_desc* desc, const crt_camera* cam, const crt_film* film,
                         crt_renderer** out);
void crt_renderer_destroy(crt_renderer* r);
/* Adds `num_samples` samples per pixel (sample indices continue from the previous call) to the
 * colour buffer; the loop `for p < params.num_samples` of main.cpp:295.  Returns when done. */
int  crt_renderer_render(crt_renderer* r, uint32_t num_samples);
/* color_buffer: [height*width*3] floats, the running SUM of per-sample radiance exactly as
 * main.cpp:139-141 keeps it (divide by *num_samples for the image, util.h genImageFromColorBuffer). */
int  crt_renderer_read(crt_renderer* r, float* color_buffer, uint32_t* num_samples);
/* Clears the colour buffer and sets the index of the next sample: GPU k of N renders the sample range
 * [k*S, (k+1)*S) of the same image, and the N buffers are summed (the reference's unit of work is the
 * sample pass, main.cpp:295; its image-state file plays this role between runs, imagestate.h). */
int  crt_renderer_reset(crt_renderer* r, uint32_t next_sample);
/* Same buffer copied to DEVICE memory of the renderer's GPU, enqueued on `stream` (all earlier
 * crt_renderer_render calls have completed): for reducing the buffers of several GPUs over RCCL. */
int  crt_renderer_read_dev(crt_renderer* r, float* d_color_buffer, void* stream);
int  crt_renderer_get_stats(crt_renderer* r, crt_render_stats* out);

/* ---- hit -> ShadeRec, host side (row a-13; the records are crt_shade_rec, above) ---------------------------- */
/* The ShadeRec each of n hits leaves behind: what `intersectTest(&sr, so, ray)` wrote into `sr` through the intersector
 * of the object that was hit -- rayIntersectTriangle / FlatTriangle / SmoothTriangle (triangle.cpp:154-258),
 * fillShadeRecSphere (sphere.cpp:3-31), rayIntersectPlane (plane.cpp:10-16), rayIntersectRect (rect.cpp:34-46),
 * rayIntersectDisk (disk.cpp:13-24), rayIntersectAABox / fillShadeRecGenericOpenCylinder / rayIntersectGenericTorus
 * (generic.cpp:3-32,:199-220,:384-401), rayIntersectInstanced (instanced.cpp:92-101: normal through the transposed inverse
 * transform, not renormalised), rayIntersectCompound (the winning part's record) -- rebuilt on the host from
 * {t, obj, part, beta, gamma}.  `in` = the insertion-order arrays the scene was built from (hit ids index them),
 * `mesh_shading` / `obj_material` may be NULL.  Plain host code, no GPU involved.
 * conventions = 0: exactly the reference's writes.  conventions != 0: additionally the fields pathTrace reads but the
 * reference never writes get this library's documented values (flag CRT_SR_BY_CONVENTION): dpdu of TRIANGLE /
 * FLAT_TRIANGLE = normalize(v1 - v0); dpdu of plane, disk, box, cylinder, torus = normalize(JITTERED_UP x normal); the
 * normal of a disk hit from behind = the disk's normal (disk.cpp:16-19 overwrites wo instead); a torus hit outside
 * the phi window (generic.cpp:207-220 returns t and writes nothing) filled as if inside.  These are what
 * crt_renderer_* shades with. */
int  crt_fill_shade_rec(const crt_scene_input* in, const crt_mesh_shading* mesh_shading, const int32_t* obj_material,
                        const crt_ray* rays, const crt_hit* hits, size_t n, int conventions, crt_shade_rec* out);

/* ---- mesh baking (part of SURVEY 8f N2), host side ----------------------------------- */
/* Mesh (mesh.h:11-22): the arrays generateMeshTriangles reads.  normals / tangents may be NULL. */
typedef struct crt_mesh {
    const float*   positions;            /* [num_vertices][3] */
    const float*   normals;              /* [num_vertices][3] */
    const float*   tangents;             /* [num_vertices][3] */
    const int32_t* indices;              /* [num_indices], three per triangle */
    int32_t        num_vertices, num_indices;
} crt_mesh;
/* MeshEntry (scene/scenedata.h): one OBJECT MESH entry of a scene file. orientation: Euler angles in degrees. */
typedef struct crt_mesh_entry { float scaling[3]; float orientation[3]; float location[3]; int32_t smooth; } crt_mesh_entry;
/* generateMeshTriangles (buildscene.h:214-311): num_indices/3 triangles, vertices scaled, rotated (eulerAngToMat3,
 * util/math.cpp:65-78), translated.  Outputs, each sized for num_indices/3 triangles, any but tri_verts may be NULL:
 * tri_verts [T][9] (crt_scene_input.tri_verts), flat_normal [T][3], vertex_normals / vertex_tangents [T][9]
 * (crt_mesh_shading), normal_mat [9] = rotation * inverse scaling, row-major. */
int  crt_bake_mesh(const crt_mesh* mesh, const crt_mesh_entry* entry, float* tri_verts, float* flat_normal,
                   float* vertex_normals, float* vertex_tangents, float* normal_mat);

/* OBJShape (objloader/objloader.h:33-43): one group of an OBJ file as an indexed mesh.  num_* count floats / ints,
 * as in the reference (num_positions = 3 * vertices). */
typedef struct crt_obj_shape {
    float*   positions; float* normals; float* texcoords; int32_t* indices;
    int32_t  num_positions, num_normals, num_texcoords, num_indices;
    char     mat_name[64];               /* the `usemtl` name in force (OBJ_NAME_LENGTH) */
} crt_obj_shape;
/* loadOBJ (objloader.h:738-939) without its MTL part: v / vn / vt / f records, fan triangulation, one vertex per
 * distinct (v, vn, vt) triple numbered in order of first use, a shape per `g` / `usemtl` / `o` section. */
int  crt_obj_load(const char* file_name, crt_obj_shape** shapes, int32_t* num_shapes);
void crt_obj_free(crt_obj_shape* shapes, int32_t num_shapes);
/* Per-vertex normals as calcTriangleNormals generates them when the shape has none (buildscene.h:58-98; normals_in
 * != NULL: copied) and per-vertex tangents of calcTangentVec (buildscene.h:101-212).  Outputs [num_vertices][3]. */
int  crt_mesh_normals_tangents(const float* positions, int32_t num_vertices, const int32_t* indices, int32_t num_indices,
                               const float* normals_in, float* normals_out, float* tangents_out);

/* ---- scene description files, host side ------------------------------ */
/* loadSceneFile + setupFilmAndCamera (buildscene.h:344-539) with the parse* functions of scene/scenefile.h:18-791:
 * film and camera header, MATERIAL entries of all seven types, OBJECT entries (SPHERE PLANE RECTANGLE TRIANGLE BOX
 * OPENCYLINDER DISK TORUS SOLIDCYLINDER MESH), ENV_LIGHT.  Objects come out in file order as a crt_scene_input (meshes
 * loaded by crt_obj_load and flattened by crt_bake_mesh), materials as crt_material records, obj_material[i] = index
 * of object i's material (-1: the name is not defined, the reference's NULL).  The older grammar of the scene files
 * the reference ships (DIFF_COLOR / AMB_COLOR materials, no film header) is read too; textures are bound by name and read
 * where they are PPM files (see the view's texture fields).
 * Everything the view points to lives until crt_scene_file_free. */
typedef struct crt_scene_file crt_scene_file;
typedef struct crt_scene_file_view {
    crt_scene_input     input;
    const int32_t*      obj_material;        /* [input.num_obj] */
    const crt_material* materials;
    const char* const*  material_names;      /* [num_materials] */
    int32_t             num_materials;
    int32_t             has_mesh_shading;    /* any OBJECT MESH entry: mesh_shading is indexed like input.tri_verts */
    crt_mesh_shading    mesh_shading;
    int32_t             has_film;            /* WINDOW_WIDTH .. FOV present */
    int32_t             window_width, window_height, image_width, image_height;
    float               fov;                 /* radians (degToRad of the file's value, buildscene.h:365) */
    int32_t             has_camera;
    float               camera_pos[3], look_point[3];
    int32_t             has_env_light;       /* an ENV_LIGHT entry; with intensity <= 0 it is no light and black (buildscene.h:525-533) */
    float               env_color[3], env_intensity;
    int32_t             num_skipped;         /* object entries dropped: unknown kind, no MATERIAL field, unreadable mesh file */
    int32_t             num_unsupported;     /* unknown material / metal types (kept as placeholders), texture files that could not be read */
    /* textures (TEXTURE / NORMAL_MAP of MATTE, scenefile.h:141-147,:161-171; KD_TEXTURE of PLASTIC, :225-233; ENV_LIGHT TYPE
     * TEXTURE, buildscene.h:466-515), in order of first mention, one entry per file name (findTexture, scenedata.h:99-112,
     * returns the first of that name).  Files are read with crt_texture_load_ppm, by the name given and then relative to
     * the scene file; textures[i].data is NULL where that failed (counted in num_unsupported) and for the environment map,
     * which the reference reads only from OpenEXR files (buildscene.h:478-483): its float texels are the caller's to supply
     * (crt_render_desc.textures is the caller's array), counted in num_external. */
    int32_t             num_textures;
    const char* const*  texture_names;       /* [num_textures] */
    const crt_texture*  textures;            /* [num_textures] */
    const int32_t*      material_diffuse_tex;/* [num_materials] Matte.diffuse / Plastic.diffuse (linkMaterialTextures, buildscene.h:315-342) or -1 */
    const int32_t*      material_normal_tex; /* [num_materials] Matte.normal_file_name or -1; the reference never applies it (buildscene.h:329) */
    int32_t             env_type;            /* 0 CONSTANT, 1 TEXTURE (an .exr name; any other extension: CONSTANT and WHITE, buildscene.h:484-488) */
    int32_t             env_map_tex;         /* index into textures[] of the map an ENV_LIGHT TYPE TEXTURE entry names, or -1 */
    float               env_transform[9];    /* mat3_rotate_y(-0.76) for a TYPE TEXTURE entry (buildscene.h:515), else zero (= identity) */
    int32_t             num_external;        /* named textures whose texels the caller supplies (the .exr environment map) */
} crt_scene_file_view;
int  crt_scene_file_load(const char* path, crt_scene_file** out);
const crt_scene_file_view* crt_scene_file_get(const crt_scene_file* sf);
void crt_scene_file_free(crt_scene_file* sf);
/* EnvLight.world_radius as preprocessLights derives it from the built accelerator (buildscene.h:839-873): twice the
 * diagonal of the BVH root / grid box, and for the QBVH the expression written there.  NONE: CRT_E_UNSUPPORTED (the
 * reference leaves the field unset). */
int  crt_env_world_radius(const crt_scene_desc* desc, float* radius);

/* ---- image state and 8-bit output, host side ---------------------- */
/* saveImageState / readImageState (imagestate.h:5-43): "<num_samples>\n<width> <height>\n" followed by
 * width*height*3 raw float32 sums -- the file the reference resumes from (main.cpp:238-253) and the
 * operand of the multi-GPU sum.  Files are interchangeable with the reference's.  load allocates
 * *color_buffer (release with crt_image_state_free). */
int  crt_image_state_save(const char* file_name, const float* color_buffer, int num_samples, int width, int height);
int  crt_image_state_load(const char* file_name, float** color_buffer, int* num_samples, int* width, int* height);
void crt_image_state_free(float* color_buffer);
/* genImageFromColorBuffer (util/util.cpp:28-45) with toneMap (shading.h:33-49): image[3*num_pixels]. */
int  crt_image_from_color_buffer(unsigned char* image, const float* color_buffer, int num_pixels, int num_samples);
/* PPM_write (imagefile.h:36-57): binary P6. */
int  crt_ppm_write(const char* file_name, const unsigned char* image, int width, int height);
/* PPM_read (imagefile.h:59-83), the one image reader the reference holds that needs no library it does not ship: binary P6,
 * maxval 255, into a three-channel 8-bit crt_texture (rows top to bottom, as loadTexture leaves them).  out->data is malloc'd:
 * release with crt_texture_free.  crt_scene_file_load reads the textures a scene file names through it. */
int  crt_texture_load_ppm(const char* file_name, crt_texture* out);
void crt_texture_free(crt_texture* t);

/* Replaces intersect_call_count / shadow_intersect_call_count (intersect.h:363-364)
 * and g_intersect_count (shapes.cpp:3).  Counting is off by default (it costs atomics);
 * when on, kernels add per-ray node / primitive-test counts. */
typedef struct crt_counters {
    uint64_t rays_closest, rays_shadow;
    uint64_t nodes_visited;       /* BVH/BVH4 nodes whose boxes were tested; GRID: cells walked */
    uint64_t prims_tested;        /* rayIntersectObject / shadowRayIntersectObject calls */
} crt_counters;
int  crt_scene_enable_counters(crt_scene* s, int on);
int  crt_scene_get_counters(crt_scene* s, crt_counters* out);   /* syncs the device */
int  crt_scene_reset_counters(crt_scene* s);

/* Duration in milliseconds of the traversal kernel of the CALLING THREAD's most recent launch through this
 * scene, measured with HIP events on the stream it was launched on (syncs on that launch).  Each launch has
 * its own events (a ring of 16 per scene), so concurrent callers read their own timings; CRT_E_INVALID when
 * the thread has not launched through `s` or 16 later launches have reused the slot. */
int  crt_scene_last_kernel_ms(crt_scene* s, float* ms);
/* Duration of the ray-binning kernels that preceded it (0 when the launch did not bin). */
int  crt_scene_last_binning_ms(crt_scene* s, float* ms);

const char* crt_strerror(int code);
const char* crt_last_hip_error(void);               /* text of the last failing HIP call */

#ifdef __cplusplus
}
#endif
#endif /* CRT_GPU_H */
