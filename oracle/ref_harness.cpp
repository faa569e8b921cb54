/*
 * ref_harness.cpp -- builds oracle/_ref/libcrt_ref.so around the REFERENCE's own code.
 *
 * TEST INFRASTRUCTURE ONLY, usable only in the build container (it compiles sources
 * where they lie under /root/reference; nothing of the reference is copied here).
 * This file contains no reference code: it constructs scenes through the reference's
 * structs (Triangle, Sphere, Rectangle, ..., InstancedShape via initBox/initTorus/
 * initOpenCylinder/initSolidCylinder), runs the reference's builders
 * (UniformGrid_create, BVH_build, BVH4_build), traversals (BVH4IntersectTest,
 * BVH4ShadowIntersectTest) and intersectors (rayIntersectObject, shadowRayIntersectObject,
 * rayIntersectAABB), and flattens what they produce into the crt_* layouts so the
 * restatement in crt_oracle.cpp can be compared with it bit for bit.
 *
 * NOT reachable from here: intersect.h (includes <GLFW/glfw3.h>, absent from the image,
 * and no stand-in is written) -> gridIntersectTest, BVHIntersectTest, intersectTest and
 * shadowIntersectTest themselves.  The few lines of intersectTest's BVH4/NONE branches
 * (intersect.h:397-432, :470-498) are re-expressed below as harness code around the
 * reference's BVH4*Test / *IntersectObject functions; this is noted in DESIGN.md.
 *
 * Object ids: rayIntersectObject / shadowRayIntersectObject calls made from inside the
 * reference's traversal are routed through a hook (macro rename applied only to the
 * accelerator headers) that logs (object pointer, t).  Every accept test in the reference
 * is a strict `<`, so the winner is the first logged call whose t equals the returned t.
 */
#include "../include/crt_gpu.h"

#include "shapes/shapes.h"
#include "shapes/instanced.h"
#include "aabb.h"
#include "util/math.h"
#include "camera.h"
#include "sampling.h"
#include "reflection.h"
#include "mesh.h"
#include "objloader/objloader.h"
#include "util/mat.h"
#include "util/util.h"
#include "imagestate.h"

#include <vector>
#include <unordered_map>

struct RefLogEntry { void* ptr; float t; };
static thread_local std::vector<RefLogEntry>* g_log = 0;

static float hook_rayIntersectObject(ShadeRec* sr, const Object_t obj, const Ray ray)
{
    float t = rayIntersectObject(sr, obj, ray);
    if (g_log) g_log->push_back(RefLogEntry{ obj.ptr, t });
    return t;
}
static float hook_shadowRayIntersectObject(const Object_t obj, const Ray ray)
{
    float t = shadowRayIntersectObject(obj, ray);
    if (g_log) g_log->push_back(RefLogEntry{ obj.ptr, t });
    return t;
}

#define rayIntersectObject hook_rayIntersectObject
#define shadowRayIntersectObject hook_shadowRayIntersectObject
#include "accelerator/uniformgrid.h"
#include "accelerator/bvh4.h"
#undef rayIntersectObject
#undef shadowRayIntersectObject

#define REF_API extern "C" __attribute__((visibility("default")))

struct ref_scene {
    int accel;
    int num_obj, num_non_grid;
    Object_t* objects;
    std::unordered_map<void*, int> id_of;
    BVHNode4* tree4;
    BVHNode* tree2;
    UniformGrid* grid;
    Material* mat;
    Mesh* mesh;
};

static void* make_shape(const crt_scene_input* in, int si, int* type_out, Material* mat);

static Object_t make_object(const crt_scene_input* in, int si, Material* mat)
{
    Object_t o;
    int type;
    o.ptr = make_shape(in, si, &type, mat);
    o.type = (ObjectType)type;
    return o;
}

static void* make_shape(const crt_scene_input* in, int si, int* type_out, Material* mat)
{
    const crt_shape& sh = in->shapes[si];
    const float* p = sh.p;
    *type_out = sh.type;
    switch (sh.type) {
    case CRT_SPHERE: {
        Sphere* s = (Sphere*)calloc(1, sizeof(Sphere));
        vec3_assign(s->center, p[0], p[1], p[2]);
        s->radius = p[3]; s->min_theta = p[4]; s->max_theta = p[5]; s->phi = p[6]; s->mat = mat;
        return s;
    }
    case CRT_PLANE: {
        Plane* s = (Plane*)calloc(1, sizeof(Plane));
        vec3_assign(s->point, p[0], p[1], p[2]);
        vec3_assign(s->normal, p[3], p[4], p[5]);
        s->mat = mat;
        return s;
    }
    case CRT_RECTANGLE: {
        Rectangle* s = (Rectangle*)calloc(1, sizeof(Rectangle));
        vec3_assign(s->point, p[0], p[1], p[2]);
        vec3_assign(s->width, p[3], p[4], p[5]);
        vec3_assign(s->height, p[6], p[7], p[8]);
        vec3_assign(s->normal, p[9], p[10], p[11]);
        s->mat = mat;
        return s;
    }
    case CRT_AABOX: {
        AABox* s = (AABox*)calloc(1, sizeof(AABox));
        vec3_assign(s->min, p[0], p[1], p[2]);
        vec3_assign(s->max, p[3], p[4], p[5]);
        s->shadow = (p[6] != 0.0f);
        s->mat = mat;
        return s;
    }
    case CRT_GENERICOPENCYLINDER: {
        GenericOpenCylinder* s = (GenericOpenCylinder*)calloc(1, sizeof(GenericOpenCylinder));
        s->half_height = p[0]; s->radius = p[1]; s->phi = p[2]; s->normal_type = (NormalType)(int)p[3]; s->mat = mat;
        return s;
    }
    case CRT_DISK: {
        Disk* s = (Disk*)calloc(1, sizeof(Disk));
        vec3_assign(s->center, p[0], p[1], p[2]);
        vec3_assign(s->normal, p[3], p[4], p[5]);
        s->radius = p[6]; s->mat = mat;
        return s;
    }
    case CRT_GENERICTORUS: {
        GenericTorus* s = (GenericTorus*)calloc(1, sizeof(GenericTorus));
        s->swept_radius = p[0]; s->tube_radius = p[1]; s->phi = p[2]; s->mat = mat;
        return s;
    }
    case CRT_COMPOUND: {
        CompoundObject* c = (CompoundObject*)calloc(1, sizeof(CompoundObject));
        c->num_obj = sh.num_parts;
        c->objects = (Object_t*)calloc(sh.num_parts, sizeof(Object_t));
        for (int i = 0; i < sh.num_parts; i++) c->objects[i] = make_object(in, sh.first_part + i, mat);
        vec3_assign(c->aabb.min, p[0], p[1], p[2]);
        vec3_assign(c->aabb.max, p[3], p[4], p[5]);
        return c;
    }
    case CRT_INSTANCED: {
        InstancedShape* is = (InstancedShape*)calloc(1, sizeof(InstancedShape));
        is->obj = make_object(in, sh.inner, mat);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) is->inv_transform[r][c] = sh.inv_xform[4*r + c];
        is->mat = mat;
        return is;
    }
    }
    return 0;
}

REF_API ref_scene* ref_scene_create(const crt_scene_input* in, int accel)
{
    ref_scene* s = new ref_scene();
    s->accel = accel;
    s->tree4 = 0; s->tree2 = 0; s->grid = 0;
    s->mat = (Material*)calloc(1, sizeof(Material));
    s->mesh = (Mesh*)calloc(1, sizeof(Mesh));
    s->num_obj = in->num_obj;
    s->objects = (Object_t*)calloc(in->num_obj > 0 ? in->num_obj : 1, sizeof(Object_t));
    for (int i = 0; i < in->num_obj; i++) {
        int type = in->obj_type ? in->obj_type[i] : (int)CRT_TRIANGLE;
        int ref = in->obj_type ? in->obj_ref[i] : i;
        Object_t o;
        if (type == CRT_TRIANGLE || type == CRT_SMOOTH_TRIANGLE || type == CRT_FLAT_TRIANGLE) {
            /* All three triangle kinds share calcTriangleIntersect (triangle.cpp:14-77) for the hit
             * arithmetic.  SmoothTriangle's wrapper dereferences mesh normals/tangents
             * (triangle.cpp:224-251) and FlatTriangle's wraps an UNINITIALISED sr->uv with
             * `while(uv > 1) uv -= 1` (triangle.cpp:195-198), which never ends for a large garbage
             * value; both are shading-side effects, so every triangle is built as a plain Triangle. */
            Triangle* t = (Triangle*)calloc(1, sizeof(Triangle));
            const float* v = in->tri_verts + (size_t)ref * 9;
            vec3_assign(t->v0, v[0], v[1], v[2]);
            vec3_assign(t->v1, v[3], v[4], v[5]);
            vec3_assign(t->v2, v[6], v[7], v[8]);
            calcTriangleNormal(t->normal, t->v0, t->v1, t->v2);
            t->mat = s->mat;
            o.type = TRIANGLE; o.ptr = t;
        } else {
            o = make_object(in, ref, s->mat);
        }
        s->objects[i] = o;
        s->id_of[o.ptr] = i;
    }
    /* mvNonGridObjToStart (scene/scenedata.h:368-382) lives in a header that pulls in the
     * OBJ loader and texture code; its 10 lines are re-expressed here around the reference's
     * isGridObjType. */
    int j = 0;
    for (int i = 0; i < s->num_obj; ++i) {
        if (!isGridObjType(s->objects[i])) {
            Object_t tmp = s->objects[i]; s->objects[i] = s->objects[j]; s->objects[j] = tmp; ++j;
        }
    }
    s->num_non_grid = j;
    int n_accel = s->num_obj - j;
    if (accel == CRT_ACCEL_GRID) {
        int n = s->num_obj;
        s->grid = UniformGrid_create(s->objects, &n, s->num_non_grid, 3);
    } else if (accel == CRT_ACCEL_BVH && n_accel > 0) {
        BVH_build(&s->tree2, &(s->objects[j]), n_accel);
    } else if (accel == CRT_ACCEL_BVH4 && n_accel > 0) {
        BVH4_build(&s->tree4, &(s->objects[j]), n_accel);
    }
    return s;
}

REF_API int ref_scene_order(const ref_scene* s, int32_t* ids, int32_t* num_non_grid)
{
    if (ids) for (int i = 0; i < s->num_obj; i++) ids[i] = s->id_of.at(s->objects[i].ptr);
    if (num_non_grid) *num_non_grid = s->num_non_grid;
    return s->num_obj;
}

static int flatten4(const ref_scene* s, const BVHNode4* n, std::vector<crt_bvh4_node>* out)
{
    int self = (int)out->size();
    out->push_back(crt_bvh4_node());
    crt_bvh4_node nd;
    memset(&nd, 0, sizeof nd);
    memcpy(nd.bbox, n->bbox, sizeof(float) * 24);
    nd.axis = n->axis;
    for (int c = 0; c < 4; c++) {
        nd.type[c] = n->type[c];
        if (n->type[c] == -1) nd.child[c] = flatten4(s, (const BVHNode4*)n->child[c], out);
        else nd.child[c] = (int)((const Object_t*)n->child[c] - s->objects);
    }
    (*out)[self] = nd;
    return self;
}
REF_API int ref_scene_nodes4(const ref_scene* s, crt_bvh4_node* out)
{
    std::vector<crt_bvh4_node> v;
    if (s->tree4) flatten4(s, s->tree4, &v);
    if (out) memcpy(out, v.data(), v.size() * sizeof(crt_bvh4_node));
    return (int)v.size();
}

static int flatten2(const ref_scene* s, const BVHNode* n, std::vector<crt_bvh2_node>* out)
{
    int self = (int)out->size();
    out->push_back(crt_bvh2_node());
    crt_bvh2_node nd;
    for (int k = 0; k < 3; k++) { nd.bmin[k] = n->aabb.min[k]; nd.bmax[k] = n->aabb.max[k]; }
    if (n->type == LEAF) {
        nd.left = -1;
        int idx = -1;
        for (int i = 0; i < s->num_obj; i++) if (s->objects[i].ptr == n->obj.ptr) { idx = i; break; }
        nd.right = idx;
    } else {
        nd.left = flatten2(s, n->left, out);
        nd.right = flatten2(s, n->right, out);
    }
    (*out)[self] = nd;
    return self;
}
REF_API int ref_scene_nodes2(const ref_scene* s, crt_bvh2_node* out)
{
    std::vector<crt_bvh2_node> v;
    if (s->tree2) flatten2(s, s->tree2, &v);
    if (out) memcpy(out, v.data(), v.size() * sizeof(crt_bvh2_node));
    return (int)v.size();
}

REF_API int ref_scene_grid(const ref_scene* s, crt_grid* g, int32_t* cell_start, int32_t* cell_items)
{
    const UniformGrid* ug = s->grid;
    if (!ug) return -1;
    if (g) {
        g->nx = ug->nx; g->ny = ug->ny; g->nz = ug->nz; g->num_cells = ug->num_cells;
        g->wx = ug->wx; g->wy = ug->wy; g->wz = ug->wz;
        for (int k = 0; k < 3; k++) { g->bmin[k] = ug->aabb.min[k]; g->bmax[k] = ug->aabb.max[k]; }
    }
    int total = 0;
    for (int c = 0; c < ug->num_cells; c++) {
        if (cell_start) cell_start[c] = total;
        for (int i = 0; i < ug->cells[c].size; i++) { if (cell_items) cell_items[total] = ug->cells[c].array[i]; total++; }
    }
    if (cell_start) cell_start[ug->num_cells] = total;
    return total;
}

static void to_ray(Ray* r, const crt_ray* in)
{
    vec3_assign(r->origin, in->o[0], in->o[1], in->o[2]);
    vec3_assign(r->direction, in->d[0], in->d[1], in->d[2]);
}

/* part index + barycentrics of the winning object, recomputed with the reference's functions */
static void fill_details(const ref_scene* s, void* ptr, const Ray ray, crt_hit* out)
{
    out->part = 0; out->beta = 0; out->gamma = 0;
    Object_t obj; obj.ptr = 0; obj.type = SPHERE;
    for (int i = 0; i < s->num_obj; i++) if (s->objects[i].ptr == ptr) { obj = s->objects[i]; break; }
    if (obj.type == TRIANGLE) {
        Triangle* t = (Triangle*)obj.ptr;
        calcTriangleIntersect(&out->beta, &out->gamma, t->v0, t->v1, t->v2, ray);
    } else if (obj.type == INSTANCED) {
        InstancedShape* is = (InstancedShape*)obj.ptr;
        if (is->obj.type == COMPOUND) {
            Ray r2;
            transformRay(&r2, is->inv_transform, ray);
            CompoundObject* co = (CompoundObject*)is->obj.ptr;
            float min_t = TMAX;
            for (int i = 0; i < co->num_obj; i++) {
                ShadeRec tmp;
                float t = rayIntersectObject(&tmp, co->objects[i], r2);
                if (t < min_t) { min_t = t; out->part = i; }
            }
        }
    } else if (obj.type == COMPOUND) {                             /* a compound that is not inside an instance */
        CompoundObject* co = (CompoundObject*)obj.ptr;
        float min_t = TMAX;
        for (int i = 0; i < co->num_obj; i++) {
            ShadeRec tmp;
            float t = rayIntersectObject(&tmp, co->objects[i], ray);
            if (t < min_t) { min_t = t; out->part = i; }
        }
    }
}

/* n x { BVH4 / NONE branch of intersectTest (intersect.h:397-432) } on the reference's functions */
REF_API int ref_intersect(const ref_scene* s, const crt_ray* rays, size_t n, crt_hit* hits)
{
    if (s->accel != CRT_ACCEL_BVH4 && s->accel != CRT_ACCEL_NONE) return CRT_E_UNSUPPORTED;
    std::vector<RefLogEntry> log;
    for (size_t i = 0; i < n; i++) {
        Ray ray; to_ray(&ray, &rays[i]);
        ShadeRec sr, tmp_sr;
        float min_t = TMAX;
        log.clear();
        g_log = &log;
        if (s->accel == CRT_ACCEL_BVH4) {
            if (s->tree4) min_t = BVH4IntersectTest(&sr, s->tree4, ray);
            for (int k = 0; k < s->num_non_grid; k++) {
                float tmp_t = hook_rayIntersectObject(&tmp_sr, s->objects[k], ray);
                if (tmp_t < min_t) min_t = tmp_t;
            }
        } else {
            for (int k = 0; k < s->num_obj; k++) {
                float tmp_t = hook_rayIntersectObject(&tmp_sr, s->objects[k], ray);
                if (tmp_t < min_t) min_t = tmp_t;
            }
        }
        g_log = 0;
        crt_hit h; h.t = min_t; h.obj = -1; h.part = -1; h.beta = 0; h.gamma = 0;
        if (min_t < TMAX) {
            for (size_t k = 0; k < log.size(); k++) if (log[k].t == min_t) {
                h.obj = s->id_of.at(log[k].ptr);
                fill_details(s, log[k].ptr, ray, &h);
                break;
            }
        }
        hits[i] = h;
    }
    return 0;
}

/* n x { BVH4 / NONE branch of shadowIntersectTest (intersect.h:470-498) } */
REF_API int ref_shadow(const ref_scene* s, const crt_ray* rays, const float* light_dist, size_t n,
                       float* t_out, uint8_t* occluded_out)
{
    if (s->accel != CRT_ACCEL_BVH4 && s->accel != CRT_ACCEL_NONE) return CRT_E_UNSUPPORTED;
    for (size_t i = 0; i < n; i++) {
        Ray ray; to_ray(&ray, &rays[i]);
        float ld = light_dist[i];
        float t = TMAX;
        if (s->accel == CRT_ACCEL_BVH4) {
            if (s->tree4) t = BVH4ShadowIntersectTest(s->tree4, ray, ld);
            if (t >= ld) {
                for (int k = 0; k < s->num_non_grid; k++) {
                    t = shadowRayIntersectObject(s->objects[k], ray);
                    if (t < ld) break;
                }
            }
        } else {
            for (int k = 0; k < s->num_obj; k++) {
                t = shadowRayIntersectObject(s->objects[k], ray);
                if (t < ld) break;
            }
        }
        if (t_out) t_out[i] = t;
        if (occluded_out) occluded_out[i] = (t < ld - K_EPSILON) ? 1 : 0;
    }
    return 0;
}

REF_API float ref_intersect_object(const ref_scene* s, int final_index, const crt_ray* ray, crt_hit* hit)
{
    Ray r; to_ray(&r, ray);
    ShadeRec sr;
    float t = rayIntersectObject(&sr, s->objects[final_index], r);
    if (hit) {
        hit->t = t; hit->obj = s->id_of.at(s->objects[final_index].ptr);
        fill_details(s, s->objects[final_index].ptr, r, hit);
    }
    return t;
}
REF_API float ref_shadow_object(const ref_scene* s, int final_index, const crt_ray* ray)
{
    Ray r; to_ray(&r, ray);
    return shadowRayIntersectObject(s->objects[final_index], r);
}
REF_API float ref_ray_aabb(const float bmin[3], const float bmax[3], const crt_ray* ray)
{
    Ray r; to_ray(&r, ray);
    AABB b;
    vec3_assign(b.min, bmin[0], bmin[1], bmin[2]);
    vec3_assign(b.max, bmax[0], bmax[1], bmax[2]);
    return rayIntersectAABB(&b, r);
}
/* rayIntersectAABB4 (bvh4.h:23-97) on one node's 24 floats */
REF_API void ref_ray_aabb4(const float bbox[24], const crt_ray* ray, float t_out[4], int32_t less_than_out[4])
{
    Ray r; to_ray(&r, ray);
    CACHE_ALIGN float bb[24];
    memcpy(bb, bbox, sizeof bb);
    CACHE_ALIGN float res[4];
    CACHE_ALIGN float lt[4];
    __m128 tmp = rayIntersectAABB4((__m128*)lt, bb, r);
    _mm_store_ps(res, tmp);
    for (int i = 0; i < 4; i++) { t_out[i] = res[i]; less_than_out[i] = lt[i] ? 1 : 0; }
}
/* ---- primary rays: the reference's own camera.cpp ---------------------------------------- */
static void cam_to_abi(crt_camera* o, const Camera& c)
{
    o->type = (int)c.camera_type; o->focal_pt_dist = c.focal_pt_dist; o->focal_length = c.focal_length; o->lens_radius = c.lens_radius;
    for (int k = 0; k < 3; k++) { o->position[k] = c.position[k]; o->x_axis[k] = c.x_axis[k]; o->y_axis[k] = c.y_axis[k];
                                  o->z_axis[k] = c.z_axis[k]; o->focal_point[k] = c.focal_point[k]; }
}
static void cam_from_abi(Camera* c, const crt_camera* o)
{
    c->camera_type = (CameraType)o->type; c->focal_pt_dist = o->focal_pt_dist; c->focal_length = o->focal_length; c->lens_radius = o->lens_radius;
    for (int k = 0; k < 3; k++) { c->position[k] = o->position[k]; c->x_axis[k] = o->x_axis[k]; c->y_axis[k] = o->y_axis[k];
                                  c->z_axis[k] = o->z_axis[k]; c->focal_point[k] = o->focal_point[k]; }
    c->samples = NULL;
}
/* init (pinhole or thin lens) -> optional setFocalDist -> cameraLookAt, all by the reference's functions */
REF_API void ref_camera_setup(crt_camera* out, int type, float focal_length, float lens_radius, float focal_pt_dist,
                              const float* position, const float* look_point, const float* up)
{
    Camera c;
    if (type == 1) initThinLensCameraDefault(&c, focal_length, lens_radius, 0, 0);
    else initPinholeCameraDefault(&c);
    if (focal_pt_dist > 0.0f) setFocalDist(&c, focal_pt_dist);
    vec3 p = { position[0], position[1], position[2] }, l = { look_point[0], look_point[1], look_point[2] }, u = { up[0], up[1], up[2] };
    cameraLookAt(&c, p, l, u);
    cam_to_abi(out, c);
    Camera_destroy(&c);
}
REF_API void ref_film_dimension(crt_film* out, float fov, int w, int h, const crt_camera* cam)
{
    Camera c; cam_from_abi(&c, cam);
    Film f; memset(&f, 0, sizeof f);
    f.fov = fov; f.frame_res_width = w; f.frame_res_height = h;
    calcFilmDimension(&f, &c);
    out->fov = f.fov; out->frame_length = f.frame_length; out->frame_height = f.frame_height; out->pixel_length = f.pixel_length;
    out->frame_res_width = w; out->frame_res_height = h;
}
REF_API void ref_primary_rays(const crt_camera* cam, const crt_film* film, unsigned first_pixel, size_t n,
                              const float* image_samples, const float* lens_samples, crt_ray* rays)
{
    Camera c; cam_from_abi(&c, cam);
    Film f; memset(&f, 0, sizeof f);
    f.fov = film->fov; f.frame_length = film->frame_length; f.frame_height = film->frame_height; f.pixel_length = film->pixel_length;
    f.frame_res_width = film->frame_res_width; f.frame_res_height = film->frame_res_height;
    for (size_t i = 0; i < n; i++) {
        vec2 is = { image_samples ? image_samples[2*i] : 0.5f, image_samples ? image_samples[2*i+1] : 0.5f };
        vec2 ls = { lens_samples ? lens_samples[2*i] : 0.5f, lens_samples ? lens_samples[2*i+1] : 0.5f };
        vec2 ic;
        calcImageCoord(ic, &f, is, (int)(first_pixel + i));          /* main.cpp:108 */
        Ray ray;
        calcCameraRay(&ray, ic, &c, ls);                              /* main.cpp:113 */
        memcpy(rays + i, &ray, sizeof(crt_ray));
    }
}

REF_API int ref_sizes(int* shaderec, int* node4, int* node2, int* object_t)
{
    *shaderec = (int)sizeof(ShadeRec); *node4 = (int)sizeof(BVHNode4); *node2 = (int)sizeof(BVHNode); *object_t = (int)sizeof(Object_t);
    return 0;
}

/* ---- pieces of the shading path that compile without trace.h ---------------- */
/* ShadeRec as the reference's intersector leaves it: out = hit_point, normal, wo, dpdu (12 floats).
 * dpdu is pre-set to a marker so that intersectors that never write it (triangle.cpp:154-204) show. */
REF_API float ref_shade_rec(const ref_scene* s, int final_index, const crt_ray* ray, float* out, float dpdu_marker)
{
    Ray r; to_ray(&r, ray);
    ShadeRec sr;
    vec3_assign(sr.dpdu, dpdu_marker, dpdu_marker, dpdu_marker);
    vec3_assign(sr.normal, dpdu_marker, dpdu_marker, dpdu_marker);
    float t = rayIntersectObject(&sr, s->objects[final_index], r);
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.wo, 12); memcpy(out + 9, sr.dpdu, 12);
    return t;
}
/* Every geometric field of the record: out = hit_point, normal, dpdu, dpdv, wo (3 each), uv (2) = 17 floats, all pre-set to
 * `marker` so that what the intersector of that shape kind never writes shows (row a-13: crt_fill_shade_rec's `written`). */
REF_API float ref_shade_rec_full(const ref_scene* s, int final_index, const crt_ray* ray, float* out, float marker)
{
    Ray r; to_ray(&r, ray);
    ShadeRec sr;
    vec3_assign(sr.hit_point, marker, marker, marker); vec3_assign(sr.normal, marker, marker, marker);
    vec3_assign(sr.dpdu, marker, marker, marker); vec3_assign(sr.dpdv, marker, marker, marker);
    vec3_assign(sr.wo, marker, marker, marker); vec2_assign(sr.uv, marker, marker);
    float t = rayIntersectObject(&sr, s->objects[final_index], r);
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.dpdu, 12); memcpy(out + 9, sr.dpdv, 12);
    memcpy(out + 12, sr.wo, 12); memcpy(out + 15, sr.uv, 8);
    return t;
}
REF_API void ref_map_sample_to_disk(const float* in, float* out) { vec2 i = { in[0], in[1] }; vec2 o; mapSampleToDisk(o, i); out[0] = o[0]; out[1] = o[1]; }
REF_API void ref_map_sample_to_hemisphere(const float* in, float* out) { vec2 i = { in[0], in[1] }; vec3 o; mapSampleToHemisphere(o, i); memcpy(out, o, 12); }
REF_API void ref_vec3_in_local_basis(const float* a, const float* normal, float* out)
{
    vec3 aa = { a[0], a[1], a[2] }, nn = { normal[0], normal[1], normal[2] }, r;
    getVec3InLocalBasis(r, aa, nn);
    memcpy(out, r, 12);
}
/* BSDF with one Lambertian lobe (BSDF_addLambertian): BSDF_sample_f and BSDF_f.  basis = tangent,
 * binormal, normal (9 floats).  Returns the pdf; f_out, wi_out from BSDF_sample_f; f_eval from BSDF_f(wi_out). */
REF_API float ref_lambertian_bsdf(const float* cd, const float* basis, const float* wo, const float* sample,
                                  float* f_out, float* wi_out, float* f_eval)
{
    static bool pool = initBSDFMem(1, 4);
    (void)pool;
    BSDF bsdf;
    bsdf.num_bxdf = 0;
    memcpy(bsdf.tangent, basis, 12); memcpy(bsdf.binormal, basis + 3, 12); memcpy(bsdf.normal, basis + 6, 12);
    vec3 c = { cd[0], cd[1], cd[2] };
    BSDF_addLambertian(&bsdf, c);
    vec3 f = { 0, 0, 0 }, wi = { 0, 0, 0 }, w = { wo[0], wo[1], wo[2] };
    vec2 smp = { sample[0], sample[1] };
    BxDFFlags flags;
    float pdf = BSDF_sample_f(f, wi, &flags, w, smp, &bsdf);
    memcpy(f_out, f, 12); memcpy(wi_out, wi, 12);
    vec3 fe = { 0, 0, 0 };
    BSDF_f(fe, wi, w, &bsdf, (BxDFFlags)(BSDF_SPECULAR | BSDF_GLOSSY));
    memcpy(f_eval, fe, 12);
    BSDF_freeBxDFs(&bsdf);
    return pdf;
}

/* BSDF with one SpecularReflection lobe (BSDF_addSpecularReflection, what computeScatteringFunc builds for MIRROR) */
REF_API float ref_mirror_bsdf(const float* cr, const float* basis, const float* wo, float* f_out, float* wi_out, int* flags_out)
{
    static bool pool = initBSDFMem(1, 4);
    (void)pool;
    BSDF bsdf;
    bsdf.num_bxdf = 0;
    memcpy(bsdf.tangent, basis, 12); memcpy(bsdf.binormal, basis + 3, 12); memcpy(bsdf.normal, basis + 6, 12);
    vec3 c = { cr[0], cr[1], cr[2] };
    BSDF_addSpecularReflection(&bsdf, c);
    vec3 f = { 0, 0, 0 }, wi = { 0, 0, 0 }, w = { wo[0], wo[1], wo[2] };
    vec2 smp = { 0.5f, 0.5f };
    BxDFFlags flags = BSDF_NONE;
    float pdf = BSDF_sample_f(f, wi, &flags, w, smp, &bsdf);
    memcpy(f_out, f, 12); memcpy(wi_out, wi, 12);
    *flags_out = (int)flags;
    BSDF_freeBxDFs(&bsdf);
    return pdf;
}

/* BSDF with one OrenNayar lobe (BSDF_addOrenNayar, MATTE with sigma != 0): BSDF_sample_f, and BSDF_f for (wi_eval, wo) */
REF_API float ref_orennayar_bsdf(const float* rr, float sigma, const float* basis, const float* wo, const float* sample,
                                 const float* wi_eval, float* f_out, float* wi_out, float* f_eval)
{
    static bool pool = initBSDFMem(1, 4);
    (void)pool;
    BSDF bsdf;
    bsdf.num_bxdf = 0;
    memcpy(bsdf.tangent, basis, 12); memcpy(bsdf.binormal, basis + 3, 12); memcpy(bsdf.normal, basis + 6, 12);
    vec3 c = { rr[0], rr[1], rr[2] };
    BSDF_addOrenNayar(&bsdf, c, sigma);
    vec3 f = { 0, 0, 0 }, wi = { 0, 0, 0 }, w = { wo[0], wo[1], wo[2] }, we = { wi_eval[0], wi_eval[1], wi_eval[2] };
    vec2 smp = { sample[0], sample[1] };
    BxDFFlags flags = BSDF_NONE;
    float pdf = BSDF_sample_f(f, wi, &flags, w, smp, &bsdf);
    memcpy(f_out, f, 12); memcpy(wi_out, wi, 12);
    vec3 fe = { 0, 0, 0 };
    BSDF_f(fe, we, w, &bsdf, (BxDFFlags)(BSDF_SPECULAR | BSDF_GLOSSY));
    memcpy(f_eval, fe, 12);
    BSDF_freeBxDFs(&bsdf);
    return pdf;
}

/* BSDF with one SpecularTransmission lobe (what computeScatteringFunc builds for TRANSPARENT).  The lobe draws
 * from libc rand(); srand(seed) first so that the caller knows the value. */
REF_API float ref_transparent_bsdf(float ior_in, float ior_out, unsigned seed, const float* basis, const float* wo,
                                   float* f_out, float* wi_out, int* flags_out)
{
    static bool pool = initBSDFMem(1, 4);
    (void)pool;
    BSDF bsdf;
    bsdf.num_bxdf = 0;
    memcpy(bsdf.tangent, basis, 12); memcpy(bsdf.binormal, basis + 3, 12); memcpy(bsdf.normal, basis + 6, 12);
    vec3 cf = { 1.0f, 1.0f, 1.0f };
    BSDF_addSpecularTransmission(&bsdf, ior_in, ior_out, cf, cf);
    vec3 f = { 0, 0, 0 }, wi = { 0, 0, 0 }, w = { wo[0], wo[1], wo[2] };
    vec2 smp = { 0.5f, 0.5f };
    BxDFFlags flags = BSDF_NONE;
    srand(seed);
    float pdf = BSDF_sample_f(f, wi, &flags, w, smp, &bsdf);
    memcpy(f_out, f, 12); memcpy(wi_out, wi, 12);
    *flags_out = (int)flags;
    BSDF_freeBxDFs(&bsdf);
    return pdf;
}

/* The BSDF computeScatteringFunc (materials.cpp:142-186) builds for PLASTIC / GLASS / METAL, put together here with the same
 * BSDF_add* calls (materials.cpp itself needs the texture loader).  BSDF_sample_f after srand(seed) (the glass lobe draws from
 * rand()), and BSDF_f for (wi_eval, wo) with SPECULAR | GLOSSY excluded, as estimateDirect calls it. */
REF_API float ref_material_bsdf(const crt_material* m, unsigned seed, const float* basis, const float* wo, const float* sample,
                                const float* wi_eval, float* f_out, float* wi_out, int* flags_out, float* f_eval)
{
    static bool pool = initBSDFMem(1, 4);
    (void)pool;
    BSDF bsdf;
    bsdf.num_bxdf = 0;
    memcpy(bsdf.tangent, basis, 12); memcpy(bsdf.binormal, basis + 3, 12); memcpy(bsdf.normal, basis + 6, 12);
    if (m->type == CRT_MAT_PLASTIC) {
        vec3 kd = { m->color[0], m->color[1], m->color[2] }, ks = { m->ks[0], m->ks[1], m->ks[2] };
        BSDF_addFresnelBlendDiffuse(&bsdf, kd, ks);
        BSDF_addFresnelBlendSpecular(&bsdf, ks, 1.5f, 1.0f, m->uroughness, m->uroughness, BECKMANN);
    } else if (m->type == CRT_MAT_GLASS) {
        vec3 color = { 1.0f, 1.0f, 1.0f };
        BSDF_addMicrofacetFresnel(&bsdf, color, m->ior_in, m->ior_out, m->uroughness, m->vroughness, BECKMANN);
    } else if (m->type == CRT_MAT_METAL) {
        vec3 color = { 1.0f, 1.0f, 1.0f }, etaI = { 1.0f, 1.0f, 1.0f };
        vec3 eta = { m->color[0], m->color[1], m->color[2] }, k = { m->k[0], m->k[1], m->k[2] };
        BSDF_addMicrofacetReflectionMetal(&bsdf, color, eta, etaI, k, m->uroughness, m->vroughness, BECKMANN);
    } else
        return -1.0f;
    vec3 f = { 0, 0, 0 }, wi = { 0, 0, 0 }, w = { wo[0], wo[1], wo[2] }, we = { wi_eval[0], wi_eval[1], wi_eval[2] };
    vec2 smp = { sample[0], sample[1] };
    BxDFFlags flags = BSDF_NONE;
    srand(seed);
    float pdf = BSDF_sample_f(f, wi, &flags, w, smp, &bsdf);
    memcpy(f_out, f, 12); memcpy(wi_out, wi, 12);
    *flags_out = (int)flags;
    vec3 fe = { 0, 0, 0 };
    BSDF_f(fe, we, w, &bsdf, (BxDFFlags)(BSDF_SPECULAR | BSDF_GLOSSY));
    memcpy(f_eval, fe, 12);
    BSDF_freeBxDFs(&bsdf);
    return pdf;
}
/* the value rand() / RAND_MAX the lobe will see after srand(seed) */
REF_API float ref_rand_float_after_srand(unsigned seed) { srand(seed); return (float)rand() / (float)RAND_MAX; }

/* Beckmann pieces (microfacet.cpp): out = { D(wh), G(wo, wi), Lambda(wo), sample_wh(wo, sample)[3] } */
REF_API void ref_beckmann(float alphax, float alphay, const float* wo, const float* wi, const float* wh, const float* sample, float* out)
{
    MicrofacetDistribution d; d.alphax = alphax; d.alphay = alphay; d.type = BECKMANN;
    vec3 o = { wo[0], wo[1], wo[2] }, i = { wi[0], wi[1], wi[2] }, h = { wh[0], wh[1], wh[2] };
    vec2 smp = { sample[0], sample[1] };
    out[0] = MicrofacetDistribution_D(h, &d);
    out[1] = MicrofacetDistribution_G(o, i, &d);
    out[2] = 1.0f / MicrofacetDistribution_G1(o, &d) - 1.0f;      /* Lambda is not declared in microfacet.h; G1 = 1 / (1 + Lambda) */
    vec3 r;
    MicrofacetDistribution_sample_wh(r, o, smp, &d);
    memcpy(out + 3, r, 12);
}
REF_API float ref_roughness_to_alpha(float roughness) { return BeckmannRoughnessToAlpha(roughness); }

/* ---- mesh triangles: the reference's own smooth / flat intersectors and matrix helpers -------- */
/* rayIntersectSmoothTriangle / rayIntersectFlatTriangle on one triangle of a three-vertex Mesh.  out = hit_point,
 * normal, wo, dpdu; bg = beta, gamma of calcTriangleIntersect (recomputed here with the same function). */
REF_API float ref_mesh_triangle_shade_rec(int smooth, const float* v, const float* vn, const float* vt, const float* flat_normal,
                                          const float* normal_mat, const crt_ray* ray, float* out, float* bg, float dpdu_marker)
{
    Ray r; to_ray(&r, ray);
    Mesh mesh; memset(&mesh, 0, sizeof mesh);
    float normals[9], texcoords[6] = {0, 0, 0, 0, 0, 0};
    vec3 tangents[3];
    memcpy(normals, vn, sizeof normals); memcpy(tangents, vt, sizeof tangents);
    mesh.normals = normals; mesh.tangents = tangents; mesh.texcoords = texcoords; mesh.num_texcoords = 0;
    mat3 nm; memcpy(nm, normal_mat, sizeof nm);
    ShadeRec sr;
    vec3_assign(sr.dpdu, dpdu_marker, dpdu_marker, dpdu_marker);
    vec2_assign(sr.uv, 0.0f, 0.0f);                                /* FlatTriangle wraps whatever uv holds (triangle.cpp:195-198) */
    float t;
    if (smooth) {
        SmoothTriangle tri; memset(&tri, 0, sizeof tri);
        tri.i0 = 0; tri.i1 = 1; tri.i2 = 2;
        memcpy(tri.v0, v, 12); memcpy(tri.v1, v + 3, 12); memcpy(tri.v2, v + 6, 12);
        tri.normal_mat = &nm; tri.mesh_ptr = &mesh; tri.mat = 0;
        Material dummy; memset(&dummy, 0, sizeof dummy); tri.mat = &dummy;
        t = rayIntersectSmoothTriangle(&sr, &tri, r);
    } else {
        FlatTriangle tri; memset(&tri, 0, sizeof tri);
        tri.i0 = 0; tri.i1 = 1; tri.i2 = 2;
        memcpy(tri.v0, v, 12); memcpy(tri.v1, v + 3, 12); memcpy(tri.v2, v + 6, 12);
        memcpy(tri.normal, flat_normal, 12);
        tri.mesh_ptr = &mesh;
        Material dummy; memset(&dummy, 0, sizeof dummy); tri.mat = &dummy;
        t = rayIntersectFlatTriangle(&sr, &tri, r);
    }
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.wo, 12); memcpy(out + 9, sr.dpdu, 12);
    vec3 v0, v1, v2; memcpy(v0, v, 12); memcpy(v1, v + 3, 12); memcpy(v2, v + 6, 12);
    calcTriangleIntersect(bg, bg + 1, v0, v1, v2, r);
    return t;
}
/* The same with texture coordinates (uv6 = the three vertices' texcoords, NULL: a mesh without any) and every field out:
 * out = hit_point, normal, dpdu, dpdv, wo, uv (17 floats), pre-set to `marker`. */
REF_API float ref_mesh_triangle_shade_rec_full(int smooth, const float* v, const float* vn, const float* vt, const float* uv6,
                                               const float* flat_normal, const float* normal_mat, const crt_ray* ray, float* out,
                                               float* bg, float marker)
{
    Ray r; to_ray(&r, ray);
    Mesh mesh; memset(&mesh, 0, sizeof mesh);
    float normals[9], texcoords[6] = {0, 0, 0, 0, 0, 0};
    vec3 tangents[3];
    memcpy(normals, vn, sizeof normals); memcpy(tangents, vt, sizeof tangents);
    if (uv6) memcpy(texcoords, uv6, sizeof texcoords);
    mesh.normals = normals; mesh.tangents = tangents; mesh.texcoords = texcoords; mesh.num_texcoords = uv6 ? 6 : 0;
    mat3 nm; memcpy(nm, normal_mat, sizeof nm);
    ShadeRec sr;
    vec3_assign(sr.hit_point, marker, marker, marker); vec3_assign(sr.normal, marker, marker, marker);
    vec3_assign(sr.dpdu, marker, marker, marker); vec3_assign(sr.dpdv, marker, marker, marker);
    vec3_assign(sr.wo, marker, marker, marker);
    vec2_assign(sr.uv, 0.25f, 0.75f);                              /* inside [0, 1]: FlatTriangle's wrap loops (triangle.cpp:195-198) leave it alone */
    Material dummy; memset(&dummy, 0, sizeof dummy);
    float t;
    if (smooth) {
        SmoothTriangle tri; memset(&tri, 0, sizeof tri);
        tri.i0 = 0; tri.i1 = 1; tri.i2 = 2;
        memcpy(tri.v0, v, 12); memcpy(tri.v1, v + 3, 12); memcpy(tri.v2, v + 6, 12);
        tri.normal_mat = &nm; tri.mesh_ptr = &mesh; tri.mat = &dummy;
        t = rayIntersectSmoothTriangle(&sr, &tri, r);
    } else {
        FlatTriangle tri; memset(&tri, 0, sizeof tri);
        tri.i0 = 0; tri.i1 = 1; tri.i2 = 2;
        memcpy(tri.v0, v, 12); memcpy(tri.v1, v + 3, 12); memcpy(tri.v2, v + 6, 12);
        memcpy(tri.normal, flat_normal, 12);
        tri.mesh_ptr = &mesh; tri.mat = &dummy;
        t = rayIntersectFlatTriangle(&sr, &tri, r);
    }
    memcpy(out, sr.hit_point, 12); memcpy(out + 3, sr.normal, 12); memcpy(out + 6, sr.dpdu, 12); memcpy(out + 9, sr.dpdv, 12);
    memcpy(out + 12, sr.wo, 12); memcpy(out + 15, sr.uv, 8);
    vec3 v0, v1, v2; memcpy(v0, v, 12); memcpy(v1, v + 3, 12); memcpy(v2, v + 6, 12);
    calcTriangleIntersect(bg, bg + 1, v0, v1, v2, r);
    return t;
}
/* The matrix steps of generateMeshTriangles (buildscene.h:216-225,:258-266,:278-284) through the reference's own
 * eulerAngToMat3 / mat3_* / calcTriangleNormal: world vertex, normal_mat, FlatTriangle.normal of one model triangle. */
REF_API void ref_bake_triangle(const float* scaling, const float* orientation, const float* location, const float* model_v,
                               float* world_v, float* normal_mat_out, float* flat_normal_out)
{
    mat3 rotation, inv_scale_mat, normal_mat;
    vec3 sc = { scaling[0], scaling[1], scaling[2] }, orient = { orientation[0], orientation[1], orientation[2] }, loc = { location[0], location[1], location[2] };
    eulerAngToMat3(rotation, orient);
    mat3_scale_inverse(inv_scale_mat, sc);
    mat3_mult(normal_mat, rotation, inv_scale_mat);
    memcpy(normal_mat_out, normal_mat, sizeof normal_mat);
    vec3 mv[3];
    for (int k = 0; k < 3; k++) {
        vec3 vv = { model_v[3*k], model_v[3*k+1], model_v[3*k+2] };
        vec3_copy(mv[k], vv);
        vec3_mult(vv, sc, vv);
        vec3 nv;
        mat3_mult_vec3(nv, rotation, vv);
        vec3_add(nv, loc, nv);
        memcpy(world_v + 3*k, nv, 12);
    }
    vec3 face_normal, new_face_normal;
    calcTriangleNormal(face_normal, mv[0], mv[1], mv[2]);
    mat3_mult_vec3(new_face_normal, normal_mat, face_normal);
    vec3_normalize(new_face_normal, new_face_normal);
    memcpy(flat_normal_out, new_face_normal, 12);
}

/* ---- OBJ files: the reference's own loadOBJ (objloader/objloader.h, implemented in mesh.cpp) -------- */
static OBJShape* g_obj_shapes = 0; static OBJMaterial* g_obj_mats = 0; static int g_obj_num = 0, g_obj_num_mat = 0;
REF_API int ref_obj_open(const char* file_name)
{
    g_obj_shapes = 0; g_obj_num = 0;
    if (!loadOBJ(&g_obj_shapes, &g_obj_mats, &g_obj_num, &g_obj_num_mat, file_name)) return -1;
    return g_obj_num;
}
REF_API void ref_obj_shape_sizes(int i, int* sizes)
{
    sizes[0] = g_obj_shapes[i].num_positions; sizes[1] = g_obj_shapes[i].num_normals;
    sizes[2] = g_obj_shapes[i].num_texcoords; sizes[3] = g_obj_shapes[i].num_indices;
}
REF_API void ref_obj_shape_copy(int i, float* pos, float* nrm, float* tex, int* idx, char* mat_name)
{
    const OBJShape& s = g_obj_shapes[i];
    memcpy(pos, s.positions, sizeof(float) * s.num_positions); memcpy(nrm, s.normals, sizeof(float) * s.num_normals);
    memcpy(tex, s.texcoords, sizeof(float) * s.num_texcoords); memcpy(idx, s.indices, sizeof(int) * s.num_indices);
    memcpy(mat_name, s.mat_name, OBJ_NAME_LENGTH);
}

/* ---- image state: the reference's own imagestate.h ------------------------- */
REF_API int ref_save_image_state(float* color_buffer, int num_samples, int width, int height, const char* file_name)
{
    return saveImageState(color_buffer, num_samples, width, height, file_name);
}
REF_API int ref_read_image_state(float* out, int capacity, int* num_samples, int* width, int* height, const char* file_name)
{
    float* buf = 0; int size = 0;
    readImageState(&buf, num_samples, &size, width, height, file_name);
    if (buf) { memcpy(out, buf, sizeof(float) * (size < capacity ? size : capacity)); free(buf); }
    return size;
}

/* the lat-long mapping getEnvLightIncRadiance applies before its texel fetch (lights.cpp:241-243): the reference's own
 * cartesianToSpherical + sphericalToUV (util/math.h:95-107) */
REF_API void ref_env_uv(const float* dir, float* uv)
{
    vec3 in = { dir[0], dir[1], dir[2] };
    vec2 spherical, out;
    cartesianToSpherical(spherical, in);
    sphericalToUV(out, spherical);
    uv[0] = out[0]; uv[1] = out[1];
}
