/*
 * crt_oracle.h -- CPU restatement of craytracer's traversal + intersection path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported baseline.  The product (craytracer_amd/) never
 * links or calls it.
 *
 * Pinning: validated in the build container against oracle/_ref (the reference's
 * own builders, QBVH traversal, box tests and primitive intersectors compiled from
 * /root/reference) by tests/test_oracle_vs_ref.py, and against the committed golden
 * vectors under tests/golden/ (generated from oracle/_ref by tests/golden/make_golden.py).
 * Not covered by oracle/_ref: the functions that live in intersect.h itself
 * (grid walk, binary-BVH recursion, dispatch) -- that header includes <GLFW/glfw3.h>,
 * which the image lacks, so it is unbuildable here; those three are restated from
 * the text and cross-checked against NONE/BVH4 results on the same rays.
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H
#include "../include/crt_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

/* mvNonGridObjToStart + buildSceneAccel restated.  sort_mode: 0 = stable sort for every
 * range (deterministic; what the product's builder does), 1 = the reference's own
 * insertion-sort / rand()-pivot quicksort (bvh.h:56-112), for comparison with oracle/_ref. */
oracle_scene* oracle_scene_create(const crt_scene_input* in, int accel, int sort_mode);
/* Adopt an already built scene (final order + flat accelerator) without rebuilding it. */
oracle_scene* oracle_scene_from_desc(const crt_scene_desc* d);
void          oracle_scene_destroy(oracle_scene* s);

/* Final objects[] order as insertion ids; returns num_obj.  ids may be NULL. */
int  oracle_scene_order(const oracle_scene* s, int32_t* ids, int32_t* num_non_grid);
/* Flattened accelerator, same numbering rule as the product (pre-order DFS, children in
 * slot order).  Pass NULL to query counts. */
int  oracle_scene_nodes4(const oracle_scene* s, crt_bvh4_node* out);
int  oracle_scene_nodes2(const oracle_scene* s, crt_bvh2_node* out);
int  oracle_scene_grid(const oracle_scene* s, crt_grid* g, int32_t* cell_start, int32_t* cell_items);

/* mode: CRT_TRAVERSE_EXACT = the reference's control flow; CRT_TRAVERSE_PRUNED = best-t
 * pruned, reference child order (the traversal the algorithmic-bytes counters are defined on).
 * counters (may be NULL) accumulates nodes_visited / prims_tested.  nthreads >= 1. */
void oracle_intersect(const oracle_scene* s, const crt_ray* rays, size_t n, crt_hit* hits,
                      int mode, crt_counters* counters, int nthreads);
void oracle_shadow(const oracle_scene* s, const crt_ray* rays, const float* light_dist, size_t n,
                   float* t_out, uint8_t* occluded_out, crt_counters* counters, int nthreads);

/* Single-primitive entry points (for leaf-level tests). */
float oracle_intersect_object(const oracle_scene* s, int final_index, const crt_ray* ray, crt_hit* hit);
float oracle_shadow_object(const oracle_scene* s, int final_index, const crt_ray* ray);
float oracle_ray_aabb(const float bmin[3], const float bmax[3], const crt_ray* ray);

/* primary rays: camera.cpp restated (checker for crt_camera_* / crt_primary_rays_dev) */
void oracle_camera_init(crt_camera* c, int type, float focal_length, float lens_radius);
void oracle_camera_look_at(crt_camera* c, const float* position, const float* look_point, const float* up);
void oracle_camera_set_focal_dist(crt_camera* c, float d);
void oracle_film_dimension(crt_film* f, float fov, int w, int h, const crt_camera* c);
void oracle_primary_rays(const crt_camera* c, const crt_film* f, unsigned first_pixel, size_t n,
                         const float* image_samples, const float* lens_samples, crt_ray* rays);

/* path tracing, restricted form (crt_oracle_trace.cpp; checker for crt_renderer_*).  color_buffer
 * [npix*3] is added to (main.cpp:139-141); samples first_sample .. first_sample + num_samples - 1.
 * Returns 0, or -1 for a scene outside the restricted form. */
float oracle_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
int   oracle_render(const oracle_scene* scene, const crt_render_desc* d, const crt_camera* cam, const crt_film* film,
                    uint32_t first_sample, uint32_t num_samples, float* color_buffer, int nthreads);

/* pieces of the above, exposed for comparison with the reference's own code (tests/test_trace_pieces.py):
 * ShadeRec tail (out = hit_point, normal, wo, dpdu), sampling maps, local basis, Lambertian BSDF */
void  oracle_trace_shade_rec(const crt_scene_input* in, int obj, const crt_ray* ray, float t, float* out);
void  oracle_trace_shade_rec_any(const crt_scene_input* in, const crt_hit* hit, const crt_ray* ray, float* out);
void  oracle_trace_shade_rec_mesh(const crt_scene_input* in, const crt_mesh_shading* ms, int obj, const crt_ray* ray, float t,
                                  float beta, float gamma, float* out);
void  oracle_trace_disk(const float* in, float* out);
void  oracle_trace_hemisphere(const float* in, float* out);
void  oracle_trace_local_basis(const float* a, const float* normal, float* out);
float oracle_trace_lambertian(const float* cd, const float* basis, const float* wo, const float* sample,
                              float* f_out, float* wi_out, float* f_eval);
float oracle_trace_mirror(const float* cr, const float* basis, const float* wo, float* f_out, float* wi_out, int* flags);
float oracle_trace_orennayar(const float* r, float sigma, const float* basis, const float* wo, const float* sample,
                             const float* wi_eval, float* f_out, float* wi_out, float* f_eval);
float oracle_trace_transparent(float ior_in, float ior_out, float rand_float, const float* basis, const float* wo,
                               float* f_out, float* wi_out, int* flags);
float oracle_trace_material_bsdf(const crt_material* mat, float rand_float, const float* basis, const float* wo, const float* sample,
                                 const float* wi_eval, float* f_out, float* wi_out, int* flags_out, float* f_eval);
void  oracle_trace_beckmann(float alphax, float alphay, const float* wo, const float* wi, const float* wh, const float* sample, float* out);
void  oracle_trace_fresnel_conductor(const float* normal, const float* wo, const float* etaT, const float* etaI, const float* k, float* out);
float oracle_trace_roughness_to_alpha(float roughness);
/* getTexColor (texture.cpp:27-86; parity unpinned, the file needs stb_image.h) and the lat-long mapping of getEnvLightIncRadiance
 * (util/math.h:95-107; pinned against oracle/_ref) */
void  oracle_trace_tex_color(const crt_texture* tex, const float* uv, float* out);
void  oracle_trace_env_uv(const float* dir, float* uv);

#ifdef __cplusplus
}
#endif
#endif
