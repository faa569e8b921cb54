/*
 * crt_shaderec.cpp -- crt_fill_shade_rec: hit records -> ShadeRec on the host.
 *
 * north_star keeps materials and shading with the C host; the host therefore needs, for every hit the GPU
 * returns, the record the reference's intersector would have left in `sr` (intersectTest(&sr, so, ray),
 * intersect.h:365).  The per-shape code is crt_shaderec.h (shared with the device renderer); this file is the
 * batch loop, threaded over contiguous chunks like the reference's pixel ranges (main.cpp:276-317).
 * Host libm is glibc's, as in the reference, so the records are the reference's bit for bit
 * (tests/test_shade_rec.py, against rayIntersectObject compiled from the reference's sources).
 */
#include <string.h>
#include <thread>
#include <vector>

#include "crt_shaderec.h"

namespace {

void fill_range(const crt_scene_input* in, const crt_mesh_shading* ms, const int32_t* obj_material, const crt_ray* rays,
                const crt_hit* hits, size_t begin, size_t end, int conventions, crt_shade_rec* out)
{
    for (size_t i = begin; i < end; i++) {
        crt_shade_rec* sr = &out[i];
        sr_fill(in, ms, &rays[i], &hits[i], conventions, sr);
        const int obj = hits[i].obj;
        sr->material = (obj_material && obj >= 0 && obj < in->num_obj) ? obj_material[obj] : -1;
    }
}

} // namespace

extern "C" int crt_fill_shade_rec(const crt_scene_input* in, const crt_mesh_shading* mesh_shading, const int32_t* obj_material,
                                  const crt_ray* rays, const crt_hit* hits, size_t n, int conventions, crt_shade_rec* out)
{
    if (!in || in->num_obj < 0 || (n > 0 && (!rays || !hits || !out))) return CRT_E_INVALID;
    if (in->num_obj > 0 && in->obj_type && !in->obj_ref) return CRT_E_INVALID;
    /* every index a hit can lead to is checked once, here: triangles into tri_verts, shapes into shapes[], nesting as crt_scene_create allows it */
    for (int i = 0; i < in->num_obj; i++) {
        const int t = in->obj_type ? in->obj_type[i] : CRT_SMOOTH_TRIANGLE, r = in->obj_ref ? in->obj_ref[i] : i;
        if (t == CRT_TRIANGLE || t == CRT_FLAT_TRIANGLE || t == CRT_SMOOTH_TRIANGLE) {
            if (r < 0 || r >= in->num_tris || !in->tri_verts) return CRT_E_INVALID;
        } else if (r < 0 || r >= in->num_shapes || !in->shapes || in->shapes[r].type != t) return CRT_E_INVALID;
    }
    for (int i = 0; i < in->num_shapes; i++) {
        const crt_shape& sh = in->shapes[i];
        if (sh.type == CRT_INSTANCED) {
            if (sh.inner < 0 || sh.inner >= in->num_shapes) return CRT_E_INVALID;
            if (in->shapes[sh.inner].type == CRT_INSTANCED) return CRT_E_UNSUPPORTED;
        }
        if (sh.type == CRT_COMPOUND) {
            if (sh.first_part < 0 || sh.num_parts < 0 || sh.first_part + sh.num_parts > in->num_shapes) return CRT_E_INVALID;
            for (int k = 0; k < sh.num_parts; k++) {
                const int pt = in->shapes[sh.first_part + k].type;
                if (pt == CRT_INSTANCED || pt == CRT_COMPOUND) return CRT_E_UNSUPPORTED;
            }
        }
    }
    if (mesh_shading && mesh_shading->vertex_normals && mesh_shading->normal_mat_index)
        for (int i = 0; i < in->num_tris; i++)
            if (mesh_shading->normal_mat_index[i] < 0 || mesh_shading->normal_mat_index[i] >= mesh_shading->num_normal_mats) return CRT_E_INVALID;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1 || n < 65536) nt = 1;
    if (nt == 1) { fill_range(in, mesh_shading, obj_material, rays, hits, 0, n, conventions, out); return CRT_OK; }
    try {
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++)
            th.emplace_back(fill_range, in, mesh_shading, obj_material, rays, hits, n * k / nt, n * (k + 1) / nt, conventions, out);
        for (std::thread& x : th) x.join();
    } catch (...) {
        return CRT_E_NOMEM;
    }
    return CRT_OK;
}
