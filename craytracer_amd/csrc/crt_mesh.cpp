/*
 * crt_mesh.cpp -- baking a mesh into world-space triangles (part of SURVEY 8f N2), host code.
 *
 * generateMeshTriangles (buildscene.h:214-311) with the face normals of calcTriangleNormals
 * (buildscene.h:28-56): every OBJECT MESH entry of a scene file becomes FLAT_TRIANGLE or SMOOTH_TRIANGLE
 * objects whose vertices are scaled, rotated (eulerAngToMat3, util/math.cpp:65-78) and translated; normals
 * go through normal_mat = rotation * inverse scaling.  float32, operation by operation.
 */
#include "../../include/crt_gpu.h"

#include <math.h>
#include <string.h>

namespace {

typedef float vec3[3];
typedef float mat3[3][3];

inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void mat3_identity(mat3 r) { for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) r[i][j] = 0.0f; r[i][i] = 1.0f; } }
inline void mat3_mult(mat3 r, const mat3 a, const mat3 b)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}
inline void mat3_mult_vec3(vec3 r, const mat3 a, const vec3 b) { for (int i = 0; i < 3; i++) r[i] = dot3(a[i], b); }
inline float degToRad(float degree) { return degree / 180.0f * (float)3.14159265358979323846264338327950288; }   /* util/math.h:65-68 */
inline void normalize3(vec3 r, const vec3 a)
{
    const float length = (float)sqrt((double)dot3(a, a));
    if (length == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    const float inv = 1.0f / length;
    r[0] = a[0] * inv; r[1] = a[1] * inv; r[2] = a[2] * inv;
}
/* mat3_rotate_x / _y / _z, util/mat.h:207-260: `(float)sin(theta)` on a float theta is C++'s float overload, i.e. sinf */
void rot(mat3 r, int axis, float theta)
{
    const float s = sinf(theta), c = cosf(theta);
    mat3_identity(r);
    if (axis == 0)      { r[1][1] = c; r[1][2] = s; r[2][1] = -s; r[2][2] = c; }
    else if (axis == 1) { r[0][0] = c; r[0][2] = s; r[2][0] = -s; r[2][2] = c; }
    else                { r[0][0] = c; r[0][1] = s; r[1][0] = -s; r[1][1] = c; }
}
void eulerAngToMat3(mat3 r, const float* euler_ang)
{
    const float z_rad = degToRad(euler_ang[2]), x_rad = degToRad(euler_ang[1]), y_rad = degToRad(euler_ang[0]);
    mat3 z_rot, y_rot, x_rot, tmp;
    rot(z_rot, 2, z_rad); rot(x_rot, 0, x_rad); rot(y_rot, 1, y_rad);
    mat3_mult(tmp, x_rot, z_rot);
    mat3_mult(r, y_rot, tmp);
}

} // namespace

extern "C" int crt_bake_mesh(const crt_mesh* mesh, const crt_mesh_entry* entry, float* tri_verts, float* flat_normal,
                             float* vertex_normals, float* vertex_tangents, float* normal_mat_out)
{
    if (!mesh || !entry || !tri_verts || mesh->num_indices < 0 || mesh->num_indices % 3 != 0 || mesh->num_vertices < 0) return CRT_E_INVALID;
    if (mesh->num_indices > 0 && (!mesh->positions || !mesh->indices)) return CRT_E_INVALID;
    if ((vertex_normals && !mesh->normals) || (vertex_tangents && !mesh->tangents)) return CRT_E_INVALID;
    for (int j = 0; j < mesh->num_indices; j++) if (mesh->indices[j] < 0 || mesh->indices[j] >= mesh->num_vertices) return CRT_E_INVALID;
    mat3 rotation, inv_scale_mat, normal_mat;
    eulerAngToMat3(rotation, entry->orientation);
    mat3_identity(inv_scale_mat);                                  /* mat3_scale_inverse, util/mat.h:331-337 */
    for (int a = 0; a < 3; a++) inv_scale_mat[a][a] = 1.0f / entry->scaling[a];
    mat3_mult(normal_mat, rotation, inv_scale_mat);
    if (normal_mat_out) memcpy(normal_mat_out, normal_mat, sizeof normal_mat);
    for (int j = 0; j < mesh->num_indices; j += 3) {
        const int tri = j / 3;
        const int idx[3] = { mesh->indices[j], mesh->indices[j + 1], mesh->indices[j + 2] };
        vec3 model[3];
        for (int k = 0; k < 3; k++) {
            const float* p = mesh->positions + 3 * (size_t)idx[k];
            model[k][0] = p[0]; model[k][1] = p[1]; model[k][2] = p[2];
            vec3 v = { entry->scaling[0] * p[0], entry->scaling[1] * p[1], entry->scaling[2] * p[2] };   /* vec3_mult(v, scaling, v) */
            vec3 nv;
            mat3_mult_vec3(nv, rotation, v);
            float* o = tri_verts + 9 * (size_t)tri + 3 * k;
            o[0] = entry->location[0] + nv[0]; o[1] = entry->location[1] + nv[1]; o[2] = entry->location[2] + nv[2];
        }
        if (flat_normal) {
            vec3 e1 = { model[1][0] - model[0][0], model[1][1] - model[0][1], model[1][2] - model[0][2] };
            vec3 e2 = { model[2][0] - model[0][0], model[2][1] - model[0][1], model[2][2] - model[0][2] };
            vec3 c = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
            vec3 face_normal, nf;
            normalize3(face_normal, c);                            /* calcTriangleNormals: model-space face normal */
            mat3_mult_vec3(nf, normal_mat, face_normal);
            normalize3(flat_normal + 3 * (size_t)tri, nf);
        }
        for (int k = 0; k < 3; k++) {
            if (vertex_normals) memcpy(vertex_normals + 9 * (size_t)tri + 3 * k, mesh->normals + 3 * (size_t)idx[k], 12);
            if (vertex_tangents) memcpy(vertex_tangents + 9 * (size_t)tri + 3 * k, mesh->tangents + 3 * (size_t)idx[k], 12);
        }
    }
    return CRT_OK;
}
