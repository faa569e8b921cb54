/*
 * crt_obj.cpp -- Wavefront OBJ loading and per-vertex normals / tangents (part of SURVEY 8f N2), host code.
 *
 * loadOBJ (objloader/objloader.h:738-939) with exportGroupToShape (:386-445) and UpdateVertexCache (:273-318):
 * `v`, `vn`, `vt`, `f` records; faces fan-triangulated; inside one shape a vertex is one distinct (v, vn, vt) triple,
 * numbered in order of first use; a shape closes at every `g`, `usemtl`, `o` line and at the end of the file.  The
 * reference's hash table only finds duplicates faster; a std::map gives the same numbering.  Quirks kept: a last
 * line without '\n' is not processed (OBJGetLine returns false on EOF before the loop body runs); an index of 0 or a
 * token that is not a number drops that face vertex (fixIndex -> -1); negative indices count from the end.  The MTL
 * file is not read (generateMeshTriangles ignores its materials, buildscene.h:230-238).
 * Then calcTriangleNormals' vertex normals for shapes without `vn` (buildscene.h:58-98) and calcTangentVec
 * (buildscene.h:101-212), float32, operation by operation.
 */
#include "../../include/crt_gpu.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <vector>

namespace {

struct VertexIndex { int v_idx, vn_idx, vt_idx; };

bool get_line(FILE* fp, std::string* line)                        /* OBJGetLine, objloader.h:132-162 */
{
    line->clear();
    for (;;) {
        const int ci = fgetc(fp);
        const char c = (char)ci;
        if (ci == EOF) return false;
        if (c == '\n') return true;
        if (c == '\r') continue;
        if (c == '\0') return false;
        if (line->size() < 1023) line->push_back(c);              /* the reference's buffer holds 1024 bytes */
    }
}
float parse_float(const char** str)
{
    *str += strspn(*str, " \t");
    const float r = (float)atof(*str);
    *str += strcspn(*str, " \t\r");
    return r;
}
int fix_index(int index, int size) { return index > 0 ? index - 1 : (index < 0 ? index + size : -1); }
VertexIndex parse_face_triple(const char** str)                   /* OBJParseFaceTriple, objloader.h:206-243 */
{
    VertexIndex vi = {0, 0, 0};
    *str += strspn(*str, " \t");
    vi.v_idx = atoi(*str);
    *str += strcspn(*str, "/ \t");
    if ((*str)[0] != '/') return vi;
    (*str)++;
    if ((*str)[0] == '/') {
        (*str)++;
        vi.vn_idx = atoi(*str);
        *str += strcspn(*str, "/ \t");
        return vi;
    }
    vi.vt_idx = atoi(*str);
    *str += strcspn(*str, "/ \t");
    if ((*str)[0] != '/') return vi;
    (*str)++;
    vi.vn_idx = atoi(*str);
    *str += strcspn(*str, "/ \t\r");
    return vi;
}

struct Shape { std::vector<float> positions, normals, texcoords; std::vector<int> indices; std::string mat_name; };

bool export_group(Shape* out, const std::vector<float>& in_pos, const std::vector<float>& in_nrm, const std::vector<float>& in_tex,
                  std::vector<std::vector<VertexIndex> >* faces, bool* bad)
{
    if (faces->empty()) return false;
    Shape s;
    std::map<std::tuple<int, int, int>, int> cache;
    auto vertex = [&](const VertexIndex& vi) -> int {
        const auto key = std::make_tuple(vi.v_idx, vi.vn_idx, vi.vt_idx);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
        if (vi.v_idx < 0 || (size_t)vi.v_idx * 3 + 2 >= in_pos.size() ||                      /* the reference asserts / reads out of range */
            (vi.vt_idx >= 0 && (size_t)vi.vt_idx * 2 + 1 >= in_tex.size()) || (vi.vn_idx >= 0 && (size_t)vi.vn_idx * 3 + 2 >= in_nrm.size())) {
            *bad = true;
            return 0;
        }
        for (int k = 0; k < 3; k++) s.positions.push_back(in_pos[(size_t)vi.v_idx * 3 + k]);
        if (vi.vt_idx >= 0) for (int k = 0; k < 2; k++) s.texcoords.push_back(in_tex[(size_t)vi.vt_idx * 2 + k]);
        if (vi.vn_idx >= 0) for (int k = 0; k < 3; k++) s.normals.push_back(in_nrm[(size_t)vi.vn_idx * 3 + k]);
        const int index = (int)(s.positions.size() / 3) - 1;
        cache[key] = index;
        return index;
    };
    for (const std::vector<VertexIndex>& f : *faces)
        for (size_t j = 2; j < f.size(); j++) {
            const int i0 = vertex(f[0]), i1 = vertex(f[j - 1]), i2 = vertex(f[j]);
            s.indices.push_back(i0); s.indices.push_back(i1); s.indices.push_back(i2);
        }
    faces->clear();
    *out = std::move(s);
    return true;
}

template <class T>
T* dup(const std::vector<T>& v)
{
    if (v.empty()) return 0;
    T* p = (T*)malloc(v.size() * sizeof(T));
    if (p) memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

typedef float vec3[3];
inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void normalize3(float* r, const float* a)
{
    const float length = (float)sqrt((double)dot3(a, a));
    if (length == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    const float inv = 1.0f / length;
    const float x = a[0] * inv, y = a[1] * inv, z = a[2] * inv;
    r[0] = x; r[1] = y; r[2] = z;
}

} // namespace

extern "C" {

void crt_obj_free(crt_obj_shape* shapes, int32_t num_shapes)
{
    if (!shapes) return;
    for (int i = 0; i < num_shapes; i++) { free(shapes[i].positions); free(shapes[i].normals); free(shapes[i].texcoords); free(shapes[i].indices); }
    free(shapes);
}

int crt_obj_load(const char* file_name, crt_obj_shape** shapes_out, int32_t* num_shapes)
{
    if (!file_name || !shapes_out || !num_shapes) return CRT_E_INVALID;
    *shapes_out = 0; *num_shapes = 0;
    FILE* fp = fopen(file_name, "r");
    if (!fp) return CRT_E_INVALID;
    try {
        std::vector<float> in_pos, in_nrm, in_tex;
        std::vector<std::vector<VertexIndex> > faces;
        std::vector<Shape> shapes;
        std::string cur_mat, line;
        bool bad = false;
        auto flush = [&](bool named) {
            Shape s;
            if (export_group(&s, in_pos, in_nrm, in_tex, &faces, &bad)) { if (named) s.mat_name = cur_mat; shapes.push_back(std::move(s)); }
        };
        while (get_line(fp, &line)) {
            const char* begin = line.c_str();
            const char* p = begin + strspn(begin, " \t");
            if (p[0] == '#') continue;
            /* the reference tests every record kind in turn on the same, advancing pointer (objloader.h:795-914) */
            if (p[0] == 'v' && p[1] == ' ') { p += 1; for (int k = 0; k < 3; k++) in_pos.push_back(parse_float(&p)); }
            if (p[0] == 'v' && p[1] == 'n') { p += 2; for (int k = 0; k < 3; k++) in_nrm.push_back(parse_float(&p)); }
            if (p[0] == 'v' && p[1] == 't') { p += 2; for (int k = 0; k < 2; k++) in_tex.push_back(parse_float(&p)); }
            if (p[0] == 'f' && p[1] == ' ') {
                p += 1;
                std::vector<VertexIndex> face;
                while (p[0] != '\0' && p < begin + line.size()) {
                    VertexIndex vi = parse_face_triple(&p);
                    vi.v_idx = fix_index(vi.v_idx, (int)(in_pos.size() / 3));
                    vi.vt_idx = fix_index(vi.vt_idx, (int)(in_tex.size() / 2));
                    vi.vn_idx = fix_index(vi.vn_idx, (int)(in_nrm.size() / 3));
                    if (vi.v_idx != -1) face.push_back(vi);
                }
                faces.push_back(face);
            }
            if (p[0] == 'g') flush(true);
            if (p[0] == 'u' && strncmp(p, "usemtl", 6) == 0) {
                flush(true);
                p += 6;
                p += strspn(p, " \t");
                const size_t len = strcspn(p, " \t\r");
                cur_mat.assign(p, len < 63 ? len : 63);
                p += len;
            }
            if (p[0] == 'o' && p[1] == ' ') flush(false);
        }
        fclose(fp); fp = 0;
        flush(true);
        if (bad) return CRT_E_INVALID;
        crt_obj_shape* out = (crt_obj_shape*)calloc(shapes.empty() ? 1 : shapes.size(), sizeof(crt_obj_shape));
        if (!out) return CRT_E_NOMEM;
        for (size_t i = 0; i < shapes.size(); i++) {
            const Shape& s = shapes[i];
            out[i].positions = dup(s.positions); out[i].normals = dup(s.normals); out[i].texcoords = dup(s.texcoords); out[i].indices = dup(s.indices);
            out[i].num_positions = (int32_t)s.positions.size(); out[i].num_normals = (int32_t)s.normals.size();
            out[i].num_texcoords = (int32_t)s.texcoords.size(); out[i].num_indices = (int32_t)s.indices.size();
            snprintf(out[i].mat_name, sizeof out[i].mat_name, "%s", s.mat_name.c_str());
            if ((!out[i].positions && !s.positions.empty()) || (!out[i].indices && !s.indices.empty())) { crt_obj_free(out, (int32_t)shapes.size()); return CRT_E_NOMEM; }
        }
        *shapes_out = out; *num_shapes = (int32_t)shapes.size();
        return CRT_OK;
    } catch (const std::exception&) {
        if (fp) fclose(fp);
        return CRT_E_NOMEM;
    }
}

int crt_mesh_normals_tangents(const float* positions, int32_t num_vertices, const int32_t* indices, int32_t num_indices,
                              const float* normals_in, float* normals_out, float* tangents_out)
{
    if (num_vertices < 0 || num_indices < 0 || num_indices % 3 != 0 || !normals_out || !tangents_out) return CRT_E_INVALID;
    if ((num_vertices > 0 && !positions) || (num_indices > 0 && !indices)) return CRT_E_INVALID;
    for (int j = 0; j < num_indices; j++) if (indices[j] < 0 || indices[j] >= num_vertices) return CRT_E_INVALID;
    const size_t nv = (size_t)num_vertices;
    if (normals_in) {
        memcpy(normals_out, normals_in, nv * 3 * sizeof(float));
    } else {                                                       /* calcTriangleNormals, buildscene.h:62-98 */
        for (size_t i = 0; i < nv * 3; i++) normals_out[i] = 0.0f;
        for (int i = 0; i < num_indices; i += 3) {
            const float* v0 = positions + 3 * (size_t)indices[i];
            const float* v1 = positions + 3 * (size_t)indices[i + 1];
            const float* v2 = positions + 3 * (size_t)indices[i + 2];
            const vec3 e1 = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, e2 = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
            const vec3 c = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
            vec3 fn;
            normalize3(fn, c);                                     /* calcTriangleNormal */
            for (int k = 0; k < 3; k++) { float* n = normals_out + 3 * (size_t)indices[i + k]; n[0] += fn[0]; n[1] += fn[1]; n[2] += fn[2]; }
        }
        for (size_t i = 0; i < nv; i++) normalize3(normals_out + 3 * i, normals_out + 3 * i);
    }
    /* calcTangentVec, buildscene.h:101-212.  Its texture coordinates are (0,0), (1,0), (1,1) for every triangle: set
     * outright when the mesh has texcoords (:157-159), and when it has none they are read from a fresh, zeroed
     * allocation and replaced by the same three (:132-140). */
    for (size_t i = 0; i < nv * 3; i++) tangents_out[i] = 0.0f;
    for (int i = 0; i < num_indices; i += 3) {
        const int i0 = indices[i], i1 = indices[i + 1], i2 = indices[i + 2];
        const float* v0 = positions + 3 * (size_t)i0; const float* v1 = positions + 3 * (size_t)i1; const float* v2 = positions + 3 * (size_t)i2;
        const vec3 q0 = { v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] }, q1 = { v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] };
        const float t0 = 0.0f - 0.0f, t1 = 1.0f - 0.0f;
        vec3 u_axis;
        for (int k = 0; k < 3; k++) { const float t1q0 = q0[k] * t1, t0q1 = q1[k] * t0; u_axis[k] = t1q0 - t0q1; }
        const int vs[3] = { i0, i1, i2 };
        for (int k = 0; k < 3; k++) { float* t = tangents_out + 3 * (size_t)vs[k]; t[0] = t[0] + u_axis[0]; t[1] = t[1] + u_axis[1]; t[2] = t[2] + u_axis[2]; }
    }
    for (size_t i = 0; i < nv; i++) {
        const float* normal = normals_out + 3 * i;
        float* t = tangents_out + 3 * i;
        const float d = dot3(normal, t);
        const vec3 displacement = { normal[0] * d, normal[1] * d, normal[2] * d };
        const vec3 tt = { t[0] - displacement[0], t[1] - displacement[1], t[2] - displacement[2] };
        normalize3(t, tt);
    }
    return CRT_OK;
}

} // extern "C"
