/*
 * crt_scenefile.cpp -- scene description files to crt_scene_input + materials, host side.
 *
 * Follows loadSceneFile / setupFilmAndCamera (buildscene.h:344-539), the parse* functions of scene/scenefile.h:18-791,
 * getNextTokenInFile (util/util.cpp:4-25), procMetalType (materials.cpp:74-109) and preprocessLights' world radius
 * (buildscene.h:835-873).  OBJECT MESH entries go through this library's crt_obj_load, crt_mesh_normals_tangents and
 * crt_bake_mesh (loadOBJ, calcTriangleNormals / calcTangentVec, generateMeshTriangles).
 *
 * PARITY UNPINNED: scenefile.h and buildscene.h include <GLFW/glfw3.h> and the texture loader (stb_image.h, not
 * shipped), so neither can be compiled into oracle/_ref.  What is checked: the shipped cornell_box.txt gives the
 * geometry and materials of scene S1 (tests/test_scene_file.py), a file in the current grammar gives the objects a
 * hand-built ShapeList gives, and the transforms equal the ones pinned through crt_bake_mesh.
 *
 * Where this reader is more tolerant than the reference's, on purpose:
 *   - the reference's own shipped files (cornell_box.txt, example_scene.txt, test_scene2.txt) are written in an OLDER
 *     grammar that the current positional parser cannot read (no film header: setupFilmAndCamera would never return;
 *     MATTE entries with DIFF_COLOR / AMB_COLOR instead of COLOR / SIGMA).  Entries are therefore read as
 *     keyword / value pairs up to END (materials) or MATERIAL <name> (objects), accepting the keys of both grammars;
 *     a well-formed file of the current grammar reads exactly as the reference reads it;
 *   - '\t' and '\r' separate tokens too (the reference splits on ' ' and '\n' only);
 *   - `#` starts a comment at the top level anywhere in the file (the reference: only after END_MATERIALS);
 *   - textures (TEXTURE / KD_TEXTURE / NORMAL_MAP, ENV_LIGHT TYPE TEXTURE) are bound by file name as the reference binds
 *     them (Scene_addTexture while parsing, linkMaterialTextures afterwards) and read with crt_texture_load_ppm -- the
 *     reference's loadTexture is stb_image, which it does not ship; PPM is the format its own imagefile.h reads.  A file
 *     that cannot be read leaves an entry without texels and view.num_unsupported counts it; the .exr environment map
 *     (read through OpenEXR in the reference) is named, its texels are the caller's to supply;
 *   - material types the current reference rejects (the old REFLECTIVE, PHONG) are kept as type 0 so that
 *     object -> material indices stay stable;
 *   - a mesh file named by two entries is baked once per entry (the reference reloads it and bakes every copy found
 *     by name again, buildscene.h:426-441).
 */
#include "../../include/crt_gpu.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>

namespace {

typedef float mat4[4][4];

struct Tokens {
    FILE* fp;
    std::string pushed; bool has_pushed;
    void push_back(const std::string& t) { pushed = t; has_pushed = true; }
    /* getNextTokenInFile, util/util.cpp:4-25 */
    bool next(std::string* out)
    {
        if (has_pushed) { *out = pushed; has_pushed = false; return true; }
        int c = fgetc(fp);
        while (c == ' ' || c == '\n' || c == '\t' || c == '\r') c = fgetc(fp);
        if (c == EOF) return false;
        out->clear();
        while (c != ' ' && c != '\n' && c != '\t' && c != '\r' && c != EOF) { out->push_back((char)c); c = fgetc(fp); }
        if (c == '\n') ungetc(c, fp);                 /* so that a comment token at the end of a line stops at this newline */
        return true;
    }
    void skip_line() { int c; while ((c = fgetc(fp)) != '\n' && c != EOF) {} }
    bool number(float* v) { std::string t; if (!next(&t)) return false; *v = (float)atof(t.c_str()); return true; }
    bool vec3(float* v) { return number(v) && number(v + 1) && number(v + 2); }          /* parseVec3, scenefile.h:64-74 */
};

/* getPresetColor, scenefile.h:18-62; the colours of util/constants.h */
bool preset_color(float* r, const std::string& s)
{
    static const struct { const char* name; float c[3]; } P[] = {
        {"RED", {1.0f, 0.0f, 0.0f}}, {"GREEN", {0.0f, 1.0f, 0.0f}}, {"BLUE", {0.0f, 0.0f, 1.0f}}, {"WHITE", {1.0f, 1.0f, 1.0f}},
        {"BLACK", {0.0f, 0.0f, 0.0f}}, {"YELLOW", {1.0f, 1.0f, 0.0f}}, {"CYAN", {0.0f, 1.0f, 1.0f}}, {"PINK", {1.0f, 0.0f, 1.0f}},
        {"GREY", {0.5f, 0.5f, 0.5f}}, {"MED_ORCHID", {0.729f, 0.333f, 0.827f}} };        /* util/constants.h:30-39 */
    for (const auto& p : P) if (s == p.name) { memcpy(r, p.c, 12); return true; }
    return false;
}
/* parseColor, scenefile.h:76-89 */
bool color(Tokens& tk, float* r)
{
    std::string t;
    if (!tk.next(&t)) return false;
    if (preset_color(r, t)) return true;
    r[0] = (float)atof(t.c_str());
    return tk.number(r + 1) && tk.number(r + 2);
}

/* procMetalType, materials.cpp:5-20,:74-109 */
bool metal_type(crt_material* m, const std::string& s)
{
    static const struct { const char* name; float eta[3], k[3]; } M[] = {
        {"GOLD", {0.14282006f, 0.37414363f, 1.43944442f}, {3.90463543f, 2.44763327f, 2.13765264f}},
        {"SILVER", {0.154935181f, 0.116475478f, 0.138087392f}, {4.81810093f, 3.11561656f, 2.1424017f}},
        {"BERYLLIUM", {4.17617416f, 3.1783011f, 2.77819276f}, {3.82729554f, 3.00373626f, 2.86292768f}},
        {"CHROMIUM", {4.36040831f, 2.9105196f, 1.65118635f}, {5.19538164f, 4.22238398f, 3.74699736f}},
        {"CESIUM", {2.14034843f, 1.69870293f, 1.65889668f}, {0.0f, 0.0f, 0.0f}},
        {"COPPER", {0.19999069f, 0.92208463f, 1.09987593f}, {3.90463543f, 2.44763327f, 2.13765264f}},
        {"MERCURY", {2.39383841f, 1.43696785f, 0.907622635f}, {6.31419611f, 4.36266136f, 3.41453838f}} };
    for (const auto& e : M) if (s == e.name) { memcpy(m->color, e.eta, 12); memcpy(m->k, e.k, 12); return true; }
    return false;
}

void mat4_identity(mat4 r) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[i][j] = i == j ? 1.0f : 0.0f; }
void mat4_mult(mat4 r, const mat4 a, const mat4 b)                  /* util/mat.h:164-173 */
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j];
}
void mat4_rotate(mat4 r, int axis, float theta)                     /* mat4_rotate_x / _y / _z, util/mat.h:218-271 */
{
    const float s = sinf(theta), c = cosf(theta);                   /* (float)sin(theta) on a float argument: the float overload */
    mat4_identity(r);
    const int a = axis == 0 ? 1 : 0, b = axis == 2 ? 1 : 2;         /* x: (1,2)  y: (0,2)  z: (0,1) */
    r[a][a] = c; r[a][b] = s; r[b][a] = -s; r[b][b] = c;
}
inline float deg_to_rad(float degree) { return degree / 180.0f * (float)3.14159265358979323846264338327950288; }   /* util/math.h:65-68 */
/* the inv_transform every instanced primitive entry composes (scenefile.h:497-503): inv_scale * inv_rotation * inv_translation,
 * rotation = eulerAngToMat4 (util/math.cpp:80-93) */
void inverse_transform(float* out16, const float* location, const float* scale, const float* orientation)
{
    mat4 inv_scale, rotation, inv_rotation, inv_translation, tmp, inv_transform, z_rot, y_rot, x_rot, t2;
    mat4_identity(inv_scale);
    inv_scale[0][0] = 1.0f / scale[0]; inv_scale[1][1] = 1.0f / scale[1]; inv_scale[2][2] = 1.0f / scale[2];
    mat4_identity(inv_translation);
    inv_translation[0][3] = -location[0]; inv_translation[1][3] = -location[1]; inv_translation[2][3] = -location[2];
    mat4_rotate(z_rot, 2, deg_to_rad(orientation[2]));
    mat4_rotate(x_rot, 0, deg_to_rad(orientation[1]));
    mat4_rotate(y_rot, 1, deg_to_rad(orientation[0]));
    mat4_mult(t2, x_rot, z_rot);
    mat4_mult(rotation, y_rot, t2);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) inv_rotation[i][j] = rotation[j][i];       /* mat4_invert_rotation = transpose */
    mat4_mult(tmp, inv_scale, inv_rotation);
    mat4_mult(inv_transform, tmp, inv_translation);
    memcpy(out16, inv_transform, 64);
}

struct Loaded {
    std::vector<int32_t> obj_type, obj_ref, obj_material, normal_mat_index;
    std::vector<float> tri_verts, flat_normal, vertex_normals, vertex_tangents, normal_mats, vertex_uv;
    bool any_uv;
    std::vector<crt_shape> shapes;
    std::vector<crt_material> materials;
    std::vector<std::string> names;
    std::vector<const char*> name_ptrs;
    std::vector<std::string> tex_names;
    std::vector<const char*> tex_name_ptrs;
    std::vector<crt_texture> textures;
    std::vector<int32_t> mat_diffuse_tex, mat_normal_tex;
    std::string dir;
    crt_scene_file_view view;
    bool any_mesh;
};

/* Scene_findTexture, else loadTexture + Scene_addTexture (parseTexture, scenefile.h:92-111): index of the entry of that name.
 * external: the texels are not ours to read (the .exr environment map). */
int bind_texture(Loaded& L, const std::string& file, bool external)
{
    for (size_t i = 0; i < L.tex_names.size(); i++) if (L.tex_names[i] == file) return (int)i;
    crt_texture t; memset(&t, 0, sizeof t);
    if (external) { t.is_float = 1; t.comp = 3; L.view.num_external++; }
    else {
        int rc = crt_texture_load_ppm(file.c_str(), &t);
        if (rc != CRT_OK && !L.dir.empty() && file[0] != '/') rc = crt_texture_load_ppm((L.dir + "/" + file).c_str(), &t);
        if (rc != CRT_OK) { memset(&t, 0, sizeof t); L.view.num_unsupported++; }
    }
    L.tex_names.push_back(file);
    L.textures.push_back(t);
    return (int)L.textures.size() - 1;
}

int find_material(const Loaded& L, const std::string& name)          /* Scene_findMaterial; -1 = NULL */
{
    for (size_t i = 0; i < L.names.size(); i++) if (L.names[i] == name) return (int)i;
    return -1;
}

int push_shape(Loaded& L, int type, const float* p, int np, int inner = -1, int first_part = 0, int num_parts = 0, const float* inv = 0)
{
    crt_shape s; memset(&s, 0, sizeof s);
    s.type = type; s.inner = inner; s.first_part = first_part; s.num_parts = num_parts;
    for (int i = 0; i < np && i < 12; i++) s.p[i] = p[i];
    mat4 id; mat4_identity(id);
    memcpy(s.inv_xform, inv ? (const void*)inv : (const void*)id, 64);
    L.shapes.push_back(s);
    return (int)L.shapes.size() - 1;
}
void push_object(Loaded& L, int type, int ref, int mat)
{
    L.obj_type.push_back(type); L.obj_ref.push_back(ref); L.obj_material.push_back(mat);
}
void push_triangle(Loaded& L, int type, const float* v9, int mat, const float* fn, const float* vn9, const float* vt9, int nmi, const float* uv6 = 0)
{
    static const float z9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    L.vertex_uv.insert(L.vertex_uv.end(), uv6 ? uv6 : z9, (uv6 ? uv6 : z9) + 6);
    if (uv6) L.any_uv = true;
    L.tri_verts.insert(L.tri_verts.end(), v9, v9 + 9);
    L.flat_normal.insert(L.flat_normal.end(), fn ? fn : z9, (fn ? fn : z9) + 3);
    L.vertex_normals.insert(L.vertex_normals.end(), vn9 ? vn9 : z9, (vn9 ? vn9 : z9) + 9);
    L.vertex_tangents.insert(L.vertex_tangents.end(), vt9 ? vt9 : z9, (vt9 ? vt9 : z9) + 9);
    L.normal_mat_index.push_back(nmi);
    push_object(L, type, (int)(L.tri_verts.size() / 9) - 1, mat);
}

/* MATERIAL <TYPE> ... END (parseMatEntry, scenefile.h:309-352, and the entries it dispatches to) */
void parse_material(Loaded& L, Tokens& tk)
{
    std::string t, type_name;
    if (!tk.next(&type_name)) return;
    crt_material m; memset(&m, 0, sizeof m);
    m.color[0] = m.color[1] = m.color[2] = 1.0f;
    m.ior_in = 1.5f; m.ior_out = 1.0f;                              /* parseGlassEntry's defaults, scenefile.h:253-254 */
    if (type_name == "MATTE") m.type = CRT_MAT_MATTE;               /* getMatTypeFromString, materials.cpp:22-49 */
    else if (type_name == "MIRROR") m.type = CRT_MAT_MIRROR;
    else if (type_name == "TRANSPARENT") m.type = CRT_MAT_TRANSPARENT;
    else if (type_name == "EMISSIVE") m.type = CRT_MAT_EMISSIVE;
    else if (type_name == "PLASTIC") m.type = CRT_MAT_PLASTIC;
    else if (type_name == "GLASS") m.type = CRT_MAT_GLASS;
    else if (type_name == "METAL") m.type = CRT_MAT_METAL;
    else { m.type = 0; L.view.num_unsupported++; }
    std::string name;
    float diff_constant = 1.0f; bool legacy_diffuse = false;
    int diffuse_tex = -1, normal_tex = -1;
    while (tk.next(&t) && t != "END") {
        if (t == "MATERIAL" || t == "END_MATERIALS" || t == "OBJECT") {     /* an entry of the old grammar without END (example_scene.txt) */
            tk.push_back(t);
            break;
        }
        float v, c3[3];
        if (t == "NAME") tk.next(&name);
        else if (t == "COLOR") { if (color(tk, c3)) memcpy(m.color, c3, 12); }
        else if (t == "TEXTURE" || t == "KD_TEXTURE") { if (tk.next(&t)) diffuse_tex = bind_texture(L, t, false); }   /* scenefile.h:141-147,:225-233 */
        else if (t == "NORMAL_MAP") { if (tk.next(&t)) normal_tex = bind_texture(L, t, false); }                       /* :161-171 */
        else if (t == "SIGMA") { if (tk.number(&v)) m.sigma = v; }
        else if (t == "IOR_IN") { if (tk.number(&v)) m.ior_in = v; }
        else if (t == "IOR_OUT") { if (tk.number(&v)) m.ior_out = v; }
        else if (t == "CF_IN" || t == "CF_OUT") color(tk, c3);               /* read and, as in the `thin` lobe, unused */
        else if (t == "KD") { if (color(tk, c3)) memcpy(m.color, c3, 12); }
        else if (t == "KS") { if (color(tk, c3)) memcpy(m.ks, c3, 12); }
        else if (t == "ROUGHNESS") { if (tk.number(&v)) { m.uroughness = v; m.vroughness = v; } }
        else if (t == "TYPE") { if (tk.next(&t) && !metal_type(&m, t)) L.view.num_unsupported++; }
        else if (t == "INTENSITY") { if (tk.number(&v)) m.intensity = v; }
        /* the older grammar of the shipped files */
        else if (t == "DIFF_COLOR") { if (color(tk, c3)) { memcpy(m.color, c3, 12); legacy_diffuse = true; } }
        else if (t == "DIFF_CONSTANT") { if (tk.number(&v)) diff_constant = v; }
        else if (t == "AMB_COLOR" || t == "SPEC_COLOR") color(tk, c3);
        else if (t == "AMB_CONSTANT" || t == "SPEC_CONSTANT" || t == "SPEC_EXPONENT") tk.number(&v);
        else if (t == "SHADOWED") tk.next(&t);
    }
    if (legacy_diffuse) for (int k = 0; k < 3; k++) m.color[k] *= diff_constant;       /* kd * cd of the old Matte */
    if (m.type != CRT_MAT_TRANSPARENT && m.type != CRT_MAT_GLASS) { m.ior_in = 1.0f; m.ior_out = 1.0f; }
    /* linkMaterialTextures (buildscene.h:315-342) links MATTE and PLASTIC only */
    if (m.type != CRT_MAT_MATTE && m.type != CRT_MAT_PLASTIC) diffuse_tex = -1;
    if (m.type != CRT_MAT_MATTE) normal_tex = -1;
    L.materials.push_back(m);
    L.names.push_back(name);
    L.mat_diffuse_tex.push_back(diffuse_tex);
    L.mat_normal_tex.push_back(normal_tex);
}

/* NormalType, shapes/generic.h:13-18; -1: invalid */
int parse_normal_type(const std::string& t) { return t == "OPEN" ? 0 : t == "CONVEX" ? 1 : t == "CONCAVE" ? 2 : -1; }

int load_mesh(Loaded& L, const std::string& file, const crt_mesh_entry& entry, int mat)
{
    crt_obj_shape* shapes = 0; int32_t ns = 0;
    int rc = crt_obj_load(file.c_str(), &shapes, &ns);
    if (rc != CRT_OK && !L.dir.empty() && file[0] != '/') rc = crt_obj_load((L.dir + "/" + file).c_str(), &shapes, &ns);
    if (rc != CRT_OK) return rc;
    for (int s = 0; s < ns; s++) {
        const crt_obj_shape& sh = shapes[s];
        const int nv = sh.num_positions / 3, ni = sh.num_indices, nt = ni / 3;
        if (nt == 0) continue;
        std::vector<float> normals(3 * (size_t)nv), tangents(3 * (size_t)nv);
        /* Mesh_copyOBJShape + calcTriangleNormals + calcTangentVec (buildscene.h:430-437) */
        rc = crt_mesh_normals_tangents(sh.positions, nv, sh.indices, ni, sh.num_normals == sh.num_positions ? sh.normals : 0,
                                       normals.data(), tangents.data());
        if (rc != CRT_OK) break;
        crt_mesh mesh = { sh.positions, normals.data(), tangents.data(), sh.indices, nv, ni };
        std::vector<float> tv(9 * (size_t)nt), fn(3 * (size_t)nt), vn(9 * (size_t)nt), vt(9 * (size_t)nt);
        float nm[9];
        rc = crt_bake_mesh(&mesh, &entry, tv.data(), fn.data(), vn.data(), vt.data(), nm);     /* generateMeshTriangles */
        if (rc != CRT_OK) break;
        L.normal_mats.insert(L.normal_mats.end(), nm, nm + 9);
        const int nmi = (int)(L.normal_mats.size() / 9) - 1;
        const bool has_uv = sh.num_texcoords == 2 * nv && sh.texcoords;      /* mesh->num_texcoords > 0, triangle.cpp:191 */
        for (int k = 0; k < nt; k++) {
            float uv6[6];
            if (has_uv) for (int c = 0; c < 3; c++) {                        /* texcoords[i*2], [i*2+1] of the triangle's three vertices */
                const int vi = sh.indices[3 * k + c];
                uv6[2 * c] = sh.texcoords[2 * (size_t)vi]; uv6[2 * c + 1] = sh.texcoords[2 * (size_t)vi + 1];
            }
            push_triangle(L, entry.smooth ? CRT_SMOOTH_TRIANGLE : CRT_FLAT_TRIANGLE, &tv[9 * (size_t)k], mat, &fn[3 * (size_t)k],
                          &vn[9 * (size_t)k], &vt[9 * (size_t)k], nmi, has_uv ? uv6 : 0);
        }
        L.any_mesh = true;
    }
    crt_obj_free(shapes, ns);
    return rc;
}

/* OBJECT <KIND> key value ... MATERIAL <name>  (parsePrimitive / parseMesh, scenefile.h:368-791) */
void parse_object(Loaded& L, Tokens& tk)
{
    std::string kind, t, file, mat_name;
    if (!tk.next(&kind)) return;
    float radius = 1.0f, center[3] = {0, 0, 0}, phi = 3.14159f, min_theta = 0.0f, max_theta = 3.14159f;
    float point[3] = {0, 0, 0}, normal[3] = {0, 1, 0}, width3[3] = {1, 0, 0}, height3[3] = {0, 1, 0}, v[9] = {0};
    float length = 1.0f, height = 1.0f, width = 1.0f, swept = 1.0f, tube = 0.25f;
    float location[3] = {0, 0, 0}, scale[3] = {1, 1, 1}, orientation[3] = {0, 0, 0};
    bool smooth = false, cast_shadow = true, got_material = false;
    int normal_type = 0;
    const bool is_box = kind == "BOX";
    while (tk.next(&t)) {
        if (t == "MATERIAL") { got_material = tk.next(&mat_name); break; }      /* always the last field of an entry */
        if (t == "RADIUS") tk.number(&radius);
        else if (t == "CENTER") tk.vec3(center);
        else if (t == "PHI") tk.number(&phi);
        else if (t == "MIN_THETA") tk.number(&min_theta);
        else if (t == "MAX_THETA") tk.number(&max_theta);
        else if (t == "POINT") tk.vec3(point);
        else if (t == "NORMAL") tk.vec3(normal);
        else if (t == "WIDTH") { if (is_box) tk.number(&width); else tk.vec3(width3); }
        else if (t == "HEIGHT") { if (is_box) tk.number(&height); else tk.vec3(height3); }
        else if (t == "LENGTH") tk.number(&length);
        else if (t == "V0") tk.vec3(v);
        else if (t == "V1") tk.vec3(v + 3);
        else if (t == "V2") tk.vec3(v + 6);
        else if (t == "LOCATION") tk.vec3(location);
        else if (t == "SCALE" || t == "SCALING") tk.vec3(scale);
        else if (t == "ORIENTATION" || t == "Orientation") tk.vec3(orientation);
        else if (t == "NORMAL_TYPE") { tk.next(&t); normal_type = parse_normal_type(t); if (normal_type < 0) return; }   /* shading only; an invalid one drops the entry (scenefile.h:538-542) */
        else if (t == "SWEPT_RADIUS") tk.number(&swept);
        else if (t == "TUBE_RADIUS") tk.number(&tube);
        else if (t == "FILE_NAME" || t == "FILE") tk.next(&file);
        else if (t == "SMOOTH") { tk.next(&t); smooth = t == "yes"; }
        else if (t == "CAST_SHADOW") { tk.next(&t); cast_shadow = t != "no"; }
        else if (t == "OBJECT" || t == "ENV_LIGHT") {               /* an entry without MATERIAL: give the token back and drop it */
            tk.push_back(t);
            break;
        }
    }
    if (!got_material) { L.view.num_skipped++; return; }
    const int mat = find_material(L, mat_name);
    float inv[16];
    if (kind == "SPHERE") {
        const float p[7] = { center[0], center[1], center[2], radius, min_theta, max_theta, phi };
        push_object(L, CRT_SPHERE, push_shape(L, CRT_SPHERE, p, 7), mat);
    } else if (kind == "PLANE") {
        const float p[6] = { point[0], point[1], point[2], normal[0], normal[1], normal[2] };
        push_object(L, CRT_PLANE, push_shape(L, CRT_PLANE, p, 6), mat);
    } else if (kind == "RECTANGLE") {                               /* normal = normalize(width x height), scenefile.h:432-433 */
        float n[3] = { width3[1] * height3[2] - width3[2] * height3[1], width3[2] * height3[0] - width3[0] * height3[2],
                       width3[0] * height3[1] - width3[1] * height3[0] };
        const float len = (float)sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);            /* vec3_normalize, util/vec.h:206-215 */
        if (len == 0) n[0] = n[1] = n[2] = 0.0f; else { const float s = 1.0f / len; n[0] *= s; n[1] *= s; n[2] *= s; }
        const float p[12] = { point[0], point[1], point[2], width3[0], width3[1], width3[2], height3[0], height3[1], height3[2], n[0], n[1], n[2] };
        push_object(L, CRT_RECTANGLE, push_shape(L, CRT_RECTANGLE, p, 12), mat);
    } else if (kind == "TRIANGLE") {
        push_triangle(L, CRT_TRIANGLE, v, mat, 0, 0, 0, 0);
    } else if (kind == "DISK") {
        float n[3] = { normal[0], normal[1], normal[2] };
        const float len = (float)sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (len == 0) n[0] = n[1] = n[2] = 0.0f; else { const float s = 1.0f / len; n[0] *= s; n[1] *= s; n[2] *= s; }
        const float p[7] = { center[0], center[1], center[2], n[0], n[1], n[2], radius };
        push_object(L, CRT_DISK, push_shape(L, CRT_DISK, p, 7), mat);
    } else if (kind == "BOX") {                                     /* initBox, shapes/box.cpp:4-20: AABox of half spans in an instance */
        inverse_transform(inv, location, scale, orientation);
        const float hx = length / 2.0f, hy = height / 2.0f, hz = width / 2.0f;
        const float p[7] = { -hx, -hy, -hz, hx, hy, hz, cast_shadow ? 1.0f : 0.0f };
        const int inner = push_shape(L, CRT_AABOX, p, 7);
        push_object(L, CRT_INSTANCED, push_shape(L, CRT_INSTANCED, 0, 0, inner, 0, 0, inv), mat);
    } else if (kind == "TORUS") {
        inverse_transform(inv, location, scale, orientation);
        const float p[3] = { swept, tube, phi };
        const int inner = push_shape(L, CRT_GENERICTORUS, p, 3);
        push_object(L, CRT_INSTANCED, push_shape(L, CRT_INSTANCED, 0, 0, inner, 0, 0, inv), mat);
    } else if (kind == "OPENCYLINDER") {                            /* initOpenCylinder, shapes/cylinder.cpp:3-18: half height 1, radius 1 */
        inverse_transform(inv, location, scale, orientation);
        const float p[4] = { 1.0f, 1.0f, phi, (float)normal_type };
        const int inner = push_shape(L, CRT_GENERICOPENCYLINDER, p, 4);
        push_object(L, CRT_INSTANCED, push_shape(L, CRT_INSTANCED, 0, 0, inner, 0, 0, inv), mat);
    } else if (kind == "SOLIDCYLINDER") {                           /* initSolidCylinder(inv, 1, 1, PI), shapes/cylinder.cpp:29-75 */
        inverse_transform(inv, location, scale, orientation);
        const float r = 1.0f, hh = 1.0f, pi = (float)3.14159265358979323846264338327950288;
        const float top[7] = { 0.0f, hh, 0.0f, 0.0f, 1.0f, 0.0f, r }, bottom[7] = { 0.0f, -hh, 0.0f, 0.0f, -1.0f, 0.0f, r };
        const float tubep[3] = { hh, r, pi }, box[6] = { -r, -hh, -r, r, hh, r };
        const int first = push_shape(L, CRT_DISK, top, 7);
        push_shape(L, CRT_DISK, bottom, 7);
        push_shape(L, CRT_GENERICOPENCYLINDER, tubep, 3);
        const int comp = push_shape(L, CRT_COMPOUND, box, 6, -1, first, 3);
        push_object(L, CRT_INSTANCED, push_shape(L, CRT_INSTANCED, 0, 0, comp, 0, 0, inv), mat);
    } else if (kind == "MESH") {
        crt_mesh_entry e;
        memcpy(e.scaling, scale, 12); memcpy(e.orientation, orientation, 12); memcpy(e.location, location, 12);
        e.smooth = smooth ? 1 : 0;
        if (file.empty() || load_mesh(L, file, e, mat) != CRT_OK) L.view.num_skipped++;
    } else
        L.view.num_skipped++;
}

/* ENV_LIGHT TYPE CONSTANT COLOR <c> INTENSITY <v> | TYPE TEXTURE COLOR <file> INTENSITY <v>  (buildscene.h:451-534) */
void parse_env_light(Loaded& L, Tokens& tk)
{
    std::string t;
    float c3[3] = {1.0f, 1.0f, 1.0f}, intensity = 0.0f;
    tk.next(&t);                                                    /* TYPE */
    tk.next(&t);
    L.view.env_type = 0; L.view.env_map_tex = -1;
    memset(L.view.env_transform, 0, sizeof L.view.env_transform);
    if (t == "CONSTANT") {
        tk.next(&t);                                                /* COLOR */
        color(tk, c3);
    } else {                                                        /* TEXTURE: COLOR <file> */
        tk.next(&t);
        std::string file;
        tk.next(&file);
        const size_t dot = file.find_last_of('.');                  /* "check extension", buildscene.h:470-477 */
        if (dot != std::string::npos && file.substr(dot + 1) == "exr") {
            L.view.env_type = 1;                                    /* readRgba1: OpenEXR, float texels, comp 3 (:478-483) */
            L.view.env_map_tex = bind_texture(L, file, true);
        }                                                           /* any other extension: CONSTANT and WHITE (:484-488) */
        const float th = -0.76f;                                    /* mat3_rotate_y(env_light->transform, -0.76), :515, util/mat.h:229-238 */
        const float sn = sinf(th), cs = cosf(th);                    /* (float)sin(theta) on a float argument: the float overload */
        const float xf[9] = { cs, 0.0f, sn, 0.0f, 1.0f, 0.0f, -sn, 0.0f, cs };
        memcpy(L.view.env_transform, xf, sizeof xf);
    }
    tk.next(&t);                                                    /* INTENSITY */
    tk.number(&intensity);
    L.view.has_env_light = 1;
    memcpy(L.view.env_color, c3, 12);
    L.view.env_intensity = intensity;
    if (!(intensity > 0.0f)) L.view.env_color[0] = L.view.env_color[1] = L.view.env_color[2] = 0.0f;   /* :525-533: not a light, black when seen */
}

} // namespace

struct crt_scene_file { Loaded L; };

extern "C" {

int crt_scene_file_load(const char* path, crt_scene_file** out)
{
    if (!path || !out) return CRT_E_INVALID;
    *out = 0;
    FILE* fp = fopen(path, "r");
    if (!fp) return CRT_E_INVALID;
    crt_scene_file* sf = new (std::nothrow) crt_scene_file;
    if (!sf) { fclose(fp); return CRT_E_NOMEM; }
    Loaded& L = sf->L;
    memset(&L.view, 0, sizeof L.view);
    L.view.env_map_tex = -1;
    L.any_mesh = false;
    L.any_uv = false;
    {
        const std::string p(path);
        const size_t slash = p.find_last_of('/');
        L.dir = slash == std::string::npos ? std::string() : p.substr(0, slash);
    }
    Tokens tk; tk.fp = fp; tk.has_pushed = false;
    std::string t;
    float v;
    bool materials_done = false;
    try {
        while (tk.next(&t)) {
            if (t[0] == '#') { tk.skip_line(); continue; }          /* buildscene.h:410-413 */
            if (t == "WINDOW_WIDTH") { if (tk.number(&v)) { L.view.window_width = (int)v; L.view.has_film = 1; } }       /* setupFilmAndCamera, :344-386 */
            else if (t == "WINDOW_HEIGHT") { if (tk.number(&v)) L.view.window_height = (int)v; }
            else if (t == "IMAGE_WIDTH") { if (tk.number(&v)) L.view.image_width = (int)v; }
            else if (t == "IMAGE_HEIGHT") { if (tk.number(&v)) L.view.image_height = (int)v; }
            else if (t == "FOV") { std::string s; if (tk.next(&s)) L.view.fov = deg_to_rad((float)atof(s.c_str())); }
            else if (t == "CAMERA_POS") { if (tk.vec3(L.view.camera_pos)) L.view.has_camera = 1; }
            else if (t == "LOOK_POINT") tk.vec3(L.view.look_point);
            else if (t == "END_MATERIALS") materials_done = true;
            else if (t == "MATERIAL") { if (!materials_done) parse_material(L, tk); }   /* after END_MATERIALS the word is data of a commented-out entry */
            else if (t == "OBJECT") parse_object(L, tk);
            else if (t == "ENV_LIGHT") parse_env_light(L, tk);
        }
    } catch (const std::bad_alloc&) {
        fclose(fp); crt_scene_file_free(sf); return CRT_E_NOMEM;
    }
    fclose(fp);
    for (const std::string& s : L.names) L.name_ptrs.push_back(s.c_str());
    crt_scene_file_view& V = L.view;
    V.input.num_obj = (int32_t)L.obj_type.size();
    V.input.obj_type = L.obj_type.data(); V.input.obj_ref = L.obj_ref.data();
    V.input.tri_verts = L.tri_verts.data(); V.input.num_tris = (int32_t)(L.tri_verts.size() / 9);
    V.input.shapes = L.shapes.data(); V.input.num_shapes = (int32_t)L.shapes.size();
    V.obj_material = L.obj_material.data();
    V.materials = L.materials.data(); V.num_materials = (int32_t)L.materials.size();
    V.material_names = L.name_ptrs.data();
    V.has_mesh_shading = L.any_mesh ? 1 : 0;
    V.mesh_shading.flat_normal = L.flat_normal.data();
    V.mesh_shading.vertex_normals = L.vertex_normals.data();
    V.mesh_shading.vertex_tangents = L.vertex_tangents.data();
    V.mesh_shading.normal_mat_index = L.normal_mat_index.data();
    V.mesh_shading.normal_mats = L.normal_mats.data();
    V.mesh_shading.num_normal_mats = (int32_t)(L.normal_mats.size() / 9);
    V.mesh_shading.vertex_uv = L.any_uv ? L.vertex_uv.data() : 0;
    for (const std::string& n : L.tex_names) L.tex_name_ptrs.push_back(n.c_str());
    V.num_textures = (int32_t)L.textures.size();
    V.texture_names = L.tex_name_ptrs.data();
    V.textures = L.textures.data();
    V.material_diffuse_tex = L.mat_diffuse_tex.data();
    V.material_normal_tex = L.mat_normal_tex.data();
    *out = sf;
    return CRT_OK;
}

const crt_scene_file_view* crt_scene_file_get(const crt_scene_file* sf) { return sf ? &sf->L.view : 0; }
void crt_scene_file_free(crt_scene_file* sf)
{
    if (!sf) return;
    for (crt_texture& t : sf->L.textures) crt_texture_free(&t);
    delete sf;
}

/* preprocessLights, buildscene.h:839-873: twice the diagonal of the accelerator's box -- with the QBVH branch as written
 * there (it starts from min = K_EPSILON, max = -HUGEVALUE and mixes components: box_min[2] takes min(., box_min[1]),
 * box_max[1] and [2] take max(., box_max[0])). */
int crt_env_world_radius(const crt_scene_desc* d, float* radius)
{
    if (!d || !radius) return CRT_E_INVALID;
    float mn[3], mx[3];
    if (d->accel == CRT_ACCEL_BVH && d->nodes2 && d->num_nodes2 > 0) {
        memcpy(mn, d->nodes2[0].bmin, 12); memcpy(mx, d->nodes2[0].bmax, 12);
    } else if (d->accel == CRT_ACCEL_GRID) {
        memcpy(mn, d->grid.bmin, 12); memcpy(mx, d->grid.bmax, 12);
    } else if (d->accel == CRT_ACCEL_BVH4 && d->nodes4 && d->num_nodes4 > 0) {
        const float* bbox = d->nodes4[0].bbox;
        const float HUGEVALUE = 1.0e10f;                            /* util/constants.h */
        mn[0] = mn[1] = mn[2] = CRT_K_EPSILON;
        mx[0] = mx[1] = mx[2] = -HUGEVALUE;
        for (int i = 0; i < 4; i++) {
            mn[0] = bbox[i + 0] < mn[0] ? bbox[i + 0] : mn[0];
            mn[1] = bbox[i + 4] < mn[1] ? bbox[i + 4] : mn[1];
            mn[2] = bbox[i + 8] < mn[1] ? bbox[i + 8] : mn[1];
            mx[0] = bbox[i + 12] > mx[0] ? bbox[i + 12] : mx[0];
            mx[1] = bbox[i + 16] > mx[0] ? bbox[i + 16] : mx[0];
            mx[2] = bbox[i + 20] > mx[0] ? bbox[i + 20] : mx[0];
        }
    } else
        return CRT_E_UNSUPPORTED;                                   /* NONE: the reference leaves world_radius unset */
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    *radius = (float)sqrt(dx * dx + dy * dy + dz * dz) * 2.0f;      /* vec3_length(dist) * 2.0f */
    return CRT_OK;
}

} // extern "C"
