// Host-side builder under AddressSanitizer + UBSan (g++, no GPU code): crt_build_scene on triangle
// soups of awkward sizes (around the leaf size 4 and the sort threshold 30), duplicated and
// degenerate triangles, equal centroids, NaN vertices, mixed analytic shapes, every accelerator;
// then the error paths.  Built and run by tests/test_host_sanitize.py.
#include "../include/crt_gpu.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <cstring>

static unsigned rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (rng_state >> 8) * (1.0f / 16777216.0f); }

/* grid_may_refuse: zero-volume or non-finite bounds have no grid resolution (CRT_E_INVALID) */
static int build_all(const crt_scene_input& in, const char* what, bool grid_may_refuse = false)
{
    for (int accel = 0; accel < 5; accel++) {           /* 4: the QBVH once more, built with CRT_BUILD_SAH */
        crt_built_scene* b = 0;
        int rc = accel == 4 ? crt_build_scene_ex(&in, CRT_ACCEL_BVH4, CRT_BUILD_SAH, &b) : crt_build_scene(&in, accel, &b);
        if (rc == CRT_E_INVALID && accel == CRT_ACCEL_GRID && grid_may_refuse && b == 0) continue;
        if (rc != CRT_OK) { printf("%s accel %d: rc %d\n", what, accel, rc); return 1; }
        const crt_scene_desc* d = crt_built_scene_desc(b);
        if (d->num_obj != in.num_obj) { printf("%s: num_obj\n", what); return 1; }
        long long sum = 0;
        for (int i = 0; i < d->num_obj; i++) sum += d->obj_id[i];
        if (sum != (long long)in.num_obj * (in.num_obj - 1) / 2) { printf("%s accel %d: obj_id is not a permutation\n", what, accel); return 1; }
        crt_built_scene_destroy(b);
    }
    return 0;
}

/* crt_obj.cpp, crt_mesh.cpp, crt_image.cpp on hostile inputs: every call must return a code, never read or write out of range */
static int write_file(const char* path, const std::string& text)
{
    FILE* fp = fopen(path, "wb");
    if (!fp) return 1;
    fwrite(text.data(), 1, text.size(), fp);
    fclose(fp);
    return 0;
}
static int obj_and_mesh_checks()
{
    const char* path = "/tmp/crt_host_sanitize.obj";
    const char* texts[] = {
        "", "\n\n", "f 1 2 3\n", "v 1 2\nf 1 1 1\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 4 5\n", "v 0 0 0\nf -5 -6 -7\n",
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/9/9 2/9/9 3/9/9\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1//-1 2//-1 3//-9\n",
        "v a b c\nv 1e999 -1e999 nan\nv 0 1 0\nf 1 2 3\nusemtl\nusemtl     \ng\no \nf 3 2 1\n",
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3      \t  \nf 1/ 2/ 3/\nf 1// 2// 3//\nf /1 /2 /3\n",
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 2147483647 2 3\nf -2147483648 2 3\nf 99999999999999999999 2 3\n",
    };
    for (const char* text : texts) {
        if (write_file(path, text)) { printf("cannot write %s\n", path); return 1; }
        crt_obj_shape* shapes = 0; int32_t n = -1;
        const int rc = crt_obj_load(path, &shapes, &n);
        if (rc == CRT_OK) {
            for (int i = 0; i < n; i++) {
                const crt_obj_shape& s = shapes[i];
                for (int k = 0; k < s.num_indices; k++) if (s.indices[k] < 0 || s.indices[k] * 3 + 2 >= s.num_positions) { printf("index out of range\n"); return 1; }
                const int nv = s.num_positions / 3;
                std::vector<float> nrm((size_t)nv * 3 + 1), tan((size_t)nv * 3 + 1);
                if (crt_mesh_normals_tangents(s.positions, nv, s.indices, s.num_indices, 0, nrm.data(), tan.data()) != CRT_OK) { printf("normals_tangents\n"); return 1; }
                crt_mesh mesh = { s.positions, nrm.data(), tan.data(), s.indices, nv, s.num_indices };
                crt_mesh_entry e = { {1.0f, 2.0f, 0.5f}, {30.0f, 45.0f, 60.0f}, {1.0f, 2.0f, 3.0f}, 1 };
                const int T = s.num_indices / 3;
                std::vector<float> tv((size_t)T * 9 + 1), fn((size_t)T * 3 + 1), vn((size_t)T * 9 + 1), vt((size_t)T * 9 + 1);
                float nm[9];
                if (crt_bake_mesh(&mesh, &e, tv.data(), fn.data(), vn.data(), vt.data(), nm) != CRT_OK) { printf("bake\n"); return 1; }
            }
            crt_obj_free(shapes, n);
        } else if (shapes != 0 || n != 0) { printf("outputs set on failure\n"); return 1; }
    }
    std::string big(5000, 'v');                                    /* a line longer than the reference's 1024-byte buffer */
    if (write_file(path, big + "\nv 0 0 0\n")) return 1;
    crt_obj_shape* shapes = 0; int32_t n = 0;
    if (crt_obj_load(path, &shapes, &n) != CRT_OK) { printf("long line\n"); return 1; }
    crt_obj_free(shapes, n);
    if (crt_obj_load("/nonexistent/x.obj", &shapes, &n) != CRT_E_INVALID || crt_obj_load(0, &shapes, &n) != CRT_E_INVALID) { printf("obj error codes\n"); return 1; }
    crt_obj_free(0, 0);
    /* mesh argument checks */
    float pos[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0}; int32_t bad_idx[3] = {0, 1, 3}; float out9[9], out3[3];
    crt_mesh m = { pos, 0, 0, bad_idx, 3, 3 };
    crt_mesh_entry e = { {1, 1, 1}, {0, 0, 0}, {0, 0, 0}, 0 };
    if (crt_bake_mesh(&m, &e, out9, out3, 0, 0, 0) != CRT_E_INVALID) { printf("bake index check\n"); return 1; }
    bad_idx[2] = 2;
    if (crt_bake_mesh(&m, &e, out9, out3, out9, 0, 0) != CRT_E_INVALID) { printf("bake normals check\n"); return 1; }   /* normals asked for, mesh has none */
    if (crt_bake_mesh(&m, &e, out9, out3, 0, 0, 0) != CRT_OK) { printf("bake ok case\n"); return 1; }
    m.num_indices = 2;
    if (crt_bake_mesh(&m, &e, out9, out3, 0, 0, 0) != CRT_E_INVALID) { printf("bake count check\n"); return 1; }
    /* image state */
    const char* ip = "/tmp/crt_host_sanitize.is";
    const char* states[] = { "", "3", "3\n", "3\n2 2", "3\n2 2\n", "3\n-2 2\n", "x\n2 2\n", "3\n99999 99999\n", "3\n2 2\nabcdefgh" };
    for (const char* st : states) {
        if (write_file(ip, st)) return 1;
        float* buf = 0; int ns = 0, w = 0, h = 0;
        if (crt_image_state_load(ip, &buf, &ns, &w, &h) == CRT_OK) { printf("truncated image state accepted: '%s'\n", st); return 1; }
        if (buf) { printf("buffer set on failure\n"); return 1; }
    }
    float cb[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 1e30f, -1.0f}; unsigned char img[12];
    if (crt_image_state_save(ip, cb, 4, 2, 2) != CRT_OK) { printf("state save\n"); return 1; }
    float* back = 0; int ns = 0, w = 0, h = 0;
    if (crt_image_state_load(ip, &back, &ns, &w, &h) != CRT_OK || ns != 4 || w != 2 || h != 2 || memcmp(back, cb, sizeof cb)) { printf("state round trip\n"); return 1; }
    crt_image_state_free(back);
    if (crt_image_from_color_buffer(img, cb, 4, 4) != CRT_OK || crt_image_from_color_buffer(0, cb, 4, 4) != CRT_E_INVALID) { printf("tone map\n"); return 1; }
    if (crt_ppm_write("/tmp/crt_host_sanitize.ppm", img, 2, 2) != CRT_OK || crt_ppm_write("/nonexistent/x.ppm", img, 2, 2) != CRT_E_INVALID) { printf("ppm\n"); return 1; }
    remove(path); remove(ip); remove("/tmp/crt_host_sanitize.ppm");
    return 0;
}

/* crt_scenefile.cpp on hostile text: truncated entries, keywords without values, numbers that are not numbers, very long
 * tokens, entries nested in each other, and random token soup; then the view of a good file built into every accelerator */
static int scene_file_checks()
{
    const char* path = "/tmp/crt_sanitize_scene.txt";
    const char* words[] = { "MATERIAL", "MATTE", "MIRROR", "TRANSPARENT", "EMISSIVE", "PLASTIC", "GLASS", "METAL", "NAME", "a", "COLOR", "RED", "0.5", "SIGMA", "END",
        "END_MATERIALS", "OBJECT", "SPHERE", "PLANE", "RECTANGLE", "TRIANGLE", "BOX", "TORUS", "OPENCYLINDER", "SOLIDCYLINDER", "DISK", "MESH", "RADIUS", "CENTER",
        "POINT", "NORMAL", "WIDTH", "HEIGHT", "LENGTH", "V0", "V1", "V2", "LOCATION", "SCALE", "ORIENTATION", "NORMAL_TYPE", "OPEN", "bogus", "FILE_NAME", "/nonexistent.obj",
        "SMOOTH", "yes", "ENV_LIGHT", "TYPE", "CONSTANT", "TEXTURE", "x.exr", "KD_TEXTURE", "NORMAL_MAP", "/tmp/crt_sanitize_t.ppm", "/tmp/crt_sanitize_bad.ppm",
        "crt_sanitize_t.ppm", "INTENSITY", "-1", "1e39", "nan", "#", "#x", "\n", "\t", "\r\n", "KD", "KS", "ROUGHNESS", "GOLD",
        "WINDOW_WIDTH", "FOV", "CAMERA_POS", "LOOK_POINT", "0", "0 0 0", "1 2", "" };
    const int nw = (int)(sizeof words / sizeof words[0]);
    /* texture files the soup can name: a good 3 x 2 PPM and hostile ones for the reader itself (PPM_read, imagefile.h:59-83) */
    {
        unsigned char px[18]; for (int i = 0; i < 18; i++) px[i] = (unsigned char)(i * 13);
        if (crt_ppm_write("/tmp/crt_sanitize_t.ppm", px, 3, 2) != CRT_OK) { printf("ppm write\n"); return 1; }
        crt_texture t; memset(&t, 0, sizeof t);
        if (crt_texture_load_ppm("/tmp/crt_sanitize_t.ppm", &t) != CRT_OK || t.width != 3 || t.height != 2 || t.comp != 3 || t.is_float ||
            memcmp(t.data, px, 18)) { printf("ppm read back\n"); return 1; }
        crt_texture_free(&t);
        const char* bad[] = { "", "P6", "P6\n", "P6\n3", "P6\n3 2\n", "P6\n3 2\n255\n", "P6\n3 2\n255\nshort", "P6\n-3 2\n255\nxxxxxxxxxxxxxxxxxx",
                              "P6\n100000 100000\n255\nx", "P6\n2147483647 2147483647\n255\nx", "P6\n0 0\n255\n", "P6\n3 2\n65535\nxxxxxxxxxxxxxxxxxx",
                              "P5\n3 2\n255\nxxxxxx", "P6\n3 x\n255\nxxxxxxxxxxxxxxxxxx", "P6 3 2 255 xxxxxxxxxxxxxxxxxx" };
        for (const char* b : bad) {
            if (write_file("/tmp/crt_sanitize_bad.ppm", b)) return 1;
            crt_texture u; memset(&u, 0, sizeof u);
            const int rc = crt_texture_load_ppm("/tmp/crt_sanitize_bad.ppm", &u);
            if (rc == CRT_OK) {                              // whatever it accepts must be consistent and readable to its end
                if (u.width <= 0 || u.height <= 0 || u.comp != 3 || !u.data) { printf("ppm accepted inconsistently: '%s'\n", b); return 1; }
                unsigned sum = 0; for (long long i = 0; i < (long long)u.width * u.height * 3; i++) sum += ((const unsigned char*)u.data)[i];
                (void)sum;
                crt_texture_free(&u);
            } else if (u.data) { printf("ppm: data set on failure\n"); return 1; }
        }
        if (crt_texture_load_ppm(0, &t) != CRT_E_INVALID || crt_texture_load_ppm("/tmp/crt_sanitize_t.ppm", 0) != CRT_E_INVALID) { printf("ppm argument checks\n"); return 1; }
        crt_texture_free(0);
        if (write_file("/tmp/crt_sanitize_bad.ppm", "P6\n3 2\n255\nshort")) return 1;     // what the soup's second name points at
    }
    for (int round = 0; round < 400; round++) {
        std::string text;
        const int len = 1 + rand() % 120;
        for (int i = 0; i < len; i++) { text += words[rand() % nw]; text += (rand() % 5 == 0) ? "\n" : " "; }
        if (round % 50 == 0) text += std::string(5000, 'Z');       // one very long token
        if (write_file(path, text)) return 1;
        crt_scene_file* sf = 0;
        if (crt_scene_file_load(path, &sf) != CRT_OK || !sf) { printf("scene file round %d refused\n", round); return 1; }
        const crt_scene_file_view* v = crt_scene_file_get(sf);
        if (v->input.num_obj < 0 || v->num_materials < 0) { printf("scene file counts\n"); return 1; }
        for (int i = 0; i < v->input.num_obj; i++)
            if (v->obj_material[i] < -1 || v->obj_material[i] >= v->num_materials) { printf("scene file material index\n"); return 1; }
        if (v->num_textures < 0 || v->env_map_tex < -1 || v->env_map_tex >= v->num_textures) { printf("scene file texture counts\n"); return 1; }
        for (int m = 0; m < v->num_materials; m++)
            if (v->material_diffuse_tex[m] < -1 || v->material_diffuse_tex[m] >= v->num_textures ||
                v->material_normal_tex[m] < -1 || v->material_normal_tex[m] >= v->num_textures) { printf("scene file texture index\n"); return 1; }
        for (int t = 0; t < v->num_textures; t++) {
            if (!v->texture_names[t]) { printf("scene file texture name\n"); return 1; }
            const crt_texture& tx = v->textures[t];
            if (tx.data) { unsigned sum = 0; for (long long i = 0; i < (long long)tx.width * tx.height * tx.comp; i++) sum += ((const unsigned char*)tx.data)[i]; (void)sum; }
        }
        char what[64]; snprintf(what, sizeof what, "scene file round %d", round);
        if (v->input.num_obj > 0 && build_all(v->input, what, true)) return 1;     // NaN / zero-extent geometry: the grid may refuse
        crt_scene_file_free(sf);
    }
    crt_scene_file* sf = 0;
    if (crt_scene_file_load(0, &sf) != CRT_E_INVALID || crt_scene_file_load("/nonexistent/scene.txt", &sf) != CRT_E_INVALID ||
        crt_scene_file_load(path, 0) != CRT_E_INVALID || crt_scene_file_get(0) != 0) { printf("scene file argument checks\n"); return 1; }
    crt_scene_file_free(0);
    float r = 0;
    if (crt_env_world_radius(0, &r) != CRT_E_INVALID) { printf("world radius null check\n"); return 1; }
    remove(path); remove("/tmp/crt_sanitize_t.ppm"); remove("/tmp/crt_sanitize_bad.ppm");
    return 0;
}

int main()
{
    const int sizes[] = { 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 29, 30, 31, 32, 33, 63, 64, 65, 127, 1000, 4099 };
    for (int n : sizes) {
        for (int flavour = 0; flavour < 4; flavour++) {
            std::vector<float> v((size_t)n * 9);
            for (int i = 0; i < n; i++) {
                float c[3] = { frand(), frand(), frand() };
                if (flavour == 1) { c[0] = 0.5f; c[1] = (i % 3) * 0.25f; c[2] = 0.5f; }          // many equal centroids
                for (int k = 0; k < 9; k++) v[(size_t)i * 9 + k] = c[k % 3] + (flavour == 2 ? 0.0f : 0.05f * (frand() - 0.5f));   // 2: degenerate points
            }
            if (flavour == 3 && n > 2) { v[4] = NAN; v[9 * (n / 2)] = INFINITY; }
            crt_scene_input in = { n, 0, 0, v.data(), n, 0, 0 };
            char what[64]; snprintf(what, sizeof what, "tris n=%d flavour=%d", n, flavour);
            if (build_all(in, what, flavour >= 2 || n == 1)) return 1;
        }
    }
    {   // mixed: sphere, plane (non-grid), instanced box, triangles
        std::vector<crt_shape> sh(3);
        for (auto& s : sh) { s = crt_shape(); for (int k = 0; k < 16; k++) s.inv_xform[k] = (k % 5 == 0) ? 1.0f : 0.0f; }
        sh[0].type = CRT_SPHERE; sh[0].p[0] = 0; sh[0].p[1] = 1; sh[0].p[2] = 0; sh[0].p[3] = 1; sh[0].p[4] = 0; sh[0].p[5] = 3.14159f; sh[0].p[6] = 3.14159f;
        sh[1].type = CRT_PLANE; sh[1].p[1] = -1; sh[1].p[4] = 1;
        sh[2].type = CRT_AABOX; sh[2].p[0] = sh[2].p[1] = sh[2].p[2] = -0.5f; sh[2].p[3] = sh[2].p[4] = sh[2].p[5] = 0.5f; sh[2].p[6] = 1;
        crt_shape inst = sh[2]; inst.type = CRT_INSTANCED; inst.inner = 2; inst.inv_xform[3] = -2.0f;
        sh.push_back(inst);
        const int nt = 37;
        std::vector<float> v((size_t)nt * 9);
        for (auto& x : v) x = frand() * 4 - 2;
        std::vector<int32_t> type, ref;
        type.push_back(CRT_SPHERE); ref.push_back(0);
        for (int i = 0; i < nt; i++) { type.push_back(i % 3 == 0 ? CRT_TRIANGLE : (i % 3 == 1 ? CRT_FLAT_TRIANGLE : CRT_SMOOTH_TRIANGLE)); ref.push_back(i); }
        type.push_back(CRT_PLANE); ref.push_back(1);
        type.push_back(CRT_INSTANCED); ref.push_back(3);
        crt_scene_input in = { (int)type.size(), type.data(), ref.data(), v.data(), nt, sh.data(), (int)sh.size() };
        if (build_all(in, "mixed")) return 1;
        // error paths
        crt_built_scene* b = 0;
        if (crt_build_scene(0, 3, &b) != CRT_E_INVALID || crt_build_scene(&in, 9, &b) != CRT_E_INVALID || crt_build_scene(&in, 3, 0) != CRT_E_INVALID) { printf("null / accel checks\n"); return 1; }
        ref[1] = nt;                                      // triangle index out of range
        if (crt_build_scene(&in, 3, &b) != CRT_E_INVALID) { printf("tri ref check\n"); return 1; }
        ref[1] = 0; ref.back() = 17;                      // shape index out of range
        if (crt_build_scene(&in, 2, &b) != CRT_E_INVALID) { printf("shape ref check\n"); return 1; }
        ref.back() = 3; sh[3].inner = 3;                  // instance wrapping itself
        if (crt_build_scene(&in, 1, &b) != CRT_E_INVALID) { printf("instance cycle check\n"); return 1; }
        crt_built_scene_destroy(0);
    }
    if (obj_and_mesh_checks()) return 1;
    if (scene_file_checks()) return 1;
    printf("host_sanitize ok\n");
    return 0;
}
