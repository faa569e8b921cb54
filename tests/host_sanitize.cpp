// Host-side builder under AddressSanitizer + UBSan (g++, no GPU code): crt_build_scene on triangle
// soups of awkward sizes (around the leaf size 4 and the sort threshold 30), duplicated and
// degenerate triangles, equal centroids, NaN vertices, mixed analytic shapes, every accelerator;
// then the error paths.  Built and run by tests/test_host_sanitize.py.
#include "../include/crt_gpu.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

static unsigned rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (rng_state >> 8) * (1.0f / 16777216.0f); }

/* grid_may_refuse: zero-volume or non-finite bounds have no grid resolution (CRT_E_INVALID) */
static int build_all(const crt_scene_input& in, const char* what, bool grid_may_refuse = false)
{
    for (int accel = 0; accel < 4; accel++) {
        crt_built_scene* b = 0;
        int rc = crt_build_scene(&in, accel, &b);
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
    printf("host_sanitize ok\n");
    return 0;
}
