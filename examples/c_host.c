/*
 * c_host.c -- a host program in the reference's own language (C) over include/crt_gpu.h, nothing else: what the
 * reference's main.cpp (:187-368) does from "scene file loaded" to "image written", with the hot path on the GPU.
 *
 *   scene file -> crt_scene_file_load            (loadSceneFile, buildscene.h:389)
 *              -> crt_build_scene_ex             (buildSceneAccel, buildscene.h:698)
 *              -> crt_scene_create               (upload; the SceneObjects handle of scenedata.h:20-28)
 *   n x intersectTest / shadowIntersectTest      -> crt_intersect_batch / crt_shadow_batch          (intersect.h:365, :443)
 *   hit -> ShadeRec                              -> crt_fill_shade_rec                               (util/shaderec.h:7-19)
 *   for p < num_samples: threads, pathTrace      -> crt_renderer_create / _reset / _render / _read  (main.cpp:295-317)
 *   saveImageState, genImageFromColorBuffer, PPM -> crt_image_state_save, crt_image_from_color_buffer, crt_ppm_write
 *
 * Several GPUs (-DWITH_RCCL, one process per GPU): rank k renders the samples [k * spp, (k + 1) * spp) of the whole frame and
 * the colour buffers -- the operand saveImageState writes (imagestate.h:5-19) -- are summed with ncclAllReduce on the
 * buffer crt_renderer_read_dev hands over.  The ncclUniqueId travels through a file rank 0 writes.
 *
 *   cc -std=c99 -Iinclude examples/c_host.c -Lcraytracer_amd/csrc -lcrt_gpu -lm -o c_host
 *   ./c_host tests/golden/scene_all.txt /tmp/out 4                         one GPU
 *   cc -std=gnu99 -DWITH_RCCL -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_host.c -Lcraytracer_amd/csrc -lcrt_gpu -L/opt/rocm/lib -lrccl -lamdhip64 -lm -o c_host_rccl
 *   ./c_host_rccl scene.txt /tmp/out 4 <rank> <nranks> /tmp/nccl.id        one process per GPU
 *
 * Files written (rank 0): <out>.rays (n x 6 float32), <out>.hits (n x crt_hit), <out>.ld (n light distances), <out>.shadow (n x {t, occluded as float}),
 * <out>.is (image state, the reference's format), <out>.ppm.  tests/test_c_host.py compares them with the Python mirror.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crt_gpu.h"

#ifdef WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>
#endif

#define CHECK(call) do { int rc_ = (call); if (rc_ != CRT_OK) { \
    fprintf(stderr, "%s:%d: %s -> %s (%s)\n", __FILE__, __LINE__, #call, crt_strerror(rc_), crt_last_hip_error()); return 1; } } while (0)

/* the LCG of glibc's rand_r, spelled out so that the ray set does not depend on the libc */
static unsigned lcg(unsigned* s) { *s = *s * 1103515245u + 12345u; return (*s >> 8) & 0xFFFFFFu; }
static float unit(unsigned* s) { return (float)lcg(s) * (1.0f / 16777216.0f); }

static int write_file(const char* prefix, const char* ext, const void* data, size_t bytes)
{
    char name[1024];
    snprintf(name, sizeof name, "%s%s", prefix, ext);
    FILE* f = fopen(name, "wb");
    if (!f) return 1;
    const size_t w = fwrite(data, 1, bytes, f);
    fclose(f);
    return w == bytes ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <scene file> <output prefix> [spp [rank nranks id-file]]\n", argv[0]); return 2; }
    const char* scene_path = argv[1];
    const char* out = argv[2];
    const unsigned spp = argc > 3 ? (unsigned)atoi(argv[3]) : 4u;
    const int rank = argc > 6 ? atoi(argv[4]) : 0, nranks = argc > 6 ? atoi(argv[5]) : 1;
    (void)nranks;
    int ndev = 0;
    CHECK(crt_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no gfx950 device: this path has no CPU fallback\n"); return 1; }
    const int device = rank % ndev;

    /* ---- scene: file -> objects in push order -> accelerator -> HBM ---- */
    crt_scene_file* sf = 0;
    CHECK(crt_scene_file_load(scene_path, &sf));
    const crt_scene_file_view* v = crt_scene_file_get(sf);
    crt_built_scene* built = 0;
    CHECK(crt_build_scene_ex(&v->input, CRT_ACCEL_BVH4, CRT_BUILD_SAH, &built));      /* or crt_build_scene: the reference's own splits */
    const crt_scene_desc* desc = crt_built_scene_desc(built);
    crt_scene* scene = 0;
    CHECK(crt_scene_create(device, desc, &scene));

    /* ---- ray queries from host arrays: n x intersectTest, n x shadowIntersectTest ---- */
    const size_t n = 1u << 16;
    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    for (int i = 0; i < v->input.num_tris * 3; i++)
        for (int a = 0; a < 3; a++) {
            const float x = v->input.tri_verts[3 * i + a];
            if (x < lo[a]) lo[a] = x;
            if (x > hi[a]) hi[a] = x;
        }
    if (v->input.num_tris == 0) for (int a = 0; a < 3; a++) { lo[a] = -1.0f; hi[a] = 1.0f; }
    crt_ray* rays = (crt_ray*)malloc(n * sizeof(crt_ray));
    crt_hit* hits = (crt_hit*)malloc(n * sizeof(crt_hit));
    float* light_dist = (float*)malloc(n * sizeof(float));
    float* sh_t = (float*)malloc(n * sizeof(float));
    uint8_t* sh_occ = (uint8_t*)malloc(n);
    float* sh_out = (float*)malloc(2 * n * sizeof(float));
    if (!rays || !hits || !light_dist || !sh_t || !sh_occ || !sh_out) return 1;
    unsigned seed = 12345u;
    for (size_t i = 0; i < n; i++) {
        float d[3], len = 0.0f;
        for (int a = 0; a < 3; a++) rays[i].o[a] = lo[a] + (hi[a] - lo[a]) * unit(&seed);
        do {
            len = 0.0f;
            for (int a = 0; a < 3; a++) { d[a] = 2.0f * unit(&seed) - 1.0f; len += d[a] * d[a]; }
        } while (len > 1.0f || len < 1e-4f);
        len = sqrtf(len);
        for (int a = 0; a < 3; a++) rays[i].d[a] = d[a] / len;
        light_dist[i] = 0.25f * (hi[0] - lo[0]) * (0.5f + unit(&seed));
    }
    CHECK(crt_intersect_batch(scene, rays, n, hits));
    CHECK(crt_shadow_batch(scene, rays, light_dist, n, sh_t, sh_occ));
    size_t n_hit = 0, n_occ = 0;
    for (size_t i = 0; i < n; i++) { n_hit += hits[i].obj >= 0; n_occ += sh_occ[i]; sh_out[2 * i] = sh_t[i]; sh_out[2 * i + 1] = (float)sh_occ[i]; }
    /* the ShadeRec intersectTest would have left behind, for the shading that stays on the host */
    crt_shade_rec sr[4];
    CHECK(crt_fill_shade_rec(&v->input, v->has_mesh_shading ? &v->mesh_shading : 0, v->obj_material, rays, hits, 4, 0, sr));
    if (rank == 0) {
        printf("rays %zu  hits %zu  occluded %zu  first hit: t %.6f obj %d normal (%.4f %.4f %.4f)\n", n, n_hit, n_occ,
               hits[0].t, hits[0].obj, sr[0].normal[0], sr[0].normal[1], sr[0].normal[2]);
        if (write_file(out, ".rays", rays, n * sizeof(crt_ray)) || write_file(out, ".hits", hits, n * sizeof(crt_hit)) ||
            write_file(out, ".ld", light_dist, n * sizeof(float)) || write_file(out, ".shadow", sh_out, 2 * n * sizeof(float))) { fprintf(stderr, "cannot write %s.*\n", out); return 1; }
    }

    /* ---- the sample loop handed over: main.cpp:295-317 ---- */
    if (!v->has_film || !v->has_camera) { fprintf(stderr, "the scene file has no film / camera block\n"); return 1; }
    crt_camera cam;
    crt_film film;
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    CHECK(crt_camera_init(&cam, 0, 0.0f, 0.0f));                                    /* Pinhole */
    CHECK(crt_camera_look_at(&cam, v->camera_pos, v->look_point, up));
    CHECK(crt_film_dimension(&film, v->fov, v->image_width, v->image_height, &cam));
    crt_render_desc rd;
    memset(&rd, 0, sizeof rd);
    rd.objects = &v->input;
    rd.obj_material = v->obj_material;
    rd.materials = v->materials;
    rd.num_materials = v->num_materials;
    rd.has_env_light = v->has_env_light;
    memcpy(rd.env_color, v->env_color, sizeof rd.env_color);
    rd.env_intensity = v->env_intensity;
    if (v->has_env_light) CHECK(crt_env_world_radius(desc, &rd.env_world_radius));  /* preprocessLights, buildscene.h:839-873 */
    rd.max_depth = 5;
    rd.seed = 3;
    rd.mesh_shading = v->has_mesh_shading ? &v->mesh_shading : 0;
    if (v->num_textures > 0 && v->num_external == 0 && v->num_unsupported == 0) {  /* every named texture was a readable PPM file */
        rd.textures = v->textures;
        rd.num_textures = v->num_textures;
        rd.material_diffuse_tex = v->material_diffuse_tex;
    }
    crt_renderer* r = 0;
    CHECK(crt_renderer_create(scene, &rd, &cam, &film, &r));
    CHECK(crt_renderer_reset(r, (uint32_t)rank * spp));                              /* this rank's sample range */
    CHECK(crt_renderer_render(r, spp));
    const size_t npix = (size_t)v->image_width * (size_t)v->image_height;
    float* color = (float*)malloc(npix * 3 * sizeof(float));
    if (!color) return 1;
    uint32_t done = 0;
#ifdef WITH_RCCL
    if (nranks > 1) {
        /* sum of the ranks' colour buffers over RCCL, on the device buffer */
        ncclUniqueId id;
        const char* id_path = argv[6];
        if (rank == 0) {
            if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
            char tmp[1024];
            snprintf(tmp, sizeof tmp, "%s.tmp", id_path);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) return 1;
            fclose(f);
            if (rename(tmp, id_path) != 0) return 1;
        } else {
            FILE* f = 0;
            for (int tries = 0; tries < 600 && !(f = fopen(id_path, "rb")); tries++) usleep(100000);
            if (!f || fread(&id, sizeof id, 1, f) != 1) return 1;
            fclose(f);
        }
        ncclComm_t comm;
        hipStream_t st;
        float* d_sum = 0;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&st) != hipSuccess) return 1;
        if (ncclCommInitRank(&comm, nranks, id, rank) != ncclSuccess) return 1;
        if (hipMalloc((void**)&d_sum, npix * 3 * sizeof(float)) != hipSuccess) return 1;
        CHECK(crt_renderer_read_dev(r, d_sum, st));
        if (ncclAllReduce(d_sum, d_sum, npix * 3, ncclFloat, ncclSum, comm, st) != ncclSuccess) return 1;
        if (hipMemcpyAsync(color, d_sum, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
        if (hipStreamSynchronize(st) != hipSuccess) return 1;
        done = spp * (uint32_t)nranks;
        (void)hipFree(d_sum);
        ncclCommDestroy(comm);
        (void)hipStreamDestroy(st);
    } else
#endif
    {
        CHECK(crt_renderer_read(r, color, &done));
        done = spp;                                                                  /* samples in the buffer (reset set the index, not a count) */
    }
    if (rank == 0) {
        char name[1024];
        snprintf(name, sizeof name, "%s.is", out);
        CHECK(crt_image_state_save(name, color, (int)done, v->image_width, v->image_height));
        unsigned char* image = (unsigned char*)malloc(npix * 3);
        if (!image) return 1;
        CHECK(crt_image_from_color_buffer(image, color, (int)npix, (int)done));
        snprintf(name, sizeof name, "%s.ppm", out);
        CHECK(crt_ppm_write(name, image, v->image_width, v->image_height));
        free(image);
        crt_render_stats stt;
        CHECK(crt_renderer_get_stats(r, &stt));
        printf("image %d x %d, %u spp: %.1f ms, %llu + %llu rays\n", v->image_width, v->image_height, done, stt.ms_total,
               (unsigned long long)stt.rays_closest, (unsigned long long)stt.rays_shadow);
    }
    crt_renderer_destroy(r);
    crt_scene_destroy(scene);
    crt_built_scene_destroy(built);
    crt_scene_file_free(sf);
    free(color); free(rays); free(hits); free(light_dist); free(sh_t); free(sh_occ); free(sh_out);
    return 0;
}
