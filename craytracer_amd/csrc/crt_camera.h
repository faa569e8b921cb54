/*
 * crt_camera.h -- primary-ray set-up, restating camera.cpp of the reference operation by operation
 * (float32, every multiply / add rounded separately; vec3_normalize = scale by 1.0f / (float)sqrt(dot),
 * util/vec.h:206-215).  Shared by the host set-up functions and the device kernel.
 */
#pragma once
#include <math.h>
#include "../../include/crt_gpu.h"

#ifdef __HIPCC__
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD inline
#endif
#include "crt_libm.h"

CRT_HD void cam_normalize(float* r, const float* a)
{
    const float len = (float)sqrt((double)(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]));
    if (len == 0) { r[0] = r[1] = r[2] = 0.0f; return; }
    const float inv = 1.0f / len;
    r[0] = a[0] * inv; r[1] = a[1] * inv; r[2] = a[2] * inv;
}

/* calcImageCoord, camera.cpp:151-157 */
CRT_HD void cam_image_coord(float* ic, const crt_film& film, float sx, float sy, unsigned pixel_index)
{
    ic[0] = -film.frame_length / 2 + film.pixel_length * ((float)(int)(pixel_index % (unsigned)film.frame_res_width) + sx);
    ic[1] = film.frame_height / 2 - film.pixel_length * ((float)(int)(pixel_index / (unsigned)film.frame_res_width) + sy);
}

/* calcRayPinhole, camera.cpp:79-91 */
CRT_HD void cam_ray_pinhole(crt_ray* ray, const float* vp, const crt_camera& c)
{
    float t1[3], t2[3], loc[3], fp[3];
    for (int k = 0; k < 3; k++) { t1[k] = c.x_axis[k] * vp[0]; t2[k] = c.y_axis[k] * vp[1]; }
    for (int k = 0; k < 3; k++) loc[k] = t1[k] + t2[k];
    for (int k = 0; k < 3; k++) fp[k] = c.z_axis[k] * c.focal_pt_dist;
    float d[3];
    for (int k = 0; k < 3; k++) d[k] = loc[k] - fp[k];
    cam_normalize(ray->d, d);
    for (int k = 0; k < 3; k++) ray->o[k] = loc[k] + c.position[k];
}

/* calcRayThinLens, camera.cpp:94-128; mapSampleToDisk, sampling.cpp:606-612 */
CRT_HD void cam_ray_thin_lens(crt_ray* ray, const float* vp, const crt_camera& c, float s0, float s1)
{
    const float phi = (float)(2.0f * M_PI * (double)s0);     /* `2.0f * M_PI * in[0]` is a double product, narrowed on assignment */
    const float radius = sqrtf(s1);
    const float dx = crt_libm_cosf(phi) * radius, dy = crt_libm_sinf(phi) * radius;          /* glibc's values to the bit: crt_libm.h */
    const float ratio = c.focal_length / c.focal_pt_dist;
    const float fpp[3] = { vp[0] * ratio, vp[1] * ratio, -c.focal_length };
    const float o[3] = { dx * c.lens_radius, dy * c.lens_radius, c.focal_pt_dist };
    float d[3] = { fpp[0] - o[0], fpp[1] - o[1], fpp[2] - o[2] };
    float dn[3];
    cam_normalize(dn, d);
    float a[3], b[3], e[3];
    for (int k = 0; k < 3; k++) { a[k] = c.x_axis[k] * dn[0]; b[k] = c.y_axis[k] * dn[1]; e[k] = c.z_axis[k] * dn[2]; }
    for (int k = 0; k < 3; k++) { a[k] = a[k] + b[k]; ray->d[k] = a[k] + e[k]; }
    for (int k = 0; k < 3; k++) { a[k] = c.x_axis[k] * o[0]; b[k] = c.y_axis[k] * o[1]; e[k] = c.z_axis[k] * o[2]; }
    for (int k = 0; k < 3; k++) { a[k] = a[k] + b[k]; a[k] = a[k] + e[k]; ray->o[k] = a[k] + c.position[k]; }
}

CRT_HD void cam_primary_ray(crt_ray* ray, const crt_camera& c, const crt_film& f, unsigned pixel_index,
                            float isx, float isy, float lsx, float lsy)
{
    float vp[2];
    cam_image_coord(vp, f, isx, isy, pixel_index);
    if (c.type == 0) cam_ray_pinhole(ray, vp, c);           /* calcCameraRay, camera.cpp:130-141 */
    else             cam_ray_thin_lens(ray, vp, c, lsx, lsy);
}

/* ---- host set-up ---- */
inline void cam_init(crt_camera* c, int type, float focal_length, float lens_radius)
{   /* camera.cpp:21-51 */
    const float z3[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) { c->position[k] = z3[k]; c->x_axis[k] = k == 0; c->y_axis[k] = k == 1; c->z_axis[k] = k == 2; }
    c->focal_pt_dist = 0.035f;
    c->focal_point[0] = 0.0f; c->focal_point[1] = 0.0f; c->focal_point[2] = c->focal_pt_dist;
    c->type = 0; c->focal_length = 0.0f; c->lens_radius = 0.0f;
    if (type == 1) { c->type = 1; c->focal_length = focal_length; c->lens_radius = lens_radius; }
}
inline void cam_look_at(crt_camera* c, const float* position, const float* look, const float* up)
{   /* camera.cpp:53-67 */
    float v[3] = { look[0] - position[0], look[1] - position[1], look[2] - position[2] }, lv[3];
    cam_normalize(lv, v);
    for (int k = 0; k < 3; k++) c->z_axis[k] = -lv[k];
    float x[3] = { c->z_axis[1]*up[2] - c->z_axis[2]*up[1], c->z_axis[2]*up[0] - c->z_axis[0]*up[2], c->z_axis[0]*up[1] - c->z_axis[1]*up[0] };
    cam_normalize(c->x_axis, x);
    for (int k = 0; k < 3; k++) c->x_axis[k] = -c->x_axis[k];
    c->y_axis[0] = c->z_axis[1]*c->x_axis[2] - c->z_axis[2]*c->x_axis[1];
    c->y_axis[1] = c->z_axis[2]*c->x_axis[0] - c->z_axis[0]*c->x_axis[2];
    c->y_axis[2] = c->z_axis[0]*c->x_axis[1] - c->z_axis[1]*c->x_axis[0];
    for (int k = 0; k < 3; k++) { c->focal_point[k] = position[k] + c->z_axis[k] * c->focal_pt_dist; c->position[k] = position[k]; }
}
inline void cam_set_focal_dist(crt_camera* c, float d)
{   /* camera.cpp:70-76 */
    c->focal_pt_dist = d;
    for (int k = 0; k < 3; k++) c->focal_point[k] = c->position[k] + c->z_axis[k] * d;
}
inline void cam_film_dimension(crt_film* f, float fov, int w, int h, const crt_camera* c)
{   /* camera.cpp:143-148: sin() in double on the float argument fov/2.0f */
    f->fov = fov; f->frame_res_width = w; f->frame_res_height = h;
    f->frame_length = (float)(2.0 * (sin((double)(fov / 2.0f)) * (double)c->focal_pt_dist));   /* one rounding, at the assignment */
    f->frame_height = f->frame_length * (float)h / (float)w;
    f->pixel_length = (float)f->frame_length / (float)w;
}
