/*
 * crt_texture.h -- texel lookup and the lat-long environment mapping, host and device.
 *
 * crt_tex_color   getTexColor, texture.cpp:27-86: nearest texel; u, v mirrored about 0, and beyond 1 reflected as
 *                 1 - frac (as written there), v flipped, the fraction rounded up above .5, index modulo the size.
 *                 Both branches divide by 255 (float texels too, texture.cpp:76-78).
 * crt_env_uv      cartesianToSpherical + sphericalToUV, util/math.h:95-107.  The reference calls atan2f / acosf; here
 *                 the double-precision function rounded once, which host and device agree on (same device / host rule
 *                 as for the microfacet transcendentals): a last-ulp difference from glibc's float routines
 *                 can move a direction that falls exactly between two texels to its neighbour.
 * Parity: texture.cpp cannot be compiled here (stb_image.h is not shipped): getTexColor is restated from the text,
 * parity unpinned; the spherical mapping is pinned against util/math.h through oracle/_ref (tests/test_texture.py).
 */
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../include/crt_gpu.h"

#if defined(__HIPCC__)
#define TX_HD __host__ __device__ inline
#else
#define TX_HD inline
#endif

/* texture.cpp:29-46 (u) and :48-68 (v, after the flip) */
TX_HD int crt_tex_index(float x_float, int size, int flip)
{
    if (x_float < 0.0f) x_float = -x_float;
    if (x_float > 1.0f) {
        x_float = x_float - (float)(int)x_float;
        x_float = 1.0f - x_float;
    }
    if (flip) x_float = 1.0f - x_float;
    x_float = x_float * (float)size;
    int x;
    if (x_float - (float)(int)x_float > 0.5f) x = (int)x_float + 1;
    else x = (int)x_float;
    return x % size;
}

TX_HD void crt_tex_color(float* color_out, int width, int height, int comp, int is_float, const void* data, const float* uv)
{
    const int u = crt_tex_index(uv[0], width, 0);
    const int v = crt_tex_index(uv[1], height, 1);
    const size_t at = ((size_t)v * (size_t)width + (size_t)u) * (size_t)comp;
    if (is_float) {
        const float* t = (const float*)data + at;
        color_out[0] = t[0] / 255.0f; color_out[1] = t[1] / 255.0f; color_out[2] = t[2] / 255.0f;
    } else {
        const unsigned char* t = (const unsigned char*)data + at;
        color_out[0] = (float)t[0] / 255.0f; color_out[1] = (float)t[1] / 255.0f; color_out[2] = (float)t[2] / 255.0f;
    }
}

/* util/math.h:95-107; PI is the DOUBLE constant of util/vec.h:6, so the sums and quotients below are taken in double and
 * rounded once when stored into the float, as the C expressions there are */
TX_HD void crt_env_uv(float* uv, const float* dir)
{
    const double pi = 3.14159265358979323846264338327950288;
    float phi = (float)atan2((double)dir[2], (double)dir[0]);      /* atan2f(in[2], in[0]) */
    phi = (float)((double)phi + pi);                               /* out[0] += PI */
    const float y = dir[1] < -1.0f ? -1.0f : (dir[1] > 1.0f ? 1.0f : dir[1]);
    const float theta = (float)acos((double)y);                    /* acosf(clamp(in[1], -1, 1)) */
    uv[0] = (float)((double)phi / (2.0 * pi));                     /* spherical[0] / (2.0f * PI) */
    uv[1] = (float)(1.0 - (double)theta / pi);                     /* 1.0f - spherical[1] / PI */
}
