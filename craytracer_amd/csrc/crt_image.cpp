/*
 * crt_image.cpp -- image state and 8-bit image output, host code.
 *
 * The colour buffer the renderer accumulates (crt_renderer_read) is the reference's color_buffer
 * (main.cpp:139-141): a running sum of per-sample radiance.  What the reference does with it:
 *   saveImageState / readImageState (imagestate.h:5-43)   "<spp>\n<w> <h>\n" + raw float32 sums
 *   genImageFromColorBuffer (util/util.cpp:28-45)          / spp, toneMap, * 255 -> unsigned char
 *   toneMap (shading.h:33-49)                              1 - expf(-2 c), then powf(., 1/2.2)
 *   PPM_write (imagefile.h:36-57)                          binary P6
 * These stay on the host on purpose: they run once per displayed frame, and expf / powf must be the
 * platform libm's for the 8-bit values to equal the reference's.
 */
#include "../../include/crt_gpu.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" {

int crt_image_state_save(const char* file_name, const float* color_buffer, int num_samples, int width, int height)
{
    if (!file_name || !color_buffer || width < 0 || height < 0) return CRT_E_INVALID;
    FILE* fp = fopen(file_name, "w");
    if (!fp) return CRT_E_INVALID;
    fprintf(fp, "%d\n", num_samples);
    fprintf(fp, "%d %d\n", width, height);
    const size_t size = (size_t)width * (size_t)height * 3;
    const size_t wrote = fwrite(color_buffer, sizeof(float), size, fp);
    const int bad = fclose(fp);
    return (wrote == size && !bad) ? CRT_OK : CRT_E_INVALID;
}

int crt_image_state_load(const char* file_name, float** color_buffer, int* num_samples, int* width, int* height)
{
    if (!file_name || !color_buffer || !num_samples || !width || !height) return CRT_E_INVALID;
    *color_buffer = 0;
    FILE* fp = fopen(file_name, "r");
    if (!fp) return CRT_E_INVALID;
    int ns = 0, w = 0, h = 0;
    /* the reference reads with "%d\n" and "%d %d\n": a trailing white-space directive eats ALL following
     * white space, so a buffer whose first bytes are 0x09-0x0d / 0x20 would lose them (imagestate.h:26-27).
     * Here exactly one '\n' is consumed after each line, which is what saveImageState wrote. */
    if (fscanf(fp, "%d", &ns) != 1 || fgetc(fp) != '\n' || fscanf(fp, "%d %d", &w, &h) != 2 || fgetc(fp) != '\n' ||
        w < 0 || h < 0 || (unsigned long long)w * (unsigned long long)h > (1ull << 28)) { fclose(fp); return CRT_E_INVALID; }
    const size_t size = (size_t)w * (size_t)h * 3;
    float* buf = (float*)malloc(size ? size * sizeof(float) : sizeof(float));
    if (!buf) { fclose(fp); return CRT_E_NOMEM; }
    const size_t got = fread(buf, sizeof(float), size, fp);
    fclose(fp);
    if (got != size) { free(buf); return CRT_E_INVALID; }
    *color_buffer = buf; *num_samples = ns; *width = w; *height = h;
    return CRT_OK;
}

void crt_image_state_free(float* color_buffer) { free(color_buffer); }

int crt_image_from_color_buffer(unsigned char* image, const float* color_buffer, int num_pixels, int num_samples)
{
    if (!image || !color_buffer || num_pixels < 0) return CRT_E_INVALID;
    const float inv = 1.0f / (float)num_samples;
    const float exposure = -2.0f;
    for (int i = 0; i < num_pixels; i++) {
        for (int k = 0; k < 3; k++) {
            float c = color_buffer[(size_t)i * 3 + k] * inv;
            c = 1.0f - expf(c * exposure);
            c = powf(c, 1.0f / 2.2f);
            image[(size_t)i * 3 + k] = (unsigned char)(c * 255.0f);
        }
    }
    return CRT_OK;
}

int crt_ppm_write(const char* file_name, const unsigned char* image, int width, int height)
{
    if (!file_name || !image || width < 0 || height < 0) return CRT_E_INVALID;
    FILE* fp = fopen(file_name, "wb");
    if (!fp) return CRT_E_INVALID;
    fprintf(fp, "P6\n");
    fprintf(fp, "%d %d\n", width, height);
    fprintf(fp, "255\n");
    const size_t size = (size_t)width * (size_t)height * 3;
    const size_t wrote = fwrite(image, 1, size, fp);
    const int bad = fclose(fp);
    return (wrote == size && !bad) ? CRT_OK : CRT_E_INVALID;
}

/* PPM_read (imagefile.h:59-83): "P6", width height, maxval, then width*height*3 bytes.  The reference reads the header with
 * fscanf("%d\n"), whose trailing white-space directive also swallows pixel bytes that happen to be 0x09-0x0d / 0x20 at the start
 * of the data (the quirk crt_image_state_load documents for readImageState); here exactly one white-space byte follows maxval,
 * as the format says and as PPM_write / crt_ppm_write leave it.  A short file is an error (the reference prints and goes on). */
int crt_texture_load_ppm(const char* file_name, crt_texture* out)
{
    if (!file_name || !out) return CRT_E_INVALID;
    memset(out, 0, sizeof *out);
    FILE* fp = fopen(file_name, "rb");
    if (!fp) return CRT_E_INVALID;
    char magic[16];
    int w = 0, h = 0, maxval = 0;
    if (fscanf(fp, "%15s", magic) != 1 || strcmp(magic, "P6") != 0 || fscanf(fp, "%d %d", &w, &h) != 2 || fscanf(fp, "%d", &maxval) != 1 ||
        w < 1 || h < 1 || w > 65536 || h > 65536 || maxval != 255) { fclose(fp); return CRT_E_INVALID; }
    const int c = fgetc(fp);                                   /* the single white-space byte after maxval */
    if (!(c == '\n' || c == ' ' || c == '\t' || c == '\r')) { fclose(fp); return CRT_E_INVALID; }
    const size_t size = (size_t)w * (size_t)h * 3;
    unsigned char* buf = (unsigned char*)malloc(size);
    if (!buf) { fclose(fp); return CRT_E_NOMEM; }
    const size_t got = fread(buf, 1, size, fp);
    fclose(fp);
    if (got != size) { free(buf); return CRT_E_INVALID; }
    out->width = w; out->height = h; out->comp = 3; out->is_float = 0; out->data = buf;
    return CRT_OK;
}
void crt_texture_free(crt_texture* t)
{
    if (!t) return;
    free(const_cast<void*>(t->data));
    t->data = 0;
}

} // extern "C"
