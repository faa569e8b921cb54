/* Entry points between the translation units of libcrt_gpu.so; not part of the C ABI (include/crt_gpu.h) and not exported. */
#ifndef CRT_INTERNAL_H
#define CRT_INTERNAL_H
#include "../../include/crt_gpu.h"

#define CRT_HIDDEN __attribute__((visibility("hidden")))

/* crt_intersect_batch_dev_counted / crt_shadow_batch_dev_counted with the ray binning chosen per call (crt_gpu.hip) */
CRT_HIDDEN int crt_internal_intersect_counted(crt_scene* s, const crt_ray* d_rays, const uint32_t* d_n, size_t n_max, crt_hit* d_hits, void* stream,
                                              bool binned);
CRT_HIDDEN int crt_internal_shadow_counted(crt_scene* s, const crt_ray* d_rays, const float* d_light_dist, const uint32_t* d_n, size_t n_max,
                                           float* d_t_out, uint8_t* d_occluded_out, void* stream, bool binned);

#endif
