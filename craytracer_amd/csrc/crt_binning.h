/*
 * crt_binning.h -- ray binning ahead of the QBVH walks (no counterpart in the reference, which
 * traces rays in whatever order the pixel loop produces them, main.cpp:88-149).
 *
 * Incoherent rays taken in arrival order touch the whole tree all the time; the same rays taken
 * cell by cell (origin quantised to a 16 x 16 x 16 grid over the scene box, cells in Morton
 * order) keep the nodes near their origins in L2 while that cell is being traced.  Only the ORDER
 * in which rays are pulled from the queue changes: the walk kernels read the ray index from
 * `perm`, trace that ray exactly as before and write its result at that index, so results do
 * not depend on the order (tests/test_gpu_fullsize.py::test_order_independence).
 *
 * One counting sort, three small kernels, HBM-bound:
 *   k_bin_hist    per-block LDS histogram of cell ids, flushed with one atomic per (block, cell)
 *   k_bin_scan    exclusive scan of the CRT_BIN_CELLS counters (one block)
 *   k_bin_scatter per-block LDS ranks + one global atomic per (block, cell) reserving a range
 * Algorithmic bytes per ray: 2 x 24 (origins read twice, whole records because of sectoring) + 4.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/crt_gpu.h"

#define CRT_BIN_BITS   4                       /* per axis */
#define CRT_BIN_CELLS  (1 << (3 * CRT_BIN_BITS))
#ifndef CRT_BIN_BLOCK
#define CRT_BIN_BLOCK  512                     /* 512 x 32 rays per block: 0.41 -> 0.34 ms per 2^24 rays against 256 x 32 (fewer, longer runs per cell and block) */
#endif
#ifndef CRT_BIN_CHUNK
#define CRT_BIN_CHUNK  (32 * CRT_BIN_BLOCK)    /* rays per block (rank inside a block's share of a cell < 2^16) */
#endif
#define CRT_BIN_PER_THREAD (CRT_BIN_CHUNK / CRT_BIN_BLOCK)

struct BinGrid { float lo[3]; float scale[3]; };      /* cell = clamp((o - lo) * scale, 0, 2^bits - 1) */

__device__ __forceinline__ unsigned bin_spread(unsigned v)      /* 4 bits -> every third bit */
{
    v = (v | (v << 4)) & 0x0C3u;
    v = (v | (v << 2)) & 0x249u;
    return v;
}
__device__ __forceinline__ unsigned bin_cell(const crt_ray* rays, unsigned i, const BinGrid& g)
{
    const float* o = rays[i].o;
    const float m = (float)((1 << CRT_BIN_BITS) - 1);
    /* NaN / inf origins land in cell 0 or the last cell: any cell is fine, only the order changes */
    float fx = (o[0] - g.lo[0]) * g.scale[0], fy = (o[1] - g.lo[1]) * g.scale[1], fz = (o[2] - g.lo[2]) * g.scale[2];
    fx = fx > 0.0f ? (fx < m ? fx : m) : 0.0f;
    fy = fy > 0.0f ? (fy < m ? fy : m) : 0.0f;
    fz = fz > 0.0f ? (fz < m ? fz : m) : 0.0f;
    return bin_spread((unsigned)fx) | (bin_spread((unsigned)fy) << 1) | (bin_spread((unsigned)fz) << 2);
}

__global__ __launch_bounds__(CRT_BIN_BLOCK) void k_bin_hist(const crt_ray* rays, unsigned n, const unsigned* n_dev, BinGrid g, unsigned* hist)
{
    if (n_dev) { const unsigned m = *n_dev; n = m < n ? m : n; }     /* counted launch: n is the upper bound */
    __shared__ unsigned h[CRT_BIN_CELLS];
    for (unsigned b = threadIdx.x; b < CRT_BIN_CELLS; b += CRT_BIN_BLOCK) h[b] = 0;
    __syncthreads();
    const unsigned first = blockIdx.x * CRT_BIN_CHUNK;
    for (unsigned k = 0; k < CRT_BIN_PER_THREAD; k++) {
        const unsigned i = first + k * CRT_BIN_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&h[bin_cell(rays, i, g)], 1u);
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < CRT_BIN_CELLS; b += CRT_BIN_BLOCK) if (h[b]) atomicAdd(&hist[b], h[b]);
}

/* hist[] -> cursor[] = exclusive prefix sums; hist is left zeroed for the next call */
__global__ __launch_bounds__(CRT_BIN_BLOCK) void k_bin_scan(unsigned* hist, unsigned* cursor)
{
    __shared__ unsigned part[CRT_BIN_BLOCK];
    const unsigned per = CRT_BIN_CELLS / CRT_BIN_BLOCK;
    unsigned v[per];
    unsigned sum = 0;
    for (unsigned k = 0; k < per; k++) { v[k] = hist[threadIdx.x * per + k]; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned d = 1; d < CRT_BIN_BLOCK; d <<= 1) {
        unsigned add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - sum;
    for (unsigned k = 0; k < per; k++) { cursor[threadIdx.x * per + k] = run; run += v[k]; hist[threadIdx.x * per + k] = 0; }
}

__global__ __launch_bounds__(CRT_BIN_BLOCK) void k_bin_scatter(const crt_ray* rays, unsigned n, const unsigned* n_dev, BinGrid g, unsigned* cursor, unsigned* perm)
{
    if (n_dev) { const unsigned m = *n_dev; n = m < n ? m : n; }
    __shared__ unsigned h[CRT_BIN_CELLS];
    for (unsigned b = threadIdx.x; b < CRT_BIN_CELLS; b += CRT_BIN_BLOCK) h[b] = 0;
    __syncthreads();
    const unsigned first = blockIdx.x * CRT_BIN_CHUNK;
    unsigned tag[CRT_BIN_PER_THREAD];                  /* cell << 16 | rank inside this block's share of the cell */
#pragma unroll
    for (unsigned k = 0; k < CRT_BIN_PER_THREAD; k++) {
        const unsigned i = first + k * CRT_BIN_BLOCK + threadIdx.x;
        tag[k] = 0xFFFFFFFFu;
        if (i < n) {
            const unsigned c = bin_cell(rays, i, g);
            tag[k] = (c << 16) | atomicAdd(&h[c], 1u);   /* rank < CRT_BIN_CHUNK = 2^13 */
        }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < CRT_BIN_CELLS; b += CRT_BIN_BLOCK) if (h[b]) h[b] = atomicAdd(&cursor[b], h[b]);
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < CRT_BIN_PER_THREAD; k++) {
        const unsigned i = first + k * CRT_BIN_BLOCK + threadIdx.x;
        if (tag[k] != 0xFFFFFFFFu) perm[h[tag[k] >> 16] + (tag[k] & 0xFFFFu)] = i;
    }
}
