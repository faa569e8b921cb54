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
 * Algorithmic bytes per ray: 24 (the origin, a whole record because of sectoring) + 2 + 2 (its cell, written by the
 * histogram pass and read by the scatter pass) + 4 (its slot in the order); 2 x 24 + 4 without the cell array.
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
/* k_bin_scatter tags a ray with cell << 16 | rank inside its block's share of the cell; 0xFFFFFFFF means "no ray" */
static_assert(CRT_BIN_CHUNK <= (1 << 16) && CRT_BIN_CELLS <= (1 << 15), "the scatter tag holds a 16-bit rank and a cell below 2^15");
static_assert(CRT_BIN_CELLS % CRT_BIN_BLOCK == 0 && CRT_BIN_CHUNK % CRT_BIN_BLOCK == 0, "whole cells / rays per thread");

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

/* cells (may be null): every ray's cell, 2 bytes, so that the scatter pass reads 2 bytes per ray instead of its 24-byte record */
__global__ __launch_bounds__(CRT_BIN_BLOCK) void k_bin_hist(const crt_ray* rays, unsigned n, const unsigned* n_dev, BinGrid g, unsigned* hist, unsigned short* cells)
{
    if (n_dev) { const unsigned m = *n_dev; n = m < n ? m : n; }     /* counted launch: n is the upper bound */
    __shared__ unsigned h[CRT_BIN_CELLS];
    for (unsigned b = threadIdx.x; b < CRT_BIN_CELLS; b += CRT_BIN_BLOCK) h[b] = 0;
    __syncthreads();
    const unsigned first = blockIdx.x * CRT_BIN_CHUNK;
    for (unsigned k = 0; k < CRT_BIN_PER_THREAD; k++) {
        const unsigned i = first + k * CRT_BIN_BLOCK + threadIdx.x;
        if (i < n) {
            const unsigned c = bin_cell(rays, i, g);
            atomicAdd(&h[c], 1u);
            if (cells) cells[i] = (unsigned short)c;
        }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < CRT_BIN_CELLS; b += CRT_BIN_BLOCK) if (h[b]) atomicAdd(&hist[b], h[b]);
}

/* A slice of the binned batch (crt_intersect_batch_dev_slice): slice k of K is the run of whole cells [c_lo, c_hi) whose queue
 * slots start at the first cell boundary at or after total * k / K and end at the first at or after total * (k + 1) / K.  The
 * boundaries are a function of the cell COUNTS alone, so K processes that bin the same batch on K GPUs cut it at the same
 * places (the order of the rays inside a cell is a matter of timing and differs from run to run; whole cells do not). */
struct BinSlice { unsigned k, count; };          /* count == 0: no slicing */
#define CRT_BIN_RANGE_WORDS 4                     /* range[] = { first slot, end slot, first cell, end cell } */

/* hist[] -> cursor[] = exclusive prefix sums; hist is left zeroed for the next call; range (when slicing): see BinSlice */
__global__ __launch_bounds__(CRT_BIN_BLOCK) void k_bin_scan(unsigned* hist, unsigned* cursor, BinSlice sl, unsigned* range)
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
    if (sl.count) {
        const unsigned total = part[CRT_BIN_BLOCK - 1];
        const unsigned t_lo = (unsigned)((unsigned long long)total * sl.k / sl.count);
        const unsigned t_hi = (unsigned)((unsigned long long)total * (sl.k + 1u) / sl.count);
        if (threadIdx.x == 0) {                       /* defaults: the boundary lies behind the last cell */
            if (t_lo == 0u) { range[0] = 0u; range[2] = 0u; } else { range[0] = total; range[2] = CRT_BIN_CELLS; }
            range[1] = total; range[3] = CRT_BIN_CELLS;
        }
        __syncthreads();
        /* the first cell whose first slot is >= t: exactly one cell has start >= t and the start of the cell before it < t */
        unsigned start = run, prev = run;             /* prev: start of the previous cell (cell 0 has none) */
        if (threadIdx.x > 0) prev = run - hist[threadIdx.x * per - 1];
        for (unsigned k = 0; k < per; k++) {
            const unsigned c = threadIdx.x * per + k;
            const bool first_cell = c == 0u;
            if (t_lo > 0u && start >= t_lo && !first_cell && prev < t_lo) { range[0] = start; range[2] = c; }
            if (sl.k + 1u < sl.count && start >= t_hi && (first_cell ? t_hi == 0u : prev < t_hi)) { range[1] = start; range[3] = c; }
            prev = start; start += v[k];
        }
        __syncthreads();
    }
    for (unsigned k = 0; k < per; k++) { cursor[threadIdx.x * per + k] = run; run += v[k]; hist[threadIdx.x * per + k] = 0; }
}

/* range (slicing): only the rays of cells [range[2], range[3]) get a slot; the others are not this launch's */
__global__ __launch_bounds__(CRT_BIN_BLOCK) void k_bin_scatter(const crt_ray* rays, unsigned n, const unsigned* n_dev, BinGrid g, unsigned* cursor, unsigned* perm,
                                                               const unsigned* range, const unsigned short* cells)
{
    if (n_dev) { const unsigned m = *n_dev; n = m < n ? m : n; }
    const unsigned c_lo = range ? range[2] : 0u, c_hi = range ? range[3] : (unsigned)CRT_BIN_CELLS;
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
            const unsigned c = cells ? (unsigned)cells[i] : bin_cell(rays, i, g);
            if (c >= c_lo && c < c_hi) tag[k] = (c << 16) | atomicAdd(&h[c], 1u);   /* rank < CRT_BIN_CHUNK <= 2^16 */
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
