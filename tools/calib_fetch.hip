// Calibration of rocprofv3 FETCH_SIZE for THIS library's access pattern (MI355X_MICROARCH.md, HBM
// section: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Each lane reads one pseudo-random 128-byte line of a 4 GiB table as 8 x 16 B,
// exactly how a traversal lane fetches a QBVH node, so the known byte count is lanes x 128 and
// nothing can be served from a cache (table >> 256 MiB Infinity Cache, each line touched ~once).
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/calib_fetch
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_calib_node_fetch(const float4* __restrict__ table, uint32_t line_mask, float* out, int rounds)
{
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    uint32_t h = gid * 2654435761u + 12345u;
    for (int r = 0; r < rounds; ++r) {
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float4* p = table + (size_t)(h & line_mask) * 8;
        float4 q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = p[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += q[i].x + q[i].y + q[i].z + q[i].w;
        h += (uint32_t)acc;                  // dependent chain, like a tree walk
    }
    if (acc == 123.456f) out[gid] = acc;
}

int main()
{
    const size_t bytes = 4ull << 30;
    const uint32_t lines = (uint32_t)(bytes / 128);
    float4* table; float* out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 4 << 24) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(table, 0, bytes);
    const int threads = 1 << 22, rounds = 4;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        k_calib_node_fetch<<<threads / 256, 256>>>(table, lines - 1, out, rounds);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("known_bytes_per_launch %zu  ms %.3f  GB/s %.1f\n", (size_t)threads * rounds * 128, ms,
               (double)threads * rounds * 128 / ms / 1e6);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipFree(table); hipFree(out);
    return 0;
}
