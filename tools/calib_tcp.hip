// How many lane-divergent 16-byte accesses per clock can one CU's L1 (TCP) serve when every lane
// reads its own 128-byte line as 7 x global_load_dwordx4 (the QBVH node fetch pattern)?  Two
// footprints: lines that fit the L1 (per-CU 8 KB window) and lines that only fit the L2 (4 MB).
//   hipcc --offload-arch=gfx950 -O3 tools/calib_tcp.hip -o tools/calib_tcp && tools/calib_tcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_tcp(const float4* __restrict__ table, uint32_t line_mask, int rounds, float* out)
{
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = gid * 2654435761u;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        h = h * 1664525u + 1013904223u;
        const float4* p = table + (size_t)((h >> 8) & line_mask) * 8;
        float4 q[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) q[i] = p[i];
#pragma unroll
        for (int i = 0; i < 7; ++i) acc += q[i].x;
        h += (acc != 12345.f);               // dependent chain like a tree walk
    }
    if (acc == 123.456f) out[gid] = acc;
}

// cooperative pattern: G consecutive lanes read the G consecutive 16-byte rows of ONE line
template <int G>
__global__ __launch_bounds__(256) void k_tcp_coop(const float4* __restrict__ table, uint32_t line_mask, int rounds, float* out)
{
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = (gid / G) * 2654435761u;
    const uint32_t row = gid % G;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float4 q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            q[i] = table[(size_t)((h >> 8) & line_mask) * 8 + row];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += q[i].x;
        h += (acc != 12345.f);
    }
    if (acc == 123.456f) out[gid] = acc;
}

template <int G>
void run_coop(const float4* table, float* out, int cus, double clk, uint32_t lines = 32768)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int rounds = 1000, blocks = cus * 4;
    k_tcp_coop<G><<<blocks, 256>>>(table, lines - 1, 10, out);
    (void)hipEventRecord(a);
    k_tcp_coop<G><<<blocks, 256>>>(table, lines - 1, rounds, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double lane_acc = (double)blocks * 256 * rounds * 8;
    printf("coop G=%d (%u lines): lane-loads/clk/CU %.3f  lines/clk/CU %.3f  GB/s %.0f\n", G, lines,
           lane_acc / (ms * 1e-3) / clk / cus, lane_acc / G / (ms * 1e-3) / clk / cus, lane_acc * 16 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = 64u << 20;
    float4* table; float* out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 4 << 22) != hipSuccess) return 1;
    (void)hipMemset(table, 0, bytes);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;                  // Hz
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int rounds = 2000;
    struct { const char* name; uint32_t lines; } cases[] = {
        {"64 lines (8 KB, L1-resident)", 64}, {"4096 lines (512 KB)", 4096}, {"32768 lines (4 MB, L2)", 32768}, {"524288 lines (64 MB, MALL)", 524288} };
    for (auto& c : cases) {
        for (int blocks_per_cu = 2; blocks_per_cu <= 8; blocks_per_cu *= 2) {
            const int blocks = cus * blocks_per_cu;
            k_tcp<<<blocks, 256>>>(table, c.lines - 1, 10, out);
            (void)hipEventRecord(a);
            k_tcp<<<blocks, 256>>>(table, c.lines - 1, rounds, out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            const double acc = (double)blocks * 256 * rounds * 7;
            printf("%-30s waves/CU %2d  ms %.3f  accesses/clk/CU %.3f  GB/s %.0f\n", c.name, blocks_per_cu * 4, ms,
                   acc / (ms * 1e-3) / clk / cus, acc * 16 / (ms * 1e-3) / 1e9);
        }
    }
    run_coop<1>(table, out, cus, clk);
    run_coop<2>(table, out, cus, clk);
    run_coop<4>(table, out, cus, clk);
    run_coop<8>(table, out, cus, clk);
    run_coop<1>(table, out, cus, clk, 64);
    run_coop<2>(table, out, cus, clk, 64);
    run_coop<4>(table, out, cus, clk, 64);
    run_coop<8>(table, out, cus, clk, 64);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    return 0;
}
