// tools/ta_mask_bench.hip -- what does a divergent 16-byte vector load cost the vector-memory path when only some lanes are active?
//
// The node loop of rt_wave.h runs at ~48 % lane utilisation.  Whether raising it would relieve the texture-addresser / L1 path
// depends on how that path charges a partially masked load.  Every lane chases its own pseudo-random 64-byte records of an
// L2-resident table (four dwordx4 loads per record, like a BNode fetch); the active-lane pattern is a launch parameter.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_mask_bench tools/ta_mask_bench.hip && /tmp/ta_mask_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct alignas(16) Rec { float4 a, b, c, d; };

__global__ void __launch_bounds__(256) chase(const Rec* __restrict__ table, uint32_t mask_records, unsigned long long lane_mask, int steps, float* out)
{
    const int lane = threadIdx.x & 63;
    if (!((lane_mask >> lane) & 1ull)) return;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const Rec r = table[idx & mask_records];
        acc += r.a.x + r.b.y + r.c.z + r.d.w;
        idx = idx * 1664525u + 1013904223u + (uint32_t) __float_as_uint(r.d.w);      // next record depends on this one, like a child link
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const uint32_t n_rec = 1u << 15;                 // 2 MiB: resident in every XCD's L2
    std::vector<Rec> host(n_rec);
    for (uint32_t i = 0; i < n_rec; i++) host[i] = Rec{ { 1, 2, 3, 4 }, { 1, 2, 3, 4 }, { 1, 2, 3, 4 }, { 1, 2, 3, 0.f } };
    Rec* table; float* out;
    CHECK(hipMalloc(&table, n_rec * sizeof(Rec)));
    CHECK(hipMalloc(&out, 16));
    CHECK(hipMemcpy(table, host.data(), n_rec * sizeof(Rec), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long mask; } pats[] = {
        { "64 lanes", ~0ull }, { "32: lower half", 0xffffffffull }, { "32: every other lane", 0x5555555555555555ull },
        { "32: every other quad", 0x0f0f0f0f0f0f0f0full }, { "16: first quarter", 0xffffull }, { "16: every fourth lane", 0x1111111111111111ull },
        { "8: every eighth lane", 0x0101010101010101ull } };
    const int steps = 2000, grid = 256 * 20;         // 20 waves per CU's worth of workgroups, like the render kernel's occupancy
    for (auto& p : pats) {
        hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, nullptr, table, n_rec - 1, p.mask, 50, out);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, nullptr, table, n_rec - 1, p.mask, steps, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double waves = (double) grid * 4, active = __builtin_popcountll(p.mask);
        const double wave_records = waves * steps;
        printf("%-24s %8.2f ms  %7.1f G lane-records/s  %6.2f G wave-records/s  %5.1f clk per wave-record per CU (2.4 GHz, 256 CUs)\n", p.name, ms,
               wave_records * active / ms * 1e-6, wave_records / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / wave_records);
    }
    return 0;
}
