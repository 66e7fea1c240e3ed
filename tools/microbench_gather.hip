// microbench_gather.hip - what a divergent per-lane record fetch costs on gfx950.
//
// Every lane of every wave follows its own pseudo-random chain through a table of 64-byte records
// (a stand-in for hierarchy nodes: the next index depends on the loaded data, so fetches of one lane
// are serialised as in a traversal).  Variants differ in how the bytes of a record reach the lane:
//   0  4 x 16-byte loads per lane (whole 64-byte record)
//   1  2 x 16-byte loads per lane (first 32 bytes)
//   2  1 x 16-byte load per lane
//   3  quad-cooperative: the 4 lanes of a quad fetch the records of lane 0..3 of the quad in 4 instructions
//      (each instruction touches 16 lines instead of 64) and hand the parts over through LDS
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_gather tools/microbench_gather.hip
// Usage: microbench_gather <records (power of two)> <steps> [waves per CU]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(64) void k_chase(const float4 *__restrict__ table, uint32_t mask, int steps, uint32_t *__restrict__ out) {
    __shared__ float4 xch[VARIANT == 3 ? 4 * 64 : 1];
    const uint32_t lane = threadIdx.x;
    uint32_t idx = (blockIdx.x * 64u + lane) * 2654435761u & mask;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        float4 a, b, c, d;
        if (VARIANT == 0) {
            a = table[4 * idx]; b = table[4 * idx + 1]; c = table[4 * idx + 2]; d = table[4 * idx + 3];
        } else if (VARIANT == 1) {
            a = table[4 * idx]; b = table[4 * idx + 1]; c = a; d = b;
        } else if (VARIANT == 2) {
            a = table[4 * idx]; b = a; c = a; d = a;
        } else {
            const uint32_t part = lane & 3u, quad_base = lane & ~3u;
            float4 p[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t owner_idx = (uint32_t)__shfl((int)idx, (int)(quad_base + k), 64);
                p[k] = table[4 * owner_idx + part];
            }
            // lane `quad_base + k` owns p[k] of the four lanes of its quad: slot [owner][part]
#pragma unroll
            for (int k = 0; k < 4; k++) xch[(quad_base + k) * 4 + part] = p[k];
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): single-wave workgroup, no barrier needed
            a = xch[lane * 4 + 0]; b = xch[lane * 4 + 1]; c = xch[lane * 4 + 2]; d = xch[lane * 4 + 3];
        }
        // every loaded dword is used, so that the compiler keeps the 16-byte loads
        const uint32_t ha = __float_as_uint(a.x) ^ __float_as_uint(a.y) ^ __float_as_uint(a.z) ^ __float_as_uint(a.w);
        const uint32_t hb = __float_as_uint(b.x) ^ __float_as_uint(b.y) ^ __float_as_uint(b.z) ^ __float_as_uint(b.w);
        const uint32_t hc = __float_as_uint(c.x) ^ __float_as_uint(c.y) ^ __float_as_uint(c.z) ^ __float_as_uint(c.w);
        const uint32_t hd = __float_as_uint(d.x) ^ __float_as_uint(d.y) ^ __float_as_uint(d.z) ^ __float_as_uint(d.w);
        acc += __uint_as_float((ha + hb) & 0x3fffffffu);
        idx = (ha + hb * 3u + hc * 5u + hd * 7u + (uint32_t)s) & mask;
    }
    out[blockIdx.x * 64u + lane] = __float_as_uint(acc) ^ idx;
}

int main(int argc, char **argv) {
    const uint32_t records = argc > 1 ? (uint32_t)atol(argv[1]) : 1024u;
    const int steps = argc > 2 ? atoi(argv[2]) : 256;
    const int wpc = argc > 3 ? atoi(argv[3]) : 16;
    const int waves = 256 * wpc;
    std::vector<float> h((size_t)records * 16);
    uint32_t r = 12345u;
    for (float &f : h) { r = r * 1664525u + 1013904223u; uint32_t bits = r; memcpy(&f, &bits, 4); }
    float4 *table;
    uint32_t *out;
    CHECK(hipMalloc(&table, h.size() * 4));
    CHECK(hipMalloc(&out, (size_t)waves * 64 * 4));
    CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int v = 0; v < 3; v++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            switch (v) {
                case 0: hipLaunchKernelGGL(k_chase<0>, dim3(waves), dim3(64), 0, 0, table, records - 1, steps, out); break;
                case 1: hipLaunchKernelGGL(k_chase<1>, dim3(waves), dim3(64), 0, 0, table, records - 1, steps, out); break;
                case 2: hipLaunchKernelGGL(k_chase<2>, dim3(waves), dim3(64), 0, 0, table, records - 1, steps, out); break;
                default: hipLaunchKernelGGL(k_chase<3>, dim3(waves), dim3(64), 0, 0, table, records - 1, steps, out); break;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double fetches = (double)waves * 64 * steps;
        printf("records %u (%u KB) waves/CU %d variant %d: %.3f ms, %.2f G lane-fetches/s, %.1f cycles per wave-step per CU at 2.4 GHz\n",
               records, records / 16, wpc, v, best, fetches / best / 1e6, best * 1e-3 * 2.4e9 / ((double)wpc * steps));
    }
    return 0;
}
