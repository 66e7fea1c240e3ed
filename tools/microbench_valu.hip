// microbench_valu.hip - sustained wave64 vector-instruction rate of gfx950 for the instruction kinds a
// traversal consists of (fma, mul/add, min/max, compare+select), at several occupancies.
// Each lane runs 8 independent chains so that no instruction waits for the previous one.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o microbench_valu tools/microbench_valu.hip
// Usage: microbench_valu [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float *__restrict__ out, float a, float b, int iters) {
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = (float)(threadIdx.x + k) * 0.001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) x[k] = __builtin_fmaf(x[k], a, b);
                else if (KIND == 1) x[k] = (r & 1) ? x[k] * a : x[k] + b;
                else if (KIND == 2) x[k] = (r & 1) ? fminf(x[k], x[(k + 1) & 7] + a) : fmaxf(x[k], b);
                else x[k] = (x[k] < x[(k + 3) & 7]) ? x[k] + a : x[(k + 5) & 7];
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));

// Packed forms: two float lanes per instruction (v_pk_fma_f32, v_pk_mul_f32 / v_pk_add_f32), 8 independent chains.
template <int KIND>
__global__ __launch_bounds__(256) void k_valu_pk(float *__restrict__ out, float a, float b, int iters) {
    float2v x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { x[k].x = (float)(threadIdx.x + k) * 0.001f; x[k].y = (float)(threadIdx.x + k) * 0.002f; }
    const float2v a2 = {a, a * 1.0001f}, b2 = {b, b * 0.5f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) x[k] = __builtin_elementwise_fma(x[k], a2, b2);
                else x[k] = (r & 1) ? x[k] * a2 : x[k] + b2;
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k].x + x[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    float *out;
    CHECK(hipMalloc(&out, (size_t)256 * 32 * 64 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[4] = {"fma", "mul/add", "min/max(+add)", "cmp+select(+add)"};
    // instructions per inner statement: fma 1, mul/add 1, min/max: odd r = add+min (2), even r = max (1) -> 1.5,
    // cmp+select: add + cmp + cndmask = 3
    const double per_stmt[4] = {1.0, 1.0, 1.5, 3.0};
    for (int kind = 0; kind < 4; kind++) {
        for (int wpc = 4; wpc <= 32; wpc *= 2) {
            const int blocks = 256 * wpc / 4;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); break;
                    case 1: hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); break;
                    case 2: hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); break;
                    default: hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); break;
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double wave_insts = (double)blocks * 4 * iters * 32 * per_stmt[kind];
            const double cycles = best * 1e-3 * 2.4e9;
            printf("%-18s waves/CU %2d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz), %.1f T lane-ops/s\n",
                   names[kind], wpc, best, cycles * 1024.0 / wave_insts, wave_insts * 64 / best / 1e9);
        }
    }
    for (int kind = 0; kind < 2; kind++) {
        for (int wpc = 4; wpc <= 32; wpc *= 2) {
            const int blocks = 256 * wpc / 4;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k_valu_pk<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
                else hipLaunchKernelGGL(k_valu_pk<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double wave_insts = (double)blocks * 4 * iters * 32;
            const double cycles = best * 1e-3 * 2.4e9;
            printf("%-18s waves/CU %2d: %.3f ms, %.2f cycles per packed wave-instruction per SIMD, %.1f T lane-ops/s\n",
                   kind == 0 ? "pk_fma" : "pk_mul/pk_add", wpc, best, cycles * 1024.0 / wave_insts, wave_insts * 128 / best / 1e9);
        }
    }
    return 0;
}
