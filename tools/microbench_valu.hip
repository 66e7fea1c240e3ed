// microbench_valu.hip - what one SIMD of gfx950 sustains in wave64 vector instructions, measured at the ISA level.
//
// Every kernel times one loop whose body is a block of inline assembly: 64 vector instructions of one kind on 8
// independent register chains (no instruction waits for the one before it).  The instruction count is therefore
// fixed by the source text and can be checked in the disassembly (tools/valu_ceiling.sh counts the mnemonics of each
// kernel with llvm-objdump).  Cycles come from s_memtime (shader clock) read by every wave around its loop, the
// clock frequency from s_memrealtime (100 MHz) read at the same two points.  A workgroup of 256 * w threads with
// more than half of a CU's LDS is the only one on its CU and puts exactly w waves on each of the 4 SIMDs
// (w = 1, 2, 4); w = 8 uses two workgroups of 1024 threads per CU.
//
// Output per instruction kind and w: cycles per instruction as one wave sees it, cycles per instruction per SIMD
// (the first divided by w), the clock, and the chip-wide rate that follows for 1024 SIMDs.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_valu tools/microbench_valu.hip
// Usage: microbench_valu [iterations]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

// One line per chain; R8(X) = 8 instructions, one per chain; the block is 8 x R8 = 64 instructions.
#define CH(op, k) op " %" #k ", %" #k ", %8, %9\n"
#define R8(op) CH(op, 0) CH(op, 1) CH(op, 2) CH(op, 3) CH(op, 4) CH(op, 5) CH(op, 6) CH(op, 7)
#define CH2(op, k) op " %" #k ", %" #k ", %8\n"
#define R8_2(op) CH2(op, 0) CH2(op, 1) CH2(op, 2) CH2(op, 3) CH2(op, 4) CH2(op, 5) CH2(op, 6) CH2(op, 7)
#define X8(s) s s s s s s s s
#define X4(s) s s s s

enum Kind { K_FMA = 0, K_MUL_ADD, K_MIN_MAX, K_MAX3, K_CMP_SEL, K_MOV, K_PK_FMA, K_PK_MUL, K_PK_ADD,
            K_INT_ADD, K_LOGIC, K_SHIFT, K_CMP_SGPR, K_SEL_SGPR, K_RCP, K_CVT, K_MUL_LO, K_FMA_MIN, K_FMA_AND, K_FMA_SEL, K_BPERMUTE, K_COUNT };

struct Stamp { unsigned long long cycles, real, start, end; unsigned int hw_id, xcc_id; };

template <int KIND>
__global__ void k_valu(Stamp *__restrict__ out, float *__restrict__ sink, float a, float b, int iters) {
    extern __shared__ float lds_claim[];
    float x0 = threadIdx.x * 0.001f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f,
          x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4},
            p7 = {x7, x6};
    const float2v a2 = {a, a}, b2 = {b, b};
    const int lane4 = (int)((threadIdx.x * 4u + 4u) & 255u);
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        if (KIND == K_FMA) {
            asm volatile(X8(R8("v_fma_f32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_MUL_ADD) {
            asm volatile(X4(R8_2("v_mul_f32") R8_2("v_add_f32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_MIN_MAX) {
            asm volatile(X4(R8_2("v_min_f32") R8_2("v_max_f32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_MAX3) {
            asm volatile(X4(R8("v_max3_f32") R8("v_min3_f32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_CMP_SEL) {
            // 32 compares and 32 selects; each select reads the mask its compare has just written
#define CS(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n v_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
            asm volatile(X4(CS(0) CS(1) CS(2) CS(3) CS(4) CS(5) CS(6) CS(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
#undef CS
        } else if (KIND == K_MOV) {
#define MV(k) "v_mov_b32 %" #k ", %8\n"
            asm volatile(X8(MV(0) MV(1) MV(2) MV(3) MV(4) MV(5) MV(6) MV(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#undef MV
        } else if (KIND == K_PK_FMA) {
            asm volatile(X8(R8("v_pk_fma_f32"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a2), "v"(b2));
        } else if (KIND == K_INT_ADD) {
            asm volatile(X8(R8_2("v_add_u32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_LOGIC) {
            asm volatile(X4(R8_2("v_and_b32") R8_2("v_or_b32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_SHIFT) {
#define SH(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
            asm volatile(X8(SH(0) SH(1) SH(2) SH(3) SH(4) SH(5) SH(6) SH(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#undef SH
        } else if (KIND == K_CMP_SGPR) {
            // compares that write their mask to a scalar register pair (what a ballot is)
#define CG(k) "v_cmp_lt_f32 s[20:21], %" #k ", %8\n"
            asm volatile(X8(CG(0) CG(1) CG(2) CG(3) CG(4) CG(5) CG(6) CG(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
#undef CG
        } else if (KIND == K_SEL_SGPR) {
            // selects by a mask that sits in a scalar register pair (not written inside the loop)
#define SG(k) "v_cndmask_b32 %" #k ", %" #k ", %9, s[20:21]\n"
            asm volatile("s_mov_b64 s[20:21], 0x5555\n" X8(SG(0) SG(1) SG(2) SG(3) SG(4) SG(5) SG(6) SG(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
#undef SG
        } else if (KIND == K_RCP) {
#define RC(k) "v_rcp_f32 %" #k ", %" #k "\n"
            asm volatile(X8(RC(0) RC(1) RC(2) RC(3) RC(4) RC(5) RC(6) RC(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#undef RC
        } else if (KIND == K_CVT) {
#define CV(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
            asm volatile(X8(CV(0) CV(1) CV(2) CV(3) CV(4) CV(5) CV(6) CV(7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#undef CV
        } else if (KIND == K_MUL_LO) {
            asm volatile(X8(R8_2("v_mul_lo_u32"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == K_FMA_MIN) {
            // mixes: do the fused-multiply-add pipe and the other vector pipe overlap?
#define FM(k, j) "v_fma_f32 %" #k ", %" #k ", %8, %9\n v_min_f32 %" #j ", %" #j ", %8\n"
            asm volatile(X8(FM(0, 4) FM(1, 5) FM(2, 6) FM(3, 7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#undef FM
        } else if (KIND == K_FMA_AND) {
#define FA(k, j) "v_fma_f32 %" #k ", %" #k ", %8, %9\n v_and_b32 %" #j ", %" #j ", %8\n"
            asm volatile(X8(FA(0, 4) FA(1, 5) FA(2, 6) FA(3, 7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
#undef FA
        } else if (KIND == K_FMA_SEL) {
#define FS(k, j) "v_fma_f32 %" #k ", %" #k ", %8, %9\n v_cndmask_b32 %" #j ", %" #j ", %9, s[20:21]\n"
            asm volatile("s_mov_b64 s[20:21], 0x5555\n" X8(FS(0, 4) FS(1, 5) FS(2, 6) FS(3, 7))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s20", "s21");
#undef FS
        } else if (KIND == K_BPERMUTE) {
            // cross-lane reads through the LDS crossbar; the address operand is the lane's own byte offset
#define BP(k) "ds_bpermute_b32 %" #k ", %10, %" #k "\n"
            asm volatile(X8(BP(0) BP(1) BP(2) BP(3) BP(4) BP(5) BP(6) BP(7)) "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(lane4));
#undef BP
        } else if (KIND == K_PK_MUL) {
            asm volatile(X8(R8_2("v_pk_mul_f32"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a2), "v"(b2));
        } else {
            asm volatile(X8(R8_2("v_pk_add_f32"))
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(a2), "v"(b2));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const float s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)) + (p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
    if (s == 12345.678f) sink[0] = s + lds_claim[threadIdx.x & 3];     // keeps the chains alive; never true in practice
    if ((threadIdx.x & 63) == 0) {
        Stamp st;
        st.cycles = t1 - t0;
        st.real = r1 - r0;
        st.start = r0;
        st.end = r1;
        st.hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, bits 0..31
        st.xcc_id = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
    }
}

typedef void (*KernelFn)(Stamp *, float *, float, float, int);

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const char *names[K_COUNT] = {"v_fma_f32", "v_mul_f32 / v_add_f32", "v_min_f32 / v_max_f32", "v_max3_f32 / v_min3_f32",
                                  "v_cmp_lt_f32 + v_cndmask_b32", "v_mov_b32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                  "v_add_u32", "v_and_b32 / v_or_b32", "v_lshlrev_b32", "v_cmp_lt_f32 -> sgpr pair", "v_cndmask_b32 (sgpr mask)",
                                  "v_rcp_f32", "v_cvt_f32_u32", "v_mul_lo_u32", "v_fma_f32 + v_min_f32", "v_fma_f32 + v_and_b32",
                                  "v_fma_f32 + v_cndmask_b32", "ds_bpermute_b32"};
    const KernelFn fns[K_COUNT] = {k_valu<K_FMA>, k_valu<K_MUL_ADD>, k_valu<K_MIN_MAX>, k_valu<K_MAX3>, k_valu<K_CMP_SEL>,
                                   k_valu<K_MOV>, k_valu<K_PK_FMA>, k_valu<K_PK_MUL>, k_valu<K_PK_ADD>,
                                   k_valu<K_INT_ADD>, k_valu<K_LOGIC>, k_valu<K_SHIFT>, k_valu<K_CMP_SGPR>, k_valu<K_SEL_SGPR>,
                                   k_valu<K_RCP>, k_valu<K_CVT>, k_valu<K_MUL_LO>, k_valu<K_FMA_MIN>, k_valu<K_FMA_AND>,
                                   k_valu<K_FMA_SEL>, k_valu<K_BPERMUTE>};
    const int lane_ops[K_COUNT] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};    // float results per lane and instruction
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs; %d loop iterations x 64 instructions per wave\n", prop.gcnArchName, cus, iters);
    printf("# w = waves per SIMD (4 SIMDs per CU, every CU loaded)\n");
    printf("# cyc/inst per SIMD = all instructions of the launch over the time from the first wave's start to the last wave's end\n"
           "#   (s_memrealtime, 100 MHz), in shader clocks (clock = s_memtime ticks per s_memrealtime tick of the median wave: s_memtime\n"
           "#   counts shader clocks on gfx950), per SIMD.  This is the rate a SIMD sustains; the rates in the next columns follow from it.\n"
           "# one wave's cyc/inst = s_memtime ticks the median wave spent per instruction of its own loop.  It is NOT w times the figure\n"
           "#   before: the waves of a SIMD do not share its issue slots evenly for the whole of their loops (the oldest wave is served\n"
           "#   first and ends early, younger ones end late), so dividing one wave's time by w overstates what the SIMD issues.\n"
           "# 'SIMDs' = distinct (XCC, SE, CU, SIMD) ids the waves report, 'w seen' = largest number of waves whose loops overlap on one SIMD\n");
    printf("%-30s %2s %20s %10s %16s %14s %10s %6s %7s %22s\n", "instruction", "w", "cyc/inst per SIMD", "clock MHz",
           "G wave-inst/s", "T lane-op/s", "span ms", "SIMDs", "w seen", "one wave's cyc/inst");
    Stamp *d_out;
    float *d_sink;
    const int max_waves = cus * 32;
    CHECK(hipMalloc(&d_out, sizeof(Stamp) * max_waves));
    CHECK(hipMalloc(&d_sink, 64));
    std::vector<Stamp> h(max_waves);
    for (int kind = 0; kind < K_COUNT; kind++) {
        CHECK(hipFuncSetAttribute((const void *)fns[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        for (int w = 1; w <= 8; w *= 2) {
            const int threads = w <= 4 ? 256 * w : 1024;
            const int blocks = w <= 4 ? cus : 2 * cus;
            const size_t lds = w <= 4 ? 100 * 1024 : 64 * 1024;      // one (two) workgroups per CU and no more
            for (int rep = 0; rep < 2; rep++) {                      // the second launch is the one reported
                hipLaunchKernelGGL(fns[kind], dim3(blocks), dim3(threads), lds, 0, d_out, d_sink, 1.0001f, 0.5f, iters);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
            }
            const int n_waves = blocks * threads / 64;
            CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * n_waves, hipMemcpyDeviceToHost));
            std::vector<double> cyc(n_waves), mhz(n_waves);
            for (int i = 0; i < n_waves; i++) {
                cyc[i] = (double)h[i].cycles / ((double)iters * 64.0);
                mhz[i] = (double)h[i].cycles / (double)h[i].real * 100.0;
            }
            std::sort(cyc.begin(), cyc.end());
            std::sort(mhz.begin(), mhz.end());
            const double c = cyc[n_waves / 2], f = mhz[n_waves / 2];

            // whole-launch view: all instructions over the time from the first start to the last end
            unsigned long long first = ~0ull, last = 0;
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> per_simd_events;
            for (int i = 0; i < n_waves; i++) {
                first = std::min(first, h[i].start);
                last = std::max(last, h[i].end);
                // HW_ID (gfx9): simd_id bits 5:4, cu_id 11:8, sh_id 12, se_id 15:13
                const unsigned long long key = ((unsigned long long)(h[i].xcc_id & 0xF) << 16) | (h[i].hw_id & 0xFF30u);
                per_simd_events[key].push_back({h[i].start, +1});
                per_simd_events[key].push_back({h[i].end, -1});
            }
            int w_seen = 0;
            for (auto &kv : per_simd_events) {
                std::sort(kv.second.begin(), kv.second.end());
                int cur = 0;
                for (auto &ev : kv.second) { cur += ev.second; w_seen = std::max(w_seen, cur); }
            }
            const double span_s = (double)(last - first) / 100e6;
            const double total_inst = (double)n_waves * iters * 64.0;
            const double span_cyc_per_inst_simd = span_s * f * 1e6 * 4.0 * cus / total_inst;
            const double rate = total_inst / span_s;                 // wave-instructions per second, whole chip
            printf("%-30s %2d %20.3f %10.0f %16.1f %14.2f %10.3f %6zu %7d %22.3f\n", names[kind], w, span_cyc_per_inst_simd, f, rate / 1e9,
                   rate * 64.0 * lane_ops[kind] / 1e12, span_s * 1e3, per_simd_events.size(), w_seen, c);
        }
    }
    return 0;
}
