// valu_issue_probe.hip — what one SIMD of gfx950 retires per cycle for the instruction kinds the dynamics kernels are
// made of, at one, two and four waves per SIMD.  Measurement tool ("lane-op roofline"), not product
// code.  Build + run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip && /tmp/valu_issue_probe
// Each kernel runs ITER iterations of 32 independent instructions (8 accumulators x 4) between two s_memtime stamps;
// the result is shader cycles per wave-instruction as seen by ONE wave (median over waves), and the SIMD's aggregate
// cycles per instruction = that / waves per SIMD.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ITER = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long* cyc, float seed)
{
    float a[8], b = seed + threadIdx.x, c = seed * 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], pb = {b, c}, pc2 = {c, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * i; p[i] = (f2){seed * i, seed + i}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc2));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 5) {   // the dynamics mix: two scalar FMAs per packed one
#define X(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %4, %5, %1" : "+v"(a[i]), "+v"(p[i]) : "v"(b), "v"(c), "v"(pb), "v"(pc2));
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (KIND == 7) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(p[i]) : "v"(pb));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_instr_in_asm)
{
    float* out; unsigned long long* cyc;
    const int maxb = 8 * 1024;
    hipMalloc(&out, maxb * 64 * sizeof(float));
    hipMalloc(&cyc, maxb * sizeof(unsigned long long));
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 1024 * wps;
        for (int rep = 0; rep < 3; ++rep) probe<KIND><<<blocks, 64>>>(out, cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double n = (double)ITER * 32 * per_instr_in_asm;
        const double med = h[blocks / 2] / n, lo = h[0] / n, hi = h[blocks - 1] / n;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_one_wave_median\": %.3f, \"min\": %.3f, \"max\": %.3f, "
               "\"simd_cycles_per_instr\": %.3f}\n", name, wps, med, lo, hi, med / wps);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 1);
    run<2>("v_pk_mul_f32", 1);
    run<3>("v_mov_b32_dpp", 1);
    run<4>("v_fmac_f32_dpp", 1);
    run<5>("mix: v_fma_f32 + v_pk_fma_f32", 2);
    run<6>("v_rcp_f32", 1);
    run<7>("v_fma_f64", 1);
    return 0;
}
