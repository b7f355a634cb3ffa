// f32_product_probe.hip — A/B of the three ways to multiply float32-accurate operands on gfx950's matrix cores, on the shape of the
// learner's layer-2 product: Y^T [256][64 samples] = W [256][256] . H^T, one 8-wave workgroup per CU (a wave owns 32 rows x 64
// samples), weights streamed from L2 in fragment order, the activation tile in LDS, ITER products back to back:
//   native   v_mfma_f32_32x32x2_f32 on float32 operands (2 048 MAC in 64 cycles per SIMD)
//   split2   two bf16 planes per operand, 3 x v_mfma_f32_32x32x16_bf16 per 16-deep k-step (16 significant bits)
//   split3   three planes, 6 MFMAs per k-step (24 bits: the float32 GEMM's accuracy) — what PPOConfig(hip_kernels="f32") runs
// Measurement tool (DESIGN.md "float32-accurate operands"), not product code.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/f32_product_probe tools/f32_product_probe.hip && /tmp/f32_product_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 64, K = 256, BM = 64;
constexpr int HS = 264;             // bf16 tile row stride (the learner's kHS)

template <int NS> struct Pairs;
template <> struct Pairs<2> { static constexpr int n = 3; static constexpr int a[3] = {0, 0, 1}, b[3] = {0, 1, 0}; };
template <> struct Pairs<3> { static constexpr int n = 6; static constexpr int a[6] = {0, 0, 1, 0, 2, 1}, b[6] = {0, 1, 0, 2, 0, 1}; };

// split: wpack [NS][8 row blocks][16 k-steps][64 lanes][8] bf16; tile planes in LDS [NS][64][HS]
template <int NS>
__global__ __launch_bounds__(512) void probe_split(const __bf16* __restrict__ wpack, const __bf16* __restrict__ h, float* out, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) __bf16 tile[NS * BM * HS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < NS * BM * HS; i += 512) tile[i] = h[i];
    __syncthreads();
    const int r = lane & 31, hh = lane >> 5;
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        const __bf16* wa = wpack + ((size_t)w * 16) * 512 + lane * 8;
        constexpr int D = 5;                                     // the product kernel's prefetch ring: fragments D - 1 k-steps ahead
        bf16x8 a[D][NS];
#pragma unroll
        for (int p = 0; p < D - 1; ++p)
#pragma unroll
            for (int s = 0; s < NS; ++s) a[p][s] = *reinterpret_cast<const bf16x8*>(wa + (size_t)s * 8 * 16 * 512 + p * 512);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks + D - 1 < 16) {
#pragma unroll
                for (int s = 0; s < NS; ++s) a[(ks + D - 1) % D][s] = *reinterpret_cast<const bf16x8*>(wa + (size_t)s * 8 * 16 * 512 + (ks + D - 1) * 512);
            }
            bf16x8 b[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) b[s][cb] = *reinterpret_cast<const bf16x8*>(tile + s * BM * HS + (32 * cb + r) * HS + 16 * ks + 8 * hh);
#pragma unroll
            for (int p = 0; p < Pairs<NS>::n; ++p)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D][Pairs<NS>::a[p]], b[Pairs<NS>::b[p]][cb], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                   // k-steps stay in order (left alone, hipcc hoists every load of the product and spills)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// native float32: weights [8 row blocks][32 groups of four k-steps][64 lanes][4] float32 (lane = row + 32 (k & 1), four k-steps per
// 16-byte load); the activation tile in LDS as [64 samples][2 parities][128] float32 (+ pad), so that a lane's four k-steps are one
// ds_read_b128
constexpr int FS = 2 * 128 + 4;
__global__ __launch_bounds__(512) void probe_native(const float* __restrict__ wf, const float* __restrict__ h, float* out, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) float tile[BM * FS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < BM * FS; i += 512) tile[i] = h[i];
    __syncthreads();
    const int r = lane & 31, hh = lane >> 5;
    f32x16 acc[2];
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        const float* wa = wf + ((size_t)w * 32) * 256 + lane * 4;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(wa + g * 256);
            f32x4 b[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) b[cb] = *reinterpret_cast<const f32x4*>(tile + (32 * cb + r) * FS + hh * 128 + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[cb][j], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class F>
static void run(const char* name, int mfma_per_product_per_wave, int cycles_per_mfma, F&& launch)
{
    const int blocks = 256;
    unsigned long long* cyc; float* out;
    hipMalloc(&cyc, blocks * sizeof(unsigned long long)); hipMalloc(&out, blocks * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) launch(blocks, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) launch(blocks, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per = (double)h[blocks / 2] / ITER;
    const double ideal = 2.0 * mfma_per_product_per_wave * cycles_per_mfma;     // two waves share a SIMD's matrix pipe
    printf("{\"form\": \"%s\", \"cycles_per_product_median\": %.0f, \"matrix_pipe_cycles_per_product\": %.0f, \"pipe_utilisation\": %.3f, "
           "\"us_per_launch_of_%d_products_per_cu\": %.2f}\n", name, per, ideal, ideal / per, ITER, ms / 10 * 1e3);
    hipFree(cyc); hipFree(out);
}

int main()
{
    __bf16* wp; __bf16* hb; float* wf; float* hf;
    const size_t nw = (size_t)3 * 8 * 16 * 512, nh = (size_t)3 * BM * HS;
    hipMalloc(&wp, nw * 2); hipMalloc(&hb, nh * 2); hipMalloc(&wf, (size_t)8 * 32 * 256 * 4); hipMalloc(&hf, (size_t)BM * FS * 4);
    std::vector<unsigned short> r16(std::max(nw, nh));
    for (size_t i = 0; i < r16.size(); ++i) r16[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22) % 512);      // bf16 values around 0.01
    hipMemcpy(wp, r16.data(), nw * 2, hipMemcpyHostToDevice); hipMemcpy(hb, r16.data(), nh * 2, hipMemcpyHostToDevice);
    std::vector<float> rf((size_t)8 * 32 * 256);
    for (size_t i = 0; i < rf.size(); ++i) rf[i] = 0.01f * (float)((i * 2654435761u >> 20) % 200) - 1.0f;
    hipMemcpy(wf, rf.data(), rf.size() * 4, hipMemcpyHostToDevice); hipMemcpy(hf, rf.data(), (size_t)BM * FS * 4, hipMemcpyHostToDevice);
    run("native v_mfma_f32_32x32x2_f32", 128 * 2, 64, [&](int b, float* o, unsigned long long* c) { probe_native<<<b, 512>>>(wf, hf, o, c); });
    run("split2: 3 x v_mfma_f32_32x32x16_bf16 per k-step", 16 * 6, 32, [&](int b, float* o, unsigned long long* c) { probe_split<2><<<b, 512>>>(wp, hb, o, c); });
    run("split3: 6 x v_mfma_f32_32x32x16_bf16 per k-step", 16 * 12, 32, [&](int b, float* o, unsigned long long* c) { probe_split<3><<<b, 512>>>(wp, hb, o, c); });
    return 0;
}
