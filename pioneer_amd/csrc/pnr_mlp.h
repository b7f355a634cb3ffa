// pnr_mlp.h — the PPO host driver's two MLPs (policy 137-256-256-12, value 137-256-256-1, tanh; the nets of
// pioneer/launch/pioneer_knm_train.py:59-61 under RLlib's FullyConnectedNetwork with vf_share_layers False) as
// hand-written bf16 MFMA kernels for gfx950: forward (three layers fused, activations never leave the CU between
// layers), backward-data (two layers fused), weight gradients (split over the batch axis into per-slice slabs,
// reduced in a fixed order: deterministic) and the weight packing.  Master weights stay float32 in the caller's
// tensors; bf16 copies (K padded 137 -> 144, head rows padded to 16, plus the transposes the backward pass reads)
// are packed once per update.
//
// Orientation.  Every GEMM is computed TRANSPOSED, Y^T = W . X^T, with the weight matrix as the MFMA's A operand
// (rows = output features; fragments are 16-byte loads straight from the packed row-major weights in L2) and the
// activation tile as the B operand (columns = samples; fragments are ds_read_b128 from a row-major [sample][feature]
// LDS tile).  A 32x32 accumulator block then holds, per lane, ONE sample (its column) and FOUR CONSECUTIVE features
// per register quad (rows (reg&3) + 8 (reg>>2) + 4 (lane>>5)): bias + tanh run in registers and each quad leaves as
// one 8-byte ds_write_b64 into the next layer's row-major tile — no 2-byte LDS stores, no lane shuffles.
// Weight gradients sum over SAMPLES, so both operands must be sample-contiguous per lane: they are read from the
// row-major LDS tiles with ds_read_b64_tr_b16 (the hardware 4x16 transpose read), rows strided 16 dwords mod 64 so
// that the transposed reads are bank-conflict-free.
//
// Work per sample and net: forward 106 496 MAC, backward-data 69 632 MAC, weight gradients 110 592 MAC.  These are
// 256-wide layers on 137-float inputs: measured (profiles/r02_*) the kernels sit at 0.6-0.8 PFLOP/s
// of the ~2.5 dense bf16 peak and 3-5 TB/s of HBM — bound by the chain of dependent products per tile and by the bytes
// of the saved activations, not by the matrix cores.
//
// Kernels here: mlp_pack_kernel, mlp_forward_kernel<FUSED> (FUSED: + the tile's loss and backward-data), mlp_gather_kernel and
// record_pack_kernel (an epoch's shuffle applied once), mlp_backward_data_kernel, mlp_wgrad_kernel, mlp_reduce_kernel,
// mlp_reduce_flat_kernel, mlp_adam_kernel (+ the update's loss means).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// The TU is compiled -ffp-contract=off for the bit-exact env integrator; nothing here needs that: the epilogues
// (bias + tanh, (1 - h^2) dz, the filter, Adam) are a quarter of the fused kernel's issue slots and fuse into FMAs.
#pragma clang fp contract(fast)

namespace pnr {

constexpr int kMlpIn = 137;       // observation entries (pioneer_knm_env.py:194-211)
constexpr int kMlpInPad = 144;    // K of the first layer, padded to a multiple of 16 (zero columns)
constexpr int kMlpHid = 256;      // fcnet_hiddens [256, 256] (pioneer_knm_train.py:60)
constexpr int kMlpAct = 6;        // action dimensions (means 0..5, log-stds 6..11 of the policy head)
constexpr int kMlpHead = 16;      // head rows: 12 (6 means + 6 log-stds) or 1 (value), zero-padded to 16
constexpr int kMlpNets = 2;       // policy, value
#ifndef PNR_MLP_BM
#define PNR_MLP_BM 64
#endif
constexpr int kMlpBM = PNR_MLP_BM;   // samples per workgroup tile (forward / backward-data): 64 -> 53 / 71 KB of LDS and <= 256
                                     // registers, i.e. two workgroups per CU whose phases overlap; 128 -> one (A/B in DESIGN.md)
constexpr int kMlpCB = kMlpBM / 32;  // 32-sample column blocks per tile
static_assert(kMlpBM == 64 || kMlpBM == 128, "tile heights 64 and 128 are implemented");
// PNR_MLP_DIAG: timing-only ablations of the forward kernel (results are wrong when set; tools/mlp_ablation.py): 1 no tanh,
// 2 no observation loads, 4 / 8 no layer-2 / layer-1 product, 16 no head, 32 no forward epilogues, 64 no tile stores (h1, h2,
// dz2, dz1), 128 no H1 reload, 256 no loss (record loads and arithmetic), 512 half the products' sample-fragment LDS reads
#ifndef PNR_MLP_RING
#define PNR_MLP_RING 5            // depth of the weight-fragment prefetch ring (A/B: tools/mlp_variant_ab.py)
#endif
#ifndef PNR_MLP_DIAG
#define PNR_MLP_DIAG 0
#endif
#ifndef PNR_MLP_LDS_PAD
#define PNR_MLP_LDS_PAD 0          // bf16 elements of unused LDS in the fused kernel: 20000 leaves ONE workgroup per CU (occupancy probe, tools/stamps.sh)
#endif
constexpr int kMlpThreads = 256;  // four waves
// PNR_MLP_STAMPS=1 (a diagnostic variant, tools/mlp_stamps.py): every wave of the fused kernel writes s_memtime at its phase
// boundaries into a buffer of its own (no output depends on it).  The product build carries none of it.
#ifndef PNR_MLP_STAMPS
#define PNR_MLP_STAMPS 0
#endif
constexpr int kMlpStampSlots = 26;   // 0..22 phase boundaries (s_memtime), 23 HW_REG_XCC_ID << 32 | HW_REG_HW_ID, 24 / 25 s_memrealtime (100 MHz) at start / end
#if PNR_MLP_STAMPS
// The stamps wait in LDS and leave for global memory at the kernel's end: written to global memory where they are taken (r03b - r03i),
// every stamp was a store that the next s_waitcnt vmcnt(..) of the wave — the weight ring's, in order — also waited for, i.e. the
// instrument stretched exactly the phases it was pointed at (found on mlp_wgrad_kernel, r03i).
#define MLP_STAMP_DECL __shared__ unsigned long long stamp_lds_[kFwdWaves][kMlpStampSlots]
#define MLP_STAMP(i) do { if (P.stamps && lane == 0) { stamp_lds_[w][(i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0) { stamp_lds_[w][24] = __builtin_amdgcn_s_memrealtime(); \
        stamp_lds_[w][23] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | __builtin_amdgcn_s_getreg(0xF804); }   /* XCC_ID | HW_ID: where the wave runs */ \
    if ((i) == 22) stamp_lds_[w][25] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define MLP_STAMP_FLUSH do { if (P.stamps && lane < kMlpStampSlots) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    P.stamps[(((size_t)blockIdx.x * stamp_ny_ + stamp_yi_) * kFwdWaves + w) * kMlpStampSlots + lane] = stamp_lds_[w][lane]; } } while (0)
#else
#define MLP_STAMP_DECL
#define MLP_STAMP(i) do { } while (0)
#define MLP_STAMP_FLUSH do { } while (0)
#endif

// packed bf16 weights of ONE net, element offsets.  Every matrix is stored FRAGMENT-NATIVE: the 32 x 16 (or 16 x 32) block
// that one MFMA consumes as its A operand is 1 KiB contiguous in lane order, so a wave's fragment load is one fully
// coalesced dwordx4 instruction (8 whole 128-byte lines).  Row-major weights made every fragment load touch 32
// different lines for 32 bytes each: 16.9 M L2 requests per 131 072-sample forward, the kernel's bottleneck
// (profiles/r02_d_mlp_forward_counters.json); the permutation costs nothing, the packing kernels apply it.
constexpr int kOffW1 = 0;                                  // [256][144]  as (rows / 32) x 9 blocks
constexpr int kOffW2 = kOffW1 + kMlpHid * kMlpInPad;       // [256][256]  as 8 x 16 blocks
constexpr int kOffW3 = kOffW2 + kMlpHid * kMlpHid;         // [16][256]   as 8 blocks of 16 x 32
constexpr int kOffW2T = kOffW3 + kMlpHead * kMlpHid;       // [256][256]  W2T[i][o] = W2[o][i], 8 x 16 blocks
constexpr int kOffW3T = kOffW2T + kMlpHid * kMlpHid;       // [256][16]   W3T[f][r] = W3[r][f], 8 x 1 blocks
constexpr int kPackElems = kOffW3T + kMlpHid * kMlpHead;   // 176 128
constexpr int kBiasElems = 2 * kMlpHid + kMlpHead;         // b1[256] b2[256] b3[16], float32
constexpr size_t kWPlane = (size_t)kMlpNets * kPackElems;  // elements between two PLANES of the packed weights ([planes][2][kPackElems]; r04, split float32 operands)

// gradient slab of ONE net and ONE batch slice, float32 element offsets (also the layout of the reduced gradient)
constexpr int kGW1 = 0;                                    // [256][144]
constexpr int kGW2 = kGW1 + kMlpHid * kMlpInPad;           // [256][256]
constexpr int kGW3 = kGW2 + kMlpHid * kMlpHid;             // [16][256]
constexpr int kGB1 = kGW3 + kMlpHead * kMlpHid;            // [256]
constexpr int kGB2 = kGB1 + kMlpHid;                       // [256]
constexpr int kGB3 = kGB2 + kMlpHid;                       // [16]
constexpr int kGradElems = kGB3 + kMlpHead;                // 107 024

// element offset of A[row][k] inside a matrix packed as 32 x 16 blocks (K = 16 KS): block (row / 32, k / 16), then the
// mfma_f32_32x32x16_bf16 A-operand lane map: lane = row % 32 + 32 * (k / 8 % 2), element k % 8
__host__ __device__ constexpr int frag32_off(int row, int k, int KS)
{
    return ((((row >> 5) * KS + (k >> 4)) * 64) + (row & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7);
}
// the same for a 16-row matrix consumed by mfma_f32_16x16x32_bf16: block k / 32, lane = row + 16 * (k / 8 % 4)
__host__ __device__ constexpr int frag16_off(int row, int k)
{
    return (((k >> 5) * 64) + row + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS row strides (bf16 elements).  Tiles read with ds_read_b128 as the B operand: rows 16-byte aligned and
// shifted 4 (12) dwords mod 64, conflict-free for the four 16-lane groups.  Tiles read with ds_read_b64_tr_b16:
// rows shifted 16 dwords mod 64 (four rows x 16 dwords cover the 64 banks).
constexpr int kXS = 152;          // row stride of the [BM][144] input tile
constexpr int kHS = 264;          // row stride of the [BM][256] hidden tile
constexpr int kGS = 24;           // row stride of the [BM][16] head-gradient tile
constexpr int kTrH = 288;         // [64][256] tile for transposed reads (144 dwords = 16 mod 64)
constexpr int kTrX = 160;         // [64][160] (80 dwords = 16 mod 64): 144 inputs, a column of ones, zeros
constexpr int kTrHalf = 160;      // [64][128] half-width hidden tile (+32 pad)
constexpr int kTrG = 32;          // [64][16] head-gradient tile (16 dwords)
constexpr int kTilePlane = PNR_MLP_BM * kXS + PNR_MLP_BM * kHS;   // elements of one LDS plane of the forward / fused kernel (input tile | hidden tile)
constexpr int kWgChunk = 64;      // samples per weight-gradient chunk
constexpr int kWgParts = 4;       // weight-gradient workgroup roles (mlp_wgrad_kernel)

struct MlpNetParams {             // float32 master parameters of one net (torch nn.Linear layouts)
    const float* w1; const float* b1;   // [256][137], [256]
    const float* w2; const float* b2;   // [256][256], [256]
    const float* w3; const float* b3;   // [n3][256], [n3]
    int n3;                             // 12 (policy) or 1 (value)
};

// tanh through one exp2 and one reciprocal: 1 - 2 / (e^{2x} + 1); saturates correctly at +-inf.  Absolute error
// ~1e-7, far below the bf16 rounding of the stored activation.
__device__ __forceinline__ float tanh_fast(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // 2 log2(e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// The same for two values, the three non-transcendental steps as packed instructions (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32:
// one issue slot for both): per element the roundings of tanh_fast (2 r is exact, so the fused last step changes nothing).
// The learner kernels are bound by vector-instruction ISSUE (r03f: ~900 vector + 112 transcendental instructions per wave and
// tile against 84 MFMAs), not by the matrix pipe: every instruction taken out of the epilogues is time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 tanh_fast2(f32x2 x)
{
    const f32x2 t = x * (f32x2){2.885390081777927f, 2.885390081777927f};
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = e + (f32x2){1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return __builtin_elementwise_fma(r, (f32x2){-2.0f, -2.0f}, (f32x2){1.0f, 1.0f});
}
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// ---- r04: float32-accurate operands as SPLIT bf16 planes.  The reference's learner is float32 torch (pioneer_knm_train.py:47
// 'framework': 'torch', no mixed precision); one bf16 operand carries 8 significant bits.  A float32 value x is written as the sum of
// NS bf16 numbers, p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) (each residual is exact in float32): 16 significant bits
// with two planes, 24 — all of float32 — with three.  A product a . b then takes the bf16 MFMAs of the plane pairs (i, j) with
// i + j < NS — 3 for NS = 2 (error ~2^-16: 8e-6 relative on the heads), 6 for NS = 3 (1.5e-7, the accuracy of a float32 GEMM) —
// accumulated in float32 like every product here.  Against the native float32 MFMA (v_mfma_f32_32x32x2_f32: 2 048 MAC in 64
// cycles per SIMD = 32 MAC/cycle) six bf16 MFMAs per 16 384 MAC (192 cycles = 85 MAC/cycle) are 2.7x the matrix-pipe rate, three
// are 5.3x (MI355X_MICROARCH.md cycle constants; tools/f32_product_probe.hip measures the three forms on the layer-2 product).
// Every tensor that is an MFMA operand exists as NS planes of the bf16 layout (packed weights, input rows, the LDS tiles, the saved
// activations and gradients); biases, heads, the loss, slabs, Adam and the master weights are float32 as before.  NS = 1 is the
// bf16 path, instruction for instruction what it was.
// ---- r05: NS = 2 is TWO FP16 PLANES, power-of-two scaled — the float32-accurate path ("f32").  An fp16 number carries 11
// significant bits, so two planes carry 22 of float32's 24 and the three products (0,0), (0,1), (1,0) leave an operand error of
// 2^-22: measured against float64 the heads are at 2.4e-7 and the weight gradients at 1.5-3.4e-7 relative, BELOW what a float32
// GEMM's own accumulation leaves (torch float32: 4.3e-7 / 4.4-6.7e-7; tools/split_accuracy.py) — at HALF the
// MFMAs and two thirds of the bytes of the three-bf16-plane form (NS = 3, kept: exact 24-bit split, no range caveat).  fp16's
// exponent range is the price: every operand tensor is stored multiplied by a power of two (exact; the accumulators are divided
// by the product of the two scales, also exact) so that residual planes stay out of the subnormals, and is clamped to +-65504:
//   weights x 2^8 (|w| < 255), activations tanh(.) x 2^8, net inputs x 2^4 (|x| < 4094: the filter clamps to +-10; raw obs <= 126),
//   gradients dZ, G x gscale = 4 * 2^ceil(log2 B) (per-sample |d loss / d z| < 16 384 before the 1 / B; larger ones saturate —
//   a float32 learner would be taking a step of that size, i.e. has already diverged).
// Buffers keep their __bf16 element type (16-bit payloads); Fmt<NS> says what the bits mean.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NS> struct Fmt {
    static constexpr bool kHalf = NS == 2;
    static constexpr float kSW = kHalf ? 256.f : 1.f;      // weights
    static constexpr float kSH = kHalf ? 256.f : 1.f;      // activations
    static constexpr float kSX = kHalf ? 16.f : 1.f;       // net inputs
};
constexpr float kHalfMax = 65504.f;
__host__ __device__ inline float mlp_grad_scale(int planes, long long B)      // gscale of a batch of B samples (1 unless planes == 2)
{
    if (planes != 2) return 1.f;
    float g = 4.f;
    for (long long n = 1; n < B; n <<= 1) g *= 2.f;
    return g;
}
__device__ __forceinline__ __bf16 half_bits(float v) { return __builtin_bit_cast(__bf16, (_Float16)v); }
__device__ __forceinline__ float half_value(__bf16 b) { return (float)__builtin_bit_cast(_Float16, b); }
// v * scale as NS planes: bf16 planes p0 = bf16(x), p1 = bf16(x - p0), .. (scale must be 1); fp16 planes of clamp(v * scale)
template <int NS>
__device__ __forceinline__ void split_quad(const float (&v)[4], bf16x4_t (&pk)[NS], float scale = 1.f)
{
    if constexpr (Fmt<NS>::kHalf) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_fmed3f(v[j] * scale, -kHalfMax, kHalfMax);
        pk[0] = (bf16x4_t){half_bits(r[0]), half_bits(r[1]), half_bits(r[2]), half_bits(r[3])};
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] -= half_value(pk[0][j]);
        pk[1] = (bf16x4_t){half_bits(r[0]), half_bits(r[1]), half_bits(r[2]), half_bits(r[3])};
    } else {
        float r[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            pk[s] = (bf16x4_t){(__bf16)r[0], (__bf16)r[1], (__bf16)r[2], (__bf16)r[3]};
            if (s + 1 < NS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] -= (float)pk[s][j];
            }
        }
    }
}
template <int NS>
__device__ __forceinline__ void split_scalar(float v, __bf16 (&p)[NS], float scale = 1.f)
{
    if constexpr (Fmt<NS>::kHalf) {
        float r = __builtin_amdgcn_fmed3f(v * scale, -kHalfMax, kHalfMax);
        p[0] = half_bits(r); r -= half_value(p[0]); p[1] = half_bits(r);
    } else {
        float r = v;
#pragma unroll
        for (int s = 0; s < NS; ++s) { p[s] = (__bf16)r; r -= (float)p[s]; }
    }
}
// the value NS planes add up to (the float32 original for bf16 planes; its 22-bit neighbour / scale for fp16 planes)
template <int NS>
__device__ __forceinline__ f32x4 planes_value(const bf16x4_t (&pk)[NS], float inv_scale = 1.f)
{
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = NS - 1; s >= 0; --s)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] += Fmt<NS>::kHalf ? half_value(pk[s][j]) : (float)pk[s][j];
    if constexpr (Fmt<NS>::kHalf) x *= inv_scale;
    return x;
}
// the MFMAs on 16-bit payloads: bf16, or fp16 (NS == 2)
template <bool HALF>
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c)
{
    if constexpr (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool HALF>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c)
{
    if constexpr (HALF) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// the plane pairs of a product in the order they are accumulated: the large term first (it may carry the bias as its C operand)
template <int NS> struct SplitPairs;
template <> struct SplitPairs<1> { static constexpr int n = 1; static constexpr int a[1] = {0}, b[1] = {0}; };
template <> struct SplitPairs<2> { static constexpr int n = 3; static constexpr int a[3] = {0, 0, 1}, b[3] = {0, 1, 0}; };
template <> struct SplitPairs<3> { static constexpr int n = 6; static constexpr int a[6] = {0, 0, 1, 0, 2, 1}, b[6] = {0, 1, 0, 2, 0, 1}; };
constexpr int kMlpMaxPlanes = 3;

struct MlpPackParams { MlpNetParams net[kMlpNets]; __bf16* wpack; float* bias; int planes; };   // wpack: [planes][2][kPackElems]

// One thread per packed element; the transposes and the zero padding happen here, once per update.
__global__ __launch_bounds__(256) void mlp_pack_kernel(const MlpPackParams P)
{
    const int net = blockIdx.y;
    const MlpNetParams& N = P.net[net];
    const int e = blockIdx.x * 256 + threadIdx.x;
    __bf16* wp = P.wpack + (size_t)net * kPackElems;
    if (e < kPackElems) {
        // e walks the SOURCE matrices in row-major order; the destination is the fragment-native position
        float v; int dst;
        if (e < kOffW2) { const int o = e / kMlpInPad, k = e % kMlpInPad; v = k < kMlpIn ? N.w1[o * kMlpIn + k] : 0.f; dst = kOffW1 + frag32_off(o, k, kMlpInPad / 16); }
        else if (e < kOffW3) { const int r = e - kOffW2, o = r / kMlpHid, i = r % kMlpHid; v = N.w2[r]; dst = kOffW2 + frag32_off(o, i, kMlpHid / 16); }
        else if (e < kOffW2T) { const int r = (e - kOffW3) / kMlpHid, f = (e - kOffW3) % kMlpHid; v = r < N.n3 ? N.w3[r * kMlpHid + f] : 0.f; dst = kOffW3 + frag16_off(r, f); }
        else if (e < kOffW3T) { const int i = (e - kOffW2T) / kMlpHid, o = (e - kOffW2T) % kMlpHid; v = N.w2[o * kMlpHid + i]; dst = kOffW2T + frag32_off(i, o, kMlpHid / 16); }
        else { const int f = (e - kOffW3T) / kMlpHead, r = (e - kOffW3T) % kMlpHead; v = r < N.n3 ? N.w3[r * kMlpHid + f] : 0.f; dst = kOffW3T + frag32_off(f, r, 1); }
        if (P.planes == 2) {                             // two fp16 planes of the weight x 2^8 (Fmt<2>)
            __bf16 b[2];
            split_scalar<2>(v, b, Fmt<2>::kSW);
            wp[dst] = b[0]; wp[kWPlane + dst] = b[1];
        } else {
            for (int pl = 0; pl < P.planes; ++pl) {      // plane pl carries what planes 0 .. pl - 1 left of the value
                const __bf16 b = (__bf16)v;
                wp[pl * kWPlane + dst] = b;
                v -= (float)b;
            }
        }
    } else if (e < kPackElems + kBiasElems) {
        const int b = e - kPackElems;
        float v;
        if (b < kMlpHid) v = N.b1[b];
        else if (b < 2 * kMlpHid) v = N.b2[b - kMlpHid];
        else v = (b - 2 * kMlpHid) < N.n3 ? N.b3[b - 2 * kMlpHid] : 0.f;
        P.bias[net * kBiasElems + b] = v;
    }
}

// tanh of accumulator quad q (four consecutive features of one sample), rounded to bf16
template <class ACC>
__device__ __forceinline__ bf16x4_t tanh_quad(const ACC& acc, int q)
{
    const f32x2 lo = tanh_fast2((f32x2){acc[4 * q], acc[4 * q + 1]}), hi = tanh_fast2((f32x2){acc[4 * q + 2], acc[4 * q + 3]});
    return (bf16x4_t){(__bf16)lo[0], (__bf16)lo[1], (__bf16)hi[0], (__bf16)hi[1]};
}
// accumulator quad q times tanh' = 1 - h^2 of the stored activations hv, rounded to bf16.  h has 8 significant bits, h * h is
// exact in float32: the fused 1 - h * h rounds once, like the product-then-subtract it replaces — same bits, half the instructions
template <class ACC>
__device__ __forceinline__ bf16x4_t dtanh_quad(const ACC& acc, int q, bf16x4_t hv)
{
    const f32x2 h0 = {(float)hv[0], (float)hv[1]}, h1 = {(float)hv[2], (float)hv[3]};
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 d0 = __builtin_elementwise_fma(-h0, h0, one), d1 = __builtin_elementwise_fma(-h1, h1, one);
    const f32x2 p0 = (f32x2){acc[4 * q], acc[4 * q + 1]} * d0, p1 = (f32x2){acc[4 * q + 2], acc[4 * q + 3]} * d1;
    return (bf16x4_t){(__bf16)p0[0], (__bf16)p0[1], (__bf16)p1[0], (__bf16)p1[1]};
}

__device__ __forceinline__ bf16x8 ld_global_bf16x8(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Workgroup rendezvous for hand-offs that go through LDS only (every tile exchange in these kernels): the wave's DS operations
// have completed (lgkmcnt), then s_barrier.  __syncthreads() additionally waits for vmcnt(0), i.e. for every global store
// and prefetched weight fragment still in flight — each barrier then costs a write acknowledgement from HBM (the tile stores of
// h1 / h2 / dz2 / dz1) or an L2 round trip (the next product's first weight fragments).
__device__ __forceinline__ void mlp_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// acc[rb][cb] += W[64 rows of this wave][K] . tile[BM samples][K]^T.  W: fragment-native packing (global, L2);
// tile: LDS, row stride STRIDE.  Two steps, so that the first D - 1 k-steps of weight fragments can be requested EARLY
// (prefetch()): ahead of a tile store to HBM — a wave's vector-memory operations return in order, so fragments requested
// behind a 32 KB store wait for its acknowledgement — and ahead of the epilogue / barrier in front of the product.
template <int K, int STRIDE>
struct MlpGemm {
    static constexpr int KS = K / 16;
    static constexpr int D = PNR_MLP_RING;     // A fragments run D - 1 k-steps ahead of their use (L2 latency ~ 2-3 k-steps of MFMAs)
    bf16x8 a[D][2];
    const __bf16* wa;
    // w_blocks: the first of this wave's two row-blocks in the fragment-native packing: block (rb, ks) at (rb KS + ks) * 512
    __device__ __forceinline__ void prefetch(const __bf16* __restrict__ w_blocks, int lane)
    {
        wa = w_blocks + lane * 8;
#pragma unroll
        for (int p = 0; p < D - 1; ++p) {
            if (p < KS) {
                a[p][0] = ld_global_bf16x8(wa + 512 * p);
                a[p][1] = ld_global_bf16x8(wa + 512 * (KS + p));
            }
        }
    }
    // `after_loads` runs once, right behind the product's LAST fragment load (k-step KS - D): the place for a tile store to
    // HBM — nothing this product still needs can queue behind it, and it drains under the remaining MFMAs and the epilogue
    template <class F>
    __device__ __forceinline__ void run(const __bf16* tile, f32x16 (&acc)[2][kMlpCB], int lane, F&& after_loads)
    {
        const int r = lane & 31, h = lane >> 5;
        const __bf16* tb = tile + r * STRIDE + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + D - 1 < KS) {
                a[(ks + D - 1) % D][0] = ld_global_bf16x8(wa + 512 * (ks + D - 1));
                a[(ks + D - 1) % D][1] = ld_global_bf16x8(wa + 512 * (KS + ks + D - 1));
            }
            if (ks == (KS > D ? KS - D : 0)) {
                __builtin_amdgcn_sched_barrier(0);
                after_loads();
                __builtin_amdgcn_sched_barrier(0);
            }
            bf16x8 b[kMlpCB];
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb) b[cb] = *reinterpret_cast<const bf16x8*>(tb + cb * 32 * STRIDE + 16 * ks);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < kMlpCB; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D][rb], b[cb], acc[rb][cb], 0, 0, 0);
        }
    }
};

template <int K, int STRIDE>
__device__ __forceinline__ void mlp_gemm_w_xt(const __bf16* __restrict__ w_blocks, const __bf16* tile, f32x16 (&acc)[2][kMlpCB], int lane)
{
    MlpGemm<K, STRIDE> g;
    g.prefetch(w_blocks, lane);
    g.run(tile, acc, lane, [] {});
}

__device__ __forceinline__ void mlp_zero_acc(f32x16 (&acc)[2][kMlpCB])
{
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;
}

// The tile's share of the rollout record, FUSED kernel, contiguous rows (no idx): requested early by all 256 threads as 16-byte
// pieces (policy net: actions | mean | log_std [64][6], adv, logp [64]; value net: vtarg, values [64]), parked in the dead input
// tile before the loss.  Read by the loss wave with per-sample loads behind the idx gather it was a dependent HBM round trip in
// the middle of the tile's chain (7.7 us of the launch in the timing-only ablation, profiles/r03_b_mlp_fused_ablation.json).
constexpr int kRecLdsFloats = 3 * kMlpBM * kMlpAct + 2 * kMlpBM;          // 1 280
template <int NT>
struct MlpRecordTile {
    static constexpr int kN = (320 + NT - 1) / NT;
    f32x4 v[kN];
    __device__ __forceinline__ static const float* piece(const float* const (&src)[5], int net, int j, long long row0, long long B, bool& ok)
    {
        // piece j of the tile: policy 0..95 actions, 96..191 mean, 192..287 log_std, 288..303 adv, 304..319 logp; value 0..15 vtarg, 16..31 values
        int arr, off;
        if (net == 0) { if (j < 288) { arr = j / 96; off = (j % 96) * 4; } else { arr = 3 + (j - 288) / 16; off = ((j - 288) % 16) * 4; } }
        else { arr = 3 + j / 16; off = (j % 16) * 4; }
        const long long per = (net == 0 && arr < 3) ? kMlpAct : 1;
        const long long e = row0 * per + off;                   // first element of the piece in its array
        ok = e + 3 < B * per;
        return src[arr] + e;
    }
    __device__ __forceinline__ void load(const float* const (&src)[5], int net, long long row0, long long B, int tid)
    {
        const int n = net == 0 ? 320 : 32;
#pragma unroll
        for (int i = 0; i < kN; ++i) {
            const int j = tid + NT * i;
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bool ok = false;
            if (j < n) {
                const float* p = piece(src, net, j, row0, B, ok);
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a caller's slice may start on any float
                if (ok) v[i] = *reinterpret_cast<const f32x4u*>(p);
                else {                                          // the batch's last, partial tile: element by element
                    const long long per = (net == 0 && j < 288) ? kMlpAct : 1;
                    const long long e0 = p - src[net == 0 ? (j < 288 ? j / 96 : 3 + (j - 288) / 16) : 3 + j / 16];
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (e0 + k < B * per) v[i][k] = p[k];
                }
            }
        }
    }
    __device__ __forceinline__ void park(float* lds, int net, int tid) const
    {
        const int n = net == 0 ? 320 : 32;
#pragma unroll
        for (int i = 0; i < kN; ++i) {
            const int j = tid + NT * i;
            if (j < n) *reinterpret_cast<f32x4*>(lds + (net == 0 ? 4 * j : 3 * kMlpBM * kMlpAct + 4 * j)) = v[i];
        }
    }
};

// copy a [BM][256] bf16 tile between LDS (row stride kHS) and row-major global rows [row0, row0 + BM) of n_rows
__device__ __forceinline__ void mlp_store_htile(const __bf16* tile, __bf16* __restrict__ dst, long long row0, long long n_rows, int tid)
{
#pragma unroll
    for (int i = 0; i < kMlpBM / 8; ++i) {
        const int ch = tid + kMlpThreads * i, row = ch >> 5, cc = ch & 31;
        if (row0 + row < n_rows)
            *reinterpret_cast<uint4*>(dst + (row0 + row) * kMlpHid + cc * 8) = *reinterpret_cast<const uint4*>(tile + row * kHS + cc * 8);
    }
}
__device__ __forceinline__ void mlp_load_htile(__bf16* tile, const __bf16* __restrict__ src, long long row0, long long n_rows, int tid)
{
#pragma unroll
    for (int i = 0; i < kMlpBM / 8; ++i) {
        const int ch = tid + kMlpThreads * i, row = ch >> 5, cc = ch & 31;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row0 + row < n_rows) v = *reinterpret_cast<const uint4*>(src + (row0 + row) * kMlpHid + cc * 8);
        *reinterpret_cast<uint4*>(tile + row * kHS + cc * 8) = v;
    }
}

struct MlpFwdParams {
    const float* obs;          // [rows][137] float32 observations (raw when the filter vectors are given)
    const long long* idx;      // [B] row of `obs` for sample b (minibatch gather), or null: row b
    const __bf16* xs_in;       // [B][144] the nets' input ALREADY filtered and rounded (mlp_gather_kernel: an epoch's shuffle applied
                               // once), or null: stage 0 makes it from obs / idx / the filter vectors
    const float* f_loc;        // [137] MeanStdFilter vectors of PPOTrainer.filter.prepare(), or null (identity):
    const float* f_inv;        //   x = clamp((obs - loc) * inv, lo, hi)
    const float* f_lo;
    const float* f_hi;
    const __bf16* wpack;       // [2][kPackElems]
    const float* bias;         // [2][kBiasElems]
    float* head;               // [2][B][16] raw head outputs (float32)
    __bf16* xs;                // [B][144] the nets' input as they saw it (saved for dW1), or null
    __bf16* h1;                // [2][B][256] tanh activations (saved for the backward pass), or null
    __bf16* h2;                // [2][B][256]
    long long B;
    int first_net, n_nets;     // blockIdx.y + first_net = net
    // the sampler's action draw, fused into the layer-3 epilogue (all null in the learner): a = mean + exp(log_std) * noise
    // with log_std = clamp(raw, -20, 2) (RLlib DiagGaussian's sample(); SquashedGaussian is not the reference's choice),
    // the env's action = clamp(a, -a_max, a_max) when a_max is given (RLlib clip_actions, the reference's default)
    const float* noise;        // [B][6] standard-normal draws
    const float* a_max;        // [6] or null
    float* mean;               // [B][6]
    float* log_std;            // [B][6] clamped
    float* actions;            // [B][6] the sampled (unclipped) action: what the log-prob is taken of
    float* env_actions;        // [B][6] what pnr_step is given (may equal `actions` when a_max is null)
    float* values;             // [B] value head
    // FUSED instantiation (pnr_mlp_train_step): the loss and the backward-data pass of the same tile follow in the same
    // launch.  Rollout record as in PpoLossParams (rows gathered by idx); each net's workgroup differentiates its own
    // half of the loss (the policy and the value terms share nothing but the sample)
    const float* rec_actions; const float* rec_logp; const float* rec_mean; const float* rec_log_std;
    const float* rec_adv; const float* rec_vtarg; const float* rec_values;
    const float* kl_coeff; const float* ent_coeff;
    float clip, vf_clip, vf_coeff;
    float* g_head;             // [2][B][16] d loss / d head (float32; the weight-gradient kernel reads it)
    float* partials;           // [tiles * nets][8] per-workgroup sums: policy rows (-surr, 0, kl, entropy), value rows (0, vf)
    float* adam_step;          // the optimiser's update count (device scalar), incremented once per launch; or null
    __bf16* dz1;               // [2][B][256]
    __bf16* dz2;
    float* w3part;             // [tiles * nets][kW3PartFloats] layer 3's weight-gradient partials per tile (then h2 may be null), or null
    size_t act_plane;          // NS > 1: elements between two planes of h1 / h2 / dz1 / dz2 ([NS][2][B][256]: 2 B 256)
    size_t xs_plane;           // NS > 1: elements between two planes of xs_in
    float gscale;              // NS == 2: the power of two the gradient planes (G, dZ2, dZ1) are stored multiplied by (mlp_grad_scale); else 1
    unsigned long long* stamps; // PNR_MLP_STAMPS builds only: [workgroups][4 waves][kMlpStampSlots] cycle stamps, or null
};

// Forward pass of one 64-sample tile through one net: grid (ceil(B / 64), nets), 256 threads.
// FUSED: followed, in the same workgroup, by the tile's loss (one wave, a thread per sample) and its backward-data
// pass — the activations are written once (for the weight-gradient kernel) and never read back, except H1, which
// returns from L2 while dZ2 is being multiplied.  One 256-column LDS tile serves H1, H2, dZ2 (in place over H2), H1
// again and dZ1 (in place), the dead input tile holds the head rows and their gradients: 53 KB as in the plain forward.
// ---- the forward / fused kernel's own geometry: EIGHT waves per 64-sample tile, each owning one 32-row block of every layer's
// output (r02 - r03b ran four waves with two row blocks each).  What decides the launch time is the length of a tile's dependent
// chain — a tile took 44 500 cycles even alone on its CU (profiles/r03_c_mlp_stamps_one_workgroup_per_cu.json: 5 600 of them
// MFMA) — and per wave that chain is products + epilogues + tile stores + reload, all proportional to the rows a wave owns.
constexpr int kFwdWaves = 8;
constexpr int kFwdThreads = 64 * kFwdWaves;
constexpr int kFusedScratchFloats = kMlpBM * kMlpHead + kMlpBM * kGS / 2 + kRecLdsFloats + 4 * 8;     // head rows, head gradients, record, loss sums: 13 KB
static_assert(kFusedScratchFloats % 4 == 0, "");
static_assert((kMlpBM * kMlpHead + kMlpBM * kGS / 2 + kRecLdsFloats + 4 * 8) * 4 <= kMlpBM * kXS * 2, "head rows, head gradients, record and loss sums fit the dead input tile");
static_assert(kFwdWaves * 32 == kMlpHid, "one 32-row block of the 256 hidden units per wave");

// the one-row-block product: acc[cb] += W[32 rows of this wave][K] . tile[BM samples][K]^T (see MlpGemm for the two steps)
template <int K, int STRIDE, int NS = 1, int PLANE = kTilePlane, int DEPTH = PNR_MLP_RING>
struct MlpGemm1 {
    static constexpr int KS = K / 16;
    static constexpr int D = DEPTH;
    bf16x8 a[D][NS];
    const __bf16* wa;
    // w_block: this wave's row block in the fragment-native packing: k-step ks at ks * 512 (plane s: + s * kWPlane)
    __device__ __forceinline__ void prefetch(const __bf16* __restrict__ w_block, int lane)
    {
        wa = w_block + lane * 8;
#pragma unroll
        for (int p = 0; p < D - 1; ++p)
            if (p < KS) {
#pragma unroll
                for (int s = 0; s < NS; ++s) a[p][s] = ld_global_bf16x8(wa + s * kWPlane + 512 * p);
            }
    }
    // init: what every column block's accumulators start from (the bias rows of this wave, the same for every sample), read as
    // the FIRST MFMA's C operand — 16 copies per column block and product less than accumulators initialised beforehand; or null
    // tile: plane 0 of the LDS tile (plane s: + s * PLANE)
    template <class F>
    __device__ __forceinline__ void run(const __bf16* tile, f32x16 (&acc)[kMlpCB], int lane, F&& after_loads, const f32x16* init = nullptr)
    {
        const int r = lane & 31, h = lane >> 5;
        const __bf16* tb = tile + r * STRIDE + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + D - 1 < KS) {
#pragma unroll
                for (int s = 0; s < NS; ++s) a[(ks + D - 1) % D][s] = ld_global_bf16x8(wa + s * kWPlane + 512 * (ks + D - 1));
            }
            if (ks == (KS > D ? KS - D : 0)) {
                __builtin_amdgcn_sched_barrier(0);
                after_loads();
                __builtin_amdgcn_sched_barrier(0);
            }
            // (reading the sample fragments one k-step ahead of their MFMAs — 16 more registers — changed nothing in the fused
            // kernel and cost the sampler's forward 9 us of 14: r03, dropped)
            bf16x8 b[NS][kMlpCB];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int cb = 0; cb < kMlpCB; ++cb) {
                    // (PNR_MLP_DIAG & 512, timing only: ONE sample-fragment read per k-step serves both column blocks — what a
                    // 4 row-groups x 2 sample-halves wave layout would save in LDS reads, without its doubled weight stream)
                    if ((PNR_MLP_DIAG & 512) && cb > 0) b[s][cb] = b[s][0];
                    else b[s][cb] = *reinterpret_cast<const bf16x8*>(tb + s * PLANE + cb * 32 * STRIDE + 16 * ks);
                }
#pragma unroll
            for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
#pragma unroll
                for (int cb = 0; cb < kMlpCB; ++cb)
                    acc[cb] = mfma32<Fmt<NS>::kHalf>(a[ks % D][SplitPairs<NS>::a[pi]], b[SplitPairs<NS>::b[pi]][cb],
                                                     (ks == 0 && pi == 0 && init) ? *init : acc[cb]);
        }
    }
};

// a [BM][256] bf16 tile between LDS (row stride kHS) and row-major global rows, by NT threads
template <int NT>
__device__ __forceinline__ void mlp_store_htile_nt(const __bf16* tile, __bf16* __restrict__ dst, long long row0, long long n_rows, int tid)
{
    if (row0 + kMlpBM <= n_rows) {
        // a whole tile (every tile but the batch's last): one uniform test, the tile's 32 KB are contiguous in global memory — thread
        // offset + a constant per piece, no per-piece row test and no 64-bit row * stride (mlp_wgrad_kernel's lesson, r03i)
        __bf16* base = dst + row0 * kMlpHid + tid * 8;
#pragma unroll
        for (int i = 0; i < kMlpBM * 32 / NT; ++i) {
            const int ch = tid + NT * i, row = ch >> 5, cc = ch & 31;
            *reinterpret_cast<uint4*>(base + (size_t)i * NT * 8) = *reinterpret_cast<const uint4*>(tile + row * kHS + cc * 8);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < kMlpBM * 32 / NT; ++i) {
        const int ch = tid + NT * i, row = ch >> 5, cc = ch & 31;
        if (row0 + row < n_rows)
            *reinterpret_cast<uint4*>(dst + (row0 + row) * kMlpHid + cc * 8) = *reinterpret_cast<const uint4*>(tile + row * kHS + cc * 8);
    }
}

// a 16x16x32 operand fragment whose k index is the SAMPLE: eight consecutive rows s0 + 8g .. +7 (g = lane >> 4) of column
// col0 + (lane & 15) of a row-major LDS tile, by two transposed 4x16 reads (cdna_hip_programming.md T10; wg_frag32 below is the
// 32x32x16 form)
__device__ __forceinline__ bf16x8 wg_frag16(const __bf16* tile, int tstride, int s0, int col0, int lane)
{
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16* a = tile + (s0 + 8 * g + q) * tstride + col0 + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * tstride));
    // whole-vector bit cast: built element by element (f[j] = bit_cast<__bf16>(lo[j])), hipcc 7.2 replicated element 0
    // of each read into all four slots (v_perm_b32 0x05040100 of one register with itself) — found in the ISA after
    // every sample = 0 mod 4 came out weighted four times and the others not at all
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}


// ---- layer 3's weight gradients per TILE (r03h).  dW3 = G^T . H2 is the only consumer of H2 outside the tile that made it: 32 KB
// of every tile's 128 KB of activation stores, read back by the weight-gradient kernel together with a second copy of dZ2 (for
// db2) — 30 % of that kernel's bytes, and it runs at HBM / Infinity-Cache bandwidth (215 MB per 32 768-sample update in 35 us).
// The fused kernel has G, H2 and dZ2 in LDS anyway: each wave multiplies its own 32 feature columns (the columns only it
// overwrites) with 16x16x32 MFMAs and writes the tile's partials — dW3 [16][256] | db2 [256] | db3 [16] float32, 17 KB — and the
// weight-gradient kernel's third role just adds a slice's 16 partial rows in tile order.  Both forms define the slice sum the same way
// (per 64-sample tile a product chained over its two 32-sample k-steps from zero, the tiles added in order), so they agree bit for bit.
constexpr int kW3PartFloats = kMlpHead * kMlpHid + kMlpHid + kMlpHead;
template <bool HALF = false>
__device__ __forceinline__ bf16x8 bf16x8_ones()
{
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = HALF ? half_bits(1.0f) : (__bf16)1.0f;
    return o;
}
// dW3 (this wave's feature columns 32 w ..) and db3 (wave 0) of one 64-sample tile: G tile [64][16] bf16, H2 tile [64][256] bf16
// (NS planes each: gplane / hplane elements apart)
template <int NS = 1>
__device__ __forceinline__ void mlp_tile_w3_products(const __bf16* gtile, int gstride, const __bf16* htile, int hstride, int lane, int w,
                                                     f32x4 (&aw3)[2], f32x4& ab3, int gplane = 0, int hplane = 0)
{
    const bf16x8 ones = bf16x8_ones<Fmt<NS>::kHalf>();
    aw3[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; aw3[1] = aw3[0]; ab3 = aw3[0];
#pragma unroll
    for (int ks = 0; ks < kMlpBM / 32; ++ks) {
        bf16x8 fg[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) fg[s] = wg_frag16(gtile + s * gplane, gstride, 32 * ks, 0, lane);            // A: rows = head entries
        if (w == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) ab3 = mfma16<Fmt<NS>::kHalf>(fg[s], ones, ab3);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            bf16x8 fh[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) fh[s] = wg_frag16(htile + s * hplane, hstride, 32 * ks, 32 * w + 16 * b, lane);
#pragma unroll
            for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
                aw3[b] = mfma16<Fmt<NS>::kHalf>(fg[SplitPairs<NS>::a[pi]], fh[SplitPairs<NS>::b[pi]], aw3[b]);
        }
    }
}
// db2 (this wave's feature columns) of one tile: every row of 1^T . dZ2
template <int NS = 1>
__device__ __forceinline__ void mlp_tile_b2_products(const __bf16* ztile, int zstride, int lane, int w, f32x4 (&ab2)[2], int zplane = 0)
{
    const bf16x8 ones = bf16x8_ones<Fmt<NS>::kHalf>();
    ab2[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; ab2[1] = ab2[0];
#pragma unroll
    for (int ks = 0; ks < kMlpBM / 32; ++ks)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bf16x8 fz = wg_frag16(ztile + s * zplane, zstride, 32 * ks, 32 * w + 16 * b, lane);
                ab2[b] = mfma16<Fmt<NS>::kHalf>(ones, fz, ab2[b]);
            }
}

// The tile's loss on all 512 threads of the fused kernels: eight lanes per sample, lane d < 6 = action dimension d of the policy head
// (ppo_policy_sample's arithmetic, its sums over the dimensions as three xor-shuffles inside the group), lane 0 the value head
// (ppo_value_sample).  hd: the tile's head rows [64][16] float32, gt: its head gradients [64][kGS] bf16 (written here, with the
// float32 copy to g_head), rl: the parked record (rec_early) — all LDS; wsum [8 waves][4]: the waves' partial loss sums.
// As one thread per sample on wave 0 the other seven waves waited 4 300 cycles of a tile's 38 000 for it (profiles/r03_d_mlp_stamps.json).
// this thread's share of the tile's record, in registers (the compact layout of the fused kernel has no LDS to park it in): requested
// early — from inside the layer-2 product — by the thread that uses it: sample tid >> 3, action dimension tid & 7
struct MlpLossRec {
    float a, m0, l0, adv, lp0;           // policy: action, old mean, old log-std (d < 6), advantage, old log-prob; value: adv = vtarg, lp0 = old value
    __device__ __forceinline__ void load(const MlpFwdParams& P, int net, long long row0, int tid)
    {
        const int sl = tid >> 3, d = tid & 7;
        const long long b = row0 + sl;
        a = m0 = l0 = adv = lp0 = 0.f;
        if (b >= P.B) return;
        if (net == 0) {
            if (d < kMlpAct) { a = P.rec_actions[b * 6 + d]; m0 = P.rec_mean[b * 6 + d]; l0 = P.rec_log_std[b * 6 + d]; }
            adv = P.rec_adv[b]; lp0 = P.rec_logp[b];
        } else if (d == 0) { adv = P.rec_vtarg[b]; lp0 = P.rec_values[b]; }
    }
};

template <int NS = 1>
__device__ __forceinline__ void mlp_tile_loss(const MlpFwdParams& P, int net, long long row0, int tid, const float* hd, __bf16* gt, const float* rl,
                                              float* wsum, bool rec_early, int gplane = kTilePlane, bool rec_regs = false, MlpLossRec rv = MlpLossRec())
{
    // (rec_regs: the record comes in registers, by value — behind a pointer that may be null it was demoted to scratch)
    const MlpLossRec* rr = rec_regs ? &rv : nullptr;
    const int lane = tid & 63, w = tid >> 6;
    const int sl = tid >> 3, d = tid & 7;                     // sample of the tile, lane of its group
    const long long b = row0 + sl;
    const bool live = b < P.B && !(PNR_MLP_DIAG & 256);
    const float invB = 1.0f / (float)P.B;
    const long long r = (!rec_early && live && P.idx) ? P.idx[b] : b;
    float g0 = 0.f, g1 = 0.f;                                 // head-gradient entries d and 6 + d (lanes 6, 7: padding 12 + ..)
    float s_surr = 0.f, s_vf = 0.f, s_kl = 0.f, s_ent = 0.f;  // the sample's loss terms (meaningful on lane 0 of the group)
    if (net == 0) {
        const bool dim = d < kMlpAct;
        float m = 0.f, raw = 0.f, a = 0.f, m0 = 0.f, l0 = 0.f, adv = 0.f, lp0 = 0.f;
        if (live) {
            if (dim) { m = hd[sl * kMlpHead + d]; raw = hd[sl * kMlpHead + kMlpAct + d]; }
            if (rr) { a = rr->a; m0 = rr->m0; l0 = rr->l0; adv = rr->adv; lp0 = rr->lp0; }
            else if (rec_early) {
                if (dim) { a = rl[sl * 6 + d]; m0 = rl[384 + sl * 6 + d]; l0 = rl[768 + sl * 6 + d]; }
                adv = rl[1152 + sl]; lp0 = rl[1216 + sl];
            } else {
                if (dim) { a = P.rec_actions[r * 6 + d]; m0 = P.rec_mean[r * 6 + d]; l0 = P.rec_log_std[r * 6 + d]; }
                adv = P.rec_adv[r]; lp0 = P.rec_logp[r];
            }
        }
        const bool pass = raw >= -20.0f && raw <= 2.0f;           // torch.clamp passes the gradient on [min, max]
        const float ls = fminf(fmaxf(raw, -20.0f), 2.0f);
        const float si = expf(-ls);
        const float z = (a - m) * si;
        const float ivar = si * si;
        const float dm = m0 - m;
        const float q = (expf(2.0f * l0) + dm * dm) * ivar;        // (var0 + (m0 - m)^2) / var
        float lp = dim ? (-0.5f * z * z - ls) : 0.f;
        float kl = dim ? (ls - l0 + 0.5f * q - 0.5f) : 0.f;
        float en = dim ? ls : 0.f;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { lp += __shfl_xor(lp, o, 64); kl += __shfl_xor(kl, o, 64); en += __shfl_xor(en, o, 64); }
        const float logp = lp - 0.5f * 6.0f * 1.8378770664093453f;  // -3 log(2 pi)
        const float ent = en + 6.0f * 1.4189385332046727f;          // 6 * 0.5 log(2 pi e)
        const float ratio = expf(logp - lp0);
        const float rc = fminf(fmaxf(ratio, 1.0f - P.clip), 1.0f + P.clip);
        const float s1 = adv * ratio, s2 = adv * rc;
        const float surr = fminf(s1, s2);
        const bool inrange = ratio >= 1.0f - P.clip && ratio <= 1.0f + P.clip;
        // torch.minimum: the smaller argument takes the gradient, a tie splits it; the clipped branch is constant outside the range
        float dsurr;
        if (s1 < s2) dsurr = s1;
        else if (s1 == s2) dsurr = 0.5f * s1 + (inrange ? 0.5f * s1 : 0.f);
        else dsurr = inrange ? s1 : 0.f;
        const float klc = *P.kl_coeff, entc = *P.ent_coeff;
        if (live && dim) {
            g0 = (-dsurr * z * si + klc * (-dm * ivar)) * invB;
            g1 = pass ? (-dsurr * (z * z - 1.0f) + klc * (1.0f - q) - entc) * invB : 0.f;
        }
        if (live) { s_surr = -surr; s_kl = kl; s_ent = ent; }
    } else if (d == 0 && live) {
        const float vt = rr ? rr->adv : (rec_early ? rl[1152 + sl] : P.rec_vtarg[r]), v0 = rr ? rr->lp0 : (rec_early ? rl[1216 + sl] : P.rec_values[r]);
        float dvf;
        ppo_value_sample(hd[sl * kMlpHead], vt, v0, P.vf_clip, s_vf, dvf);
        g0 = P.vf_coeff * dvf * invB;
    }
    // head gradients: entries d and 6 + d of the sample's row (lanes 6, 7: the zero padding 12 .. 15), float32 for the
    // weight-gradient kernel, bf16 for this tile's backward products
    const int e0 = d < kMlpAct ? d : 12 + 2 * (d - 6), e1 = d < kMlpAct ? kMlpAct + d : 13 + 2 * (d - 6);
    if (b < P.B && P.g_head) {          // (null when nothing outside the tile reads it: layer 3's products are made by the tile itself)
        float* gp = P.g_head + ((size_t)net * P.B + b) * kMlpHead;
        gp[e0] = g0; gp[e1] = g1;
    }
    if constexpr (NS == 1) {
        gt[sl * kGS + e0] = (__bf16)g0;
        gt[sl * kGS + e1] = (__bf16)g1;
    } else {                              // the gradient rows as NS planes (plane s of the tile: + s * kTilePlane)
        __bf16 p0[NS], p1[NS];
        split_scalar<NS>(g0, p0, P.gscale); split_scalar<NS>(g1, p1, P.gscale);
#pragma unroll
        for (int s = 0; s < NS; ++s) { gt[s * gplane + sl * kGS + e0] = p0[s]; gt[s * gplane + sl * kGS + e1] = p1[s]; }
    }
    // the tile's sums: lane 0 of every group, then across the wave's eight samples; the waves' partial sums meet in LDS
    float sums[4] = {d == 0 ? s_surr : 0.f, d == 0 ? s_vf : 0.f, d == 0 ? s_kl : 0.f, d == 0 ? s_ent : 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = sums[k];
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
        sums[k] = x;
    }
    if (lane == 0) *reinterpret_cast<f32x4*>(wsum + 4 * w) = (f32x4){sums[0], sums[1], sums[2], sums[3]};
}

// Forward pass of one 64-sample tile through one net: grid (ceil(B / 64), nets), 512 threads.
// FUSED: followed, in the same workgroup, by the tile's loss (one wave, a thread per sample) and its backward-data
// pass — the activations are written once (for the weight-gradient kernel) and never read back; H1 also stays in the
// registers of the lanes that made it, for the dZ1 epilogue.  One 256-column LDS tile serves H1, H2, dZ2 (in place over H2)
// and dZ1, the dead input tile holds the head rows, their gradients and the tile's record: 53 KB.
// NS: bf16 planes per operand (1: the bf16 path; 2, 3: split float32 operands, one workgroup per CU — the LDS tile exists NS times,
// plane s at + s * kTilePlane; the packed weights at + s * kWPlane; the saved tiles at plane stride P.act_plane)
// COMPACT (r05; the fused kernel with fp16 planes, NS = 2): TWO workgroups per CU, as the bf16 form has them.  One tile's chain leaves
// a CU idle at every barrier — stamps at one workgroup per CU: 57 500 cycles per tile of which 21 800 are products, 16 000 barrier waits
// (profiles/r05_c_mlp_stamps_f32_one_workgroup_per_cu.json) — and only a second resident tile fills that.  78 KB of LDS instead of 106:
// the input planes ALIAS the hidden planes (a barrier between layer 1's product and its epilogue), the head rows / head gradients /
// loss sums get 10 KB of their own, the tile's record waits in registers (MlpLossRec) instead of LDS; <= 128 registers: layer 1's
// activations are not kept for the dZ1 epilogue but read back from the H1 planes this workgroup stored (L2), the weight ring is 3 deep.
#ifndef PNR_MLP_COMPACT
#define PNR_MLP_COMPACT 1
#endif
#ifndef PNR_MLP_COMPACT_RING
#define PNR_MLP_COMPACT_RING 3
#endif
template <bool FUSED, int NS = 1>
__global__ __launch_bounds__(kFwdThreads, ((FUSED && NS == 1) || (NS == 2 && PNR_MLP_COMPACT)) ? 4 : 2) void mlp_forward_kernel(const MlpFwdParams P)
{
    constexpr bool kCompact = PNR_MLP_COMPACT && NS == 2;          // (the plain forward too: the sampler's pnr_mlp_act then runs its 512 (tile, net) units in one round)
    constexpr int XPL = kCompact ? kMlpBM * kXS : kTilePlane;          // plane strides (elements) of the input, hidden and head-gradient tiles
    constexpr int HPL = kCompact ? kMlpBM * kHS : kTilePlane;
    constexpr int GPL = kCompact ? kMlpBM * kGS : kTilePlane;
    constexpr int RING = kCompact ? PNR_MLP_COMPACT_RING : PNR_MLP_RING;
    constexpr int kScrElems = kMlpBM * kMlpHead * 2 + NS * kMlpBM * kGS + 2 * 4 * kFwdWaves;      // head rows (float32) | gradient planes | loss sums
    constexpr int kLdsElems = kCompact ? NS * kMlpBM * kHS + kScrElems : NS * kTilePlane + ((FUSED && NS == 1) ? PNR_MLP_LDS_PAD : 0);
    __shared__ __attribute__((aligned(16))) __bf16 lds[kLdsElems];
    static_assert(NS >= 1 && NS <= kMlpMaxPlanes && kLdsElems * 2 + (PNR_MLP_STAMPS ? 2048 : 0) <= 160 * 1024, "the planes' tiles fit one CU");
    static_assert(!kCompact || (2 * (kLdsElems * 2 + (PNR_MLP_STAMPS ? 2048 : 0)) <= 160 * 1024 && NS * kMlpBM * kXS <= NS * kMlpBM * kHS), "two compact workgroups per CU");
    MLP_STAMP_DECL;
    __bf16* xt = lds;
    __bf16* ht = kCompact ? lds : lds + kMlpBM * kXS;
    const long long row0 = (long long)blockIdx.x * kMlpBM;
    // the dead input tile: head rows, head gradients, record, loss sums (compact: a block of their own behind the hidden planes)
    float* const scr = kCompact ? reinterpret_cast<float*>(lds + NS * kMlpBM * kHS) : reinterpret_cast<float*>(xt);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    // this wave's accumulators hold row block w of a layer's [256 rows][64 samples] output, column blocks 0 and 1
    constexpr int NRB = 1;                                        // row blocks (and bias sets) of a wave
    const int rowblk0 = w;
    const auto rowblk = [&](int) { return w; };
    const auto colblk = [&](int j) { return j; };
    // a wave's four 16-byte bias pieces (rows 32 w + 8 k + 4 h ..): requested in FRONT of the product's weight-fragment prefetch,
    // so that they are the older operations (vector-memory results return in order: asked for behind the fragments, as the
    // accumulators' initial values, each bias load made the compiler wait for vmcnt(0), draining the prefetch ring inside the product)
    const auto bias_load = [&](const float* b, f32x4 (&q)[NRB][4]) {
#pragma unroll
        for (int r = 0; r < NRB; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) q[r][k] = *reinterpret_cast<const f32x4*>(b + 32 * rowblk(r) + 8 * k + 4 * h);
    };
    // .. as ONE 16-register value: the first MFMA of each column block reads it as its C operand (MlpGemm1::run's `init`)
    // (fp16 planes: times the product's scale — the accumulators hold scale * (W . x + b) until the epilogue divides, both exact)
    const auto bias16 = [&](const f32x4 (&q)[NRB][4], int r, float scale) {
        f32x16 b;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) b[4 * k + j] = Fmt<NS>::kHalf ? q[r][k][j] * scale : q[r][k][j];
        return b;
    };
    typedef Fmt<NS> F;
    constexpr float kS1 = F::kSW * F::kSX, kS2 = F::kSW * F::kSH;     // scale of layer 1's / layer 2's and the head's accumulators
    const auto bias_init = [&](f32x16 (&acc)[kMlpCB], const f32x16 (&b)[NRB]) {       // (timing-only builds that skip a product)
#pragma unroll
        for (int cb = 0; cb < kMlpCB; ++cb) acc[cb] = b[0];
    };
    const int yi = (int)blockIdx.y;                               // index of this (tile, net) unit among the tile's units
    const int net = yi + P.first_net;
    [[maybe_unused]] const int stamp_yi_ = yi, stamp_ny_ = (int)gridDim.y;
    const __bf16* wp = P.wpack + (size_t)net * kPackElems;
    const float* bias = P.bias + net * kBiasElems;
    // layer 1's bias and first weight fragments do not depend on the tile: requested before anything else
    MLP_STAMP(0);
    f32x4 bq1[NRB][4];
    bias_load(bias, bq1);
    __builtin_amdgcn_sched_barrier(0);
    MlpGemm1<kMlpInPad, kXS, NS, XPL, RING> g1;
    g1.prefetch(wp + kOffW1 + rowblk0 * (kMlpInPad / 16) * 512, lane);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSED) {       // this launch is one optimiser update: counted here, read by the Adam kernel two launches on
        if (blockIdx.x == 0 && yi == 0 && tid == 0 && P.adam_step) *P.adam_step += 1.0f;
    }

    // ---- stage 0: the tile's observations, filtered, as bf16 [BM][144] (columns 137.. zero)
    {
        float* fv = kCompact ? scr : reinterpret_cast<float*>(ht);   // loc | inv | lo | hi, 4 x 144 floats, in the idle tile (compact: the scratch block)
        static_assert(4 * kMlpInPad * 2 <= kScrElems, "the filter vectors fit the scratch block");
        if (P.xs_in) {                                            // the tile's 64 rows are 18 KB of contiguous bf16: a plain copy,
            // every load of the thread in flight before the first LDS write
            constexpr int kCh = kMlpBM * (kMlpInPad / 8), kIt = (kCh + kFwdThreads - 1) / kFwdThreads;
            uint4 v[NS][kIt];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
            const __bf16* xin = P.xs_in + (size_t)s * P.xs_plane;               // plane s of the pre-gathered rows
            if (row0 + kMlpBM <= P.B) {            // a whole tile: 18 KB contiguous — thread offset + constant per piece, one uniform test
                const __bf16* base = xin + row0 * kMlpInPad + tid * 8;
#pragma unroll
                for (int i = 0; i < kIt; ++i) {
                    v[s][i] = make_uint4(0u, 0u, 0u, 0u);
                    if (i < kCh / kFwdThreads || tid + kFwdThreads * i < kCh) v[s][i] = *reinterpret_cast<const uint4*>(base + (size_t)i * kFwdThreads * 8);
                }
            } else {
#pragma unroll
                for (int i = 0; i < kIt; ++i) {
                    const int ch = tid + kFwdThreads * i, row = ch / (kMlpInPad / 8), cc = ch % (kMlpInPad / 8);
                    v[s][i] = make_uint4(0u, 0u, 0u, 0u);
                    if (ch < kCh && row0 + row < P.B) v[s][i] = *reinterpret_cast<const uint4*>(xin + (row0 + row) * kMlpInPad + cc * 8);
                }
            }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < kIt; ++i) {
                const int ch = tid + kFwdThreads * i, row = ch / (kMlpInPad / 8), cc = ch % (kMlpInPad / 8);
                if (ch < kCh) *reinterpret_cast<uint4*>(xt + s * XPL + row * kXS + cc * 8) = v[s][i];
            }
        } else {
        if (P.f_loc) {
            for (int i = tid; i < 4 * kMlpInPad; i += kFwdThreads) {
                const int which = i / kMlpInPad, k = i % kMlpInPad;
                const float* src = which == 0 ? P.f_loc : (which == 1 ? P.f_inv : (which == 2 ? P.f_lo : P.f_hi));
                fv[i] = k < kMlpIn ? src[k] : 0.f;
            }
        }
        mlp_barrier();
        // eight threads per row, 18 columns each: every load of the thread (a row starts on a 4-byte boundary only) is issued
        // before the first use.  Written as a loop over (row, column pair) with one dependent idx -> row load per iteration this
        // stage was a chain of ~36 memory round trips per thread: 35.8 us of a 16 384-sample launch (rocprof r02_b).
        {
            constexpr int TPR = kFwdThreads / kMlpBM;             // threads per row: 8
            constexpr int CPT = kMlpInPad / TPR;                  // 18 columns: four 16-byte loads + one 8-byte load
            static_assert(CPT == 18, "stage 0 is written for 18 columns per thread");
            const int row = tid / TPR, part = tid % TPR;
            const long long b = row0 + row;
            const bool live = b < P.B;
            const float* src = P.obs + (live ? (P.idx ? P.idx[b] : b) : 0) * kMlpIn + CPT * part;
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
            float x[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) x[j] = 0.f;
            if (live && !(PNR_MLP_DIAG & 2)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = CPT * part + 4 * j;
                    if (col + 3 < kMlpIn) { const f32x4 v = *reinterpret_cast<const f32x4u*>(src + 4 * j); x[4 * j] = v[0]; x[4 * j + 1] = v[1]; x[4 * j + 2] = v[2]; x[4 * j + 3] = v[3]; }
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (col + k < kMlpIn) x[4 * j + k] = src[4 * j + k];
                    }
                }
                const int col = CPT * part + 16;
                if (col + 1 < kMlpIn) { const f32x2u v = *reinterpret_cast<const f32x2u*>(src + 16); x[16] = v[0]; x[17] = v[1]; }
                else if (col < kMlpIn) x[16] = src[16];
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int col = CPT * part + j;
                float y = x[j];
                if (P.f_loc) y = fminf(fmaxf((y - fv[col]) * fv[kMlpInPad + col], fv[2 * kMlpInPad + col]), fv[3 * kMlpInPad + col]);
                x[j] = (live && col < kMlpIn) ? y : 0.f;
            }
            // 18 bf16 = 36 bytes per thread, 4-byte aligned in the tile: nine dword stores (per plane: the residual goes on)
#pragma unroll
            for (int j = 0; j < CPT; j += 2) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                __bf16 p0[NS], p1[NS];
                split_scalar<NS>(x[j], p0, F::kSX); split_scalar<NS>(x[j + 1], p1, F::kSX);
#pragma unroll
                for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x2*>(xt + s * XPL + row * kXS + CPT * part + j) = (bf16x2){p0[s], p1[s]};
            }
        }
        }
        mlp_barrier();
        if (P.xs && net == 0) {                                   // the input is the same for both nets: saved once
            for (int ch = tid; ch < kMlpBM * (kMlpInPad / 8); ch += kFwdThreads) {
                const int row = ch / (kMlpInPad / 8), cc = ch % (kMlpInPad / 8);
                if (row0 + row < P.B)
                    *reinterpret_cast<uint4*>(P.xs + (row0 + row) * kMlpInPad + cc * 8) = *reinterpret_cast<const uint4*>(xt + row * kXS + cc * 8);
            }
        }
    }

    MLP_STAMP(1);                         // stage 0 done (tile in LDS, barrier passed)
    f32x16 acc[kMlpCB];
    // tanh in registers (the bias is what the accumulators started from), each register quad = four consecutive features
    // of one sample -> one ds_write_b64
    // FUSED: layer 1's activations additionally STAY in this lane's registers (32 bf16 = 16 registers: exactly the values the
    // dZ1 epilogue multiplies its own accumulators with), instead of coming back from L2 behind a vmcnt(0), two barriers and an
    // LDS round trip (3 500 of a tile's 44 000 cycles in the phase stamps)
    bf16x4 h1keep[(FUSED && NS == 1) ? kMlpCB * 4 : 1];
    f32x4 h1keep_f[(FUSED && NS > 1 && !kCompact) ? kMlpCB * 4 : 1];        // NS > 1: the float32 values themselves (what the planes add up to)
    const auto epilogue = [&](bool keep, [[maybe_unused]] float inv_scale) {
#pragma unroll
        for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (NS == 1) {
                bf16x4 pk;
                if (PNR_MLP_DIAG & 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = (__bf16)acc[cb][4 * q + j];
                } else {
                    pk = tanh_quad(acc[cb], q);
                }
                *reinterpret_cast<bf16x4*>(ht + (32 * colblk(cb) + c) * kHS + 32 * rowblk(cb) + 8 * q + 4 * h) = pk;
                if constexpr (FUSED) { if (keep) h1keep[4 * cb + q] = pk; }
                } else {
                    f32x2 z0 = {acc[cb][4 * q], acc[cb][4 * q + 1]}, z1 = {acc[cb][4 * q + 2], acc[cb][4 * q + 3]};
                    if constexpr (F::kHalf) { z0 *= inv_scale; z1 *= inv_scale; }        // exact: a power of two
                    const f32x2 lo = tanh_fast2(z0), hi = tanh_fast2(z1);
                    const float t[4] = {lo[0], lo[1], hi[0], hi[1]};
                    bf16x4_t pk[NS];
                    split_quad<NS>(t, pk, F::kSH);
#pragma unroll
                    for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(ht + s * HPL + (32 * cb + c) * kHS + 32 * w + 8 * q + 4 * h) = pk[s];
                    if constexpr (FUSED && !kCompact) { if (keep) h1keep_f[4 * cb + q] = (f32x4){t[0], t[1], t[2], t[3]}; }
                }
            }
    };
    // a [BM][256] tile's NS planes to global rows (plane s of a saved tensor: + s * P.act_plane elements)
    const auto store_planes = [&](__bf16* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            mlp_store_htile_nt<kFwdThreads>(ht + s * HPL, dst + (size_t)s * P.act_plane + (size_t)net * P.B * kMlpHid, row0, P.B, tid);
    };

    // ---- layer 1: H1^T = tanh(W1 . X^T + b1)
    {
        f32x16 b16[NRB];
#pragma unroll
        for (int r = 0; r < NRB; ++r) b16[r] = bias16(bq1, r, kS1);
        if (PNR_MLP_DIAG & 8) bias_init(acc, b16);
        else g1.run(xt, acc, lane, [] {}, &b16[0]);
    }
    if constexpr (kCompact) mlp_barrier();      // H1 is written over the input planes: every wave is done reading them
    MLP_STAMP(2);                         // layer-1 product issued
    if (!(PNR_MLP_DIAG & 32)) epilogue(true, 1.f / kS1);
    MLP_STAMP(3);                         // layer-1 epilogue
    // layer 2's bias and first weight fragments are requested ahead of the barrier (and of the tile store behind it)
    f32x4 bq2[NRB][4];
    bias_load(bias + kMlpHid, bq2);
    __builtin_amdgcn_sched_barrier(0);
    MlpGemm1<kMlpHid, kHS, NS, HPL, RING> g2;
    g2.prefetch(wp + kOffW2 + rowblk0 * (kMlpHid / 16) * 512, lane);
    __builtin_amdgcn_sched_barrier(0);
    mlp_barrier();
    MLP_STAMP(4);                         // barrier after the layer-1 epilogue
    MLP_STAMP(5);

    // ---- layer 2: H2^T = tanh(W2 . H1^T + b2); the tile is overwritten once every wave has read it.  The H1 tile leaves for
    // HBM from INSIDE the product, behind its last weight-fragment load, and the tile's record is requested there too
    f32x16 b16_2[NRB];
#pragma unroll
    for (int r = 0; r < NRB; ++r) b16_2[r] = bias16(bq2, r, kS2);
    if (PNR_MLP_DIAG & 4) bias_init(acc, b16_2);
    MlpRecordTile<kCompact ? 100000 : kFwdThreads> rect;      // (compact: unused, no registers)
    MlpLossRec lrec;
    const bool rec_early = FUSED && !P.idx && !(PNR_MLP_DIAG & 256);
    const auto l2_hook = [&] {
        if constexpr (kCompact) {
            if (rec_early) lrec.load(P, net, row0, tid);
        } else if constexpr (FUSED) {
            if (rec_early) {
                // policy: actions, mean, log_std, adv, logp; value: -, -, -, vtarg, values
                const float* const src[5] = {P.rec_actions, P.rec_mean, P.rec_log_std, net == 0 ? P.rec_adv : P.rec_vtarg, net == 0 ? P.rec_logp : P.rec_values};
                rect.load(src, net, row0, P.B, tid);
            }
        }
        if (P.h1 && !(PNR_MLP_DIAG & 64)) store_planes(P.h1);
    };
    if (!(PNR_MLP_DIAG & 4)) {
        g2.run(ht, acc, lane, l2_hook, &b16_2[0]);
    }
    if constexpr (FUSED && !kCompact) {      // the input tile is dead since the barrier above: the record waits there, behind the head rows and gradients
        if (rec_early) rect.park(scr + kMlpBM * kMlpHead + kMlpBM * kGS / 2, net, tid);
    }
    MLP_STAMP(6);                         // layer-2 product issued
    mlp_barrier();
    MLP_STAMP(7);
    if (!(PNR_MLP_DIAG & 32)) epilogue(false, 1.f / kS2);
    MLP_STAMP(8);                         // layer-2 epilogue
    mlp_barrier();
    MLP_STAMP(9);
    MLP_STAMP(10);

    // ---- layer 3: head^T [16][samples] = W3 . H2^T + b3 with 16x16x32 MFMAs; waves 0-3 own 16 samples each (waves 4-7 go on to
    // the H2 store)
    if (!(PNR_MLP_DIAG & 16) && w < 4) {
        const int r16 = lane & 15, g = lane >> 4;
        f32x4 a3 = *reinterpret_cast<const f32x4*>(bias + 2 * kMlpHid + 4 * g);    // rows 4g .. 4g+3: the accumulators' start
        if constexpr (F::kHalf) a3 *= kS2;
        const __bf16* w3 = wp + kOffW3 + lane * 8;                    // fragment-native: block ks at ks * 512
        bf16x8 w3f[kMlpHid / 32][NS];
#pragma unroll
        for (int ks = 0; ks < kMlpHid / 32; ++ks)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                w3f[ks][s] = ld_global_bf16x8(w3 + s * kWPlane + 512 * ks);
            }
#pragma unroll
        for (int ks = 0; ks < kMlpHid / 32; ++ks) {
            bf16x8 b[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) b[s] = *reinterpret_cast<const bf16x8*>(ht + s * HPL + (16 * w + r16) * kHS + 32 * ks + 8 * g);
#pragma unroll
            for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
                a3 = mfma16<F::kHalf>(w3f[ks][SplitPairs<NS>::a[pi]], b[SplitPairs<NS>::b[pi]], a3);
        }
        if constexpr (F::kHalf) a3 *= 1.f / kS2;
        {
            const long long b = row0 + 16 * w + r16;                  // column = sample, rows 4g .. 4g+3 = head entries
            const f32x4 hq = a3;
            if (b < P.B && P.head) *reinterpret_cast<f32x4*>(P.head + ((size_t)net * P.B + b) * kMlpHead + 4 * g) = hq;
            if constexpr (FUSED) {                                    // head rows of the tile, float32 [64][16], in the dead input tile
                *reinterpret_cast<f32x4*>(scr + (16 * w + r16) * kMlpHead + 4 * g) = hq;
            } else if (P.noise) {
                if (net == 1) {
                    if (b < P.B && g == 0) P.values[b] = hq[0];
                } else {
                    // policy rows: g = 0 holds means 0..3, g = 1 means 4, 5 and raw log-stds 0, 1, g = 2 raw log-stds 2..5.
                    // Six cross-lane reads put each mean next to its log-std (executed by all lanes: no divergence around them).
                    const auto ls = [](float x) { return fminf(fmaxf(x, -20.f), 2.f); };
                    const float l0 = ls(__shfl(hq[2], r16 + 16)), l1 = ls(__shfl(hq[3], r16 + 16));
                    const float l2 = ls(__shfl(hq[0], r16 + 32)), l3 = ls(__shfl(hq[1], r16 + 32));
                    const float l4 = ls(__shfl(hq[2], r16 + 32)), l5 = ls(__shfl(hq[3], r16 + 32));
                    if (b < P.B && g <= 2) {
                        const size_t o = (size_t)b * kMlpAct;
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const auto st2 = [](float* dst, float x, float y) { *reinterpret_cast<f32x2*>(dst) = (f32x2){x, y}; };
                        const auto draw = [&](int j, float m, float l, float& a, float& e) {
                            a = fmaf(expf(l), P.noise[o + j], m);
                            e = P.a_max ? fminf(fmaxf(a, -P.a_max[j]), P.a_max[j]) : a;
                        };
                        if (g == 0) {
                            float a[4], e[4];
                            draw(0, hq[0], l0, a[0], e[0]); draw(1, hq[1], l1, a[1], e[1]);
                            draw(2, hq[2], l2, a[2], e[2]); draw(3, hq[3], l3, a[3], e[3]);
                            st2(P.mean + o, hq[0], hq[1]); st2(P.mean + o + 2, hq[2], hq[3]);
                            st2(P.actions + o, a[0], a[1]); st2(P.actions + o + 2, a[2], a[3]);
                            if (P.env_actions != P.actions) { st2(P.env_actions + o, e[0], e[1]); st2(P.env_actions + o + 2, e[2], e[3]); }
                        } else if (g == 1) {
                            float a[2], e[2];
                            draw(4, hq[0], l4, a[0], e[0]); draw(5, hq[1], l5, a[1], e[1]);
                            st2(P.mean + o + 4, hq[0], hq[1]);
                            st2(P.actions + o + 4, a[0], a[1]);
                            if (P.env_actions != P.actions) st2(P.env_actions + o + 4, e[0], e[1]);
                            st2(P.log_std + o, ls(hq[2]), ls(hq[3]));
                        } else {
                            st2(P.log_std + o + 2, ls(hq[0]), ls(hq[1])); st2(P.log_std + o + 4, ls(hq[2]), ls(hq[3]));
                        }
                    }
                }
            }
        }
    }
    MLP_STAMP(11);                        // head product + its stores

    if constexpr (!FUSED) {
        if (P.h2 && !(PNR_MLP_DIAG & 64)) store_planes(P.h2);
    }
    if constexpr (FUSED) {
        float* hd = scr;                                                  // [64][16] float32 head rows (written above)
        __bf16* gt = reinterpret_cast<__bf16*>(scr) + kMlpBM * kMlpHead * 2;    // [64][kGS] bf16 head gradients, behind them
        const float* rl = scr + kMlpBM * kMlpHead + kMlpBM * kGS / 2;     // the parked record (compact: none, MlpLossRec)
        float* wsum = kCompact ? scr + kMlpBM * kMlpHead + NS * kMlpBM * kGS / 2      // [8 waves][4] loss sums
                               : scr + kMlpBM * kMlpHead + kMlpBM * kGS / 2 + kRecLdsFloats;
        // W3^T's fragment for the first backward product: requested before the H2 store and the loss
        bf16x8 w3t[NS];                                                  // [plane]
#pragma unroll
        for (int s = 0; s < NS; ++s) w3t[s] = ld_global_bf16x8(wp + s * kWPlane + kOffW3T + w * 512 + lane * 8);
        __builtin_amdgcn_sched_barrier(0);
        if (P.h2 && !(PNR_MLP_DIAG & 64)) store_planes(P.h2);      // (null: layer 3's gradients are made here, below)
        mlp_barrier();
        MLP_STAMP(12);                    // barrier before the loss
        // ---- the tile's loss on all 512 threads (mlp_tile_loss)
        mlp_tile_loss<NS>(P, net, row0, tid, hd, gt, rl, wsum, rec_early, GPL, kCompact && rec_early, lrec);
        MLP_STAMP(13);                    // loss done
        mlp_barrier();
        MLP_STAMP(14);                    // barrier after the loss
        if (tid == 0) {                   // the eight waves' sums in wave order: one row of partial sums per workgroup
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kFwdWaves; ++k) t += *reinterpret_cast<const f32x4*>(wsum + 4 * k);
            float* pr = P.partials + ((size_t)blockIdx.x * P.n_nets + yi) * 8;
            *reinterpret_cast<f32x4*>(pr) = t;
            *reinterpret_cast<f32x4*>(pr + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

        // acc * (1 - h^2) with h read from the tile at this lane's own quads and the product written over it
        // acc * (1 - h^2) of quad q with h the float32 values hf, as NS planes into the tile at this lane's quads
        const auto dtanh_split = [&](int cb, int q, const f32x4& hf) {
            float d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = (F::kHalf ? acc[cb][4 * q + j] * (1.f / F::kSW) : acc[cb][4 * q + j]) * __builtin_fmaf(-hf[j], hf[j], 1.0f);
            bf16x4_t pk[NS];
            split_quad<NS>(d, pk);               // (fp16 planes: d is the gradient times gscale already; the split clamps)
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(ht + s * HPL + (32 * cb + c) * kHS + 32 * w + 8 * q + 4 * h) = pk[s];
        };
        const auto bwd_epilogue = [&]() {
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __bf16* at = ht + (32 * colblk(cb) + c) * kHS + 32 * rowblk(cb) + 8 * q + 4 * h;
                    if constexpr (NS == 1) {
                    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(at);
                    *reinterpret_cast<bf16x4*>(at) = dtanh_quad(acc[cb], q, hv);
                    } else {
                        bf16x4_t hv[NS];                                          // the planes add up to the activation (bf16 planes: exactly)
#pragma unroll
                        for (int s = 0; s < NS; ++s) hv[s] = *reinterpret_cast<const bf16x4*>(at + s * HPL);
                        dtanh_split(cb, q, planes_value<NS>(hv, 1.f / F::kSH));
                    }
                }
        };
        const auto zero_acc = [&]() {
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
        };
        // ---- layer 3's weight-gradient partials of this tile (dW3 = G^T . H2, db3 = G^T . 1), while H2 is still in the tile: this wave's
        // 32 feature columns are the ones only it overwrites below
        float* w3p = P.w3part ? P.w3part + ((size_t)blockIdx.x * P.n_nets + yi) * kW3PartFloats : nullptr;
        [[maybe_unused]] const float inv_g = 1.f / P.gscale;       // (a power of two: exact)
        if (w3p) {
            f32x4 aw3[2], ab3;
            mlp_tile_w3_products<NS>(gt, kGS, ht, kHS, lane, w, aw3, ab3, GPL, HPL);
            const int c16 = lane & 15, g = lane >> 4;              // C: col = lane & 15, rows 4g .. 4g+3
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) w3p[(4 * g + j) * kMlpHid + 32 * w + 16 * b + c16] = F::kHalf ? aw3[b][j] * (inv_g * (1.f / F::kSH)) : aw3[b][j];
            if (w == 0 && c16 == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w3p[kMlpHead * kMlpHid + kMlpHid + 4 * g + j] = F::kHalf ? ab3[j] * inv_g : ab3[j];     // every column of G^T . 1 is db3
            }
        }
        // ---- dH2^T = W3^T . G^T (one k-step of 16; the padded head rows are zero), dZ2 in place over H2
        zero_acc();
        {
            bf16x8 b[NS][kMlpCB];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int cb = 0; cb < kMlpCB; ++cb) b[s][cb] = *reinterpret_cast<const bf16x8*>(gt + s * GPL + (32 * cb + c) * kGS + 8 * h);
#pragma unroll
            for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
#pragma unroll
                for (int cb = 0; cb < kMlpCB; ++cb)
                    acc[cb] = mfma32<F::kHalf>(w3t[SplitPairs<NS>::a[pi]], b[SplitPairs<NS>::b[pi]][cb], acc[cb]);
        }
        bwd_epilogue();
        const auto b2_products = [&] {                             // db2 = 1^T . dZ2 of this wave's columns 32 w .., now that they hold dZ2
            if (w3p) {
                f32x4 ab2[2];
                mlp_tile_b2_products<NS>(ht, kHS, lane, w, ab2, HPL);
                if ((lane >> 4) == 0) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) w3p[kMlpHead * kMlpHid + 32 * w + 16 * b + (lane & 15)] = F::kHalf ? ab2[b][0] * inv_g : ab2[b][0];
                }
            }
        };
        b2_products();                                             // (only this wave wrote these columns)
        MLP_STAMP(15);                    // dH2 product + its epilogue
        MlpGemm1<kMlpHid, kHS, NS, HPL, RING> g4;    // W2^T's first fragments ahead of the barrier
        g4.prefetch(wp + kOffW2T + rowblk0 * (kMlpHid / 16) * 512, lane);
        __builtin_amdgcn_sched_barrier(0);
        mlp_barrier();
        MLP_STAMP(16);                    // barrier after it

        // ---- dH1^T = W2^T . dZ2^T (the dZ2 tile leaves from inside the product), then H1 into the tile and dZ1 in place over it
        zero_acc();
        const auto dz2_hook = [&] { if (!(PNR_MLP_DIAG & 64)) store_planes(P.dz2); };
        g4.run(ht, acc, lane, dz2_hook);
        MLP_STAMP(17);                    // dZ2 store + W2^T product issued
        // compact: layer 1's activations, this lane's quads, back from the H1 planes this workgroup stored during layer 2's product (the
        // stores have completed: every wave has since waited for later loads of its own, vector-memory operations complete in order,
        // and barriers followed) — requested here, consumed behind the barrier
        bf16x4 h1back[kCompact ? kMlpCB * 4 : 1][kCompact ? NS : 1];
        if constexpr (kCompact) {
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb) {
                const long long row = row0 + 32 * cb + c;
                const __bf16* src = P.h1 + ((size_t)net * P.B + (row < P.B ? row : 0)) * kMlpHid + 32 * w + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int s = 0; s < NS; ++s) h1back[4 * cb + q][s] = *reinterpret_cast<const bf16x4*>(src + (size_t)s * P.act_plane + 8 * q);
            }
        }
        MLP_STAMP(18);
        mlp_barrier();                         // every read of dZ2 (the product and the store inside it) is done: the tile is free
        MLP_STAMP(19);
        MLP_STAMP(20);
        // dZ1 = dH1 * (1 - H1^2) with H1 from this lane's own registers, written into the free tile for the coalesced store
#pragma unroll
        for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (NS == 1) *reinterpret_cast<bf16x4*>(ht + (32 * colblk(cb) + c) * kHS + 32 * rowblk(cb) + 8 * q + 4 * h) = dtanh_quad(acc[cb], q, h1keep[4 * cb + q]);
                else if constexpr (kCompact) dtanh_split(cb, q, planes_value<NS>(h1back[4 * cb + q], 1.f / F::kSH));
                else dtanh_split(cb, q, h1keep_f[4 * cb + q]);
            }
        MLP_STAMP(21);
        mlp_barrier();
        if (!(PNR_MLP_DIAG & 64)) store_planes(P.dz1);
        MLP_STAMP(22);                    // end
        MLP_STAMP_FLUSH;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// An SGD epoch's shuffle applied ONCE: row i of the outputs is row idx[i] of the rollout — the observation filtered and
// rounded exactly as stage 0 above does it ([B][144] bf16, columns 137.. zero) and the rollout record (22 floats).  The
// epoch's 16 minibatch updates then read contiguous rows (mlp_forward_kernel's xs_in path: a plain 18 KB copy per tile
// instead of 64 scattered 548-byte rows and the filter arithmetic, no idx gather in the loss, and no `xs` store: the
// weight-gradient kernel reads these rows directly).  One block = 64 samples, four threads per row, like stage 0.
// ---------------------------------------------------------------------------------------------------------------
// The rollout record of one sample as one 96-byte row: actions 0..5 | mean 6..11 | log_std 12..17 | logp, adv, vtarg, value |
// 2 pad.  Packed once per iteration (contiguous reads and writes), with the advantages standardised on the way
// ((adv - mu) / den, PPO's batch standardisation, the same float32 operations as the element-wise form): the epoch
// gathers then touch one or two cache lines per sample for the record instead of seven.
constexpr int kRecAos = 24;
struct RecordPackParams {
    const float* actions; const float* logp; const float* mean; const float* log_std;
    const float* adv; const float* vtarg; const float* values;
    const float* adv_mu; const float* adv_den;      // device scalars, or null: advantages as they are
    float* aos;                                     // [rows][24]
    long long rows;
};

__global__ __launch_bounds__(256) void record_pack_kernel(const RecordPackParams P)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= P.rows) return;
    float o[kRecAos];
#pragma unroll
    for (int j = 0; j < 6; ++j) { o[j] = P.actions[r * 6 + j]; o[6 + j] = P.mean[r * 6 + j]; o[12 + j] = P.log_std[r * 6 + j]; }
    o[18] = P.logp[r];
    const float a = P.adv[r];
    o[19] = P.adv_mu ? __fdiv_rn(__fsub_rn(a, *P.adv_mu), *P.adv_den) : a;
    o[20] = P.vtarg[r]; o[21] = P.values[r]; o[22] = 0.f; o[23] = 0.f;
    f32x4* dst = reinterpret_cast<f32x4*>(P.aos + r * kRecAos);
#pragma unroll
    for (int q = 0; q < 6; ++q) dst[q] = (f32x4){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
}

struct MlpGatherParams {
    const float* obs; const long long* idx;
    const float* f_loc; const float* f_inv; const float* f_lo; const float* f_hi;     // all four or none
    const float* actions; const float* logp; const float* mean; const float* log_std;
    const float* adv; const float* vtarg; const float* values;
    const float* rec_aos;      // [rows][24] the same record as ONE 96-byte row per sample (record_pack_kernel), or null: the seven arrays
    const __bf16* xs_src;      // [rows][144] the nets' inputs as the sampler saw them (pnr_mlp_act's xs_out), or null: made from obs
    __bf16* xs_out;            // [planes][B][144]
    int planes;                // 1, or 2 / 3 split planes of the filtered float32 input (then xs_src must be null)
    float* actions_out; float* logp_out; float* mean_out; float* log_std_out; float* adv_out; float* vtarg_out; float* values_out;
    long long B;
};

__global__ __launch_bounds__(kMlpThreads) void mlp_gather_kernel(const MlpGatherParams P)
{
    __shared__ __attribute__((aligned(16))) float fv[4 * kMlpInPad];
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 64;
    if (P.f_loc) {
        for (int i = tid; i < 4 * kMlpInPad; i += kMlpThreads) {
            const int which = i / kMlpInPad, k = i % kMlpInPad;
            const float* src = which == 0 ? P.f_loc : (which == 1 ? P.f_inv : (which == 2 ? P.f_lo : P.f_hi));
            fv[i] = k < kMlpIn ? src[k] : 0.f;
        }
    }
    __syncthreads();
    constexpr int TPR = kMlpThreads / 64, CPT = kMlpInPad / TPR, NV = CPT / 4;
    const int row = tid / TPR, part = tid % TPR;
    const long long b = row0 + row;
    if (b >= P.B) return;
    const long long r = P.idx ? P.idx[b] : b;
    if (P.xs_src) {     // a 288-byte row copied as 18 16-byte pieces (thread `part` takes pieces part, part + 4, ...): 3 lines, not 5
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int cc = part + TPR * j;
            if (cc < kMlpInPad / 8)
                *reinterpret_cast<uint4*>(P.xs_out + b * kMlpInPad + cc * 8) = *reinterpret_cast<const uint4*>(P.xs_src + r * kMlpInPad + cc * 8);
        }
    }
    // a row's four threads take its 16-byte pieces INTERLEAVED (thread `part`: pieces part, part + 4, ...), so that one load instruction
    // reads 64 contiguous bytes of each of the wave's 16 random rows; with a contiguous 36-column share per thread (r02 - r04) every
    // instruction touched 64 different cache lines for 16 bytes each, four times the L2 sectors per row
    const float* src = P.obs + r * kMlpIn;
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = 4 * TPR * j + 4 * part;
        v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (P.xs_src) continue;
        if (col + 3 < kMlpIn) v[j] = *reinterpret_cast<const f32x4u*>(src + col);
        else if (col < kMlpIn) v[j][0] = src[col];
    }
    float recv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};             // this thread's share of the record: part 0 actions, 1 mean, 2 log_std
    float sc[4] = {0.f, 0.f, 0.f, 0.f};                         // part 3: logp, adv, vtarg, values
    if (P.rec_aos) {                                            // the row's four threads read its 96 contiguous bytes
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2* a2 = reinterpret_cast<const f32x2*>(P.rec_aos + r * kRecAos + 6 * part);
        const f32x2 x0 = a2[0], x1 = a2[1], x2 = a2[2];
        if (part < 3) { recv[0] = x0[0]; recv[1] = x0[1]; recv[2] = x1[0]; recv[3] = x1[1]; recv[4] = x2[0]; recv[5] = x2[1]; }
        else { sc[0] = x0[0]; sc[1] = x0[1]; sc[2] = x1[0]; sc[3] = x1[1]; }
    } else if (part < 3) {
        const float* a = (part == 0 ? P.actions : (part == 1 ? P.mean : P.log_std)) + r * 6;
#pragma unroll
        for (int j = 0; j < 6; ++j) recv[j] = a[j];
    } else { sc[0] = P.logp[r]; sc[1] = P.adv[r]; sc[2] = P.vtarg[r]; sc[3] = P.values[r]; }
    if (!P.xs_src) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = 4 * TPR * j + 4 * part;
        f32x4 x = v[j];
        if (P.f_loc) {
            const f32x4 loc = *reinterpret_cast<const f32x4*>(fv + col), inv = *reinterpret_cast<const f32x4*>(fv + kMlpInPad + col);
            const f32x4 lo = *reinterpret_cast<const f32x4*>(fv + 2 * kMlpInPad + col), hi = *reinterpret_cast<const f32x4*>(fv + 3 * kMlpInPad + col);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = fminf(fmaxf((x[k] - loc[k]) * inv[k], lo[k]), hi[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) if (col + k >= kMlpIn) x[k] = 0.f;
        if (P.planes == 2) {                             // two fp16 planes of the input x 2^4 (Fmt<2>)
            const float xv[4] = {x[0], x[1], x[2], x[3]};
            bf16x4_t hp[2];
            split_quad<2>(xv, hp, Fmt<2>::kSX);
            *reinterpret_cast<bf16x4*>(P.xs_out + b * kMlpInPad + col) = hp[0];
            *reinterpret_cast<bf16x4*>(P.xs_out + ((size_t)P.B + b) * kMlpInPad + col) = hp[1];
        } else {
        bf16x4 pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) pk[k] = (__bf16)x[k];
        *reinterpret_cast<bf16x4*>(P.xs_out + b * kMlpInPad + col) = pk;
        for (int pl = 1; pl < P.planes; ++pl) {          // the residual planes of the split float32 input
#pragma unroll
            for (int k = 0; k < 4; ++k) { x[k] = x[k] - (float)pk[k]; pk[k] = (__bf16)x[k]; }
            *reinterpret_cast<bf16x4*>(P.xs_out + ((size_t)pl * P.B + b) * kMlpInPad + col) = pk;
        }
        }
    }
    }
    if (part < 3) {
        float* o = (part == 0 ? P.actions_out : (part == 1 ? P.mean_out : P.log_std_out)) + b * 6;
#pragma unroll
        for (int j = 0; j < 6; ++j) o[j] = recv[j];
    } else { P.logp_out[b] = sc[0]; P.adv_out[b] = sc[1]; P.vtarg_out[b] = sc[2]; P.values_out[b] = sc[3]; }
}

struct MlpBwdParams {
    const float* g_head;       // [2][B][16] d loss / d head (float32)
    const __bf16* wpack;       // [2][kPackElems]
    const __bf16* h1;          // [2][B][256]
    const __bf16* h2;          // [2][B][256]
    __bf16* dz1;               // [2][B][256] d loss / d (pre-activation of layer 1)
    __bf16* dz2;               // [2][B][256]
    long long B;
};

// Backward-data of one BM-sample tile: dZ2 = (G W3) * (1 - H2^2), dZ1 = (dZ2 W2) * (1 - H1^2).
__global__ __launch_bounds__(kMlpThreads) void mlp_backward_data_kernel(const MlpBwdParams P)
{
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * kMlpBM * kHS + kMlpBM * kGS];
    __bf16* ht = lds;                       // H2, then H1
    __bf16* dz = lds + kMlpBM * kHS;        // dZ2, then dZ1
    __bf16* gt = lds + 2 * kMlpBM * kHS;    // head gradients as bf16 [BM][16]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int net = blockIdx.y;
    const long long row0 = (long long)blockIdx.x * kMlpBM;
    const __bf16* wp = P.wpack + (size_t)net * kPackElems;
    const int c = lane & 31, h = lane >> 5;

    if (tid < 2 * kMlpBM) {   // head gradients: thread = (row, half): eight floats -> one ds_write_b128
        const int row = tid >> 1, half = tid & 1;
        bf16x8 pk;
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
        if (row0 + row < P.B) {
            const float* gp = P.g_head + ((size_t)net * P.B + row0 + row) * kMlpHead + 8 * half;
            g0 = *reinterpret_cast<const f32x4*>(gp); g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { pk[j] = (__bf16)g0[j]; pk[4 + j] = (__bf16)g1[j]; }
        *reinterpret_cast<bf16x8*>(gt + row * kGS + 8 * half) = pk;
    }
    mlp_load_htile(ht, P.h2 + (size_t)net * P.B * kMlpHid, row0, P.B, tid);
    mlp_barrier();

    f32x16 acc[2][kMlpCB];
    // acc * (1 - h^2) with h from the activation tile, packed into the gradient tile (same quad layout as forward)
    const auto epilogue = [&]() {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = (32 * cb + c) * kHS + 64 * w + 32 * rb + 8 * q + 4 * h;
                    const bf16x4 hv = *reinterpret_cast<const bf16x4*>(ht + off);
                    bf16x4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float hf = (float)hv[j]; pk[j] = (__bf16)(acc[rb][cb][4 * q + j] * (1.0f - hf * hf)); }
                    *reinterpret_cast<bf16x4*>(dz + off) = pk;
                }
    };

    // ---- dH2^T = W3^T . G^T: one k-step of 16 (the padded head rows are zero)
    mlp_zero_acc(acc);
    {
        const __bf16* wa = wp + kOffW3T + 2 * w * 512 + lane * 8;     // fragment-native, one k-step per row-block
        bf16x8 a[2] = {ld_global_bf16x8(wa), ld_global_bf16x8(wa + 512)};
        bf16x8 b[4];
#pragma unroll
        for (int cb = 0; cb < kMlpCB; ++cb) b[cb] = *reinterpret_cast<const bf16x8*>(gt + (32 * cb + c) * kGS + 8 * h);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < kMlpCB; ++cb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
    }
    // the H1 tile is requested now (16 x 16 bytes per thread, held in registers) and lands under the epilogue below
    uint4 h1r[kMlpBM / 8];
    {
        const __bf16* src = P.h1 + (size_t)net * P.B * kMlpHid;
#pragma unroll
        for (int i = 0; i < kMlpBM / 8; ++i) {
            const int ch = tid + kMlpThreads * i, row = ch >> 5, cc = ch & 31;
            h1r[i] = make_uint4(0u, 0u, 0u, 0u);
            if (row0 + row < P.B) h1r[i] = *reinterpret_cast<const uint4*>(src + (row0 + row) * kMlpHid + cc * 8);
        }
    }
    epilogue();
    mlp_barrier();                                                               // every wave is done with H2
    mlp_store_htile(dz, P.dz2 + (size_t)net * P.B * kMlpHid, row0, P.B, tid);
#pragma unroll
    for (int i = 0; i < kMlpBM / 8; ++i) {
        const int ch = tid + kMlpThreads * i, row = ch >> 5, cc = ch & 31;
        *reinterpret_cast<uint4*>(ht + row * kHS + cc * 8) = h1r[i];
    }
    mlp_barrier();

    // ---- dH1^T = W2^T . dZ2^T
    mlp_zero_acc(acc);
    mlp_gemm_w_xt<kMlpHid, kHS>(wp + kOffW2T + 2 * w * (kMlpHid / 16) * 512, dz, acc, lane);
    mlp_barrier();                         // all reads of dZ2 done before it is overwritten
    epilogue();
    mlp_barrier();
    mlp_store_htile(dz, P.dz1 + (size_t)net * P.B * kMlpHid, row0, P.B, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradients: dW = dZ^T . H over the samples of one batch slice, written to that slice's slab.
// grid (slices, 4 parts, nets): part 0 / 1 = the two 128-column halves of dW2, part 2 = dW1 and db1 (the input tile
// carries a column of ones at k = 144), part 3 = dW3, db3 and db2 (16x16x32 MFMAs, a fragment of ones).  One workgroup
// per CU.  (What was tried and dropped here in r02 / r03 — more roles, two chunks in flight, register-staged chunks: DESIGN_HISTORY.md.)
// ---------------------------------------------------------------------------------------------------------------
struct MlpWgradParams {
    const float* g_head;       // [2][B][16]
    const __bf16* xs;          // [B][144]
    const __bf16* h1;          // [2][B][256]
    const __bf16* h2;
    const __bf16* dz1;
    const __bf16* dz2;
    float* slabs;              // [slices][2][kGradElems]
    long long B;
    long long slice_rows;      // samples per slice, a multiple of kWgChunk
    int first_net;             // blockIdx.z + first_net = net
    const float* w3part;       // [tiles * n_nets][kW3PartFloats] the fused kernel's per-tile layer-3 partials (then h2 is not read), or null
    int n_nets;                // nets of the launch that wrote w3part (its row index is tile * n_nets + blockIdx.z)
    unsigned long long* stamps; // PNR_MLP_STAMPS builds only (tools/wgrad_stamps.py): [nets][roles][slices][8 waves][kMlpStampSlots], or null
    size_t act_plane;          // NS > 1: elements between two planes of h1 / dz1 / dz2
    size_t xs_plane;           // .. and of xs
    float gscale;              // NS == 2: what the gradient planes are stored multiplied by (MlpFwdParams::gscale); else 1
};
#if PNR_MLP_STAMPS
#define WG_STAMP(i) do { if (P.stamps && lane == 0) { unsigned long long* sp_ = P.stamps + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + w) * kMlpStampSlots; \
    sp_[(i)] = __builtin_amdgcn_s_memtime(); if ((i) == 0) sp_[24] = __builtin_amdgcn_s_memrealtime(); if ((i) == 22) sp_[25] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#endif

// A 64-row chunk of COLS bf16 columns (a multiple of 8) on its way from row-major global memory (row stride src_stride)
// into an LDS tile: loaded into registers first (every load of the chunk in flight together), written later — the
// weight-gradient loop requests chunk c + 1 before it multiplies chunk c.
// (eight waves per workgroup: two per SIMD — with the four of r02 every MFMA chain, LDS read and chunk hand-over of a workgroup
// was exposed on a SIMD that had nothing else to run)
constexpr int kWgThreads = 512;
template <int COLS>
struct WgChunk {
    static constexpr int kPieces = kWgChunk * (COLS / 8);
    static constexpr int kPerThread = (kPieces + kWgThreads - 1) / kWgThreads;
    static constexpr int kFull = kPieces / kWgThreads;             // iterations in which every thread has a piece
    uint4 v[kPerThread];
    // A WHOLE chunk (the common case: slices are multiples of 64 rows, only the batch's last chunk can be short) is requested with no
    // per-thread test and one per-thread offset that does not depend on the chunk: uniform base + thread offset + constant.  Written
    // with a bounds test, a zero fill and a 64-bit row * stride per piece, requesting a chunk's nine pieces cost ~1 000 of its
    // 3 150 cycles in address arithmetic alone (tools/wgrad_stamps.py, r03i).
    __device__ __forceinline__ void load(const __bf16* __restrict__ src, long long src_stride, long long row0, long long n_rows, int tid)
    {
        constexpr int cpr = COLS / 8;
        if (row0 + kWgChunk <= n_rows) {                            // uniform
            const __bf16* base = src + row0 * src_stride;
            const unsigned off = (unsigned)(tid / cpr) * (unsigned)src_stride + (unsigned)(tid % cpr) * 8u;
            const unsigned step = (unsigned)(kWgThreads / cpr) * (unsigned)src_stride;
            static_assert(kWgThreads % cpr == 0 || kFull * kWgThreads == kPieces || true, "");
#pragma unroll
            for (int i = 0; i < kPerThread; ++i) {
                if constexpr (kWgThreads % cpr == 0) {
                    // rows advance by kWgThreads / cpr per iteration, the column piece stays
                    if (i < kFull || tid < kPieces - kFull * kWgThreads) v[i] = *reinterpret_cast<const uint4*>(base + off + (size_t)i * step);
                    else v[i] = make_uint4(0u, 0u, 0u, 0u);
                } else {
                    const int ch = tid + kWgThreads * i, row = ch / cpr, cc = ch % cpr;
                    if (i < kFull || ch < kPieces) v[i] = *reinterpret_cast<const uint4*>(base + (unsigned)row * (unsigned)src_stride + (unsigned)cc * 8u);
                    else v[i] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const int ch = tid + kWgThreads * i, row = ch / cpr, cc = ch % cpr;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (ch < kWgChunk * cpr && row0 + row < n_rows) v[i] = *reinterpret_cast<const uint4*>(src + (row0 + row) * src_stride + cc * 8);
        }
    }
    __device__ __forceinline__ void store(__bf16* tile, int tstride, int tid) const
    {
        constexpr int cpr = COLS / 8;
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const int ch = tid + kWgThreads * i, row = ch / cpr, cc = ch % cpr;
            if (ch < kWgChunk * cpr) *reinterpret_cast<uint4*>(tile + row * tstride + cc * 8) = v[i];
        }
    }
};

// a 32x32x16 operand fragment whose k index is the SAMPLE: eight consecutive rows s0 + 8h .. +7 of column
// col0 + (lane & 31) of a row-major tile, by two transposed 4x16 reads (cdna_hip_programming.md T10)
__device__ __forceinline__ bf16x8 wg_frag32(const __bf16* tile, int tstride, int s0, int col0, int lane)
{
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16* a = tile + (s0 + 8 * (G >> 1) + q) * tstride + col0 + 16 * (G & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * tstride));
    // whole-vector bit cast: built element by element (f[j] = bit_cast<__bf16>(lo[j])), hipcc 7.2 replicated element 0
    // of each read into all four slots (v_perm_b32 0x05040100 of one register with itself) — found in the ISA after
    // every sample = 0 mod 4 came out weighted four times and the others not at all
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define PNR_SLAB_STORE(p, v) __builtin_nontemporal_store((v), (p))      // slabs are written once and read by another kernel: streaming stores
// store a 32x32 accumulator block to a row-major float32 matrix: rows row0.., cols col0.. (cols < ncols kept)
__device__ __forceinline__ void wg_store_block(float* __restrict__ m, int ld, int row0, int col0, int ncols, const f32x16& a, int lane)
{
    const int c = lane & 31, h = lane >> 5;
    if (col0 + c < ncols) {
#pragma unroll
        for (int i = 0; i < 16; ++i) PNR_SLAB_STORE(m + (size_t)(row0 + (i & 3) + 8 * (i >> 2) + 4 * h) * ld + col0 + c, a[i]);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// r04: the hot roles of the weight-gradient kernel (dW2 halves; dW1 halves) stage their chunks with DIRECT-TO-LDS loads
// (global_load_lds_dwordx4, "glds": 1 KiB per wave-instruction, no VGPR destination, no ds_write pass) into a THREE-stage ring,
// one raw barrier per chunk and counted s_waitcnt vmcnt(N): chunk c + 2 is requested at the start of chunk c's products and has two
// chunks of MFMAs to arrive, where the register-staged form of r03 (bit-identical; git history) had one chunk of prefetch, two
// barriers and a VGPR -> LDS write pass per chunk: 3 150 cycles per 64-sample chunk for 16 MFMAs per wave (1 024 cycles of matrix
// pipe per SIMD), the same for every chunk (profiles/r03_i_wgrad_stamps.json) — a workgroup alone on its CU has nothing else to run
// while it waits (cdna_hip_programming.md section 5, "Pipelining across barriers": the regime where the 3-buffer span pays).
// A glds writes LDS lane-linearly (wave-uniform base + 16 lane), so the conflict-free image for the transposed reads cannot be
// made by padding rows: it is an XOR swizzle applied on the SOURCE address and again on the read (16-byte piece j of row r sits
// at piece j ^ 4 (r & 3): the four rows a ds_read_b64_tr_b16 half-wave touches land in four different 64-byte bank groups).
// X rows are 288 bytes (32 mod 256): stored as they are, the four rows overlap pairwise in the banks (2-way conflict on the
// B-operand reads of the dW1 roles, ~2 of 32 cycles per MFMA gap); db1 comes from a fragment of ones in registers instead of a
// column of ones in the tile.  The products, their k order and the chunk order are those of the register-staged form: same bits.
// ---------------------------------------------------------------------------------------------------------------
// the ring's geometry by the number of operand planes (NS > 1: float32-accurate split operands): chunks of 64 samples and three
// stages for bf16; 32-sample chunks for split operands (a stage holds every plane's tiles), three stages with two planes, two with three
template <int NS> struct WgGeom {
    static constexpr int CH = NS == 1 ? 64 : 32;                        // samples per chunk
    static constexpr int KS = CH / 16;                                  // k-steps per chunk
    static constexpr int STAGES = NS <= 2 ? 3 : 2;
    static constexpr int kPlane2 = CH * 512 + CH * 256;                 // dW2 roles, one plane: dZ2 [CH][256] | H1 half [CH][128], bf16
    static constexpr int kPlane1 = CH * 256 + CH * 288;                 // dW1 roles, one plane: dZ1 half [CH][128] | X [CH][144]
    static constexpr int kStage2 = NS * kPlane2, kStage1 = NS * kPlane1;
    static constexpr int kRing = STAGES * kStage2 + 64;                 // (+ slack: the last X block reads 32 bytes past its row)
    static constexpr int nA2 = CH / 16, nB2 = CH / 32;                  // a wave's pieces per plane and chunk: dZ2, H1 half
    static constexpr int nA1 = CH / 32, kXPieces = CH * 288 / 1024;     // dZ1 half; X pieces of the whole workgroup (18 or 9)
};
constexpr int kWgRingBytes = WgGeom<1>::kRing > WgGeom<2>::kRing ? (WgGeom<1>::kRing > WgGeom<3>::kRing ? WgGeom<1>::kRing : WgGeom<3>::kRing)
                                                                 : (WgGeom<2>::kRing > WgGeom<3>::kRing ? WgGeom<2>::kRing : WgGeom<3>::kRing);
static_assert(kWgRingBytes <= 150 * 1024, "the ring fits one CU beside the stamps");
constexpr int kWgLdsBytesOld = (kWgChunk * kTrH + kWgChunk * kTrH + kWgChunk * kTrG) * 2;
constexpr int kWgLdsBytes = kWgRingBytes > kWgLdsBytesOld ? kWgRingBytes : kWgLdsBytesOld;

// one direct-to-LDS piece: lane l's 16 bytes at sbase + voff land at LDS byte address lds_dst + 16 l (lds_dst, sbase wave-uniform).
// M0 carries the LDS base and is compiler-reserved: written and restored in the same statement (cdna_hip_programming.md, inline asm)
__device__ __forceinline__ void wg_glds16(unsigned voff, const void* sbase, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
template <int N> __device__ __forceinline__ void wg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ const void* wg_uniform_ptr(const void* p)
{
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
// LDS byte address of a __shared__ pointer
__device__ __forceinline__ unsigned wg_lds_addr(const void* p)
{
    typedef char __attribute__((address_space(3))) * lds_c;
    return (unsigned)(unsigned long long)(lds_c)(p);
}
// the fragment of wg_frag32 from a SWIZZLED tile (rows of ROWB bytes, 16-byte piece j of row r at piece j ^ 4 (r & 3)); s0 a multiple of 16
template <int ROWB>
__device__ __forceinline__ bf16x8 wg_frag32_swz(const char* tile, int s0, int col0, int lane)
{
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = s0 + 8 * (G >> 1) + q;                      // row & 3 == q, and so for row + 4
    const int cb = (col0 + 16 * (G & 1) + 4 * p) * 2;           // byte of the column inside the row
    const char* a = tile + row * ROWB + (cb ^ (q << 6));
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * ROWB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// .. and from a LINEAR tile with rows of ROWB bytes
template <int ROWB>
__device__ __forceinline__ bf16x8 wg_frag32_lin(const char* tile, int s0, int col0, int lane)
{
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = tile + (s0 + 8 * (G >> 1) + q) * ROWB + (col0 + 16 * (G & 1) + 4 * p) * 2;
    typedef s16x4 __attribute__((address_space(3))) * lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * ROWB));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// a swizzled tile's piece i (1 KiB of LDS = 1024 / ROWB rows): the source byte offset of lane `lane` relative to the chunk's first row
// (global rows GROWB bytes apart, the tile's columns starting at byte col_off of a row)
template <int ROWB>
__device__ __forceinline__ unsigned wg_piece_src_swz(int i, int lane, int GROWB, int col_off, int& row)
{
    constexpr int kLanesPerRow = ROWB / 16;
    row = i * (1024 / ROWB) + lane / kLanesPerRow;
    const int jp = lane % kLanesPerRow;
    return (unsigned)(row * GROWB + col_off + ((jp ^ (4 * (row & 3))) << 4));
}

// The ring's schedule, shared by both roles.  NP = this wave's glds instructions per chunk.  Per chunk c:
//   wait until chunk c's pieces of THIS wave have landed (counted: chunk c + 1's may stay in flight) -> barrier (every wave's pieces
//   landed; every wave is done reading chunk c - 1) -> request chunk c + 2 into the stage chunk c - 1 used -> multiply chunk c.
// A chunk that is not whole (only the batch's last one can be) is staged by plain loads and ds_write into the same image, zeros
// for the rows past the end, at the place its glds would have been issued.
#if PNR_MLP_STAMPS
// (diagnostic build) phase stamps of the ring, parked in LDS and flushed at the kernel's end: a stamp written to global memory would be
// one more operation on the VM counter that the ring's counted waits are written against
#define WG_RING_STAMP(i) do { if (stamps && (threadIdx.x & 63) == 0) stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_RING_STAMP(i) do { } while (0)
#endif
template <int CH, int STAGES, int MAXP, class ISSUE, class SYNC, class MUL, class PRO>
__device__ __forceinline__ void wg_ring_loop(long long s_begin, long long s_end, int np, ISSUE&& issue, SYNC&& stage_sync, MUL&& multiply,
                                             PRO&& prologue_work, unsigned long long* stamps = nullptr)
{
    constexpr int D = STAGES - 1;                                     // chunks requested ahead of the one being multiplied
    const int nch = (int)((s_end - s_begin + CH - 1) / CH);
    const auto whole = [&](int c) { return s_begin + (long long)(c + 1) * CH <= s_end; };
    const auto request = [&](int c, int stage) {
        if (whole(c)) {
#pragma unroll
            for (int k = 0; k < MAXP; ++k) issue(c, stage, k);
        } else stage_sync(c, stage);
    };
#pragma unroll
    for (int c = 0; c < D; ++c)
        if (c < nch) request(c, c);
    // `late_work` (the requests of the layer-3 partial sums: sixteen 16-byte loads of a few threads, consumed after the accumulators'
    // stores) is issued at the top of the LAST chunk's products.  In front of the loop — even behind the first chunks' pieces — its 16
    // vector-memory instructions per wave queued up with the 12 pieces and the loop started 5 500 cycles later (8 250 against 2 750
    // cycles from the kernel's start, profiles/r04_b_wgrad_ring_stamps.json); no counted wait follows the last chunk's, so nothing
    // waits for these loads but their use.
    bool late_done = false;
    int stage = 0;
    WG_RING_STAMP(1);
    for (int c = 0; c < nch; ++c) {
        // glds of this wave that may stay in flight: those of the chunks c + 1 .. c + D - 1 (requested by glds; none with two stages)
        const int ahead = (D > 1 && c + 1 < nch && whole(c + 1)) ? np : 0;
        if (c >= 4 && c < 8) WG_RING_STAMP(2 + 4 * (c - 4));          // top of the chunk
        if (ahead == 0) wg_wait_vm<0>();
        else if (ahead == 4) wg_wait_vm<4>();
        else if (ahead == 5) wg_wait_vm<5>();
        else if (ahead == 6) wg_wait_vm<6>();
        else wg_wait_vm<0>();
        if (c >= 4 && c < 8) WG_RING_STAMP(3 + 4 * (c - 4));          // its pieces landed
        mlp_barrier();
        if (c >= 4 && c < 8) WG_RING_STAMP(4 + 4 * (c - 4));          // barrier passed
        const int nstage = stage == 0 ? STAGES - 1 : stage - 1;       // the stage chunk c - 1 used: every wave is done with it
        const bool glds_next = c + D < nch && whole(c + D);
        if (c + D < nch && !glds_next) stage_sync(c + D, nstage);
        if (c >= 4 && c < 8) WG_RING_STAMP(5 + 4 * (c - 4));
        if (c == nch - 1) { prologue_work(); late_done = true; }
        // chunk c + D's pieces are requested from INSIDE the products, a few behind each k-step's MFMAs: issuing the six
        // of them in one go cost 650 cycles per chunk in which the wave issued no MFMA (profiles/r04_a_wgrad_ring_stamps_issue_in_one_go.json)
        multiply(stage, [&](int k) { if (glds_next) issue(c + D, nstage, k); });
        stage = stage == STAGES - 1 ? 0 : stage + 1;
    }
    if (!late_done) prologue_work();
    WG_RING_STAMP(20);
}

// A wave's 32x32 accumulator block to a row-major float32 matrix through a wave-private LDS tile: 16-byte stores, eight lanes per
// 128-byte row segment (wg_store_block's one dword per lane cost ~96 cycles of issue per instruction: 6 000 cycles per dW2 wave).
constexpr int kWgTrS = 36;                                       // floats per row of the transposing tile (144 B: 16-byte aligned)
// (unscale: the power of two the accumulators are multiplied by on their way out — fp16 planes; 1 otherwise)
__device__ __forceinline__ void wg_store_block_lds(float* scratch, float* __restrict__ m, int ld, int row0, int col0, int ncols, const f32x16& a, int lane,
                                                   float unscale = 1.f)
{
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) scratch[((i & 3) + 8 * (i >> 2) + 4 * h) * kWgTrS + c] = a[i] * unscale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private tile: the wave's own DS operations complete in order
    const int r = lane >> 3, q = lane & 7;
#pragma unroll
    for (int pss = 0; pss < 4; ++pss) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (r + 8 * pss) * kWgTrS + 4 * q);
        if (col0 + 4 * q < ncols) {
            f32x4* dst = reinterpret_cast<f32x4*>(m + (size_t)(row0 + r + 8 * pss) * ld + col0 + 4 * q);
            __builtin_nontemporal_store(v, dst);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the reads are done before the next block overwrites the tile
}

// The fused kernel's per-tile layer-3 products of this slice (dW3 | db2 | db3, kW3PartFloats per tile), added in tile order: part `part`
// of `parts` takes that share of the elements, a thread ONE quad of them.  Two steps: the (at most 16) tiles' 16-byte pieces are
// REQUESTED before the ring starts and ADDED after it — one memory round trip (2-4 us under load) for a handful of threads, which as
// a serial step cost the whole workgroup that time wherever it stood (profiles/r04_b_wgrad_ring_stamps.json: 8 000 cycles in front of
// the loop, 7 400 behind it); now it travels under the products.
struct WgW3Sums {
    static constexpr int kTiles = 16;
    f32x4 x[kTiles];
    int q;                    // this thread's quad, or -1
    long long extra0, t1;     // tiles beyond the first 16 (none at the loop's slice size): added synchronously in finish()
    const float* pp; size_t stride;
    __device__ __forceinline__ void request(const MlpWgradParams& P, long long s_begin, long long s_end, int part, int parts, int tid)
    {
        const int quads = kW3PartFloats / 4 / parts;              // 273 with four parts: one per thread
        q = tid < quads ? part * quads + tid : -1;
        const long long t0 = s_begin / kWgChunk;
        t1 = (s_end + kWgChunk - 1) / kWgChunk;
        stride = (size_t)P.n_nets * kW3PartFloats;
        pp = P.w3part + ((size_t)t0 * P.n_nets + blockIdx.z) * kW3PartFloats + 4 * (q < 0 ? 0 : q);
        extra0 = t0 + kTiles;
#pragma unroll
        for (int j = 0; j < kTiles; ++j) x[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < 0) return;
        if (t0 + kTiles <= t1) {                                  // the usual case, one uniform test: sixteen requests back to back (written
            // with a test per tile, hipcc branched around every load and made the first one wait for its data before the next was issued)
            const float* a = pp;
#pragma unroll
            for (int j = 0; j < kTiles; ++j) { x[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a)); a += stride; }
        } else {
#pragma unroll
            for (int j = 0; j < kTiles; ++j)
                if (t0 + j < t1) x[j] = *reinterpret_cast<const f32x4*>(pp + (size_t)j * stride);
        }
    }
    __device__ __forceinline__ void finish(float* slab) const
    {
        if (q < 0) return;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        const long long t0 = extra0 - kTiles;
#pragma unroll
        for (int j = 0; j < kTiles; ++j) if (t0 + j < t1) sum += x[j];
        for (long long t = extra0; t < t1; ++t) sum += *reinterpret_cast<const f32x4*>(pp + (size_t)(t - t0) * stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 4 * q + j;
            slab[e < kMlpHead * kMlpHid ? kGW3 + e : (e < kMlpHead * kMlpHid + kMlpHid ? kGB2 + (e - kMlpHead * kMlpHid) : kGB3 + (e - kMlpHead * kMlpHid - kMlpHid))] = sum[j];
        }
    }
};
static_assert(kW3PartFloats / 4 / kWgParts <= kWgThreads && kW3PartFloats % (4 * kWgParts) == 0, "one quad of the layer-3 partials per thread and role");

// dW2[:, 128 part .. +128] = dZ2^T . H1[:, that half] of one slice; waves 4 x 2, each 64 (o) x 64 (i)
template <int NS>
__device__ __forceinline__ void wgrad_dw2_glds(const MlpWgradParams& P, char* ring, int part, size_t nb, long long s_begin, long long s_end,
                                               float* slab, int tid, unsigned long long* stamps = nullptr)
{
    typedef WgGeom<NS> G;
    constexpr int CH = G::CH, NPP = G::nA2 + G::nB2, MAXP = NS * NPP;   // pieces per plane / per chunk of one wave
    const int lane = tid & 63, w = tid >> 6;
    const int wo = w >> 1, wi = w & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // this wave's pieces of a chunk and plane: A (dZ2, pieces of two 512-byte rows) w, w + 8, ..; B (H1 half, pieces of four 256-byte
    // rows) w, ..  The per-lane source offsets depend on neither the chunk nor the plane.
    unsigned offA[G::nA2], offB[G::nB2];
    int rowA[G::nA2], rowB[G::nB2];
#pragma unroll
    for (int k = 0; k < G::nA2; ++k) offA[k] = wg_piece_src_swz<512>(w + 8 * k, lane, kMlpHid * 2, 0, rowA[k]);
#pragma unroll
    for (int k = 0; k < G::nB2; ++k) offB[k] = wg_piece_src_swz<256>(w + 8 * k, lane, kMlpHid * 2, 256 * part, rowB[k]);
    const char* gA = reinterpret_cast<const char*>(P.dz2 + nb);
    const char* gB = reinterpret_cast<const char*>(P.h1 + nb);
    const size_t plane_b = P.act_plane * 2;                          // bytes between two planes of a saved tensor
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(wg_lds_addr(ring));
    // piece k of this wave's MAXP of chunk c into stage `stage`: plane k / NPP, then A pieces, then B pieces
    const auto issue = [&](int c, int stage, int k) {
        const long long s = s_begin + (long long)c * CH;
        const int pl = k / NPP, r = k % NPP;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + stage * G::kStage2 + pl * G::kPlane2 + w * 1024);
        if (r < G::nA2) wg_glds16(offA[r], wg_uniform_ptr(gA + pl * plane_b + s * (kMlpHid * 2)), dst + r * 8192);
        else wg_glds16(offB[r - G::nA2], wg_uniform_ptr(gB + pl * plane_b + s * (kMlpHid * 2)), dst + CH * 512 + (r - G::nA2) * 8192);
    };
    const auto stage_sync = [&](int c, int stage) {
        const long long s = s_begin + (long long)c * CH;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
            char* dst = ring + stage * G::kStage2 + pl * G::kPlane2 + w * 1024 + lane * 16;
#pragma unroll
            for (int k = 0; k < G::nA2; ++k) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (s + rowA[k] < s_end) v = *reinterpret_cast<const uint4*>(gA + pl * plane_b + s * (kMlpHid * 2) + offA[k]);
                *reinterpret_cast<uint4*>(dst + k * 8192) = v;
            }
#pragma unroll
            for (int k = 0; k < G::nB2; ++k) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (s + rowB[k] < s_end) v = *reinterpret_cast<const uint4*>(gB + pl * plane_b + s * (kMlpHid * 2) + offB[k]);
                *reinterpret_cast<uint4*>(dst + CH * 512 + k * 8192) = v;
            }
        }
    };
    const auto multiply = [&](int stage, auto&& piece) {
        const char* ta = ring + stage * G::kStage2;
        const char* tb = ta + CH * 512;
        if constexpr (NS == 1) {
            bf16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[0][a] = wg_frag32_swz<512>(ta, 0, 64 * wo + 32 * a, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[0][b] = wg_frag32_swz<256>(tb, 0, 64 * wi + 32 * b, lane);
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                if (ks + 1 < G::KS) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) fa[(ks + 1) & 1][a] = wg_frag32_swz<512>(ta, 16 * (ks + 1), 64 * wo + 32 * a, lane);
#pragma unroll
                    for (int b = 0; b < 2; ++b) fb[(ks + 1) & 1][b] = wg_frag32_swz<256>(tb, 16 * (ks + 1), 64 * wi + 32 * b, lane);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = mfma32<false>(fa[ks & 1][a], fb[ks & 1][b], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = ks; k < MAXP; k += G::KS) piece(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                bf16x8 fa[NS][2], fb[NS][2];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) fa[pl][a] = wg_frag32_swz<512>(ta + pl * G::kPlane2, 16 * ks, 64 * wo + 32 * a, lane);
#pragma unroll
                    for (int b = 0; b < 2; ++b) fb[pl][b] = wg_frag32_swz<256>(tb + pl * G::kPlane2, 16 * ks, 64 * wi + 32 * b, lane);
                }
#pragma unroll
                for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = mfma32<Fmt<NS>::kHalf>(fa[SplitPairs<NS>::a[pi]][a], fb[SplitPairs<NS>::b[pi]][b], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = ks; k < MAXP; k += G::KS) piece(k);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    WgW3Sums w3;
    w3.q = -1;
    wg_ring_loop<CH, G::STAGES, MAXP>(s_begin, s_end, MAXP, issue, stage_sync, multiply,
                                      [&] { if (P.w3part) w3.request(P, s_begin, s_end, part, kWgParts, tid); }, stamps);
    mlp_barrier();                                               // every wave is done with the ring: its memory carries the stores' tiles
    float* scratch = reinterpret_cast<float*>(ring) + w * (32 * kWgTrS);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            wg_store_block_lds(scratch, slab + kGW2, kMlpHid, 64 * wo + 32 * a, 128 * part + 64 * wi + 32 * b, kMlpHid, acc[a][b], lane,
                               Fmt<NS>::kHalf ? 1.f / (P.gscale * Fmt<NS>::kSH) : 1.f);
    if (P.w3part) w3.finish(slab);                               // (behind the accumulators' stores: its loads have had that long to arrive)
}

// dW1[128 half .. +128, :] = dZ1[:, that half]^T . X and db1 of one slice; waves 4 (row blocks) x 2 (column groups: X blocks 0-2 | blocks
// 3-4 and db1 from a fragment of ones).  Every element's products are accumulated in the order of the register-staged form.
template <int NS>
__device__ __forceinline__ void wgrad_dw1_glds(const MlpWgradParams& P, char* ring, int half, size_t nb, long long s_begin, long long s_end,
                                               float* slab, int tid, unsigned long long* stamps = nullptr)
{
    typedef WgGeom<NS> G;
    constexpr int CH = G::CH, NXMAX = (G::kXPieces + 7) / 8, NPP = G::nA1 + NXMAX, MAXP = NS * NPP;
    const int lane = tid & 63, w = tid >> 6;
    const int rb = w & 3, cg = w >> 2;
    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    // pieces per plane: A (dZ1 half, pieces of four 256-byte rows) w, ..; X (CH rows of 288 bytes, copied as they lie) w, w + 8, ..
    // below kXPieces (some waves have one piece less)
    unsigned offA[G::nA1];
    int rowA[G::nA1];
#pragma unroll
    for (int k = 0; k < G::nA1; ++k) offA[k] = wg_piece_src_swz<256>(w + 8 * k, lane, kMlpHid * 2, 256 * half, rowA[k]);
    const int nx = (G::kXPieces - w + 7) / 8;
    const char* gA = reinterpret_cast<const char*>(P.dz1 + nb);
    const char* gX = reinterpret_cast<const char*>(P.xs);
    const size_t plane_b = P.act_plane * 2, xplane_b = P.xs_plane * 2;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(wg_lds_addr(ring));
    const auto issue = [&](int c, int stage, int k) {
        const long long s = s_begin + (long long)c * CH;
        const int pl = k / NPP, r = k % NPP;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + stage * G::kStage1 + pl * G::kPlane1 + w * 1024);
        if (r < G::nA1) wg_glds16(offA[r], wg_uniform_ptr(gA + pl * plane_b + s * (kMlpHid * 2)), dst + r * 8192);
        else if (r - G::nA1 < nx)
            wg_glds16((unsigned)((w + 8 * (r - G::nA1)) * 1024 + lane * 16), wg_uniform_ptr(gX + pl * xplane_b + s * (kMlpInPad * 2)), dst + CH * 256 + (r - G::nA1) * 8192);
    };
    const auto stage_sync = [&](int c, int stage) {
        const long long s = s_begin + (long long)c * CH;
#pragma unroll
        for (int pl = 0; pl < NS; ++pl) {
            char* dst = ring + stage * G::kStage1 + pl * G::kPlane1 + w * 1024 + lane * 16;
#pragma unroll
            for (int k = 0; k < G::nA1; ++k) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (s + rowA[k] < s_end) v = *reinterpret_cast<const uint4*>(gA + pl * plane_b + s * (kMlpHid * 2) + offA[k]);
                *reinterpret_cast<uint4*>(dst + k * 8192) = v;
            }
#pragma unroll
            for (int k = 0; k < NXMAX; ++k) {
                if (k < nx) {
                    const int o = (w + 8 * k) * 1024 + lane * 16;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (s + o / (kMlpInPad * 2) < s_end) v = *reinterpret_cast<const uint4*>(gX + pl * xplane_b + s * (kMlpInPad * 2) + o);
                    *reinterpret_cast<uint4*>(dst + CH * 256 + k * 8192) = v;
                }
            }
        }
    };
    const bf16x8 ones = bf16x8_ones<Fmt<NS>::kHalf>();
    const auto multiply = [&](int stage, auto&& piece) {
        const char* ta = ring + stage * G::kStage1;
        const char* tb = ta + CH * 256;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            bf16x8 fa[NS];
#pragma unroll
            for (int pl = 0; pl < NS; ++pl) fa[pl] = wg_frag32_swz<256>(ta + pl * G::kPlane1, 16 * ks, 32 * rb, lane);
            if (cg == 0) {
                bf16x8 fb[NS][3];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                    for (int b = 0; b < 3; ++b) fb[pl][b] = wg_frag32_lin<kMlpInPad * 2>(tb + pl * G::kPlane1, 16 * ks, 32 * b, lane);
#pragma unroll
                for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
#pragma unroll
                    for (int b = 0; b < 3; ++b) acc[b] = mfma32<Fmt<NS>::kHalf>(fa[SplitPairs<NS>::a[pi]], fb[SplitPairs<NS>::b[pi]][b], acc[b]);
            } else {
                bf16x8 fb[NS][2];
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
#pragma unroll
                    for (int b = 0; b < 2; ++b) fb[pl][b] = wg_frag32_lin<kMlpInPad * 2>(tb + pl * G::kPlane1, 16 * ks, 32 * (3 + b), lane);
#pragma unroll
                for (int pi = 0; pi < SplitPairs<NS>::n; ++pi)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[b] = mfma32<Fmt<NS>::kHalf>(fa[SplitPairs<NS>::a[pi]], fb[SplitPairs<NS>::b[pi]][b], acc[b]);
#pragma unroll
                for (int pl = 0; pl < NS; ++pl)
                    acc[2] = mfma32<Fmt<NS>::kHalf>(fa[pl], ones, acc[2]);      // every column: db1 of this wave's rows
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = ks; k < MAXP; k += G::KS) piece(k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    WgW3Sums w3;
    // (np differs by wave — some have one X piece less per plane — wave-uniform)
    wg_ring_loop<CH, G::STAGES, MAXP>(s_begin, s_end, NS * (G::nA1 + nx), issue, stage_sync, multiply,
                                      [&] { w3.request(P, s_begin, s_end, 2 + half, kWgParts, tid); }, stamps);
    mlp_barrier();
    float* scratch = reinterpret_cast<float*>(ring) + w * (32 * kWgTrS);
    const int row0 = 128 * half + 32 * rb;
    const float inv_g = 1.f / P.gscale, un1 = Fmt<NS>::kHalf ? inv_g * (1.f / Fmt<NS>::kSX) : 1.f;      // (powers of two: exact)
    if (cg == 0) {
#pragma unroll
        for (int b = 0; b < 3; ++b) wg_store_block_lds(scratch, slab + kGW1, kMlpInPad, row0, 32 * b, kMlpInPad, acc[b], lane, un1);
    } else {
#pragma unroll
        for (int b = 0; b < 2; ++b) wg_store_block_lds(scratch, slab + kGW1, kMlpInPad, row0, 32 * (3 + b), kMlpInPad, acc[b], lane, un1);
        if ((lane & 31) == 0) {
            const int hh = lane >> 5;
#pragma unroll
            for (int i = 0; i < 16; ++i) slab[kGB1 + row0 + (i & 3) + 8 * (i >> 2) + 4 * hh] = Fmt<NS>::kHalf ? acc[2][i] * inv_g : acc[2][i];
        }
    }
    w3.finish(slab);
}

template <int NS = 1>
__global__ __launch_bounds__(kWgThreads) void mlp_wgrad_kernel(const MlpWgradParams P)
{
    __shared__ __attribute__((aligned(1024))) char lds_raw[kWgLdsBytes];
    __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int slice = blockIdx.x, part = blockIdx.y, net = blockIdx.z + P.first_net;
    const long long s_begin = (long long)slice * P.slice_rows;
    long long s_end = s_begin + P.slice_rows;
    if (s_end > P.B) s_end = P.B;
    float* slab = P.slabs + ((size_t)slice * kMlpNets + net) * kGradElems;
    const size_t nb = (size_t)net * P.B * kMlpHid;

#if PNR_MLP_STAMPS
    __shared__ unsigned long long wg_stamp_lds[8][kMlpStampSlots];
    unsigned long long* my_stamps = P.stamps ? wg_stamp_lds[w] : nullptr;
    if (my_stamps && lane < kMlpStampSlots) my_stamps[lane] = 0ull;
    const auto flush_stamps = [&]() {
        if (my_stamps) {
            if (lane == 0) { my_stamps[22] = __builtin_amdgcn_s_memtime(); my_stamps[25] = __builtin_amdgcn_s_memrealtime(); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < kMlpStampSlots)
                P.stamps[((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + w) * kMlpStampSlots + lane] = my_stamps[lane];
        }
    };
    if (my_stamps && lane == 0) { my_stamps[0] = __builtin_amdgcn_s_memtime(); my_stamps[24] = __builtin_amdgcn_s_memrealtime(); }
#else
    unsigned long long* my_stamps = nullptr;
    const auto flush_stamps = [] {};
#endif
    if (part < 2) {
        wgrad_dw2_glds<NS>(P, lds_raw, part, nb, s_begin, s_end, slab, tid, my_stamps);
        flush_stamps();
    } else if (P.w3part) {
        // ---- with the fused kernel's layer-3 partials the fourth role has next to nothing to do (9 us of adding 16 partial rows), and dW1 was the
        // longest role (27 us against dW2's 23.5: five MFMAs per wave and k-step against four, tools/wgrad_stamps.py): roles 2 and 3 each take
        // HALF of dW1's rows (128 output units = columns 128 (part - 2) .. of dZ1) and then adds half of the partials' elements.  Waves 4 (row blocks) x 2
        // (column groups: blocks 0-2 | blocks 3-4 of X's 160 columns).  Every element's products are accumulated in the same order as in the
        // one-role form below: the same bits.
        const int half = part - 2;
        wgrad_dw1_glds<NS>(P, lds_raw, half, nb, s_begin, s_end, slab, tid, my_stamps);
        flush_stamps();
    } else if (part == 2) {
        // dW1 = dZ1^T . X (144 columns) and db1 = dZ1^T . 1 (the tile's column 144 is all ones); wave w: rows 32w..
        __bf16* ta = lds;                        // dZ1 chunk [64][256]
        __bf16* tb = lds + kWgChunk * kTrH;      // X chunk [64][160]: 144 inputs | 1 | 15 zeros
        WG_STAMP(0);
        f32x16 acc[5];
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        WgChunk<kMlpHid> ca; WgChunk<kMlpInPad> cb;
        ca.load(P.dz1 + nb, kMlpHid, s_begin, s_end, tid);
        cb.load(P.xs, kMlpInPad, s_begin, s_end, tid);
        for (long long s = s_begin; s < s_end; s += kWgChunk) {
            mlp_barrier();
            ca.store(ta, kTrH, tid); cb.store(tb, kTrX, tid);
            if (tid < kWgChunk) {                                     // columns 144..159: a one (real rows only), zeros
                bf16x8 one = {(__bf16)((s + tid < s_end) ? 1.0f : 0.0f), (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                *reinterpret_cast<bf16x8*>(tb + tid * kTrX + kMlpInPad) = one;
                *reinterpret_cast<bf16x8*>(tb + tid * kTrX + kMlpInPad + 8) = zero;
            }
            mlp_barrier();
            if (s + kWgChunk < s_end) {
                ca.load(P.dz1 + nb, kMlpHid, s + kWgChunk, s_end, tid);
                cb.load(P.xs, kMlpInPad, s + kWgChunk, s_end, tid);
            }
#pragma unroll
            for (int ks = 0; ks < kWgChunk / 16; ++ks) {
                bf16x8 fb[5];
                const bf16x8 fa = wg_frag32(ta, kTrH, 16 * ks, 32 * w, lane);
#pragma unroll
                for (int b = 0; b < 5; ++b) fb[b] = wg_frag32(tb, kTrX, 16 * ks, 32 * b, lane);
#pragma unroll
                for (int b = 0; b < 5; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[b], acc[b], 0, 0, 0);
            }
        }
        {
#pragma unroll
            for (int b = 0; b < 5; ++b) wg_store_block(slab + kGW1, kMlpInPad, 32 * w, 32 * b, kMlpInPad, acc[b], lane);
            // column 144 of the product = db1: lane c == 16 of block b == 4
            if ((lane & 31) == 16) {
                const int hh = lane >> 5;
#pragma unroll
                for (int i = 0; i < 16; ++i) slab[kGB1 + 32 * w + (i & 3) + 8 * (i >> 2) + 4 * hh] = acc[4][i];
            }
        }
        WG_STAMP(22);
    } else {
        // dW3 [16][256] = G^T . H2, db3 = G^T . 1, db2 = 1^T . dZ2 of the slice from the stored H2 / dZ2 / G (no w3part: the weight-stationary
        // variant, pnr_mlp_backward) with 16x16x32 MFMAs — per 64-sample chunk a product chained over its two 32-sample k-steps from zero,
        // the chunks added in order: the same sums, bit for bit, as the fused kernel's per-tile products added in tile order above.
        __bf16* th = lds;                        // H2 chunk [64][256]
        __bf16* tz = lds + kWgChunk * kTrH;      // dZ2 chunk [64][256]
        __bf16* tg = lds + 2 * kWgChunk * kTrH;  // G chunk [64][16] as bf16
        f32x4 aw3[2], ab2[2], ab3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 2; ++b) { aw3[b] = ab3; ab2[b] = ab3; }
        WgChunk<kMlpHid> ch2, cz2;
        ch2.load(P.h2 + nb, kMlpHid, s_begin, s_end, tid);
        cz2.load(P.dz2 + nb, kMlpHid, s_begin, s_end, tid);
        for (long long s = s_begin; s < s_end; s += kWgChunk) {
            mlp_barrier();
            ch2.store(th, kTrH, tid); cz2.store(tz, kTrH, tid);
            if (tid < 2 * kWgChunk) {
                const int row = tid >> 1, half = tid & 1;
                f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
                if (s + row < s_end) {
                    const float* gp = P.g_head + ((size_t)net * P.B + s + row) * kMlpHead + 8 * half;
                    g0 = *reinterpret_cast<const f32x4*>(gp); g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                }
                bf16x8 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pk[j] = (__bf16)g0[j]; pk[4 + j] = (__bf16)g1[j]; }
                *reinterpret_cast<bf16x8*>(tg + row * kTrG + 8 * half) = pk;
            }
            mlp_barrier();
            if (s + kWgChunk < s_end) {
                ch2.load(P.h2 + nb, kMlpHid, s + kWgChunk, s_end, tid);
                cz2.load(P.dz2 + nb, kMlpHid, s + kWgChunk, s_end, tid);
            }
            f32x4 tw3[2], tb2[2], tb3;
            mlp_tile_w3_products(tg, kTrG, th, kTrH, lane, w, tw3, tb3);
            mlp_tile_b2_products(tz, kTrH, lane, w, tb2);
#pragma unroll
            for (int b = 0; b < 2; ++b) { aw3[b] += tw3[b]; ab2[b] += tb2[b]; }
            ab3 += tb3;
        }
        const int c16 = lane & 15, g = lane >> 4;                  // C: col = lane & 15, rows 4g .. 4g+3
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) slab[kGW3 + (4 * g + j) * kMlpHid + 32 * w + 16 * b + c16] = aw3[b][j];
            if (g == 0) slab[kGB2 + 32 * w + 16 * b + c16] = ab2[b][0];            // every row of 1^T . dZ2 is db2
        }
        if (w == 0 && c16 == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) slab[kGB3 + 4 * g + j] = ab3[j];           // every column of G^T . 1 is db3
        }
    }
}

struct MlpReduceParams {
    const float* slabs;        // [slices][2][kGradElems]
    int slices;
    float* gw1[kMlpNets]; float* gb1[kMlpNets];    // gradients in the master parameters' layouts
    float* gw2[kMlpNets]; float* gb2[kMlpNets];
    float* gw3[kMlpNets]; float* gb3[kMlpNets];
    int n3[kMlpNets];
    int accumulate;            // 1: add to what the gradient tensors hold (autograd accumulation), 0: overwrite
    const float* scale;        // device scalar multiplied into the sums (the upstream d / d loss), or null: 1
};

// Sum the slices' slabs in slice order (deterministic) and scatter into the parameter-shaped gradients.
__global__ __launch_bounds__(256) void mlp_reduce_kernel(const MlpReduceParams P)
{
    const int net = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kGradElems) return;
    float* dst = nullptr;
    if (e < kGW2) { const int o = e / kMlpInPad, k = e % kMlpInPad; if (k < kMlpIn) dst = P.gw1[net] + o * kMlpIn + k; }
    else if (e < kGW3) dst = P.gw2[net] + (e - kGW2);
    else if (e < kGB1) { const int r = (e - kGW3) / kMlpHid; if (r < P.n3[net]) dst = P.gw3[net] + (e - kGW3); }
    else if (e < kGB2) dst = P.gb1[net] + (e - kGB1);
    else if (e < kGB3) dst = P.gb2[net] + (e - kGB2);
    else if (e - kGB3 < P.n3[net]) dst = P.gb3[net] + (e - kGB3);
    if (!dst) return;
    float s = 0.f;
    const float* p = P.slabs + (size_t)net * kGradElems + e;
    for (int k = 0; k < P.slices; ++k) s += p[(size_t)k * kMlpNets * kGradElems];
    if (P.scale) s *= *P.scale;
    *dst = P.accumulate ? (*dst + s) : s;
}

// slabs -> one flat gradient [2][kGradElems] (the bucket a multi-GPU run all-reduces), summed in slice order
// (elements [first, first + count) of the bucket: one net's half when the nets are driven as two chains)
__global__ __launch_bounds__(256) void mlp_reduce_flat_kernel(const float* __restrict__ slabs, int slices, float* __restrict__ flat, int first, int count)
{
    const int i = first + blockIdx.x * 256 + threadIdx.x;
    if (i >= first + count) return;
    constexpr size_t kStride = (size_t)kMlpNets * kGradElems;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= slices; k += 8) {                       // slice order, eight loads in flight (see mlp_adam_kernel)
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = slabs[(size_t)(k + j) * kStride + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    for (; k < slices; ++k) s += slabs[(size_t)k * kStride + i];
    flat[i] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// Adam on the float32 master parameters, fused with the slab reduction in front of it and with the bf16 weight
// packing behind it: one launch turns per-slice partial gradients into the next forward's operands.  torch.optim.Adam
// semantics (no weight decay, no amsgrad): m += (g - m)(1 - b1); v = b2 v + (1 - b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), t = *step (incremented once per update by the loss
// kernel's finishing launch).  State m, v live in the padded slab layout [2][kGradElems].
// ---------------------------------------------------------------------------------------------------------------
struct MlpAdamParams {
    const float* grad;         // slices x [2][kGradElems] partial gradients (slabs), or one flat all-reduced gradient
    int slices;                // 1 for a flat gradient
    float grad_scale;          // e.g. 1 / world size after a summing all-reduce
    float* w1[kMlpNets]; float* b1[kMlpNets];      // master parameters (the caller's tensors)
    float* w2[kMlpNets]; float* b2[kMlpNets];
    float* w3[kMlpNets]; float* b3[kMlpNets];
    int n3[kMlpNets];
    float* m; float* v;        // [2][kGradElems]
    const float* step;         // device scalar: number of updates including this one
    float lr, beta1, beta2, eps;
    __bf16* wpack;             // [2][kPackElems]: refreshed in place
    float* bias;               // [2][kBiasElems]
    // the loss means of the update ride in this launch (one extra block; null partials: none)
    const float* partials;     // [loss_rows][8] of the fused forward + loss + backward kernel
    long long loss_rows, batch;
    float* means;              // [8]
    const float* kl_coeff; const float* ent_coeff; float vf_coeff;
    int first_net;             // blockIdx.y + first_net = net
    int planes;                // bf16 planes of the packed weights that are refreshed (1: bf16 operands; 2, 3: split float32)
};

constexpr int kAdamVec = 4;          // consecutive gradient-layout elements per thread (every region of the layout starts on a multiple of 4)
constexpr int kAdamBlocks = (kGradElems / kAdamVec + 255) / 256;     // + 1: the loss-means block
static_assert(kGradElems % kAdamVec == 0 && kGW2 % 4 == 0 && kGW3 % 4 == 0 && kGB1 % 4 == 0 && kGB2 % 4 == 0 && kGB3 % 4 == 0 && kMlpInPad % 4 == 0, "");

// r04: FOUR elements per thread — the bias corrections' two powf once per four elements, the 32 slabs as 32 16-byte loads in one batch
// (slice order in the sum: same bits as before), one 8-byte store per fragment-native bf16 group; 840 waves instead of 6 712.  Measured
// (tools/adam_floor.py, profiles/r04_g_adam_four_per_thread_ab.txt): 10.6 -> 10.1 us with the learner's 32 slabs, 8.0 -> 7.7 us back to
// back with ONE flat gradient — neither the instruction stream (~700 per wave before) nor the slabs' four dependent round trips were
// the bound: ~6 us of the launch are its floor (launch, one load round trip from HBM, the stores' drain) and ~4 us the 27 MB of slabs.
__global__ __launch_bounds__(256) void mlp_adam_kernel(const MlpAdamParams P)
{
    const int net = blockIdx.y + P.first_net;
    if (blockIdx.x == 0) {
        // the extra block (the FIRST one, so that it starts with the launch and not as its tail): the five loss means from the fused
        // kernel's per-workgroup rows (ppo_loss_finish_split_kernel's job), in the shadow of the other blocks instead of a launch of its own
        __shared__ float red[4][kPpoSums];
        if (blockIdx.y != 0 || !P.partials) return;
        ppo_loss_means_block(P.partials, P.loss_rows, P.batch, P.means, P.kl_coeff, P.ent_coeff, P.vf_coeff, red);
        return;
    }
    const int e = ((blockIdx.x - 1) * 256 + threadIdx.x) * kAdamVec;
    if (e >= kGradElems) return;
    const size_t si = (size_t)net * kGradElems + e;
    // requested first, so that they arrive under the slabs' round trip
    const float t = *P.step;
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(P.m + si), v4 = *reinterpret_cast<const f32x4*>(P.v + si);

    // where the four elements live: master parameter (dst, `valid` of them exist), fragment-native bf16 copies (wp0: four consecutive
    // elements of one fragment row; wp1[j]: the transposed copy, one row each), bias copy
    float* dst = nullptr;
    int valid = kAdamVec, wp0 = -1, bp = -1;
    int wp1[kAdamVec] = {-1, -1, -1, -1};
    if (e < kGW2) {
        const int o = e / kMlpInPad, k = e % kMlpInPad;
        wp0 = kOffW1 + frag32_off(o, k, kMlpInPad / 16);
        dst = P.w1[net] + o * kMlpIn + k; valid = kMlpIn - k;
    } else if (e < kGW3) {
        const int r = e - kGW2, o = r / kMlpHid, i = r % kMlpHid;
        dst = P.w2[net] + r; wp0 = kOffW2 + frag32_off(o, i, kMlpHid / 16);
#pragma unroll
        for (int j = 0; j < kAdamVec; ++j) wp1[j] = kOffW2T + frag32_off(i + j, o, kMlpHid / 16);
    } else if (e < kGB1) {
        const int r = e - kGW3, row = r / kMlpHid, f = r % kMlpHid;
        wp0 = kOffW3 + frag16_off(row, f);
#pragma unroll
        for (int j = 0; j < kAdamVec; ++j) wp1[j] = kOffW3T + frag32_off(f + j, row, 1);
        dst = P.w3[net] + r; valid = row < P.n3[net] ? kAdamVec : 0;
    } else if (e < kGB2) { dst = P.b1[net] + (e - kGB1); bp = e - kGB1; }
    else if (e < kGB3) { dst = P.b2[net] + (e - kGB2); bp = kMlpHid + (e - kGB2); }
    else { bp = 2 * kMlpHid + (e - kGB3); dst = P.b3[net] + (e - kGB3); valid = P.n3[net] - (e - kGB3); }
    float p0[kAdamVec];
#pragma unroll
    for (int j = 0; j < kAdamVec; ++j) p0[j] = j < valid ? dst[j] : 0.f;

    // the slices' partial gradients, summed in slice order per element (the order mlp_reduce_flat_kernel uses: same bits)
    f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
    if (valid > 0) {
        constexpr size_t kStride = (size_t)kMlpNets * kGradElems;
        const float* gp = P.grad + si;
        int k = 0;
        if (P.slices == 32) {                               // the learner's slab count: one batch of loads, one round trip
            f32x4 x[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gp + (size_t)j * kStride));
#pragma unroll
            for (int j = 0; j < 32; ++j) g4 += x[j];
            k = 32;
        }
        for (; k + 8 <= P.slices; k += 8) {
            f32x4 x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const f32x4*>(gp + (size_t)(k + j) * kStride);
#pragma unroll
            for (int j = 0; j < 8; ++j) g4 += x[j];
        }
        for (; k < P.slices; ++k) g4 += *reinterpret_cast<const f32x4*>(gp + (size_t)k * kStride);
    }
    const float bc1 = 1.0f - powf(P.beta1, t), bc2 = 1.0f - powf(P.beta2, t);
    const float rbc2 = sqrtf(bc2), lr1 = P.lr / bc1;
    f32x4 mo = m4, vo = v4;
    float pv[kAdamVec];
#pragma unroll
    for (int j = 0; j < kAdamVec; ++j) {
        pv[j] = 0.f;
        if (j < valid) {
            const float g = g4[j] * P.grad_scale;
            const float m = m4[j] + (g - m4[j]) * (1.0f - P.beta1);
            const float v = P.beta2 * v4[j] + (1.0f - P.beta2) * g * g;
            const float denom = sqrtf(v) / rbc2 + P.eps;
            pv[j] = p0[j] - lr1 * (m / denom);
            mo[j] = m; vo[j] = v; dst[j] = pv[j];
        }
    }
    if (valid > 0) { *reinterpret_cast<f32x4*>(P.m + si) = mo; *reinterpret_cast<f32x4*>(P.v + si) = vo; }
    __bf16* wp = P.wpack + (size_t)net * kPackElems;
    float r[kAdamVec] = {pv[0], pv[1], pv[2], pv[3]};
    bf16x4_t hp[2];
    if (P.planes == 2) split_quad<2>(r, hp, Fmt<2>::kSW);          // two fp16 planes of the weight x 2^8 (Fmt<2>)
    for (int pl = 0; pl < P.planes; ++pl) {
        bf16x4 b;
        if (P.planes == 2) b = hp[pl];
        else {
#pragma unroll
            for (int j = 0; j < kAdamVec; ++j) { b[j] = (__bf16)r[j]; r[j] -= (float)b[j]; }
        }
        __bf16* w = wp + (size_t)pl * kWPlane;
        if (wp0 >= 0) *reinterpret_cast<bf16x4*>(w + wp0) = b;
#pragma unroll
        for (int j = 0; j < kAdamVec; ++j) if (wp1[j] >= 0) w[wp1[j]] = b[j];
    }
    if (bp >= 0) *reinterpret_cast<f32x4*>(P.bias + net * kBiasElems + bp) = f32x4{pv[0], pv[1], pv[2], pv[3]};
}

}  // namespace pnr

#pragma clang fp contract(off)
