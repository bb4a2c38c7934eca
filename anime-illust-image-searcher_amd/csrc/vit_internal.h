// vit_internal.h -- shared declarations of the ViT forward kernels (gemm.hip, attn.hip, vit.hip).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace hipts {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// The 16-bit MFMA operand type is bf16 (default, what BASELINE.json names) or IEEE half (F16 = true,
// hipts_vit_config_t.operand_f16): same MFMA rate, 3 more mantissa bits.  Buffers are declared bf16_t
// either way (2-byte elements); only the conversions and the MFMA opcode differ.
template <bool F16>
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    if constexpr (F16) {
        f16x4 h;
        h[0] = (_Float16)a; h[1] = (_Float16)b; h[2] = (_Float16)c; h[3] = (_Float16)d;
        return __builtin_bit_cast(bf16x4, h);
    } else {
        bf16x4 o;
        o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
        return o;
    }
}
template <bool F16>
__device__ __forceinline__ bf16_t to_op(float a) {
    if constexpr (F16) return __builtin_bit_cast(bf16_t, (_Float16)a);
    else return (bf16_t)a;
}
template <bool F16>
__device__ __forceinline__ float from_op(bf16_t a) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, a);
    else return (float)a;
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- 8-bit operands (OCP e4m3fn, the gfx950 fp8; BASELINE.json configs[4]) ----
// v_cvt_pk_fp8_f32 rounds to nearest even and turns everything above 464 into NaN (measured, tools/micro/
// fp8_mfma.hip), so values are clamped to the largest finite e4m3 (448) first.
__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
    b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f);
    d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}
// D += A B with e4m3 operands, K = 128: lane (r = lane & 15, q = lane >> 4) holds bytes 32 q .. 32 q + 31 of row r of
// both operands (checked with exact data, tools/micro/fp8_mfma.hip); the E8M0 scale bytes multiply the products by
// 2^(scale - 127) in the instruction -- how the per-tensor power-of-two weight scale is undone for free.
__device__ __forceinline__ f32x4 mfma_16x16x128_e4m3(i32x8 a, i32x8 b, f32x4 c, int scale_a, int scale_b) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
}

// ---- host-side operand conversion (weights are converted once at upload) ----
// float -> e4m3fn bits, round to nearest even, saturating at +-448 (NaN -> 0x7f)
inline uint8_t f32_to_e4m3_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (uint8_t)(sign | 0x7f);
    float a;
    memcpy(&a, &u, 4);
    if (a >= 448.0f) return (uint8_t)(sign | 0x7e);
    if (a < 0.0009765625f) return sign;                      // < 2^-10: below half of the smallest subnormal (ties to even -> 0)
    int e;
    (void)frexpf(a, &e);                                     // a = m 2^e, m in [0.5, 1)
    int ex = e - 1;                                          // a = 1.x * 2^ex
    if (ex < -6) ex = -6;                                    // subnormal range: fixed quantum 2^-9
    const float q = ldexpf(1.0f, ex - 3);                    // quantum
    const float n = nearbyintf(a / q);                       // RNE under the default rounding mode; exact (a / q has <= 24 bits)
    float v = n * q;
    if (v >= 448.0f) return (uint8_t)(sign | 0x7e);
    if (v < 0.015625f) return (uint8_t)(sign | (uint8_t)n);  // subnormal: mantissa = n (n = 8 is the first normal and encodes alike)
    int e2;
    const float m2 = frexpf(v, &e2);                         // v = m2 2^e2
    const int bexp = e2 - 1 + 7;
    const int mant = (int)((m2 * 2.0f - 1.0f) * 8.0f);
    return (uint8_t)(sign | (bexp << 3) | mant);
}
inline float e4m3_bits_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) {
        const uint32_t qnan = 0x7fc00000u;      // (not the NAN macro: attn.hip is built with -fno-honor-nans)
        memcpy(&f, &qnan, 4);
    } else if (e == 0) f = ldexpf((float)m, -9);
    else f = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -f : f;
}
inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// float -> IEEE half bits, round to nearest even (subnormals kept)
inline uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                     // rounds to >= 65520 -> inf
    if (x < 0x38800000u) {                                                       // below the smallest normal half
        if (x < 0x33000000u) return (uint16_t)sign;                              // < 2^-25 -> 0
        const int shift = 126 - (int)(x >> 23);                                  // 14..24: value = mant24 * 2^-24 >> shift
        uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        const uint32_t round = (1u << (shift - 1)) - 1 + ((mant >> shift) & 1u);
        mant += round;
        return (uint16_t)(sign | (mant >> shift));
    }
    const uint32_t round = 0xfffu + ((x >> 13) & 1u);
    x += round;
    return (uint16_t)(sign | ((x - 0x38000000u) >> 13));
}
inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, out;
    if (e == 0) {
        if (m == 0) out = sign;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            out = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) out = sign | 0x7f800000u | (m << 13);
    else out = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &out, 4);
    return f;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// rows x cols float matrix -> bf16 / half bits, zero padded to rows_pad rows, uploaded into buf
inline int upload_matrix16(DevBuf& buf, const float* data, int rows, int cols, int rows_pad, bool f16) {
    std::vector<uint16_t> h((size_t)rows_pad * cols, 0);
    if (f16) for (size_t i = 0; i < (size_t)rows * cols; ++i) h[i] = f32_to_f16_rne(data[i]);
    else for (size_t i = 0; i < (size_t)rows * cols; ++i) h[i] = f32_to_bf16_rne(data[i]);
    HIPTS_TRY(buf.alloc(h.size() * 2));
    return upload(buf.p, h.data(), h.size() * 2);
}

// rows x cols float matrix -> e4m3 bytes of W * 2^w_exp (per-tensor power-of-two scale that puts max |W| in
// (224, 448]; the GEMM undoes it through the MFMA scale operand), zero padded to rows_pad rows
inline int upload_matrix8(DevBuf& buf, const float* data, int rows, int cols, int rows_pad, int* w_exp) {
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)rows * cols; ++i) mx = fmaxf(mx, fabsf(data[i]));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) {
        e = (int)floorf(log2f(448.0f / mx));
        if (ldexpf(mx, e) > 448.0f) --e;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    std::vector<uint8_t> h((size_t)rows_pad * cols, 0);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) h[i] = f32_to_e4m3_rne(ldexpf(data[i], e));
    *w_exp = e;
    HIPTS_TRY(buf.alloc(h.size()));
    return upload(buf.p, h.data(), h.size());
}

// GEMM  C[M,N] = A[M,K] (bf16, row-major) x W[N,K]^T (bf16, row-major: torch Linear layout), fp32
// accumulate on MFMA, fused epilogue.  K % 64 == 0; W must be allocated (zero padded) up to a
// multiple of 256 rows; A rows beyond M are never read (row index clamped), stores are masked.
enum GemmEpilogue {
    EPI_PATCH = 0,   // x[m][n]  = acc * qscale + bias[n] + pos[(m % tokens)][n]        (fp32 out)
    EPI_QK = 1,      // q/k[b][h][t][d] = bf16((acc + bias[n]) * (n < dim ? qscale : 1)) (n in [0, 2*dim)); with N = 3*dim also v (out3_bf16)
    EPI_VT = 2,      // vT[b][h][d][t] = bf16(acc + bias[dim2 + n])                      (n in [0, dim))
    EPI_RESID = 3,   // x[m][n] += acc + bias[n]                                        (fp32 in/out)
    EPI_GELU = 4,    // out[m][n] = bf16(gelu(acc + bias[n]))
    EPI_HEAD = 5,    // logits[m][n] = acc + bias[n]; probs[m][n] = sigmoid(logits)     (fp32 out, n < N)
    // MetaFormer (CCIP encoder) epilogues
    EPI_STAR = 6,    // out[m][n] = bf16(star_scale * relu(acc + bias[n])^2 + star_bias)  (StarReLU)
    EPI_RESCALE = 7, // x[m][n] = x[m][n] * res_scale[n] + acc + bias[n]                (fp32 in/out)
    EPI_BIAS = 8,    // x[m][n] = acc + bias[n]                                         (fp32 out)
    EPI_QK_ROPE = 10, // EPI_QK with the EVA02 2-D rotary embedding applied to the fp32 result before the single rounding: token t
                      // (1 <= t <= rope_tokens) of an image rotates each column pair (2i, 2i+1) of a 64-wide head by the angle in
                      // rope[t - 1][i] = (sin, cos); token 0 (class) and padding rows pass through.  Staged epilogue only.
    EPI_SWIGLU = 11,  // W rows interleaved per 32 hidden units [gate 0..31 | value 0..31]: out[m][u] = bf16(silu(gate + bias) * (value + bias)),
                      // N / 2 output columns (row stride ld_out).  Staged epilogue only.
    EPI_RESID_ROWSTAT = 12, // x[m][n] += rowstat[m].x * acc - rowstat[m].y * col_u[n] + bias[n]: the residual GEMM of a LayerNorm-ed operand with
                            // the LayerNorm folded in -- A holds gamma * p (raw rows times the LayerNorm weight, written by the producer),
                            // col_u[n] = sum_k W[n][k] gamma[k], bias[n] = W beta + b, rowstat[m] = (rstd, rstd * mean) of the raw row m
                            // (from the stat_part partials of EPI_SWIGLU).  W itself is unchanged: folding gamma into W would round W again.
    EPI_RESID_XG = 13, // (with res_scale set: x[m][n] = x[m][n] * res_scale[n] + acc + bias[n], the CAFormer's scaled residual; general form only)
                       // the residual GEMM with the NEXT LayerNorm prepared in its epilogue (and optionally the fold of EPI_RESID_ROWSTAT on its
                       // input): x[m][n] += rowstat[m].x * acc - rowstat[m].y * col_u[n] + bias[n] (rowstat null: += acc + bias), then
                       // out_bf16[m][n] = 16bit(x[m][n] * ln_gamma[n]) and per (256-column tile, row) partial (sum x, sum x^2) into stat_part
                       // (the four waves of a row reduce through LDS; persistent loop only).
                       // A consumer GEMM (QK, QK_ROPE, VT, GELU, SWIGLU with rowstat / col_u set) then computes
                       // W LN(x) + b = rstd (W (gamma x)) - rstd mean (W gamma) + (W beta + b) without a LayerNorm pass over x.
    EPI_RESID_XGI = 14, // EPI_RESID_XG with the fold on its input (rowstat / stat_in + col_u); plain EPI_RESID_XG ignores them, which
                        // leaves it the registers for four row blocks of residual loads in flight instead of two
    EPI_RESID_LN = 9 // x[m][n] = x[m][n] * (res_scale ? res_scale[n] : 1) + acc + bias[n]  (fp32 in/out), AND the LayerNorm
                     // of the new row: xn[m][n] = bf16((x - mean) * rstd * ln_gamma[n] (+ ln_beta[n])).  Needs the whole
                     // row in one tile: N <= 256.  Saves the separate LayerNorm pass over x (HBM-bound).
};

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* W;
    int M, N, K;
    const float* bias;
    // epilogue specific
    float* out_f32 = nullptr;       // PATCH / RESID / HEAD(logits)
    float* out2_f32 = nullptr;      // HEAD(probs, may be null)
    bf16_t* out_bf16 = nullptr;     // GELU; QK: q base; VT: vT base
    bf16_t* out2_bf16 = nullptr;    // QK: k base
    bf16_t* out3_bf16 = nullptr;    // QK / QK_ROPE with N = 3 * dim: v base, same [b][h][t][d] layout as q and k (no scale, no rotation)
    const float* pos = nullptr;     // PATCH
    int tokens = 0;                 // tokens per image (PATCH, QK, VT)
    int tokens_pad = 0;             // padded token count of the q/k/vT layouts
    int heads = 0, dim = 0;         // QK / VT
    int hd_log2 = 6;                // QK / VT: log2(head_dim), 6 (ViT) or 5 (CAFormer)
    const float* res_scale = nullptr;   // RESCALE (required) / RESID_LN (optional): per-column scale of the residual
    const float* ln_gamma = nullptr;    // RESID_LN
    const float* ln_beta = nullptr;     // RESID_LN, optional
    float ln_eps = 1e-6f;               // RESID_LN
    float star_scale = 1.0f, star_bias = 0.0f;   // STAR
    int star_kind = 0;                  // STAR: 0 = StarReLU, 1 = SiLU (x * sigmoid(x)), 2 = identity (bias only)
    float qscale = 1.0f;
    float* stat_part = nullptr;     // SWIGLU (optional): per (64-column block, row) partial (sum, sum of squares) of the fp32 products,
    int stat_stride = 0;            //   float2 at stat_part[2 * (block * stat_stride + m)]
    const float* stat_in = nullptr; // folded LayerNorm on the A operand, statistics still as the producer's partials: float2 (sum x, sum x^2) at
    int stat_in_blocks = 0;         //   stat_in[2 * (b * stat_in_stride + m)], b < stat_in_blocks; the epilogue finishes them (ln_dim columns, ln_eps)
    int stat_in_stride = 0;
    int ln_dim = 0;
    const float* rowstat = nullptr; // ... or finished: folded LayerNorm on the A operand (RESID_ROWSTAT, RESID_XG, QK, QK_ROPE, VT, GELU, SWIGLU): float2 per row
    const float* col_u = nullptr;   //   (rstd, rstd * mean), and W gamma per output column; `bias` then holds W beta + b
    const float* rope = nullptr;    // QK_ROPE: [rope_tokens][32] (sin, cos) pairs
    int rope_tokens = 0;
    int gelu_tanh = 1;
    int ld_out = 0;                 // row stride of out (elements); 0 = N
    int f16 = 0;                    // operands (and 16-bit outputs) are IEEE half instead of bf16
    int op8 = 0;                    // A and W are e4m3 bytes (K counts elements, K % 128 == 0; W holds W * 2^w_exp); 16-bit outputs are half
    int w_exp = 0;                  // op8: the weight scale exponent
    int out8 = 0;                   // STAR / RESID_LN: the 16-bit output (out_bf16) is written as e4m3 bytes instead (ld_out in bytes)
    // Split-K tail (round 4; persistent loop, residual epilogues only): the last tiles of the launch order -- the partial last round, or all
    // tiles of a launch smaller than the chip -- are cut into sk_slices K ranges, each a work item of its own.  A slice leaves its fp32
    // accumulators in a slab of the workspace (write-through stores) and draws a ticket; the LAST arriver adds the slabs in slice order
    // (the bits do not depend on who arrives last) and runs the epilogue.  Nobody waits for anybody: no residency assumption.
    void* sk_ws = nullptr;                  // caller-owned, zeroed once, one per stream: [4 KB of tickets][slabs]; null = never split
    size_t sk_ws_bytes = 0;
    int sk_first = 0, sk_slices = 1;        // set by the launcher: tiles [sk_first, tiles) are split sk_slices ways
    int raster_gn = 0;                      // > 0: column groups of this many column tiles outermost, row-major inside (set by the launcher)
    int raster_gm = 0;                      // > 0: tile order in groups of this many row panels, column-major inside (set by the launcher)
    int x_blocked = 0;                      // out_f32 of the residual epilogues (RESID, RESCALE, RESID_ROWSTAT, RESID_XG / XGI) is stored as 16 x 16 blocks of
                                            // 1 KB, [m / 16][n / 16][m % 16][n % 16] (gemm_epi.h::x_off); ld % 16 == 0, rows allocated up to a multiple of 16
    float* resid_rowmajor_out = nullptr;    // x_blocked, EPI_RESID: the new rows are written here, row-major, instead of back into the blocked stream
    int epi_prefetch = 0;                   // interior residual epilogue: its fp32 tile is requested into L2 during the last K-tile (HIPTS_EPI_PREFETCH)
    int epi_prio = 0;                       // the two waves of a SIMD alternate s_setprio through the epilogue's steps (set by the launcher from HIPTS_EPI_PRIO)
    int shared_chip = 0;                    // another stream's kernels run concurrently (sub-batch streams)
    int trace = 0;                          // diagnostic: per-workgroup timeline records instead of stamps (dw loop)
    unsigned long long* stamps = nullptr;   // diagnostic build only (tools/gemm_bench.py): s_memtime stamps of block 0
};

int launch_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s);
// gemm4.hip: the 4-wave (one wave per SIMD) main loop for launches of whole 256 x 256 tiles with an even number of K-tiles and the
// GELU / QK / RESID_XG epilogues; *handled = false: not built for this launch, the caller goes on to gemm_pp_kernel
int launch_gemm_q4(GemmEpilogue epi, const GemmArgs& a, hipStream_t s, bool* handled);
constexpr unsigned GEMM_Q4_DEFAULT_MASK = 0u;      // epilogues whose eligible launches take gemm4.hip by default (HIPTS_GEMM_Q4 overrides)
void set_gemm_q4_mask(unsigned mask);
long long gemm_q4_launch_count();              // launches gemm4.hip has taken so far (tests: the comparison really compared)
constexpr size_t GEMM_SK_WS_BYTES = 4096 + 256 * (size_t)(256 * 256 * 4);      // tickets + one 256 x 256 fp32 slab per work item of a full round

// softmax(Q K^T) V for every (image, head): q,k [B*H][tokens_pad][64] bf16 (q pre-scaled by
// head_dim^-0.5), vT [B*H][64][tokens_pad] bf16, out [B*tokens][H*64] bf16.
// out_tokens_stride: rows an image owns in `out` (0 = tokens).
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int batch, int heads, int tokens,
                     int tokens_pad, bool f16, hipStream_t s, int head_dim = 64, int out_tokens_stride = 0);

// The same for head_dim 64 with V in its natural layout: q, k, v [B*H][tokens_pad][64] (attn2.hip: LDS-DMA ring, transposed LDS reads of V).
// variant 0 = default geometry (HIPTS_ATTN2 overrides): 1: 4 waves x 64 query rows, 2: 8 waves x 32, 3: 4 waves x 32.
// split_lo != 0: rows of out are [hi | lo * lo_scale], 2 * heads * 64 wide -- each output value as two 16-bit halves (the consumer GEMM runs
// K = 2 * dim against [W | W / lo_scale]; split_lo_scale(f16) is the scale both sides use).
int launch_attention2(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int batch, int heads, int tokens, int tokens_pad, bool f16,
                      hipStream_t s, int out_tokens_stride = 0, int variant = 0, int split_lo = 0, float lo_scale = 1.0f);
// The power of two that the low half of a hi | lo operand pair is multiplied by (and the matching weight copy divided by).  IEEE half: the
// low half of a value v is about v * 2^-11, subnormal (< 2^-14) for every |v| < 1/8 -- which covers most activations; 64 moves it back into
// the normal range while W / 64 stays normal for |W| >= 2^-8 and exact (a power of two) otherwise down to 2^-18.  Measured, ViT-B/16 batch
// 64: with scale 1 the forward ran 26 % slower than without the split (5295 -> 3919 images/s; bf16 operands, whose low halves are normal
// numbers: 6 %) -- subnormal half operands slow the matrix pipe down.  HIPTS_SPLIT_LO_SCALE overrides (A/B).
inline float split_lo_scale(bool f16) {
    static const float env = getenv("HIPTS_SPLIT_LO_SCALE") ? (float)atof(getenv("HIPTS_SPLIT_LO_SCALE")) : 0.0f;
    if (env > 0.0f) return env;
    return f16 ? 64.0f : 1.0f;
}

int attention2_read_stamps(unsigned long long* host, int n);      // measurement-only builds (HIPTS_X_STAMPS)

// out[row][:] = bf16((x[row][:] - mean) * rstd * g + b)  (b may be null: bias-free LayerNorm); D % 4 == 0, D <= 1024
int launch_layernorm(const float* x, const float* g, const float* b, bf16_t* out, int64_t rows, int D, float eps, bool f16,
                     hipStream_t s);
// rowstat[m] = (rstd, rstd * mean) from `blocks` partial (sum, sum of squares) pairs per row: part[2 * (b * stride + m)], D columns in all
int launch_rowstat(const float* part, float* rowstat, int M, int stride, int blocks, int D, float eps, hipStream_t s);
// u[n] = sum_k W[n][k] gamma[k], c[n] = sum_k W[n][k] beta[k] + bias[n] (beta / bias may be null) from the uploaded 16-bit W
int launch_fold_ln(const bf16_t* W, bool f16, const float* gamma, const float* beta, const float* bias, float* u, float* c, int N, int K,
                   hipStream_t s);
// mlp.hip: fc1 -> StarReLU -> fc2 -> scaled residual -> LayerNorm of a CAFormer block in one kernel (C = 128 / 256, hidden 4 C; xn_out may be xn)
std::vector<uint16_t> mlp_weight_image(const float* w1, const float* w2, int C);
bool mlp_fused_supports(int C);
int launch_mlp_fused(const bf16_t* xn, const void* wimg, float* x, const float* res_scale, const float* gamma, bf16_t* xn_out, int M, int C,
                     float star_s, float star_b, float eps, hipStream_t s, int waves = 0,       // waves per workgroup: 0 = chosen by the launch's size, 4, 8
                     int xblk = 0);                                                             // x is stored as 16 x 16 blocks (gemm_epi.h::x_off)
// the same with e4m3 output bytes (the A operand of an op8 GEMM)
int launch_layernorm8(const float* x, const float* g, const float* b, uint8_t* out, int64_t rows, int D, float eps, hipStream_t s);

}  // namespace hipts
