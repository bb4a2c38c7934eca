// vit_internal.h -- shared declarations of the ViT forward kernels (gemm.hip, attn.hip, vit.hip).
#pragma once
#include "common.h"

namespace hipts {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// GEMM  C[M,N] = A[M,K] (bf16, row-major) x W[N,K]^T (bf16, row-major: torch Linear layout), fp32
// accumulate on MFMA, fused epilogue.  K % 64 == 0; W must be allocated (zero padded) up to a
// multiple of 256 rows; A rows beyond M are never read (row index clamped), stores are masked.
enum GemmEpilogue {
    EPI_PATCH = 0,   // x[m][n]  = acc * qscale + bias[n] + pos[(m % tokens)][n]        (fp32 out)
    EPI_QK = 1,      // q/k[b][h][t][d] = bf16((acc + bias[n]) * (n < dim ? qscale : 1)) (n in [0, 2*dim))
    EPI_VT = 2,      // vT[b][h][d][t] = bf16(acc + bias[dim2 + n])                      (n in [0, dim))
    EPI_RESID = 3,   // x[m][n] += acc + bias[n]                                        (fp32 in/out)
    EPI_GELU = 4,    // out[m][n] = bf16(gelu(acc + bias[n]))
    EPI_HEAD = 5     // logits[m][n] = acc + bias[n]; probs[m][n] = sigmoid(logits)     (fp32 out, n < N)
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
    const float* pos = nullptr;     // PATCH
    int tokens = 0;                 // tokens per image (PATCH, QK, VT)
    int tokens_pad = 0;             // padded token count of the q/k/vT layouts
    int heads = 0, dim = 0;         // QK / VT
    float qscale = 1.0f;
    int gelu_tanh = 1;
    int ld_out = 0;                 // row stride of out (elements); 0 = N
    int f16 = 0;                    // operands (and 16-bit outputs) are IEEE half instead of bf16
    int shared_chip = 0;                    // another stream's kernels run concurrently (sub-batch streams)
    int trace = 0;                          // diagnostic: per-workgroup timeline records instead of stamps (dw loop)
    unsigned long long* stamps = nullptr;   // diagnostic build only (tools/gemm_bench.py): s_memtime stamps of block 0
};

int launch_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s);

// softmax(Q K^T) V for every (image, head): q,k [B*H][tokens_pad][64] bf16 (q pre-scaled by
// head_dim^-0.5), vT [B*H][64][tokens_pad] bf16, out [B*tokens][H*64] bf16.
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int batch, int heads, int tokens,
                     int tokens_pad, bool f16, hipStream_t s);

}  // namespace hipts
