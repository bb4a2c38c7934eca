// vit_internal.h -- shared declarations of the ViT forward kernels (gemm.hip, attn.hip, vit.hip).
#pragma once
#include "common.h"

namespace hipts {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
    unsigned long long* stamps = nullptr;   // diagnostic build only (tools/gemm_bench.py): s_memtime stamps of block 0
};

int launch_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s);

// softmax(Q K^T) V for every (image, head): q,k [B*H][tokens_pad][64] bf16 (q pre-scaled by
// head_dim^-0.5), vT [B*H][64][tokens_pad] bf16, out [B*tokens][H*64] bf16.
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int batch, int heads, int tokens,
                     int tokens_pad, hipStream_t s);

}  // namespace hipts
