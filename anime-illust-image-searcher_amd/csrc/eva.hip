// eva.hip -- EVA02 tagger forward (the model tagging.py:45 really loads: wd-eva02-large-tagger-v3 =
// timm eva02_large_patch14_448) behind hipts_eva_*.
//
// Replaces `model.forward(batched_tensor)` + `F.sigmoid` (tagging.py:174-176) for that model family:
//   patch-embed conv 14x14 s14 (+bias) -> [cls | patches] + pos_embed -> depth x {
//       x += proj(softmax(rope(q) rope(k)^T / 8) v)     q/k/v = Linear(LN(x)) (q, v biased, k not), 2-D axial
//                                                        RoPE on the patch tokens, head_dim 64
//       x += fc2(LN(silu(fc1_g(LN(x))) * fc1_x(LN(x))))  SwiGLU with inner LayerNorm }
//   -> mean over the patch tokens -> fc_norm -> head (+ sigmoid).            (oracle/eva.py restates the same graph)
//
// Built from the pieces of the ViT-B/16 path: every Linear is the persistent MFMA GEMM of gemm.hip (q|k and v^T
// epilogues write the attention layouts, +bias+residual epilogues do the fp32 read-modify-write, the
// bias+SiLU / bias-only epilogues feed the SwiGLU), attention is attn.hip<64> with the 1025 real tokens masked
// inside a 1088-token padded layout.  An image owns TS = 1032 rows of the token matrices (1025 rounded up to 8):
// the 7 spare rows are zero-initialised, stay finite and are never read as keys (masked) or pooled.
// First version of this model: RoPE and SwiGLU-product + LayerNorm are separate HBM-bound kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "vit_internal.h"

using namespace hipts;

namespace {

struct EvaLayer {
    DevBuf ln1_g, ln1_b, ln2_g, ln2_b;
    DevBuf qkv_w, qkv_b, proj_w, proj_b, gx_w, gx_b;     // gx: fc1_g | fc1_x rows interleaved per 32 hidden units
    // fc2 with the inner LayerNorm (mlp.norm) folded in: fc2(LN(p)) = rstd (W (gamma p)) - rstd mean u + c with
    // u[n] = sum_k W[n][k] gamma[k] (of the rounded operand values of W), c = W beta + b; the fc1 epilogue stores gamma p.
    // Built when all four tensors are set; the host copies stay so that any of them can be set again.
    DevBuf fc2_w, fc2_u, fc2_c, mn_g;
    // pre-LayerNorms folded into the GEMM epilogues (EPI_RESID_XG): W gamma and W beta + b of the GEMMs that consume norm1 / norm2
    DevBuf qkv_u, qkv_c, gx_u, gx_c;
    std::vector<float> h_fc2_w, h_fc2_b, h_mn_g, h_mn_b;
};

}  // namespace

struct hipts_eva {
    int device = 0;
    hipts_eva_config_t cfg{};
    int grid = 0, np = 0, T = 0, TS = 0, Tp = 0, PK = 0, HN = 0, HK = 0;
    std::vector<EvaLayer> layers;
    DevBuf patch_w, patch_b, cls, pos, fcn_g, fcn_b, head_w, head_b, rope;
    std::vector<std::string> missing;
    DevBuf img_in, a0, tmp, x, xn, q, k, v, att, g1, stat_part, rowstat, xstat, pool_part, pooled2, logits, probs;
    DevBuf x0, x_rm;               // blocked residual stream (GemmArgs::x_blocked): the assembled rows row-major for layer 0's LayerNorm, the last layer's rows row-major for the pool
    bool fold_ln = false, fold_dirty = true;      // as in the ViT forward (vit.hip)
    bool split_att = false;                       // cfg.operand_f16 bit 4 (HIPTS_OPERAND_SPLIT_ATT), as in the ViT forward
    DevBuf sk_ws;                                 // split-K workspaces of proj / fc2 (GemmArgs::sk_ws), one per sub-batch stream, zeroed once
    hipStream_t sub[2] = {};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {};
};

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Patch matrix, hi | lo halves of the normalised pixel (K = 2 PK, PK = P*P*3 rounded up to 64, pad columns stay
// zero).  Column (ky*P + kx)*3 + c holds memory channel c (RGB); the BGR flip of tagging.py:243 lives in the
// weight permutation.  U8: ToTensor (/255) and Normalize ((x - .5) / .5) in float32 like the reference.
template <bool U8, bool F16>
__global__ __launch_bounds__(256) void eva_patchify_kernel(const void* __restrict__ img, bf16_t* __restrict__ a0, int batch, int S, int P,
                                                           int grid, int PK) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)batch * grid * grid * P;
    if (idx >= total) return;
    const int ky = (int)(idx % P);
    const int64_t tok = idx / P;
    const int px = (int)(tok % grid), py = (int)((tok / grid) % grid);
    const int64_t b = tok / ((int64_t)grid * grid);
    bf16_t* dst = a0 + tok * (int64_t)(2 * PK) + ky * P * 3;
    const int iy = py * P + ky;
    for (int kx = 0; kx < P; ++kx) {
        const int ix = px * P + kx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v;
            if constexpr (U8) {
                const float u = (float)reinterpret_cast<const uint8_t*>(img)[((b * S + iy) * S + ix) * 3 + c];
                v = (u / 255.0f - 0.5f) / 0.5f;
            } else {
                v = reinterpret_cast<const float*>(img)[((b * 3 + (2 - c)) * S + iy) * (int64_t)S + ix];
            }
            const bf16_t hi = to_op<F16>(v);
            dst[kx * 3 + c] = hi;
            dst[PK + kx * 3 + c] = to_op<F16>(v - from_op<F16>(hi));
        }
    }
}

// x[b*TS + 0] = cls + pos[0]; x[b*TS + 1 + t] = tmp[b*np + t] (conv + bias + pos[1 + t], from the GEMM epilogue);
// x[b*TS + T .. TS) = 0.   One thread per float4.
// xblk (optional): the same rows a second time in the residual stream's blocked layout, [m / 16][n / 16][m % 16][n % 16] (gemm_epi.h::x_off)
__global__ __launch_bounds__(256) void eva_assemble_kernel(const float* __restrict__ tmp, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x, int batch, int np, int TS,
                                                           int D, float* __restrict__ xblk) {
    const int dq = D >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)batch * TS * dq;
    if (idx >= total) return;
    const int c4 = (int)(idx % dq);
    const int r = (int)((idx / dq) % TS);
    const int64_t b = idx / ((int64_t)dq * TS);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r == 0) {
        const float4 a = reinterpret_cast<const float4*>(cls)[c4], p = reinterpret_cast<const float4*>(pos)[c4];
        v = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    } else if (r <= np) {
        v = reinterpret_cast<const float4*>(tmp)[(b * np + (r - 1)) * dq + c4];
    }
    reinterpret_cast<float4*>(x)[idx] = v;
    if (xblk) {
        const int64_t m = idx / dq;
        *reinterpret_cast<float4*>(xblk + ((((m >> 4) * (int64_t)(D >> 4) + (c4 >> 2)) << 8) + (m & 15) * 16 + (c4 & 3) * 4)) = v;
    }
}

// part[b][split][:] = sum of the patch-token rows of split `split` of image b (grid (POOL_SPLITS, batch)): the whole chip
// reads the 4 MB an image's tokens occupy instead of one workgroup per image (312 -> ~20 us at batch 32).
constexpr int POOL_SPLITS = 16;
__global__ __launch_bounds__(256) void eva_colsum_kernel(const float* __restrict__ x, float* __restrict__ part, int np, int TS, int D) {
    const int split = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int per = (np + POOL_SPLITS - 1) / POOL_SPLITS;
    const int r0 = split * per, r1 = min(np, r0 + per);
    if (tid * 4 >= D) return;
    const float4* xb = reinterpret_cast<const float4*>(x + ((int64_t)b * TS + 1) * D) + tid;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int dq = D >> 2;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        const float4 v = xb[(int64_t)r * dq];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(part + ((int64_t)b * POOL_SPLITS + split) * D)[tid] = acc;
}

// pooled2[b] = hi | lo of fc_norm(mean over the patch tokens of x[b]) from the POOL_SPLITS partial row sums.  One
// 1024-thread workgroup per image; x = part, np = POOL_SPLITS rows, TS = POOL_SPLITS, first row 0, count = patch tokens.
template <bool F16>
__global__ __launch_bounds__(1024) void eva_pool_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta,
                                                        bf16_t* __restrict__ out, int np, int TS, int D, float eps, int count) {
    __shared__ float part[4][1024];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x & 255, rg = threadIdx.x >> 8;
    const float* xb = x + (int64_t)b * TS * D;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = rg; r < np; r += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;
            if (c < D) m[u] += xb[(int64_t)r * D + c];
        }
#pragma unroll
    for (int u = 0; u < 4; ++u) part[rg][tid + 256 * u] = m[u];
    __syncthreads();
    float s = 0.f;
    if (rg == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;
            m[u] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / (float)count;
            if (c < D) s += m[u];
        }
        s = wsum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
    }
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)D;
    __syncthreads();
    if (rg == 0) {
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + 256 * u < D) ss += (m[u] - mean) * (m[u] - mean);
        ss = wsum(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
    }
    __syncthreads();
    if (rg != 0) return;
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + 256 * u;
        if (c < D) {
            const float f = (m[u] - mean) * rstd * g[c] + bta[c];
            const bf16_t hi = to_op<F16>(f);
            out[(int64_t)b * 2 * D + c] = hi;
            out[(int64_t)b * 2 * D + D + c] = to_op<F16>(f - from_op<F16>(hi));
        }
    }
}

int up_f32(DevBuf& buf, const float* data, size_t n) {
    HIPTS_TRY(buf.alloc(n * 4));
    return upload(buf.p, data, n * 4);
}

// rows x cols float block -> 16-bit operand bits at row offset `row0` of a [*, ld] matrix that was allocated zeroed
int put_rows16(DevBuf& buf, const float* data, int rows, int cols, int row0, int ld, bool f16) {
    std::vector<uint16_t> h((size_t)rows * ld, 0);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) h[(size_t)r * ld + c] = f16 ? f32_to_f16_rne(data[(size_t)r * cols + c]) : f32_to_bf16_rne(data[(size_t)r * cols + c]);
    return upload(buf.as<uint16_t>() + (size_t)row0 * ld, h.data(), h.size() * 2);
}

int alloc_zero(DevBuf& buf, size_t bytes) {
    HIPTS_TRY(buf.alloc(bytes));
    HIPTS_HIP(hipMemset(buf.p, 0, bytes));
    return HIPTS_OK;
}

// The whole kernel sequence for images [i0, i0 + batch) on stream s (every buffer is indexed by image)
int eva_run_images(hipts_eva* h, const void* in_dev, bool is_u8, int i0, int batch, float* lg, float* pr, hipStream_t s, bool shared_chip,
                   int sub = 0) {
    const auto& c = h->cfg;
    const int S = c.image_size, D = c.dim, P = c.patch, H = c.heads, T = h->T, TS = h->TS, Tp = h->Tp, np = h->np;
    const bool f16 = (c.operand_f16 & 1) != 0;
    const int M = batch * TS;
    in_dev = (const char*)in_dev + (size_t)i0 * S * S * 3 * (is_u8 ? 1 : 4);
    const size_t r0 = (size_t)i0 * TS, qo = (size_t)i0 * H * Tp * 64;
    bf16_t* a0_p = h->a0.as<bf16_t>() + (size_t)i0 * np * 2 * h->PK;
    float* tmp_p = h->tmp.as<float>() + (size_t)i0 * np * D;
    // the fp32 residual stream as 16 x 16 blocks (round 5; csrc/gemm_epi.h::x_off, the ViT forward has the measurements): a sub-batch's
    // region starts at a multiple of 16 rows of its own (1032 rows per image: odd image offsets are not)
    static const bool xb_env = !(getenv("HIPTS_X_BLOCKED") && atoi(getenv("HIPTS_X_BLOCKED")) == 0);
    const bool xb = xb_env && h->fold_ln && D % 16 == 0;
    float* x = h->x.as<float>() + (xb ? ((r0 + 15) / 16 * 16 + (size_t)16 * sub) : r0) * D;
    float* x0 = xb ? h->x0.as<float>() + r0 * D : x;                // what eva_assemble_kernel writes row-major (layer 0's LayerNorm reads it)
    float* x_rm = h->x_rm.as<float>() + r0 * D;                     // the last layer's rows for the column sums
    bf16_t* xn = h->xn.as<bf16_t>() + r0 * D;
    bf16_t* q_p = h->q.as<bf16_t>() + qo;
    bf16_t* k_p = h->k.as<bf16_t>() + qo;
    bf16_t* v_p = h->v.as<bf16_t>() + qo;
    const int att_k = h->split_att ? 2 * D : D;
    bf16_t* att_p = h->att.as<bf16_t>() + r0 * att_k;
    bf16_t* g1_p = h->g1.as<bf16_t>() + r0 * h->HK;
    const int sblocks = (2 * h->HK + 255) / 256;                               // 256-column tiles of the fc1 launch: one partial (sum, sum of squares) pair per tile and row
    float* stat_p = h->stat_part.as<float>() + 2 * (size_t)sblocks * r0;      // [sblocks][M] float2, this sub-batch's region
    float* rowstat_p = h->rowstat.as<float>() + 2 * r0;
    bf16_t* pooled2_p = h->pooled2.as<bf16_t>() + (size_t)i0 * 2 * D;
    if (lg) lg += (size_t)i0 * c.num_classes;
    if (pr) pr += (size_t)i0 * c.num_classes;
    GemmArgs g;
    {
        const int64_t total = (int64_t)batch * np * P;
        const int blocks = ceil_div(total, 256);
        if (is_u8) {
            if (f16) eva_patchify_kernel<true, true><<<blocks, 256, 0, s>>>(in_dev, a0_p, batch, S, P, h->grid, h->PK);
            else eva_patchify_kernel<true, false><<<blocks, 256, 0, s>>>(in_dev, a0_p, batch, S, P, h->grid, h->PK);
        } else {
            if (f16) eva_patchify_kernel<false, true><<<blocks, 256, 0, s>>>(in_dev, a0_p, batch, S, P, h->grid, h->PK);
            else eva_patchify_kernel<false, false><<<blocks, 256, 0, s>>>(in_dev, a0_p, batch, S, P, h->grid, h->PK);
        }
        HIPTS_LAUNCH_CHECK();
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = a0_p; g.W = h->patch_w.as<bf16_t>(); g.M = batch * np; g.N = D; g.K = 2 * h->PK;
        g.bias = h->patch_b.as<float>(); g.out_f32 = tmp_p; g.pos = h->pos.as<float>() + D; g.tokens = np; g.qscale = 1.0f;
        HIPTS_TRY(launch_gemm(EPI_PATCH, g, s));
        const int64_t tot4 = (int64_t)batch * TS * (D / 4);
        eva_assemble_kernel<<<ceil_div(tot4, 256), 256, 0, s>>>(tmp_p, h->cls.as<float>(), h->pos.as<float>(), x0, batch, np, TS, D, xb ? x : nullptr);
        HIPTS_LAUNCH_CHECK();
    }
    // pre-LayerNorms folded into the neighbouring GEMM epilogues, as in the ViT forward: the residual GEMM writes gamma * x (16-bit)
    // and the per-tile row sums of x; the consumer finishes the statistics per tile and applies rstd / mean / beta
    const bool fold = h->fold_ln;
    const int xblocks = (D + 255) / 256;
    float* xstat_p = fold ? h->xstat.as<float>() + 2 * (size_t)xblocks * r0 : nullptr;
    auto folded = [&](GemmArgs& ga, const float* u, const float* cvec) {
        ga.stat_in = xstat_p; ga.stat_in_blocks = xblocks; ga.stat_in_stride = M; ga.ln_dim = D; ga.ln_eps = c.ln_eps;
        ga.col_u = u; ga.bias = cvec;
    };
    for (int li = 0; li < c.depth; ++li) {
        EvaLayer& L = h->layers[li];
        const bool ln1_folded = fold && li > 0;
        if (!ln1_folded) HIPTS_TRY(launch_layernorm(li == 0 ? x0 : x, L.ln1_g.as<float>(), L.ln1_b.as<float>(), xn, M, D, c.ln_eps, f16, s));
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = xn; g.W = L.qkv_w.as<bf16_t>(); g.M = M; g.N = 3 * D; g.K = D; g.bias = L.qkv_b.as<float>();
        g.out_bf16 = q_p; g.out2_bf16 = k_p; g.out3_bf16 = v_p;       // one launch: q and k rotated, v as it is, all [image][head][token][64]
        g.tokens = TS; g.tokens_pad = Tp; g.heads = H; g.dim = D;
        g.qscale = 0.125f * 1.4426950408889634f;           // 64^-0.5 * log2(e); linear, so it commutes with the rotation
        g.rope = h->rope.as<float>(); g.rope_tokens = np;  // 2-D rotary embedding on the fp32 result, in the epilogue
        if (ln1_folded) folded(g, L.qkv_u.as<float>(), L.qkv_c.as<float>());
        HIPTS_TRY(launch_gemm(EPI_QK_ROPE, g, s));
        HIPTS_TRY(launch_attention2(q_p, k_p, v_p, att_p, batch, H, T, Tp, f16, s, TS, 0, h->split_att ? 1 : 0, split_lo_scale(f16)));
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = att_p; g.W = L.proj_w.as<bf16_t>(); g.M = M; g.N = D; g.K = att_k; g.bias = L.proj_b.as<float>(); g.out_f32 = x;
        g.x_blocked = xb ? 1 : 0;
        g.sk_ws = h->sk_ws.as<char>() + (size_t)sub * GEMM_SK_WS_BYTES; g.sk_ws_bytes = GEMM_SK_WS_BYTES;      // split-K when the launch under-fills the chip
        if (fold) {
            g.out_bf16 = xn; g.ln_gamma = L.ln2_g.as<float>(); g.stat_part = xstat_p; g.stat_stride = M;
            HIPTS_TRY(launch_gemm(EPI_RESID_XG, g, s));
        } else {
            HIPTS_TRY(launch_gemm(EPI_RESID, g, s));
            HIPTS_TRY(launch_layernorm(x, L.ln2_g.as<float>(), L.ln2_b.as<float>(), xn, M, D, c.ln_eps, f16, s));
        }
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        // fc1_g and fc1_x in one launch (rows interleaved per 32 hidden units); silu(gate) * value in the epilogue
        g.A = xn; g.W = L.gx_w.as<bf16_t>(); g.M = M; g.N = 2 * h->HK; g.K = D; g.bias = L.gx_b.as<float>();
        g.out_bf16 = g1_p; g.ld_out = h->HK;
        g.stat_part = stat_p; g.stat_stride = M;             // row sums of the product for the LayerNorm folded into fc2
        g.ln_gamma = L.mn_g.as<float>();
        if (fold) folded(g, L.gx_u.as<float>(), L.gx_c.as<float>());
        HIPTS_TRY(launch_gemm(EPI_SWIGLU, g, s));
        // (round 3) no rowstat kernel between the two GEMMs: the SwiGLU epilogue leaves ONE partial pair per (256-column tile, row) -- the four
        // waves of a row meet in LDS -- and fc2's workgroups finish the 22 pairs of their tile's rows into an LDS table before their epilogue
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = g1_p; g.W = L.fc2_w.as<bf16_t>(); g.M = M; g.N = D; g.K = h->HK; g.bias = L.fc2_c.as<float>(); g.out_f32 = x;
        g.x_blocked = xb ? 1 : 0;
        g.sk_ws = h->sk_ws.as<char>() + (size_t)sub * GEMM_SK_WS_BYTES; g.sk_ws_bytes = GEMM_SK_WS_BYTES;
        static const bool rowstat_kernel = getenv("HIPTS_EVA_ROWSTAT_KERNEL") && atoi(getenv("HIPTS_EVA_ROWSTAT_KERNEL"));      // A/B: finish the pairs in a kernel of its own
        if (rowstat_kernel) {
            HIPTS_TRY(launch_rowstat(stat_p, rowstat_p, M, M, sblocks, c.mlp_hidden, c.ln_eps, s));
            g.rowstat = rowstat_p;
        } else {
            g.stat_in = stat_p; g.stat_in_blocks = sblocks; g.stat_in_stride = M; g.ln_dim = c.mlp_hidden; g.ln_eps = c.ln_eps;
        }
        g.col_u = L.fc2_u.as<float>();
        if (fold && li + 1 < c.depth) {      // the next layer's norm1 prepared here
            g.out_bf16 = xn; g.ln_gamma = h->layers[li + 1].ln1_g.as<float>(); g.stat_part = xstat_p; g.stat_stride = M;
            HIPTS_TRY(launch_gemm(EPI_RESID_XGI, g, s));
        } else {
            if (xb) g.resid_rowmajor_out = x_rm;        // the forward's last residual launch (fold: the last layer): rows for the column sums
            HIPTS_TRY(launch_gemm(EPI_RESID_ROWSTAT, g, s));
        }
    }
    float* part_p = h->pool_part.as<float>() + (size_t)i0 * POOL_SPLITS * D;
    eva_colsum_kernel<<<dim3(POOL_SPLITS, batch), 256, 0, s>>>(xb ? x_rm : x, part_p, np, TS, D);
    if (f16) eva_pool_kernel<true><<<batch, 1024, 0, s>>>(part_p, h->fcn_g.as<float>(), h->fcn_b.as<float>(), pooled2_p, POOL_SPLITS, POOL_SPLITS, D, c.ln_eps, np);
    else eva_pool_kernel<false><<<batch, 1024, 0, s>>>(part_p, h->fcn_g.as<float>(), h->fcn_b.as<float>(), pooled2_p, POOL_SPLITS, POOL_SPLITS, D, c.ln_eps, np);
    HIPTS_LAUNCH_CHECK();
    g = GemmArgs{};
    g.f16 = f16;
    g.shared_chip = shared_chip;
    g.A = pooled2_p; g.W = h->head_w.as<bf16_t>(); g.M = batch; g.N = c.num_classes; g.K = 2 * D;
    g.sk_ws = h->sk_ws.as<char>() + (size_t)sub * GEMM_SK_WS_BYTES; g.sk_ws_bytes = GEMM_SK_WS_BYTES;      // split-K: 43 tiles on 256 CUs
    g.bias = h->head_b.as<float>(); g.out_f32 = lg; g.out2_f32 = pr;
    HIPTS_TRY(launch_gemm(EPI_HEAD, g, s));
    return HIPTS_OK;
}

int eva_forward_impl(hipts_eva* h, const void* input, int in_memspace, bool is_u8, int batch, float* logits_out, float* probs_out,
                     int out_memspace, hipStream_t s) {
    HIPTS_REQUIRE(h && input && batch >= 1, "hipts_eva_forward: bad arguments");
    HIPTS_REQUIRE(batch <= h->cfg.max_batch, "batch %d exceeds max_batch %d", batch, h->cfg.max_batch);
    if (!h->missing.empty())
        return set_error(HIPTS_ERR_STATE, "hipts_eva_forward: %zu checkpoint tensors not set (first: %s)", h->missing.size(),
                         h->missing[0].c_str());
    HIPTS_TRY(use_device(h->device));
    const auto& c = h->cfg;
    const int S = c.image_size;
    const void* in_dev = input;
    if (in_memspace != HIPTS_DEVICE) {
        const size_t bytes = (size_t)batch * S * S * 3 * (is_u8 ? 1 : 4);
        HIPTS_TRY(h->img_in.reserve(bytes));
        HIPTS_HIP(hipMemcpyAsync(h->img_in.p, input, bytes, hipMemcpyHostToDevice, s));
        in_dev = h->img_in.p;
    }
    const bool dev_out = out_memspace == HIPTS_DEVICE;
    float* lg = (dev_out && logits_out) ? logits_out : h->logits.as<float>();
    float* pr = (probs_out || !dev_out) ? ((dev_out && probs_out) ? probs_out : h->probs.as<float>()) : nullptr;
    if (h->fold_ln && h->fold_dirty) {
        const bool f16w = (c.operand_f16 & 1) != 0;
        for (auto& L : h->layers) {
            HIPTS_TRY(launch_fold_ln(L.qkv_w.as<bf16_t>(), f16w, L.ln1_g.as<float>(), L.ln1_b.as<float>(), L.qkv_b.as<float>(), L.qkv_u.as<float>(),
                                     L.qkv_c.as<float>(), 3 * c.dim, c.dim, s));
            HIPTS_TRY(launch_fold_ln(L.gx_w.as<bf16_t>(), f16w, L.ln2_g.as<float>(), L.ln2_b.as<float>(), L.gx_b.as<float>(), L.gx_u.as<float>(),
                                     L.gx_c.as<float>(), 2 * h->HK, c.dim, s));
        }
        h->fold_dirty = false;
    }
    static const int want_streams = getenv("HIPTS_EVA_STREAMS") ? atoi(getenv("HIPTS_EVA_STREAMS")) : 2;
    static const int min_sub = getenv("HIPTS_EVA_MINSUB") ? atoi(getenv("HIPTS_EVA_MINSUB")) : 5;      // images per sub-batch needed to split (the reference batch of 10 as two halves: 980 -> 1005 images/s)
    const int ns = std::min({want_streams, 2, batch / (min_sub > 0 ? min_sub : 5)});
    if (ns >= 2) {
        // two sub-batches on two internal streams, as in the ViT forward (partial last rounds of the persistent GEMMs)
        if (!h->ev_fork) {
            HIPTS_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            for (int i = 0; i < 2; ++i) {
                HIPTS_HIP(hipStreamCreateWithFlags(&h->sub[i], hipStreamNonBlocking));
                HIPTS_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
            }
        }
        HIPTS_HIP(hipEventRecord(h->ev_fork, s));
        const int nb0 = (batch + 1) / 2;
        for (int i = 0; i < 2; ++i) {
            HIPTS_HIP(hipStreamWaitEvent(h->sub[i], h->ev_fork, 0));
            HIPTS_TRY(eva_run_images(h, in_dev, is_u8, i ? nb0 : 0, i ? batch - nb0 : nb0, lg, pr, h->sub[i], true, i));
            HIPTS_HIP(hipEventRecord(h->ev_join[i], h->sub[i]));
            HIPTS_HIP(hipStreamWaitEvent(s, h->ev_join[i], 0));
        }
    } else {
        HIPTS_TRY(eva_run_images(h, in_dev, is_u8, 0, batch, lg, pr, s, false));
    }
    if (!dev_out) {
        const size_t bytes = (size_t)batch * c.num_classes * 4;
        if (logits_out) HIPTS_HIP(hipMemcpyAsync(logits_out, lg, bytes, hipMemcpyDeviceToHost, s));
        if (probs_out) HIPTS_HIP(hipMemcpyAsync(probs_out, pr, bytes, hipMemcpyDeviceToHost, s));
        HIPTS_HIP(hipStreamSynchronize(s));
    }
    return HIPTS_OK;
}

}  // namespace

extern "C" {

int hipts_eva_create(const hipts_eva_config_t* cfg, int device, hipts_eva_t** out) {
    HIPTS_REQUIRE(cfg && out, "hipts_eva_create: null argument");
    HIPTS_REQUIRE(cfg->patch >= 1 && cfg->image_size % cfg->patch == 0, "image_size %d must be a multiple of patch %d", cfg->image_size, cfg->patch);
    HIPTS_REQUIRE(cfg->dim % 64 == 0 && cfg->dim <= 1024 && cfg->heads * 64 == cfg->dim, "dim %d / heads %d: head_dim must be 64, dim <= 1024", cfg->dim, cfg->heads);
    HIPTS_REQUIRE(cfg->mlp_hidden >= 8 && cfg->mlp_hidden <= 4096, "mlp_hidden %d must be 8 .. 4096", cfg->mlp_hidden);
    HIPTS_REQUIRE(cfg->depth >= 1 && cfg->num_classes >= 1 && cfg->max_batch >= 1 && cfg->rope_ref_grid >= 1, "bad configuration");
    HIPTS_TRY(use_device(device));
    auto* h = new hipts_eva();
    h->device = device;
    h->cfg = *cfg;
    h->split_att = (cfg->operand_f16 & HIPTS_OPERAND_SPLIT_ATT) != 0;
    const int D = cfg->dim, B = cfg->max_batch;
    h->grid = cfg->image_size / cfg->patch;
    h->np = h->grid * h->grid;
    h->T = h->np + 1;
    h->TS = round_up(h->T, 8);
    h->Tp = round_up(h->T, 64);
    h->PK = round_up(cfg->patch * cfg->patch * 3, 64);
    h->HN = round_up(cfg->mlp_hidden, 16);
    h->HK = round_up(cfg->mlp_hidden, 64);
    h->fold_ln = cfg->dim % 64 == 0 && !getenv("HIPTS_GEMM") && !(getenv("HIPTS_LN_FOLD") && atoi(getenv("HIPTS_LN_FOLD")) == 0);
    h->layers.resize(cfg->depth);
    const size_t M = (size_t)B * h->TS;
    const size_t qk = (size_t)B * cfg->heads * h->Tp * 64 * 2;
    int st = 0;
    if ((st = alloc_zero(h->a0, (size_t)B * h->np * 2 * h->PK * 2)) || (st = h->tmp.alloc((size_t)B * h->np * D * 4)) || (st = alloc_zero(h->x, (M + 16 * 9) * D * 4)) || (st = h->x0.alloc(M * D * 4)) || (st = h->x_rm.alloc(M * D * 4)) ||
        (st = alloc_zero(h->xn, M * D * 2)) || (st = alloc_zero(h->q, qk)) || (st = alloc_zero(h->k, qk)) || (st = alloc_zero(h->v, qk)) ||
        (st = alloc_zero(h->att, M * D * 2 * (h->split_att ? 2 : 1))) || (st = alloc_zero(h->g1, M * h->HK * 2)) ||
        (st = h->stat_part.alloc((size_t)((2 * h->HK + 255) / 256) * M * 8)) || (st = h->xstat.alloc((size_t)((D + 255) / 256) * M * 8)) || (st = h->rowstat.alloc(M * 8)) || (st = h->pooled2.alloc((size_t)B * 2 * D * 2)) || (st = h->pool_part.alloc((size_t)B * 16 * D * 4)) ||
        (st = h->logits.alloc((size_t)B * cfg->num_classes * 4)) || (st = h->probs.alloc((size_t)B * cfg->num_classes * 4)) ||
        (st = alloc_zero(h->sk_ws, 2 * GEMM_SK_WS_BYTES))) {      // split-K tickets start at zero (the kernels leave them so)
        delete h;
        return st;
    }
    // RoPE tables (timm RotaryEmbeddingCat, in_pixels = False, positions rescaled to the reference grid)
    {
        const int nb = 16;      // head_dim / 4
        std::vector<float> sn((size_t)h->np * 64), cs((size_t)h->np * 64), tab((size_t)h->np * 64);
        for (int y = 0; y < h->grid; ++y)
            for (int xx = 0; xx < h->grid; ++xx)
                for (int ax = 0; ax < 2; ++ax)
                    for (int i = 0; i < nb; ++i) {
                        const float band = 1.0f / powf(10000.0f, (float)i / (float)nb);
                        const float t = (float)(ax == 0 ? y : xx) / (float)h->grid * (float)cfg->rope_ref_grid;
                        const float a = t * band;
                        const size_t o = ((size_t)y * h->grid + xx) * 64 + (size_t)(ax * nb + i) * 2;
                        sn[o] = sn[o + 1] = sinf(a);
                        cs[o] = cs[o + 1] = cosf(a);
                    }
        for (size_t t = 0; t < (size_t)h->np; ++t)
            for (int i = 0; i < 32; ++i) {          // (sin, cos) of column pair (2i, 2i+1)
                tab[t * 64 + 2 * i] = sn[t * 64 + 2 * i];
                tab[t * 64 + 2 * i + 1] = cs[t * 64 + 2 * i];
            }
        if ((st = up_f32(h->rope, tab.data(), tab.size()))) {
            delete h;
            return st;
        }
    }
    for (auto& L : h->layers) {
        // assembled from three tensors each: allocate zeroed now (k has no bias; pad rows / columns stay zero)
        if ((st = alloc_zero(L.qkv_w, (size_t)(round_up(2 * D, 256) + round_up(D, 256) + 256) * D * 2)) || (st = alloc_zero(L.qkv_b, (size_t)3 * D * 4)) ||
            (st = alloc_zero(L.gx_w, (size_t)round_up(2 * h->HK, 256) * D * 2)) || (st = alloc_zero(L.gx_b, (size_t)round_up(2 * h->HK, 256) * 4)) ||
            (st = alloc_zero(L.fc2_w, (size_t)round_up(D, 256) * h->HK * 2)) || (st = alloc_zero(L.fc2_u, (size_t)round_up(D, 256) * 4)) ||
            (st = alloc_zero(L.fc2_c, (size_t)round_up(D, 256) * 4)) ||
            (st = alloc_zero(L.mn_g, (size_t)h->HK * 4)) || (st = alloc_zero(L.qkv_u, (size_t)3 * D * 4)) || (st = alloc_zero(L.qkv_c, (size_t)3 * D * 4)) ||
            (st = alloc_zero(L.gx_u, (size_t)2 * h->HK * 4)) || (st = alloc_zero(L.gx_c, (size_t)2 * h->HK * 4))) {
            delete h;
            return st;
        }
    }
    if ((st = alloc_zero(h->patch_w, (size_t)round_up(D, 256) * 2 * h->PK * 2))) {
        delete h;
        return st;
    }
    auto need = [&](const std::string& k) { h->missing.push_back(k); };
    for (const char* k : {"patch_embed.proj.weight", "patch_embed.proj.bias", "cls_token", "pos_embed", "fc_norm.weight", "fc_norm.bias", "head.weight",
                          "head.bias"})
        need(k);
    for (int i = 0; i < cfg->depth; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        for (const char* k : {"norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.q_proj.weight", "attn.q_proj.bias", "attn.k_proj.weight",
                              "attn.v_proj.weight", "attn.v_proj.bias", "attn.proj.weight", "attn.proj.bias", "mlp.fc1_g.weight", "mlp.fc1_g.bias",
                              "mlp.fc1_x.weight", "mlp.fc1_x.bias", "mlp.norm.weight", "mlp.norm.bias", "mlp.fc2.weight", "mlp.fc2.bias"})
            need(p + k);
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_eva_destroy(hipts_eva_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < 2; ++i) {
            if (h->sub[i]) (void)hipStreamDestroy(h->sub[i]);
            if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
        }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_eva_set_tensor(hipts_eva_t* h, const char* key_c, const float* data, int64_t numel) {
    HIPTS_REQUIRE(h && key_c && data, "hipts_eva_set_tensor: null argument");
    HIPTS_TRY(use_device(h->device));
    const std::string key(key_c);
    const auto& c = h->cfg;
    const bool f16 = (c.operand_f16 & 1) != 0;
    const int D = c.dim, P = c.patch, Hd = c.mlp_hidden, C = c.num_classes;
    int st = HIPTS_OK;
#define EXPECT(n)                                                                                                         \
    do {                                                                                                                  \
        if (numel != (int64_t)(n)) return set_error(HIPTS_ERR_INVALID, "tensor %s: %lld elements, expected %lld", key_c, (long long)numel, (long long)(n)); \
    } while (0)
    if (key == "patch_embed.proj.weight") {
        EXPECT((int64_t)D * 3 * P * P);
        // [n][c_model][ky][kx] -> [n][(ky*P + kx)*3 + c_mem], c_model = 2 - c_mem (BGR flip), duplicated for hi | lo
        const int K1 = P * P * 3;
        std::vector<float> w2((size_t)D * 2 * h->PK, 0.f);
        for (int n = 0; n < D; ++n)
            for (int cm = 0; cm < 3; ++cm)
                for (int t = 0; t < P * P; ++t) {
                    const float v = data[((size_t)n * 3 + (2 - cm)) * P * P + t];
                    w2[(size_t)n * 2 * h->PK + t * 3 + cm] = v;
                    w2[(size_t)n * 2 * h->PK + h->PK + t * 3 + cm] = v;
                }
        (void)K1;
        st = put_rows16(h->patch_w, w2.data(), D, 2 * h->PK, 0, 2 * h->PK, f16);
    } else if (key == "patch_embed.proj.bias") { EXPECT(D); st = up_f32(h->patch_b, data, D); }
    else if (key == "cls_token") { EXPECT(D); st = up_f32(h->cls, data, D); }
    else if (key == "pos_embed") { EXPECT((int64_t)h->T * D); st = up_f32(h->pos, data, (size_t)h->T * D); }
    else if (key == "fc_norm.weight") { EXPECT(D); st = up_f32(h->fcn_g, data, D); }
    else if (key == "fc_norm.bias") { EXPECT(D); st = up_f32(h->fcn_b, data, D); }
    else if (key == "head.bias") { EXPECT(C); st = up_f32(h->head_b, data, C); }
    else if (key == "head.weight") {
        EXPECT((int64_t)C * D);
        std::vector<float> dup((size_t)C * 2 * D);
        for (int n = 0; n < C; ++n)
            for (int k2 = 0; k2 < D; ++k2) dup[(size_t)n * 2 * D + k2] = dup[(size_t)n * 2 * D + D + k2] = data[(size_t)n * D + k2];
        st = upload_matrix16(h->head_w, dup.data(), C, 2 * D, round_up(C, 256), f16);
    } else if (key.rfind("blocks.", 0) == 0) {
        const size_t d1 = key.find('.', 7);
        if (d1 == std::string::npos) return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
        const int li = atoi(key.substr(7, d1 - 7).c_str());
        if (li < 0 || li >= c.depth) return set_error(HIPTS_ERR_INVALID, "tensor %s: block out of range", key_c);
        EvaLayer& L = h->layers[li];
        const std::string t = key.substr(d1 + 1);
        if (t == "norm1.weight") { EXPECT(D); st = up_f32(L.ln1_g, data, D); }
        else if (t == "norm1.bias") { EXPECT(D); st = up_f32(L.ln1_b, data, D); }
        else if (t == "norm2.weight") { EXPECT(D); st = up_f32(L.ln2_g, data, D); }
        else if (t == "norm2.bias") { EXPECT(D); st = up_f32(L.ln2_b, data, D); }
        else if (t == "attn.q_proj.weight") { EXPECT((int64_t)D * D); st = put_rows16(L.qkv_w, data, D, D, 0, D, f16); }
        else if (t == "attn.k_proj.weight") { EXPECT((int64_t)D * D); st = put_rows16(L.qkv_w, data, D, D, D, D, f16); }
        else if (t == "attn.v_proj.weight") { EXPECT((int64_t)D * D); st = put_rows16(L.qkv_w, data, D, D, 2 * D, D, f16); }
        else if (t == "attn.q_proj.bias") { EXPECT(D); st = upload(L.qkv_b.as<float>(), data, (size_t)D * 4); }
        else if (t == "attn.v_proj.bias") { EXPECT(D); st = upload(L.qkv_b.as<float>() + 2 * D, data, (size_t)D * 4); }
        else if (t == "attn.proj.weight") {
            EXPECT((int64_t)D * D);
            if (h->split_att) {      // [W | W] against the (hi | lo) halves of the attention output
                std::vector<float> dup((size_t)D * 2 * D);
                const float inv = 1.0f / split_lo_scale(f16);      // the low halves arrive multiplied by the scale
                for (int n = 0; n < D; ++n) {
                    memcpy(&dup[(size_t)n * 2 * D], &data[(size_t)n * D], (size_t)D * 4);
                    for (int kk = 0; kk < D; ++kk) dup[(size_t)n * 2 * D + D + kk] = data[(size_t)n * D + kk] * inv;
                }
                st = upload_matrix16(L.proj_w, dup.data(), D, 2 * D, round_up(D, 256), f16);
            } else {
                st = upload_matrix16(L.proj_w, data, D, D, round_up(D, 256), f16);
            }
        }
        else if (t == "attn.proj.bias") { EXPECT(D); st = up_f32(L.proj_b, data, D); }
        else if (t == "mlp.fc1_g.weight" || t == "mlp.fc1_x.weight") {
            // hidden unit u -> physical row 64 (u / 32) + (u % 32), + 32 for the value half (EPI_SWIGLU)
            EXPECT((int64_t)Hd * D);
            const int half = t == "mlp.fc1_x.weight" ? 32 : 0;
            for (int u0 = 0; u0 < Hd && !st; u0 += 32)
                st = put_rows16(L.gx_w, data + (size_t)u0 * D, std::min(32, Hd - u0), D, (u0 / 32) * 64 + half, D, f16);
        }
        else if (t == "mlp.fc1_g.bias" || t == "mlp.fc1_x.bias") {
            EXPECT(Hd);
            const int half = t == "mlp.fc1_x.bias" ? 32 : 0;
            for (int u0 = 0; u0 < Hd && !st; u0 += 32)
                st = upload(L.gx_b.as<float>() + (u0 / 32) * 64 + half, data + u0, (size_t)std::min(32, Hd - u0) * 4);
        }
        else if (t == "mlp.norm.weight" || t == "mlp.norm.bias" || t == "mlp.fc2.weight" || t == "mlp.fc2.bias") {
            if (t == "mlp.norm.weight") { EXPECT(Hd); L.h_mn_g.assign(data, data + Hd); }
            else if (t == "mlp.norm.bias") { EXPECT(Hd); L.h_mn_b.assign(data, data + Hd); }
            else if (t == "mlp.fc2.weight") { EXPECT((int64_t)D * Hd); L.h_fc2_w.assign(data, data + (size_t)D * Hd); }
            else { EXPECT(D); L.h_fc2_b.assign(data, data + D); }
            if (!L.h_mn_g.empty() && !L.h_mn_b.empty() && !L.h_fc2_w.empty() && !L.h_fc2_b.empty()) {
                // u and c from the rounded operand values of W, so that the mean term cancels against what the MFMA sums
                std::vector<float> u(D), cc(D);
                for (int n = 0; n < D; ++n) {
                    double su = 0.0, sc = 0.0;
                    for (int k = 0; k < Hd; ++k) {
                        const float w = L.h_fc2_w[(size_t)n * Hd + k];
                        float r;
                        if (f16) r = f16_bits_to_f32(f32_to_f16_rne(w));
                        else { const uint32_t bits = (uint32_t)f32_to_bf16_rne(w) << 16; memcpy(&r, &bits, 4); }
                        su += (double)r * (double)L.h_mn_g[k];
                        sc += (double)r * (double)L.h_mn_b[k];
                    }
                    u[n] = (float)su;
                    cc[n] = (float)(sc + (double)L.h_fc2_b[n]);
                }
                st = put_rows16(L.fc2_w, L.h_fc2_w.data(), D, Hd, 0, h->HK, f16);
                if (!st) st = upload(L.mn_g.as<float>(), L.h_mn_g.data(), (size_t)Hd * 4);      // pad entries stay zero
                if (!st) st = upload(L.fc2_u.as<float>(), u.data(), (size_t)D * 4);
                if (!st) st = upload(L.fc2_c.as<float>(), cc.data(), (size_t)D * 4);
            }
        }
        else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
    } else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
#undef EXPECT
    if (st) return st;
    h->fold_dirty = true;
    auto it = std::find(h->missing.begin(), h->missing.end(), key);
    if (it != h->missing.end()) h->missing.erase(it);
    return HIPTS_OK;
}

int hipts_eva_forward_u8(hipts_eva_t* h, const uint8_t* images, int images_memspace, int batch, float* logits_out, float* probs_out,
                         int out_memspace, void* stream) {
    return eva_forward_impl(h, images, images_memspace, true, batch, logits_out, probs_out, out_memspace, (hipStream_t)stream);
}

int hipts_eva_forward_f32(hipts_eva_t* h, const float* x, int x_memspace, int batch, float* logits_out, float* probs_out, int out_memspace,
                          void* stream) {
    return eva_forward_impl(h, x, x_memspace, false, batch, logits_out, probs_out, out_memspace, (hipStream_t)stream);
}

int hipts_eva_flops_per_image(const hipts_eva_t* h, double* flops) {
    HIPTS_REQUIRE(h && flops, "null argument");
    const auto& c = h->cfg;
    const double T = h->T, D = c.dim, Hd = c.mlp_hidden;
    double f = 2.0 * h->np * D * (double)(c.patch * c.patch * 3);
    f += c.depth * (2.0 * T * 3 * D * D + 4.0 * T * T * D + 2.0 * T * D * D + 2.0 * T * D * Hd * 3);
    f += 2.0 * D * c.num_classes;
    *flops = f;
    return HIPTS_OK;
}

}  // extern "C"
