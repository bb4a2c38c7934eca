// vit.hip -- ViT tagger forward: handle, checkpoint upload, small kernels, orchestration.
//
// Replaces `model.forward(batched_tensor)` + `F.sigmoid` of tagging.py:174-176 for the contract
// model (ViT-B/16 @448, no class token, mean pool).  Kernel sequence per forward:
//   patchify (ToTensor + Normalize + BGR flip fused, tagging.py:241-243) -> GEMM(+bias+pos) ->
//   depth x [ LN -> GEMM(QK) + GEMM(V^T) -> attention -> GEMM(+bias+residual) ->
//             LN -> GEMM(+bias, GELU) -> GEMM(+bias+residual) ] ->
//   final LN + token mean (partial sums) -> finalize (hi/lo bf16 split) -> head GEMM (+sigmoid).
// The residual stream, LN statistics, softmax and every accumulation are float32; only MFMA
// operands are bf16.
#include <algorithm>
#include "../../include/hip_tagsearch_debug.h"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vit_internal.h"

using namespace hipts;

namespace {

struct Layer {
    DevBuf ln1_g, ln1_b, ln2_g, ln2_b;
    DevBuf qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b;
    // folded LayerNorms (EPI_RESID_XG): W gamma and W beta + b of the GEMMs that consume norm1 / norm2
    DevBuf qkv_u, qkv_c, fc1_u, fc1_c;
};


}  // namespace

enum ProfCat { PC_PATCHIFY = 0, PC_GEMM_PATCH, PC_LAYERNORM, PC_GEMM_QK, PC_GEMM_VT, PC_ATTENTION, PC_GEMM_RESID,
               PC_GEMM_GELU, PC_POOL, PC_GEMM_HEAD, PC_OTHER, PC_COUNT };
static_assert(PC_COUNT == HIPTS_VIT_PROF_CATEGORIES, "category count");
static const char* const kProfNames[PC_COUNT] = {
    "patchify_kernel", "gemm_kernel<EPI_PATCH>", "layernorm_kernel", "gemm_kernel<EPI_QKV>", "gemm_kernel<EPI_VT>",
    "attn2_kernel", "gemm_kernel<EPI_RESID>", "gemm_kernel<EPI_GELU>", "pool_kernels", "gemm_kernel<EPI_HEAD>", "other"};

struct ProfRec {
    int cat;
    hipEvent_t a, b;
    double flops, bytes;
};

struct hipts_vit {
    bool prof = false;          // events are recorded during THIS forward call
    int prof_every = 0;         // 0 = off, n = record every n-th forward call (sampling keeps event overhead out of the step time)
    uint32_t prof_mask = 0xffffffffu;      // categories recorded (hipts_vit_profile_select)
    long prof_calls = 0;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;      // recycled events
    double acc_ms[PC_COUNT] = {0}, acc_flops[PC_COUNT] = {0}, acc_bytes[PC_COUNT] = {0};
    int64_t acc_n[PC_COUNT] = {0};
    int device = 0;
    hipts_vit_config_t cfg{};
    int grid = 0, tokens = 0, tokens_pad = 0, patch_k = 0;
    std::vector<Layer> layers;
    DevBuf patch_w, patch_w2, patch_b, patch_b_u8, pos, norm_g, norm_b, head_w, head_b;
    std::vector<float> h_patch_bias, h_patch_rowsum;   // bias and sum_k W[n][k] (of the bf16 values)
    std::vector<std::string> missing;   // tensors not yet set
    // workspace (sized for cfg.max_batch)
    DevBuf img_in, a0, x, xn, q, k, v, att, hmid, pool_part, pooled2, logits, probs, stat_part;
    DevBuf x_rm;                   // the last residual launch's rows, row-major, when the stream itself is blocked (GemmArgs::x_blocked)
    DevBuf sk_ws;                                 // split-K workspaces of the residual GEMMs (GemmArgs::sk_ws), one per sub-batch stream, zeroed once
    bool fold_ln = false;                         // LayerNorms folded into the GEMM epilogues (default; HIPTS_LN_FOLD=0 turns it off)
    bool fold_dirty = true;                       // a tensor changed: the folded vectors are rebuilt at the next forward
    bool split_att = false;                       // cfg.operand_f16 bit 4: the attention output travels as a hi | lo pair, proj runs K = 2 dim against [W | W]
    int pool_splits = 1;
    static constexpr int kMaxSub = 4;
    int want_sub = 0;                             // hipts_vit_set_sub_batches; 0 = default
    bool deferred_join = false;                   // hipts_vit_set_deferred_join
    int last_ns = 0;                              // sub-batch streams the last forward used (0: none to join)
    int pend_ns = 0, pend_batch = 0;              // an UNJOINED forward (deferred join) of pend_batch images on pend_ns streams may still run
    hipStream_t sub[kMaxSub] = {};                // internal streams of the sub-batches
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSub] = {};
    hipEvent_t ev_stagger = nullptr;              // HIPTS_VIT_STAGGER: recorded on sub-batch stream 0 after a chosen launch, waited for by the later streams
};

namespace {

// ---------------------------------------------------------------------------------------------
// patch matrix: A0[m][(ky*P + kx)*3 + c] = bf16(normalised pixel), c in memory (RGB) order; the
// BGR flip of tagging.py:243 is folded into the weight permutation at upload time.
// One thread per (token, ky): reads P*3 contiguous bytes, writes P*3 contiguous bf16.
// ---------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void patchify_u8_kernel(const uint8_t* __restrict__ img, bf16_t* __restrict__ a0, int batch,
                                                          int size, int P, int grid) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)batch * grid * grid * P;
    if (idx >= total) return;
    const int ky = (int)(idx % P);
    const int64_t tok = idx / P;
    const int px = (int)(tok % grid), py = (int)((tok / grid) % grid), b = (int)(tok / ((int64_t)grid * grid));
    const uint8_t* src = img + (((int64_t)b * size + (py * P + ky)) * size + px * P) * 3;
    bf16_t* dst = a0 + tok * (int64_t)(P * P * 3) + ky * P * 3;
    // The pixel is stored as the exact integer 0..255 (exact in bf16).  ToTensor + Normalize,
    // x = (u/255 - .5)/.5 = u*(2/255) - 1, is affine, so it is applied to the fp32 accumulator in the
    // GEMM epilogue: W.x = (2/255) W.u - rowsum(W).  Rounding x itself to bf16 would put the same
    // 256 rounding errors on every token -- a systematic error that mean-pooling does not average out.
    if (P == 16) {          // 48 contiguous bytes in, 96 contiguous bytes out: three 16-B loads, six 16-B stores
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const uint4 in = s4[v];
            const uint32_t w[4] = {in.x, in.y, in.z, in.w};
            bf16x8 lo, hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lo[e] = to_op<F16>((float)((w[e >> 2] >> (8 * (e & 3))) & 0xffu));
                hi[e] = to_op<F16>((float)((w[2 + (e >> 2)] >> (8 * (e & 3))) & 0xffu));
            }
            d4[2 * v] = *reinterpret_cast<const uint4*>(&lo);
            d4[2 * v + 1] = *reinterpret_cast<const uint4*>(&hi);
        }
        return;
    }
    for (int i = 0; i < P * 3; ++i) dst[i] = to_op<F16>((float)src[i]);
}

// The same for 16 x 16 patches with every global access coalesced (round 4): a workgroup takes up to 16 neighbouring tokens of one patch
// row -- 16 image rows x (tokens * 48) contiguous bytes -- with linear 16-byte loads, converts, lays the values out token-major in LDS
// ([token][ky][kx * 3 + c], 1536 B per token) and writes the tokens' rows, which are contiguous in the patch matrix, with linear 16-byte
// stores.  (patchify_u8_kernel reads 48-byte pieces 1344 B apart per lane: 39 us per 32 images for 58 MB.)
template <bool F16>
__global__ __launch_bounds__(256) void patchify_u8_p16_kernel(const uint8_t* __restrict__ img, bf16_t* __restrict__ a0, int size, int grid,
                                                              int xgroups) {
    __shared__ __attribute__((aligned(16))) char image[16 * 1536];
    const int tid = threadIdx.x;
    const int xg = blockIdx.x % xgroups, py = (blockIdx.x / xgroups) % grid, b = blockIdx.x / (xgroups * grid);
    const int px0 = xg * 16, ntok = min(16, grid - px0);
    const int chunks_per_row = ntok * 3;                               // 16-byte chunks of one image row inside the group
    const int total = 16 * chunks_per_row;
    const uint8_t* src0 = img + (((int64_t)b * size + (int64_t)py * 16) * size + px0 * 16) * 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int L = j * 256 + tid;
        if (L < total) {
            const int ky = L / chunks_per_row, c16 = L - ky * chunks_per_row;
            const uint4 in = *reinterpret_cast<const uint4*>(src0 + (int64_t)ky * size * 3 + c16 * 16);
            const uint32_t w[4] = {in.x, in.y, in.z, in.w};
            bf16x8 lo, hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                lo[e] = to_op<F16>((float)((w[e >> 2] >> (8 * (e & 3))) & 0xffu));
                hi[e] = to_op<F16>((float)((w[2 + (e >> 2)] >> (8 * (e & 3))) & 0xffu));
            }
            const int tok = c16 / 3, part = c16 - tok * 3;             // a 16-byte chunk never straddles two tokens (48 = 3 x 16)
            char* d = image + tok * 1536 + ky * 96 + part * 32;
            *reinterpret_cast<bf16x8*>(d) = lo;
            *reinterpret_cast<bf16x8*>(d + 16) = hi;
        }
    }
    __syncthreads();
    const int out_chunks = ntok * 96;                                   // 1536 B per token
    uint4* dst = reinterpret_cast<uint4*>(a0 + (((int64_t)b * grid + py) * grid + px0) * 768);
    for (int L = tid; L < out_chunks; L += 256) dst[L] = *reinterpret_cast<const uint4*>(image + L * 16);
}

// x: float32 [B][3][S][S] (BGR, already normalised).  Channel c of the patch matrix (memory/RGB
// order) is model channel 2 - c.  Each value is split into bf16 hi + bf16 lo (row = [hi(K) | lo(K)],
// multiplied against [W | W]) so the float32 input keeps ~16 significant bits.
template <bool F16>
__global__ __launch_bounds__(256) void patchify_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ a0, int batch,
                                                           int size, int P, int grid) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)batch * grid * grid * P;
    if (idx >= total) return;
    const int ky = (int)(idx % P);
    const int64_t tok = idx / P;
    const int px = (int)(tok % grid), py = (int)((tok / grid) % grid), b = (int)(tok / ((int64_t)grid * grid));
    const int K = P * P * 3;
    bf16_t* dst = a0 + tok * (int64_t)(2 * K) + ky * P * 3;
    for (int c = 0; c < 3; ++c) {
        const float* src = x + (((int64_t)b * 3 + (2 - c)) * size + (py * P + ky)) * size + px * P;
        for (int kx = 0; kx < P; ++kx) {
            const float v = src[kx];
            const bf16_t hi = to_op<F16>(v);
            dst[kx * 3 + c] = hi;
            dst[K + kx * 3 + c] = to_op<F16>(v - from_op<F16>(hi));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (eps inside the sqrt, biased variance -- torch F.layer_norm), one
// wave per row, float4 loads, two-pass statistics in registers, bf16 output.  D % 4 == 0, D <= 1024.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <bool F16, bool OUT8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ bta, bf16_t* __restrict__ out, int64_t rows,
                                                        int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nvec ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane + 64 * i < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    const float4* gr = reinterpret_cast<const float4*>(g);
    const float4* br = reinterpret_cast<const float4*>(bta);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float4 gg = gr[c], bb = bta ? br[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (OUT8)
                reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + row * D)[c] =
                    pack4_e4m3((v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                               (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w);
            else
            *reinterpret_cast<bf16x4*>(out + row * D + 4 * c) =
                pack4<F16>((v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                           (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// final norm + mean pool.  grid (splits, batch); each workgroup sums, over its share of the
// image's tokens, either (x - mean) * rstd (norm-then-pool: affine applied after the mean, it is
// linear) or x itself (pool-then-norm).  Partials [batch][splits][D] float32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* __restrict__ x, float* __restrict__ part, int tokens,
                                                           int D, float eps, int normalize, int splits) {
    __shared__ float red[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int split = blockIdx.x, b = blockIdx.y;
    const int per = (tokens + splits - 1) / splits;
    const int t0 = split * per, t1 = min(tokens, t0 + per);
    const int nvec = D >> 2;
    float4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the next row's loads are in flight under this row's two wave reductions (round 4: the kernel was a chain of ~25 dependent
    // load -> reduce -> reduce steps per wave, 58 us for 15 us of bytes; with 28 splits instead of 8 a wave owns 7 rows)
    auto load_row = [&](int t, float4 (&v)[4]) {
        const float4* xr = reinterpret_cast<const float4*>(x + ((int64_t)b * tokens + (t < t1 ? t : t1 - 1)) * D);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nvec ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 vn[4];
    if (t0 + wave < t1) load_row(t0 + wave, vn);
    for (int t = t0 + wave; t < t1; t += 4) {
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = vn[i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        if (t + 4 < t1) load_row(t + 4, vn);
        float mean = 0.f, rstd = 1.f;
        if (normalize) {
            mean = wave_sum(s) / (float)D;
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lane + 64 * i < nvec) {
                    const float a = v[i].x - mean, bq = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                    ss += (a * a + bq * bq) + (c * c + d * d);
                }
            rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i].x += (v[i].x - mean) * rstd;
            acc[i].y += (v[i].y - mean) * rstd;
            acc[i].z += (v[i].z - mean) * rstd;
            acc[i].w += (v[i].w - mean) * rstd;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            red[wave][4 * c + 0] = acc[i].x;
            red[wave][4 * c + 1] = acc[i].y;
            red[wave][4 * c + 2] = acc[i].z;
            red[wave][4 * c + 3] = acc[i].w;
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256)
        part[((int64_t)b * splits + split) * D + d] = (red[0][d] + red[1][d]) + (red[2][d] + red[3][d]);
}

// Sum the partials, apply the affine (or the LayerNorm for pool-then-norm), and split the float32
// feature into bf16 hi + bf16 lo so the head GEMM (K = 2D against [W | W]) keeps ~16 bits of the
// feature: out[b][0..D) = hi, out[b][D..2D) = lo.   One workgroup per image.
template <bool F16>
__global__ __launch_bounds__(256) void pool_finalize_kernel(const float* __restrict__ part, const float* __restrict__ g,
                                                            const float* __restrict__ bta, bf16_t* __restrict__ out, int tokens,
                                                            int D, float eps, int normalize_after, int splits) {
    __shared__ float feat[1024];
    __shared__ float red[8];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float a = 0.f;
        for (int sp = 0; sp < splits; ++sp) a += part[((int64_t)b * splits + sp) * D + d];
        a = a / (float)tokens;
        feat[d] = a;
        s += a;
    }
    float mean = 0.f, rstd = 1.f;
    if (normalize_after) {
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
        float ss = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) {
            const float c = feat[d] - mean;
            ss += c * c;
        }
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = ss;
        __syncthreads();
        rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)D + eps);
    }
    for (int d = threadIdx.x; d < D; d += 256) {
        const float f = (feat[d] - mean) * rstd * g[d] + bta[d];
        const bf16_t hi = to_op<F16>(f);
        const bf16_t lo = to_op<F16>(f - from_op<F16>(hi));
        out[(int64_t)b * 2 * D + d] = hi;
        out[(int64_t)b * 2 * D + D + d] = lo;
    }
}

int set_f32(DevBuf& buf, const float* data, size_t n) {
    HIPTS_TRY(buf.alloc(n * 4));
    return upload(buf.p, data, n * 4);
}

// rows x cols float32 -> bf16 (or half), rows zero-padded to rows_pad
bool g_upload_f16 = false;      // set from the handle around hipts_vit_set_tensor (single caller per handle)
int set_bf16_matrix(DevBuf& buf, const float* data, int rows, int cols, int rows_pad) {
    std::vector<uint16_t> h((size_t)rows_pad * cols, 0);
    if (g_upload_f16) for (size_t i = 0; i < (size_t)rows * cols; ++i) h[i] = f32_to_f16_rne(data[i]);
    else for (size_t i = 0; i < (size_t)rows * cols; ++i) h[i] = f32_to_bf16_rne(data[i]);
    HIPTS_TRY(buf.alloc(h.size() * 2));
    return upload(buf.p, h.data(), h.size() * 2);
}

bool erase_missing(hipts_vit* h, const std::string& key) {
    for (size_t i = 0; i < h->missing.size(); ++i)
        if (h->missing[i] == key) {
            h->missing.erase(h->missing.begin() + i);
            return true;
        }
    return false;
}

}  // namespace

extern "C" {

int hipts_vit_create(const hipts_vit_config_t* cfg, int device, hipts_vit_t** out) {
    HIPTS_REQUIRE(cfg && out, "hipts_vit_create: null argument");
    HIPTS_REQUIRE(cfg->patch >= 4 && cfg->image_size % cfg->patch == 0, "image_size must be a multiple of patch");
    HIPTS_REQUIRE(cfg->heads >= 1 && cfg->dim == cfg->heads * 64, "head dim must be 64 (dim = heads * 64)");
    HIPTS_REQUIRE(cfg->dim % 64 == 0 && cfg->dim <= 1024, "dim must be a multiple of 64, <= 1024");
    HIPTS_REQUIRE(cfg->mlp_dim % 64 == 0, "mlp_dim must be a multiple of 64");
    HIPTS_REQUIRE((cfg->patch * cfg->patch * 3) % 64 == 0, "patch*patch*3 must be a multiple of 64");
    HIPTS_REQUIRE(cfg->depth >= 1 && cfg->num_classes >= 1 && cfg->max_batch >= 1, "bad depth / classes / max_batch");
    const int grid = cfg->image_size / cfg->patch;
    HIPTS_REQUIRE((grid * grid) % 4 == 0, "token count must be a multiple of 4");
    HIPTS_TRY(use_device(device));
    auto* h = new hipts_vit();
    h->device = device;
    h->cfg = *cfg;
    h->grid = grid;
    h->tokens = grid * grid;
    h->tokens_pad = round_up(h->tokens, 64);
    h->patch_k = cfg->patch * cfg->patch * 3;
    h->split_att = (cfg->operand_f16 & HIPTS_OPERAND_SPLIT_ATT) != 0;
    h->layers.resize(cfg->depth);
    h->missing = {"patch_embed.proj.weight", "patch_embed.proj.bias", "pos_embed", "norm.weight", "norm.bias",
                  "head.weight", "head.bias"};
    for (int i = 0; i < cfg->depth; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        for (const char* s : {"norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                              "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                              "mlp.fc2.weight", "mlp.fc2.bias"})
            h->missing.push_back(p + s);
    }
    const size_t B = cfg->max_batch, M = B * h->tokens, D = cfg->dim;
    const size_t qkv_elems = B * cfg->heads * (size_t)h->tokens_pad * 64;
    h->pool_splits = h->tokens >= 64 ? std::min(32, std::max(8, h->tokens / 28)) : 1;      // 784 tokens: 28 splits of 28 tokens, 7 per wave
    if (getenv("HIPTS_POOL_SPLITS") && h->tokens >= 64) h->pool_splits = std::max(1, std::min(32, atoi(getenv("HIPTS_POOL_SPLITS"))));      // A/B
    int st = HIPTS_OK;
    if ((st = h->a0.alloc(M * h->patch_k * 2 * 2)) || (st = h->x.alloc(M * D * 4)) || (st = h->x_rm.alloc(M * D * 4)) || (st = h->xn.alloc(M * D * 2)) ||
        (st = h->q.alloc(qkv_elems * 2)) || (st = h->k.alloc(qkv_elems * 2)) || (st = h->v.alloc(qkv_elems * 2)) ||
        (st = h->att.alloc(M * D * 2 * (h->split_att ? 2 : 1))) || (st = h->hmid.alloc(M * (size_t)cfg->mlp_dim * 2)) ||
        (st = h->pool_part.alloc(B * h->pool_splits * D * 4)) || (st = h->pooled2.alloc(B * 2 * D * 2)) ||
        (st = h->logits.alloc(B * (size_t)cfg->num_classes * 4)) || (st = h->probs.alloc(B * (size_t)cfg->num_classes * 4))) {
        delete h;
        return st;
    }
    // On by default (HIPTS_LN_FOLD=0 restores the separate LayerNorm kernels): +3.4 % images/s (4.57 -> 4.73 k, two A/B pairs on one
    // box) once the residual epilogue interleaved the 16-bit copy with its read-modify-write; the residual GEMM's own launch
    // grows from 174 to 207 us for the same flops, which is what `roofline` in the bench line then shows for that kernel.
    h->fold_ln = D % 64 == 0 && h->tokens % 8 == 0 && cfg->mlp_dim % 8 == 0 && !getenv("HIPTS_GEMM") &&
                 !(getenv("HIPTS_LN_FOLD") && atoi(getenv("HIPTS_LN_FOLD")) == 0);
    if (h->fold_ln) {
        if ((st = h->stat_part.alloc(((D + 255) / 256) * M * 8))) {      // (sum x, sum x^2) per row and 256-column tile
            delete h;
            return st;
        }
        for (auto& L : h->layers)
            if ((st = L.qkv_u.alloc(3 * D * 4)) || (st = L.qkv_c.alloc(3 * D * 4)) || (st = L.fc1_u.alloc((size_t)cfg->mlp_dim * 4)) ||
                (st = L.fc1_c.alloc((size_t)cfg->mlp_dim * 4))) {
                delete h;
                return st;
            }
    }
    {   // split-K tail of the residual GEMMs: tickets must start at zero (the kernels leave them so)
        hipError_t e2;
        if ((st = h->sk_ws.alloc(hipts_vit::kMaxSub * GEMM_SK_WS_BYTES)) || (e2 = hipMemset(h->sk_ws.p, 0, h->sk_ws.bytes)) != hipSuccess) {
            delete h;
            return st ? st : set_error(HIPTS_ERR_HIP, "hipMemset failed");
        }
    }
    // padded token rows of q / k / v must be finite (zero): cleared once, never written afterwards
    hipError_t e;
    if ((e = hipMemset(h->q.p, 0, h->q.bytes)) != hipSuccess || (e = hipMemset(h->k.p, 0, h->k.bytes)) != hipSuccess ||
        (e = hipMemset(h->v.p, 0, h->v.bytes)) != hipSuccess) {
        delete h;
        return set_error(HIPTS_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_vit_destroy(hipts_vit_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        for (auto st : h->sub)
            if (st) (void)hipStreamDestroy(st);
        for (auto e : h->ev_join)
            if (e) (void)hipEventDestroy(e);
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        if (h->ev_stagger) (void)hipEventDestroy(h->ev_stagger);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_vit_set_tensor(hipts_vit_t* h, const char* key_c, const float* data, int64_t numel) {
    HIPTS_REQUIRE(h && key_c && data, "hipts_vit_set_tensor: null argument");
    HIPTS_TRY(use_device(h->device));
    g_upload_f16 = (h->cfg.operand_f16 & 1) != 0;
    const std::string key(key_c);
    const auto& c = h->cfg;
    const int D = c.dim, P = c.patch, Mlp = c.mlp_dim, C = c.num_classes;
#define EXPECT(n)                                                                                                      \
    HIPTS_REQUIRE(numel == (int64_t)(n), "tensor %s: expected %lld elements, got %lld", key_c, (long long)(n),         \
                  (long long)numel)
    int st = HIPTS_ERR_INVALID;
    if (key == "patch_embed.proj.weight") {
        EXPECT((int64_t)D * 3 * P * P);
        // [D][3][P][P] (model channel order = BGR) -> [D][(ky*P + kx)*3 + c_rgb], model channel 2 - c_rgb
        std::vector<float> perm((size_t)D * h->patch_k);
        for (int n = 0; n < D; ++n)
            for (int ky = 0; ky < P; ++ky)
                for (int kx = 0; kx < P; ++kx)
                    for (int cr = 0; cr < 3; ++cr)
                        perm[(size_t)n * h->patch_k + (ky * P + kx) * 3 + cr] = data[(((size_t)n * 3 + (2 - cr)) * P + ky) * P + kx];
        st = set_bf16_matrix(h->patch_w, perm.data(), D, h->patch_k, round_up(D, 256));
        if (st == HIPTS_OK) {
            std::vector<float> dup((size_t)D * 2 * h->patch_k);
            h->h_patch_rowsum.assign(D, 0.f);
            for (int n = 0; n < D; ++n) {
                double rs = 0.0;
                for (int kk = 0; kk < h->patch_k; ++kk) {
                    float wv;
                    if (g_upload_f16) {
                        wv = f16_bits_to_f32(f32_to_f16_rne(perm[(size_t)n * h->patch_k + kk]));
                    } else {
                        const uint32_t bits = (uint32_t)f32_to_bf16_rne(perm[(size_t)n * h->patch_k + kk]) << 16;
                        memcpy(&wv, &bits, 4);
                    }
                    rs += (double)wv;
                }
                h->h_patch_rowsum[n] = (float)rs;
                memcpy(&dup[(size_t)n * 2 * h->patch_k], &perm[(size_t)n * h->patch_k], (size_t)h->patch_k * 4);
                memcpy(&dup[(size_t)n * 2 * h->patch_k + h->patch_k], &perm[(size_t)n * h->patch_k], (size_t)h->patch_k * 4);
            }
            st = set_bf16_matrix(h->patch_w2, dup.data(), D, 2 * h->patch_k, round_up(D, 256));
        }
    } else if (key == "patch_embed.proj.bias") {
        EXPECT(D);
        h->h_patch_bias.assign(data, data + D);
        st = set_f32(h->patch_b, data, D);
    } else if (key == "pos_embed") {
        EXPECT((int64_t)h->tokens * D);
        st = set_f32(h->pos, data, (size_t)h->tokens * D);
    } else if (key == "norm.weight") {
        EXPECT(D);
        st = set_f32(h->norm_g, data, D);
    } else if (key == "norm.bias") {
        EXPECT(D);
        st = set_f32(h->norm_b, data, D);
    } else if (key == "head.weight") {
        EXPECT((int64_t)C * D);
        // [C][D] -> [C][2D] = [W | W]: multiplies the (hi | lo) split of the pooled feature
        std::vector<float> dup((size_t)C * 2 * D);
        for (int n = 0; n < C; ++n) {
            memcpy(&dup[(size_t)n * 2 * D], &data[(size_t)n * D], (size_t)D * 4);
            memcpy(&dup[(size_t)n * 2 * D + D], &data[(size_t)n * D], (size_t)D * 4);
        }
        st = set_bf16_matrix(h->head_w, dup.data(), C, 2 * D, round_up(C, 256));
    } else if (key == "head.bias") {
        EXPECT(C);
        st = set_f32(h->head_b, data, C);
    } else if (key.rfind("blocks.", 0) == 0) {
        const size_t dot = key.find('.', 7);
        HIPTS_REQUIRE(dot != std::string::npos, "unknown tensor key %s", key_c);
        const int li = atoi(key.substr(7, dot - 7).c_str());
        HIPTS_REQUIRE(li >= 0 && li < c.depth, "tensor %s: block index out of range", key_c);
        Layer& L = h->layers[li];
        const std::string sub = key.substr(dot + 1);
        if (sub == "norm1.weight") { EXPECT(D); st = set_f32(L.ln1_g, data, D); }
        else if (sub == "norm1.bias") { EXPECT(D); st = set_f32(L.ln1_b, data, D); }
        else if (sub == "norm2.weight") { EXPECT(D); st = set_f32(L.ln2_g, data, D); }
        else if (sub == "norm2.bias") { EXPECT(D); st = set_f32(L.ln2_b, data, D); }
        else if (sub == "attn.qkv.weight") { EXPECT((int64_t)3 * D * D); st = set_bf16_matrix(L.qkv_w, data, 3 * D, D, round_up(2 * D, 256) + round_up(D, 256) + 256); }
        else if (sub == "attn.qkv.bias") { EXPECT(3 * D); st = set_f32(L.qkv_b, data, 3 * D); }
        else if (sub == "attn.proj.weight") {
            EXPECT((int64_t)D * D);
            if (h->split_att) {      // [D][2D] = [W | W]: multiplies the (hi | lo) halves of the attention output
                std::vector<float> dup((size_t)D * 2 * D);
                const float inv = 1.0f / split_lo_scale(g_upload_f16);      // the low halves arrive multiplied by the scale
                for (int n = 0; n < D; ++n) {
                    memcpy(&dup[(size_t)n * 2 * D], &data[(size_t)n * D], (size_t)D * 4);
                    for (int kk = 0; kk < D; ++kk) dup[(size_t)n * 2 * D + D + kk] = data[(size_t)n * D + kk] * inv;
                }
                st = set_bf16_matrix(L.proj_w, dup.data(), D, 2 * D, round_up(D, 256));
            } else {
                st = set_bf16_matrix(L.proj_w, data, D, D, round_up(D, 256));
            }
        }
        else if (sub == "attn.proj.bias") { EXPECT(D); st = set_f32(L.proj_b, data, D); }
        else if (sub == "mlp.fc1.weight") { EXPECT((int64_t)Mlp * D); st = set_bf16_matrix(L.fc1_w, data, Mlp, D, round_up(Mlp, 256)); }
        else if (sub == "mlp.fc1.bias") { EXPECT(Mlp); st = set_f32(L.fc1_b, data, Mlp); }
        else if (sub == "mlp.fc2.weight") { EXPECT((int64_t)D * Mlp); st = set_bf16_matrix(L.fc2_w, data, D, Mlp, round_up(D, 256)); }
        else if (sub == "mlp.fc2.bias") { EXPECT(D); st = set_f32(L.fc2_b, data, D); }
        else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
    } else {
        return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
    }
#undef EXPECT
    if (st == HIPTS_OK) erase_missing(h, key);
    if (st == HIPTS_OK) h->fold_dirty = true;
    if (st == HIPTS_OK && (key == "patch_embed.proj.weight" || key == "patch_embed.proj.bias") &&
        !h->h_patch_bias.empty() && !h->h_patch_rowsum.empty()) {
        std::vector<float> eff(D);
        for (int n = 0; n < D; ++n) eff[n] = (float)((double)h->h_patch_bias[n] - (double)h->h_patch_rowsum[n]);
        st = set_f32(h->patch_b_u8, eff.data(), D);   // bias of the u8 path: b - rowsum(W)
    }
    return st;
}

int hipts_vit_profile_name(int category, char* buf, size_t n) {
    HIPTS_REQUIRE(buf && n > 0 && category >= 0 && category < PC_COUNT, "bad category");
    snprintf(buf, n, "%s", kProfNames[category]);
    return HIPTS_OK;
}

int hipts_vit_flops_per_image(const hipts_vit_t* h, double* flops) {
    HIPTS_REQUIRE(h && flops, "null argument");
    const double N = h->tokens, D = h->cfg.dim, Ml = h->cfg.mlp_dim, C = h->cfg.num_classes, Kp = h->patch_k;
    const double per_layer = 2 * N * D * 3 * D + 2 * 2 * N * N * D + 2 * N * D * D + 2 * 2 * N * D * Ml;
    *flops = 2 * N * Kp * D + h->cfg.depth * per_layer + 2 * D * C;
    return HIPTS_OK;
}

}  // extern "C"

namespace {

struct ProfScope {
    hipts_vit* h;
    hipStream_t s;
    ProfRec r{};
    bool on;
    ProfScope(hipts_vit* h_, hipStream_t s_, int cat, double flops, double bytes) : h(h_), s(s_), on(h_->prof && ((h_->prof_mask >> cat) & 1u)) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!h->pool.empty()) {
                e = h->pool.back();
                h->pool.pop_back();
            } else if (hipEventCreate(&e) != hipSuccess) {
                e = nullptr;
            }
            return e;
        };
        r.cat = cat;
        r.flops = flops;
        r.bytes = bytes;
        r.a = get();
        r.b = get();
        if (r.a) (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        if (r.b) (void)hipEventRecord(r.b, s);
        h->recs.push_back(r);
    }
};

int prof_resolve(hipts_vit* h) {
    for (auto& r : h->recs) {
        if (r.a && r.b) {
            HIPTS_HIP(hipEventSynchronize(r.b));
            float ms = 0.f;
            HIPTS_HIP(hipEventElapsedTime(&ms, r.a, r.b));
            h->acc_ms[r.cat] += ms;
            h->acc_n[r.cat] += 1;
            h->acc_flops[r.cat] += r.flops;
            h->acc_bytes[r.cat] += r.bytes;
        }
        if (r.a) h->pool.push_back(r.a);
        if (r.b) h->pool.push_back(r.b);
    }
    h->recs.clear();
    return HIPTS_OK;
}

// The whole kernel sequence for images [i0, i0 + nb) of the current call, on stream s.  Every workspace
// buffer is indexed by image (rows of M = batch * tokens, or (image, head) blocks), so disjoint image
// ranges can run on different streams at the same time.
int vit_run_images(hipts_vit* h, const void* in_dev, bool is_u8, int i0, int nb, float* lg, float* pr, hipStream_t s, bool shared_chip,
                   int stagger_at = 0, int sub = 0) {
    const auto& c = h->cfg;
    const int D = c.dim, P = c.patch, S = c.image_size, T = h->tokens, Tp = h->tokens_pad, H = c.heads;
    const int M = nb * T;
    const bool f16 = (c.operand_f16 & 1) != 0;
    const size_t r0 = (size_t)i0 * T;                                   // first token row
    const int a0_ld = is_u8 ? h->patch_k : 2 * h->patch_k;
    bf16_t* a0 = h->a0.as<bf16_t>() + r0 * a0_ld;
    float* x = h->x.as<float>() + r0 * D;
    float* x_rm = h->x_rm.as<float>() + r0 * D;
    bf16_t* xn = h->xn.as<bf16_t>() + r0 * D;
    const int att_k = h->split_att ? 2 * D : D;                          // row width of the attention output: [hi | lo] when split
    bf16_t* att = h->att.as<bf16_t>() + r0 * att_k;
    bf16_t* hmid = h->hmid.as<bf16_t>() + r0 * c.mlp_dim;
    const size_t qoff = (size_t)i0 * H * Tp * 64;
    bf16_t* q = h->q.as<bf16_t>() + qoff;
    bf16_t* k = h->k.as<bf16_t>() + qoff;
    bf16_t* v = h->v.as<bf16_t>() + qoff;
    float* pool_part = h->pool_part.as<float>() + (size_t)i0 * h->pool_splits * D;
    bf16_t* pooled2 = h->pooled2.as<bf16_t>() + (size_t)i0 * 2 * D;
    const size_t img_bytes = (size_t)S * S * 3 * (is_u8 ? 1 : 4);
    const char* in_p = (const char*)in_dev + (size_t)i0 * img_bytes;

    const double dM = (double)M, dD = (double)D, dMlp = (double)c.mlp_dim, dT = (double)T;
    {
        ProfScope ps(h, s, PC_PATCHIFY, 0.0, dM * h->patch_k * (is_u8 ? 3.0 : 6.0));
        const int64_t total = (int64_t)M * P;
        const int blocks = ceil_div(total, 256);
        if (is_u8 && P == 16 && (S * 3) % 16 == 0) {
            const int xgroups = (h->grid + 15) / 16;
            const int pblocks = nb * h->grid * xgroups;
            if (f16) patchify_u8_p16_kernel<true><<<pblocks, 256, 0, s>>>((const uint8_t*)in_p, a0, S, h->grid, xgroups);
            else patchify_u8_p16_kernel<false><<<pblocks, 256, 0, s>>>((const uint8_t*)in_p, a0, S, h->grid, xgroups);
        } else if (is_u8) {
            if (f16) patchify_u8_kernel<true><<<blocks, 256, 0, s>>>((const uint8_t*)in_p, a0, nb, S, P, h->grid);
            else patchify_u8_kernel<false><<<blocks, 256, 0, s>>>((const uint8_t*)in_p, a0, nb, S, P, h->grid);
        } else {
            if (f16) patchify_f32_kernel<true><<<blocks, 256, 0, s>>>((const float*)in_p, a0, nb, S, P, h->grid);
            else patchify_f32_kernel<false><<<blocks, 256, 0, s>>>((const float*)in_p, a0, nb, S, P, h->grid);
        }
        HIPTS_LAUNCH_CHECK();
    }
    GemmArgs g;
    // patch embedding: x = A0 W^T + b + pos
    g = GemmArgs{};
    g.f16 = f16;
    g.shared_chip = shared_chip;
    g.A = a0; g.M = M; g.N = D; g.out_f32 = x; g.pos = h->pos.as<float>(); g.tokens = T;
    if (is_u8) {   // exact integer pixels; affine normalisation folded into the epilogue
        g.W = h->patch_w.as<bf16_t>(); g.K = h->patch_k; g.bias = h->patch_b_u8.as<float>(); g.qscale = 2.0f / 255.0f;
    } else {       // hi | lo split of the float input against [W | W]
        g.W = h->patch_w2.as<bf16_t>(); g.K = 2 * h->patch_k; g.bias = h->patch_b.as<float>(); g.qscale = 1.0f;
    }
    const bool fold = h->fold_ln;
    // The fp32 residual stream as 16 x 16 blocks (round 5, gemm_epi.h::x_off): every epilogue load / store instruction then moves one
    // contiguous kilobyte instead of sixteen half lines.  Only the residual epilogues touch the stream (folded LayerNorms); the last
    // launch writes its rows row-major for the pool.  A sub-batch starts at a multiple of 16 rows when tokens % 16 == 0.
    // HIPTS_X_BLOCKED=0: row-major (A/B).
    static const bool xb_env = !(getenv("HIPTS_X_BLOCKED") && atoi(getenv("HIPTS_X_BLOCKED")) == 0);
    const bool xb = xb_env && fold && D % 16 == 0 && T % 16 == 0 && c.depth >= 1;
    const int sblocks = (D + 255) / 256;                 // partial (sum, sum of squares) pairs per row: one per 256-column tile
    float* stat_p = fold ? h->stat_part.as<float>() + 2 * (size_t)sblocks * r0 : nullptr;
    {
        ProfScope ps(h, s, PC_GEMM_PATCH, 2.0 * dM * dD * h->patch_k, dM * h->patch_k * 2 + dM * dD * 4 + (fold ? dM * dD * 2 : 0.0));
        if (fold) {      // the first norm1 is prepared by this epilogue as well (EPI_RESID_XG with `pos`: x = acc * qscale + bias + pos)
            g.out_bf16 = xn; g.ln_gamma = h->layers[0].ln1_g.as<float>(); g.stat_part = stat_p; g.stat_stride = M;
            g.x_blocked = xb ? 1 : 0;
            HIPTS_TRY(launch_gemm(EPI_RESID_XG, g, s));
        } else {
            HIPTS_TRY(launch_gemm(EPI_PATCH, g, s));
        }
    }

    const int ln_blocks = ceil_div(M, 4);
    // Folded LayerNorms (EPI_RESID_XG): the residual GEMM that produces a row also writes gamma * x as the next GEMM's 16-bit
    // operand and the row's partial sums; the consumer applies rstd / mean / beta in its epilogue.  No LayerNorm pass over the
    // fp32 stream (HBM-bound, 24 per forward) remains: the first norm1 is prepared by the patch GEMM's epilogue.
    auto folded = [&](GemmArgs& ga, const float* u, const float* cvec) {
        ga.stat_in = stat_p; ga.stat_in_blocks = sblocks; ga.stat_in_stride = M; ga.ln_dim = D; ga.ln_eps = c.ln_eps;
        ga.col_u = u; ga.bias = cvec;
    };
    auto layernorm = [&](const float* gamma, const float* beta) -> int {
        ProfScope ps(h, s, PC_LAYERNORM, 0.0, dM * dD * 6);
        if (f16) layernorm_kernel<true><<<ln_blocks, 256, 0, s>>>(x, gamma, beta, xn, M, D, c.ln_eps);
        else layernorm_kernel<false><<<ln_blocks, 256, 0, s>>>(x, gamma, beta, xn, M, D, c.ln_eps);
        HIPTS_LAUNCH_CHECK();
        return HIPTS_OK;
    };
    // x += A W^T + b; with next_gamma also xn = 16bit(gamma * x) and the row statistics of x for the consumer of that norm
    auto residual = [&](const bf16_t* A, const bf16_t* W, const float* bias, int K, const float* next_gamma, double flops, double bytes) -> int {
        GemmArgs r{};
        r.f16 = f16;
        r.shared_chip = shared_chip;
        r.A = A; r.W = W; r.M = M; r.N = D; r.K = K; r.bias = bias; r.out_f32 = x;
        r.sk_ws = h->sk_ws.as<char>() + (size_t)sub * GEMM_SK_WS_BYTES; r.sk_ws_bytes = GEMM_SK_WS_BYTES;      // this stream's split-K workspace
        r.x_blocked = xb ? 1 : 0;
        if (next_gamma) {
            r.out_bf16 = xn; r.ln_gamma = next_gamma; r.stat_part = stat_p; r.stat_stride = M;
        } else if (xb) {
            r.resid_rowmajor_out = x_rm;        // the forward's last residual launch: rows for the final norm + pool
        }
        {
            ProfScope ps(h, s, PC_GEMM_RESID, flops, bytes + (next_gamma ? dM * dD * 2 : 0.0));
            HIPTS_TRY(launch_gemm(next_gamma ? EPI_RESID_XG : EPI_RESID, r, s));
        }
        return HIPTS_OK;
    };
    for (int li = 0; li < c.depth; ++li) {
        Layer& L = h->layers[li];
        const bool ln1_folded = fold;                  // prepared by the previous layer's fc2 epilogue (layer 0: by the patch GEMM's)
        if (!ln1_folded) HIPTS_TRY(layernorm(L.ln1_g.as<float>(), L.ln1_b.as<float>()));
        // q, k, v in ONE launch over the fused qkv weight (N = 3 D; round 3): all three leave in the [image][head][token][64] layout, V in
        // its natural orientation -- the attention kernel reads it transposed out of LDS (attn2.hip), so no transposing epilogue, one
        // launch and one pass over the activations less per layer.  q pre-scaled for the base-2 softmax.
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = xn; g.W = L.qkv_w.as<bf16_t>(); g.M = M; g.N = 3 * D; g.K = D;
        g.bias = L.qkv_b.as<float>(); g.out_bf16 = q; g.out2_bf16 = k; g.out3_bf16 = v;
        if (ln1_folded) folded(g, L.qkv_u.as<float>(), L.qkv_c.as<float>());
        g.tokens = T; g.tokens_pad = Tp; g.heads = H; g.dim = D;
        g.qscale = 0.125f * 1.4426950408889634f;   // head_dim^-0.5 (64^-0.5) * log2(e): attention works in base 2
        {
            ProfScope ps(h, s, PC_GEMM_QK, 2.0 * dM * 3 * dD * dD, dM * dD * 2 + dM * 3 * dD * 2);
            HIPTS_TRY(launch_gemm(EPI_QK, g, s));
        }
        // HIPTS_VIT_STAGGER (A/B): the later sub-batch streams start only when stream 0 is this far into its FIRST layer, so that the
        // streams run out of phase -- one's attention (32 KB of LDS per workgroup) beside the other's GEMMs (128 KB) on the same CUs
        auto stagger = [&](int pos) {
            if (li == 0 && stagger_at == pos && h->ev_stagger) (void)hipEventRecord(h->ev_stagger, s);
        };
        stagger(1);
        {
            ProfScope ps(h, s, PC_ATTENTION, 4.0 * nb * H * dT * dT * 64, dM * dD * 2 * 4);
            HIPTS_TRY(launch_attention2(q, k, v, att, nb, H, T, Tp, f16, s, 0, 0, h->split_att ? 1 : 0, split_lo_scale(f16)));
        }
        stagger(2);
        // x += att Wp^T + b  (+ norm2 prepared)
        HIPTS_TRY(residual(att, L.proj_w.as<bf16_t>(), L.proj_b.as<float>(), att_k, fold ? L.ln2_g.as<float>() : nullptr, 2.0 * dM * dD * att_k,
                           dM * att_k * 2 + dM * dD * 8));
        stagger(3);
        if (!fold) HIPTS_TRY(layernorm(L.ln2_g.as<float>(), L.ln2_b.as<float>()));
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = xn; g.W = L.fc1_w.as<bf16_t>(); g.M = M; g.N = c.mlp_dim; g.K = D;
        g.bias = L.fc1_b.as<float>(); g.out_bf16 = hmid; g.gelu_tanh = c.gelu_tanh;
        if (fold) folded(g, L.fc1_u.as<float>(), L.fc1_c.as<float>());
        {
            ProfScope ps(h, s, PC_GEMM_GELU, 2.0 * dM * dD * dMlp, dM * dD * 2 + dM * dMlp * 2);
            HIPTS_TRY(launch_gemm(EPI_GELU, g, s));
        }
        stagger(4);
        // x += hmid W2^T + b  (+ the next layer's norm1 prepared)
        HIPTS_TRY(residual(hmid, L.fc2_w.as<bf16_t>(), L.fc2_b.as<float>(), c.mlp_dim,
                           (fold && li + 1 < c.depth) ? h->layers[li + 1].ln1_g.as<float>() : nullptr, 2.0 * dM * dD * dMlp,
                           dM * dMlp * 2 + dM * dD * 8));
        stagger(5);
    }
    // final norm + mean pool (+ hi/lo split)
    {
        ProfScope ps(h, s, PC_POOL, 0.0, dM * dD * 4);
        pool_partial_kernel<<<dim3(h->pool_splits, nb), 256, 0, s>>>(xb ? x_rm : x, pool_part, T, D, c.ln_eps, c.pool_then_norm ? 0 : 1, h->pool_splits);
        HIPTS_LAUNCH_CHECK();
        if (f16)
            pool_finalize_kernel<true><<<nb, 256, 0, s>>>(pool_part, h->norm_g.as<float>(), h->norm_b.as<float>(), pooled2, T, D, c.ln_eps,
                                                          c.pool_then_norm ? 1 : 0, h->pool_splits);
        else
            pool_finalize_kernel<false><<<nb, 256, 0, s>>>(pool_part, h->norm_g.as<float>(), h->norm_b.as<float>(), pooled2, T, D, c.ln_eps,
                                                           c.pool_then_norm ? 1 : 0, h->pool_splits);
        HIPTS_LAUNCH_CHECK();
    }
    // head (+ sigmoid, tagging.py:176)
    g = GemmArgs{};
    g.f16 = f16;
    g.shared_chip = shared_chip;
    g.A = pooled2; g.W = h->head_w.as<bf16_t>(); g.M = nb; g.N = c.num_classes; g.K = 2 * D;
    g.sk_ws = h->sk_ws.as<char>() + (size_t)sub * GEMM_SK_WS_BYTES; g.sk_ws_bytes = GEMM_SK_WS_BYTES;      // split-K: 43 tiles on 256 CUs
    g.bias = h->head_b.as<float>(); g.out_f32 = lg ? lg + (size_t)i0 * c.num_classes : nullptr;
    g.out2_f32 = pr ? pr + (size_t)i0 * c.num_classes : nullptr;
    {
        ProfScope ps(h, s, PC_GEMM_HEAD, 2.0 * nb * dD * c.num_classes, (double)c.num_classes * 2 * dD * 2);
        HIPTS_TRY(launch_gemm(EPI_HEAD, g, s));
    }
    return HIPTS_OK;
}

int vit_forward_impl(hipts_vit* h, const void* input, int in_memspace, bool is_u8, int batch, float* logits_out,
                     float* probs_out, int out_memspace, hipStream_t s) {
    HIPTS_REQUIRE(h && input && batch >= 1, "hipts_vit_forward: bad arguments");
    HIPTS_REQUIRE(batch <= h->cfg.max_batch, "batch %d exceeds max_batch %d", batch, h->cfg.max_batch);
    if (!h->missing.empty())
        return set_error(HIPTS_ERR_STATE, "hipts_vit_forward: %zu checkpoint tensors not set (first: %s)", h->missing.size(),
                         h->missing[0].c_str());
    HIPTS_TRY(use_device(h->device));
    h->prof = h->prof_every > 0 && (h->prof_calls++ % h->prof_every) == 0;
    const auto& c = h->cfg;
    const int S = c.image_size;

    // Two half-batches on two internal streams: a GEMM grid's partial last round, an epilogue that is
    // waiting on HBM and every kernel boundary of one half are filled with work of the other half.
    static const int want_streams = getenv("HIPTS_VIT_STREAMS") ? atoi(getenv("HIPTS_VIT_STREAMS")) : 2;
    const int ns = std::min({h->want_sub > 0 ? h->want_sub : want_streams, (int)hipts_vit::kMaxSub, batch / 8});
    // Workspaces are carved by image offset, not by sub-batch stream.  A previous forward left unjoined (deferred join) is
    // ordered against this one only stream by stream, which is enough exactly when sub-batch i covers the same images as
    // before and nothing is staged through the shared input buffer; otherwise this call waits for all of it first.
    if (h->pend_ns > 0 && (h->pend_ns != ns || h->pend_batch != batch || in_memspace != HIPTS_DEVICE))
        for (int i = 0; i < h->pend_ns; ++i) HIPTS_HIP(hipStreamWaitEvent(s, h->ev_join[i], 0));
    h->pend_ns = 0;

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
            HIPTS_TRY(launch_fold_ln(L.fc1_w.as<bf16_t>(), f16w, L.ln2_g.as<float>(), L.ln2_b.as<float>(), L.fc1_b.as<float>(), L.fc1_u.as<float>(),
                                     L.fc1_c.as<float>(), c.mlp_dim, c.dim, s));
        }
        h->fold_dirty = false;
    }

    if (ns >= 2) {
        if (!h->ev_fork) HIPTS_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < ns; ++i)
            if (!h->sub[i]) {
                HIPTS_HIP(hipStreamCreateWithFlags(&h->sub[i], hipStreamNonBlocking));
                if (!h->ev_join[i]) HIPTS_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
            }
        HIPTS_HIP(hipEventRecord(h->ev_fork, s));
        static const int stagger_at = getenv("HIPTS_VIT_STAGGER") ? atoi(getenv("HIPTS_VIT_STAGGER")) : 0;
        if (stagger_at > 0 && !h->ev_stagger) HIPTS_HIP(hipEventCreateWithFlags(&h->ev_stagger, hipEventDisableTiming));
        for (int i = 0; i < ns; ++i) {
            const int i0 = (int)((int64_t)batch * i / ns), i1 = (int)((int64_t)batch * (i + 1) / ns);
            HIPTS_HIP(hipStreamWaitEvent(h->sub[i], h->ev_fork, 0));
            if (i > 0 && stagger_at > 0) HIPTS_HIP(hipStreamWaitEvent(h->sub[i], h->ev_stagger, 0));
            HIPTS_TRY(vit_run_images(h, in_dev, is_u8, i0, i1 - i0, lg, pr, h->sub[i], true, i == 0 ? stagger_at : 0, i));
            HIPTS_HIP(hipEventRecord(h->ev_join[i], h->sub[i]));
            if (!h->deferred_join || !dev_out) HIPTS_HIP(hipStreamWaitEvent(s, h->ev_join[i], 0));
        }
        h->last_ns = (h->deferred_join && dev_out) ? ns : 0;
        h->pend_ns = h->last_ns;
        h->pend_batch = batch;
    } else {
        HIPTS_TRY(vit_run_images(h, in_dev, is_u8, 0, batch, lg, pr, s, false));
        h->last_ns = 0;
        if (h->deferred_join && dev_out) {
            // one stream (small batch, HIPTS_VIT_STREAMS=1): the work sits on the caller's stream, but hipts_vit_join's contract is that
            // ANY consuming stream may join -- give it an event to wait for
            if (!h->ev_join[0]) HIPTS_HIP(hipEventCreateWithFlags(&h->ev_join[0], hipEventDisableTiming));
            HIPTS_HIP(hipEventRecord(h->ev_join[0], s));
            h->last_ns = 1;
        }
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

namespace hipts {
namespace {
// rowstat[m] = (rstd, rstd * mean) of row m from the per-64-column partial (sum, sum of squares) pairs an epilogue wrote:
// part[b * stride + m].  64 rows x 4 quarters per workgroup: quarter q sums blocks q, q + 4, ..., then the quarters are
// combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void rowstat_kernel(const float2* __restrict__ part, float2* __restrict__ rowstat, int M, int stride, int blocks,
                                                      int D, float eps) {
    const int q = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + (threadIdx.x & 63);
    __shared__ float2 red[4][64];
    float s1 = 0.f, s2 = 0.f;
    if (m < M) {
#pragma unroll 4
        for (int b = q; b < blocks; b += 4) {
            const float2 v = part[(size_t)b * stride + m];
            s1 += v.x;
            s2 += v.y;
        }
    }
    red[q][threadIdx.x & 63] = make_float2(s1, s2);
    __syncthreads();
    if (q != 0 || m >= M) return;
    const float2 a = red[0][threadIdx.x], b2 = red[1][threadIdx.x], c = red[2][threadIdx.x], d = red[3][threadIdx.x];
    s1 = (a.x + b2.x) + (c.x + d.x);
    s2 = (a.y + b2.y) + (c.y + d.y);
    const float mean = s1 / (float)D;
    const float var = fmaxf(s2 / (float)D - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    rowstat[m] = make_float2(rstd, rstd * mean);
}

// u[n] = sum_k W[n][k] gamma[k], c[n] = sum_k W[n][k] beta[k] + bias[n] in float64 from the 16-bit operand values of W (so that the
// mean term cancels against what the MFMA sums).  One wave per output row; runs once per checkpoint.
template <bool F16>
__global__ __launch_bounds__(256) void fold_ln_kernel(const bf16_t* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias, float* __restrict__ u, float* __restrict__ c, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    double su = 0.0, sc = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double w = (double)from_op<F16>(W[(size_t)n * K + k]);
        su += w * (double)gamma[k];
        if (beta) sc += w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        su += __shfl_xor(su, o);
        sc += __shfl_xor(sc, o);
    }
    if (lane == 0) {
        u[n] = (float)su;
        c[n] = (float)(sc + (bias ? (double)bias[n] : 0.0));
    }
}
}  // namespace

int launch_rowstat(const float* part, float* rowstat, int M, int stride, int blocks, int D, float eps, hipStream_t s) {
    rowstat_kernel<<<(M + 63) / 64, 256, 0, s>>>(reinterpret_cast<const float2*>(part), reinterpret_cast<float2*>(rowstat), M, stride, blocks, D, eps);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

int launch_fold_ln(const bf16_t* W, bool f16, const float* gamma, const float* beta, const float* bias, float* u, float* c, int N, int K,
                   hipStream_t s) {
    if (f16) fold_ln_kernel<true><<<(N + 3) / 4, 256, 0, s>>>(W, gamma, beta, bias, u, c, N, K);
    else fold_ln_kernel<false><<<(N + 3) / 4, 256, 0, s>>>(W, gamma, beta, bias, u, c, N, K);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

int launch_layernorm(const float* x, const float* g, const float* b, bf16_t* out, int64_t rows, int D, float eps, bool f16,
                     hipStream_t s) {
    HIPTS_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm: D=%d must be a multiple of 4, at most 1024", D);
    const int blocks = (int)((rows + 3) / 4);
    if (f16) layernorm_kernel<true><<<blocks, 256, 0, s>>>(x, g, b, out, rows, D, eps);
    else layernorm_kernel<false><<<blocks, 256, 0, s>>>(x, g, b, out, rows, D, eps);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

int launch_layernorm8(const float* x, const float* g, const float* b, uint8_t* out, int64_t rows, int D, float eps, hipStream_t s) {
    HIPTS_REQUIRE(D % 4 == 0 && D >= 4 && D <= 1024, "layernorm: D=%d must be a multiple of 4, at most 1024", D);
    const int blocks = (int)((rows + 3) / 4);
    layernorm_kernel<true, true><<<blocks, 256, 0, s>>>(x, g, b, reinterpret_cast<bf16_t*>(out), rows, D, eps);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}
}  // namespace hipts

extern "C" {

int hipts_vit_set_sub_batches(hipts_vit_t* h, int n) {
    HIPTS_REQUIRE(h, "null handle");
    HIPTS_REQUIRE(n >= 0 && n <= hipts_vit::kMaxSub, "hipts_vit_set_sub_batches: n must be 0 .. %d", (int)hipts_vit::kMaxSub);
    h->want_sub = n;
    return HIPTS_OK;
}

int hipts_vit_set_deferred_join(hipts_vit_t* h, int on) {
    HIPTS_REQUIRE(h, "null handle");
    h->deferred_join = on != 0;
    return HIPTS_OK;
}

int hipts_vit_join(hipts_vit_t* h, void* stream) {
    HIPTS_REQUIRE(h, "null handle");
    HIPTS_TRY(use_device(h->device));
    for (int i = 0; i < h->last_ns; ++i) HIPTS_HIP(hipStreamWaitEvent((hipStream_t)stream, h->ev_join[i], 0));
    return HIPTS_OK;
}

int hipts_vit_profile_enable(hipts_vit_t* h, int enable) {
    HIPTS_REQUIRE(h, "null handle");
    HIPTS_TRY(use_device(h->device));
    HIPTS_TRY(prof_resolve(h));
    for (int i = 0; i < PC_COUNT; ++i) {
        h->acc_ms[i] = h->acc_flops[i] = h->acc_bytes[i] = 0.0;
        h->acc_n[i] = 0;
    }
    h->prof_every = enable > 0 ? enable : 0;
    h->prof_calls = 0;
    h->prof = false;
    return HIPTS_OK;
}

int hipts_vit_profile_select(hipts_vit_t* h, uint32_t category_mask) {
    HIPTS_REQUIRE(h, "null handle");
    h->prof_mask = category_mask;
    return HIPTS_OK;
}

int hipts_vit_profile_read(hipts_vit_t* h, int category, double* total_ms, int64_t* launches, double* total_flops,
                           double* total_bytes) {
    HIPTS_REQUIRE(h && category >= 0 && category < PC_COUNT, "bad category");
    HIPTS_TRY(use_device(h->device));
    HIPTS_TRY(prof_resolve(h));
    if (total_ms) *total_ms = h->acc_ms[category];
    if (launches) *launches = h->acc_n[category];
    if (total_flops) *total_flops = h->acc_flops[category];
    if (total_bytes) *total_bytes = h->acc_bytes[category];
    return HIPTS_OK;
}

int hipts_vit_forward_u8(hipts_vit_t* h, const uint8_t* images, int images_memspace, int batch, float* logits_out,
                         float* probs_out, int out_memspace, void* stream) {
    return vit_forward_impl(h, images, images_memspace, true, batch, logits_out, probs_out, out_memspace, (hipStream_t)stream);
}

int hipts_vit_forward_f32(hipts_vit_t* h, const float* x, int x_memspace, int batch, float* logits_out, float* probs_out,
                          int out_memspace, void* stream) {
    return vit_forward_impl(h, x, x_memspace, false, batch, logits_out, probs_out, out_memspace, (hipStream_t)stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Development aid (not part of the public ABI, not declared in include/hip_tagsearch.h): time one
// GEMM shape with random bf16 operands.  Used by tools/gemm_bench.py under gpurun.
// ---------------------------------------------------------------------------------------------
static float g_last_loop_ghz = 0.f;     // hiptsdbg_gemm_clock: shader clock during the main loop of the last stamped launch


// Diagnostic (bench.py): the shader clock the chip sustains inside the GEMM main loop -- cycle counter against the 100 MHz
// wall clock between the loop's first and last barrier, one workgroup's stamps of the last of `iters` back-to-back launches.
extern "C" int hiptsdbg_gemm_clock(int M, int N, int K, int epi, int iters, float* ms_out, float* loop_ghz) {
    HIPTS_REQUIRE(ms_out && loop_ghz, "null argument");
    const char* had = getenv("HIPTS_GEMM_STAMPS");
    setenv("HIPTS_GEMM_STAMPS", "quiet", 1);
    g_last_loop_ghz = 0.f;
    const int st = hiptsdbg_gemm_time(M, N, K, epi, iters, ms_out);
    if (!had) unsetenv("HIPTS_GEMM_STAMPS");
    *loop_ghz = g_last_loop_ghz;
    return st;
}

extern "C" int hiptsdbg_gemm_time(int M, int N, int K, int epi, int iters, float* ms_out) {
    HIPTS_TRY(use_device(0));
    const int Np = round_up(N, 256);
    DevBuf A, W, bias, of32, obf, obf2, pos;
    HIPTS_TRY(A.alloc((size_t)M * K * 2));
    HIPTS_TRY(W.alloc((size_t)Np * K * 2));
    HIPTS_TRY(bias.alloc((size_t)Np * 4));
    HIPTS_TRY(of32.alloc((size_t)M * Np * 4));
    // q/k/vT layouts pad 784 tokens to 832 per image: size the bf16 outputs for the padded layout
    const size_t padded_rows = ((size_t)M / 784 + 1) * 832;
    const size_t obytes = (padded_rows > (size_t)M ? padded_rows : (size_t)M) * Np * 2 + (1 << 20);
    HIPTS_TRY(obf.alloc(obytes));
    HIPTS_TRY(obf2.alloc(obytes));
    HIPTS_TRY(pos.alloc((size_t)1024 * Np * 4));
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)Np * K);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    const bool dbg_f16 = getenv("HIPTS_DBG_GEMM_F16") != nullptr;      // IEEE-half operands (the product's default)
    for (auto& v : ha) v = dbg_f16 ? f32_to_f16_rne(rnd()) : f32_to_bf16_rne(rnd());
    for (auto& v : hw) v = dbg_f16 ? f32_to_f16_rne(rnd() * 0.05f) : f32_to_bf16_rne(rnd() * 0.05f);
    if (getenv("HIPTS_DBG_GEMM_ZERO")) {      // all-zero operands: no switching in the matrix pipe, the clock stays up -- the loop's issue rate without the power limit
        for (auto& v : ha) v = 0;
        for (auto& v : hw) v = 0;
    }
    const bool op8 = getenv("HIPTS_GEMM_OP8") != nullptr;      // e4m3 operands: the same bytes, two finite codes per element
    if (op8) {
        for (auto& v : ha) v &= 0xbfbf;
        for (auto& v : hw) v &= 0xbfbf;
    }
    HIPTS_TRY(upload(A.p, ha.data(), ha.size() * 2));
    HIPTS_TRY(upload(W.p, hw.data(), hw.size() * 2));
    HIPTS_HIP(hipMemset(bias.p, 0, bias.bytes));
    HIPTS_HIP(hipMemset(of32.p, 0, of32.bytes));
    HIPTS_HIP(hipMemset(pos.p, 0, pos.bytes));
    GemmArgs g{};
    g.A = A.as<bf16_t>(); g.W = W.as<bf16_t>(); g.M = M; g.N = N; g.K = K; g.bias = bias.as<float>();
    g.out_f32 = of32.as<float>(); g.out_bf16 = obf.as<bf16_t>(); g.out2_bf16 = obf2.as<bf16_t>(); g.pos = pos.as<float>();
    g.tokens = 784; g.tokens_pad = 832; g.heads = N / 128 > 0 ? N / 128 : 1; g.dim = N / 2; g.qscale = 0.125f;
    if (epi == EPI_VT) { g.heads = N / 64; g.dim = N; }
    DevBuf statp;
    if (epi == EPI_RESID_XG) {      // the ViT's residual launches: x += ..., the next LayerNorm's 16-bit gamma * x copy and row sums
        HIPTS_TRY(statp.alloc((size_t)(Np / 256) * M * 8));
        g.pos = nullptr; g.ln_gamma = bias.as<float>(); g.stat_part = statp.as<float>(); g.stat_stride = M;
    }
    if (op8) { g.op8 = 1; g.f16 = 1; g.w_exp = 3; g.out8 = (epi == EPI_STAR && getenv("HIPTS_GEMM_OUT8")) ? 1 : 0; }
    if (dbg_f16) g.f16 = 1;
    if (getenv("HIPTS_DBG_X_BLOCKED")) g.x_blocked = 1;         // the fp32 stream as 16 x 16 blocks (timing only: the buffer is scratch here)
    if (getenv("HIPTS_DBG_GEMM_SHARED")) g.shared_chip = 1;      // as under sub-batch streams: 256-row tiles whatever the round count
    if (epi == EPI_QK && N % 3 == 0) { g.dim = N / 3; g.heads = g.dim / 64; g.out3_bf16 = obf2.as<bf16_t>(); }      // fused q | k | v (timing only: v shares k's buffer)
    DevBuf stamps;
    if (getenv("HIPTS_GEMM_STAMPS")) {
        HIPTS_TRY(stamps.alloc(4096 * 8 * 8));
        HIPTS_HIP(hipMemset(stamps.p, 0, 4096 * 8 * 8));
        g.stamps = stamps.as<unsigned long long>();
        g.trace = getenv("HIPTS_GEMM_TRACE") ? 1 : 0;
    }
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    DevBuf skws;
    if (getenv("HIPTS_DBG_GEMM_SK")) {
        HIPTS_TRY(skws.alloc(GEMM_SK_WS_BYTES));
        HIPTS_HIP(hipMemset(skws.p, 0, skws.bytes));
        g.sk_ws = skws.p; g.sk_ws_bytes = GEMM_SK_WS_BYTES; g.shared_chip = 1;
    }
    for (int i = 0; i < 3; ++i) HIPTS_TRY(launch_gemm((GemmEpilogue)epi, g, nullptr));
    HIPTS_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) HIPTS_TRY(launch_gemm((GemmEpilogue)epi, g, nullptr));
    HIPTS_HIP(hipEventRecord(e1, nullptr));
    HIPTS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    if (g.stamps) {
        // trace mode: the last launch's records
        unsigned long long h[8 * 64];
        HIPTS_HIP(hipMemcpy(h, g.stamps, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long base = ~0ull;
        for (int w = 0; w < 8; ++w) if (h[w * 64] && h[w * 64] < base) base = h[w * 64];
        if (gemm_q4_launch_count() > 0 && getenv("HIPTS_GEMM_Q4")) {
            // gemm4.hip's records of workgroup 8: per wave and tile [top, loop start, loop end, epilogue end, sum phase 0, sum wait + barrier,
            // sum phase 1, loop wall ticks of 10 ns]
            fprintf(stderr, "4-wave loop, workgroup 8: per tile start -> loop | loop (phase 0 | wait + barrier | phase 1) | epilogue, cycles\n");
            for (int w = 0; w < 4; ++w)
                for (int t = 0; t < 8; ++t) {
                    const unsigned long long* o = h + w * 64 + t * 8;
                    if (!o[0]) continue;
                    fprintf(stderr, "wave %d tile %d: top +%6lld  start %5lld  loop %7lld (p0 %7lld | wait %6lld | p1 %7lld)  epilogue %6lld  clock %.2f GHz\n", w, t,
                            (long long)(o[0] - base), (long long)(o[1] - o[0]), (long long)(o[2] - o[1]), (long long)o[4], (long long)o[5], (long long)o[6],
                            (long long)(o[3] - o[2]), o[7] ? (o[2] - o[1]) / (o[7] * 10.0) : 0.0);
                }
        } else
        if (getenv("HIPTS_GEMM_TRACE")) {
            std::vector<unsigned long long> tr(4096 * 8);
            HIPTS_HIP(hipMemcpy(tr.data(), g.stamps, tr.size() * 8, hipMemcpyDeviceToHost));
            FILE* f = fopen(getenv("HIPTS_GEMM_TRACE"), "w");
            if (f) {
                for (int b = 0; b < 4096; ++b)
                    if (tr[b * 8 + 2])
                        fprintf(f, "%d %llu %llu %llu %llu %llu\n", b, tr[b * 8], tr[b * 8 + 1], tr[b * 8 + 2], tr[b * 8 + 3], tr[b * 8 + 4]);
                fclose(f);
            }
        } else
        if (getenv("HIPTS_GEMM") && strcmp(getenv("HIPTS_GEMM"), "dw") == 0) {
            for (int w = 0; w < 4; ++w)
                fprintf(stderr, "dw wave %d: loop %lld cycles, waitcnt %lld, barrier %lld, wall %lld x 10 ns -> %.2f GHz\n", w, (long long)h[w * 64],
                        (long long)h[w * 64 + 1], (long long)h[w * 64 + 2], (long long)h[w * 64 + 3], h[w * 64] / (h[w * 64 + 3] * 10.0));
        }
        const bool quiet = strcmp(getenv("HIPTS_GEMM_STAMPS"), "quiet") == 0 || getenv("HIPTS_GEMM_TRACE") ||
                           (getenv("HIPTS_GEMM") && strcmp(getenv("HIPTS_GEMM"), "dw") == 0);
        if (h[2] && h[7] > h[6]) g_last_loop_ghz = (float)((h[3] - h[2]) / ((h[7] - h[6]) * 10.0));      // wave 0, tile 0
        if (!quiet) fprintf(stderr, "stamps of workgroup 8 (cycles since first): per tile [top, K-tile 0 landed, loop start, loop end, next prologue issued, epilogue issued]\n");
        for (int w = 0; w < 8 && !quiet; w += 4) {
            for (int t = 0; t < 8; ++t) {
                if (!h[w * 64 + t * 8]) continue;
                fprintf(stderr, "wave %d tile %d:", w, t);
                for (int i = 0; i < 6; ++i) fprintf(stderr, " %7lld", (long long)(h[w * 64 + t * 8 + i] - base));
                fprintf(stderr, "  loop clock %.2f GHz", (h[w * 64 + t * 8 + 3] - h[w * 64 + t * 8 + 2]) / ((h[w * 64 + t * 8 + 7] - h[w * 64 + t * 8 + 6]) * 10.0));
                fprintf(stderr, "\n");
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return HIPTS_OK;
}

// Development/test aid: the same GEMM launch through gemm_pp_kernel (8 waves) and through gemm4.hip's 4-wave loop on identical random
// operands; every output buffer of the two must agree byte for byte (the MFMA chain of an element runs over K in the same order).
// epi: EPI_GELU (4), EPI_QK (1; N % 3 == 0: the fused q | k | v launch), EPI_RESID_XG (13).  *mismatch_out = number of differing bytes;
// HIPTS_ERR_INVALID when the launch is not one gemm4.hip is built for.  ms_out[0 / 1]: average device time of `iters` launches each way.
extern "C" int hiptsdbg_gemm_q4_compare(int M, int N, int K, int epi, int f16, int iters, long long* mismatch_out, float* ms_out) {
    HIPTS_TRY(use_device(0));
    HIPTS_REQUIRE(mismatch_out && ms_out && (epi == EPI_GELU || epi == EPI_QK || epi == EPI_RESID_XG), "hiptsdbg_gemm_q4_compare: epilogue %d", epi);
    HIPTS_REQUIRE(M % 256 == 0 && N % 256 == 0 && K % 128 == 0 && M % 784 == 0, "hiptsdbg_gemm_q4_compare: whole tiles, even K-tile count, whole images");
    DevBuf A, W, bias, gamma, x0, of32, obf[3], statp;
    HIPTS_TRY(A.alloc((size_t)M * K * 2));
    HIPTS_TRY(W.alloc((size_t)N * K * 2));
    HIPTS_TRY(bias.alloc((size_t)N * 4));
    HIPTS_TRY(gamma.alloc((size_t)N * 4));
    HIPTS_TRY(x0.alloc((size_t)M * N * 4));
    HIPTS_TRY(of32.alloc((size_t)M * N * 4));
    const size_t obytes = ((size_t)M / 784) * 832 * N * 2 + (1 << 20);
    for (auto& b : obf) HIPTS_TRY(b.alloc(obytes));
    HIPTS_TRY(statp.alloc((size_t)(N / 256) * M * 8));
    {
        std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
        std::vector<float> hb((size_t)N), hg((size_t)N), hx((size_t)M * N);
        uint32_t st = 2468u + (uint32_t)(M + 3 * N + 7 * K);
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
        for (auto& v : ha) v = f16 ? f32_to_f16_rne(rnd()) : f32_to_bf16_rne(rnd());
        for (auto& v : hw) v = f16 ? f32_to_f16_rne(rnd() * 0.05f) : f32_to_bf16_rne(rnd() * 0.05f);
        for (auto& v : hb) v = rnd() * 0.3f;
        for (auto& v : hg) v = 1.0f + rnd() * 0.2f;
        for (auto& v : hx) v = rnd() * 3.0f;
        HIPTS_TRY(upload(A.p, ha.data(), ha.size() * 2));
        HIPTS_TRY(upload(W.p, hw.data(), hw.size() * 2));
        HIPTS_TRY(upload(bias.p, hb.data(), hb.size() * 4));
        HIPTS_TRY(upload(gamma.p, hg.data(), hg.size() * 4));
        HIPTS_TRY(upload(x0.p, hx.data(), hx.size() * 4));
    }
    GemmArgs g{};
    g.A = A.as<bf16_t>(); g.W = W.as<bf16_t>(); g.M = M; g.N = N; g.K = K; g.bias = bias.as<float>(); g.f16 = f16 ? 1 : 0;
    g.out_f32 = of32.as<float>(); g.out_bf16 = obf[0].as<bf16_t>(); g.out2_bf16 = obf[1].as<bf16_t>();
    g.tokens = 784; g.tokens_pad = 832; g.qscale = 0.125f;
    if (epi == EPI_QK) {
        if (N % 3 == 0) { g.dim = N / 3; g.out3_bf16 = obf[2].as<bf16_t>(); } else g.dim = N / 2;
        g.heads = g.dim / 64;
        HIPTS_REQUIRE(g.dim % 64 == 0, "hiptsdbg_gemm_q4_compare: q | k | v widths must be multiples of 64");
    }
    if (epi == EPI_RESID_XG) { g.ln_gamma = gamma.as<float>(); g.stat_part = statp.as<float>(); g.stat_stride = M; }
    g.shared_chip = 1;      // as the forwards launch it (sub-batch streams): 256-row tiles whatever the round count
    const long long q4_before = gemm_q4_launch_count();
    std::vector<std::vector<char>> keep[2];
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    const unsigned mask_before = 0;
    (void)mask_before;
    for (int pass = 0; pass < 2; ++pass) {
        set_gemm_q4_mask(pass ? (1u << epi) : 0u);
        HIPTS_HIP(hipMemcpy(of32.p, x0.p, of32.bytes, hipMemcpyDeviceToDevice));
        for (auto& b : obf) HIPTS_HIP(hipMemset(b.p, 0, b.bytes));
        HIPTS_HIP(hipMemset(statp.p, 0, statp.bytes));
        HIPTS_TRY(launch_gemm((GemmEpilogue)epi, g, nullptr));
        HIPTS_HIP(hipDeviceSynchronize());
        DevBuf* outs[5] = {&of32, &obf[0], &obf[1], &obf[2], &statp};
        for (DevBuf* b : outs) {
            keep[pass].emplace_back(b->bytes);
            HIPTS_HIP(hipMemcpy(keep[pass].back().data(), b->p, b->bytes, hipMemcpyDeviceToHost));
        }
        // timing (the residual stream keeps accumulating: values do not matter here)
        HIPTS_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) HIPTS_TRY(launch_gemm((GemmEpilogue)epi, g, nullptr));
        HIPTS_HIP(hipEventRecord(e1, nullptr));
        HIPTS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms_out[pass] = iters > 0 ? ms / iters : 0.f;
    }
    set_gemm_q4_mask(getenv("HIPTS_GEMM_Q4") ? (unsigned)strtoul(getenv("HIPTS_GEMM_Q4"), nullptr, 0) : GEMM_Q4_DEFAULT_MASK);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPTS_REQUIRE(gemm_q4_launch_count() == q4_before + 1 + iters, "hiptsdbg_gemm_q4_compare: the launcher did not take the 4-wave loop for this shape");
    long long bad = 0;
    for (size_t b = 0; b < keep[0].size(); ++b)
        for (size_t i = 0; i < keep[0][b].size(); ++i) bad += keep[0][b][i] != keep[1][b][i];
    *mismatch_out = bad;
    return HIPTS_OK;
}

// Development/test aid (not part of the public ABI): C = A W^T through the production GEMM kernel
// (EPI_RESID with a zero residual and zero bias), fp32 result copied to the host.
extern "C" int hiptsdbg_gemm_run(int M, int N, int K, const uint16_t* a_bf16, const uint16_t* w_bf16, float* out_host) {
    HIPTS_TRY(use_device(0));
    HIPTS_REQUIRE(M >= 1 && N >= 16 && N % 16 == 0 && K >= 64 && K % 64 == 0, "hiptsdbg_gemm_run: unsupported shape");
    const int Np = round_up(N, 256);
    DevBuf A, W, bias, out;
    HIPTS_TRY(A.alloc((size_t)M * K * 2));
    HIPTS_TRY(W.alloc((size_t)Np * K * 2));
    HIPTS_TRY(bias.alloc((size_t)Np * 4));
    HIPTS_TRY(out.alloc((size_t)M * N * 4));
    HIPTS_HIP(hipMemset(W.p, 0, W.bytes));
    HIPTS_HIP(hipMemset(bias.p, 0, bias.bytes));
    HIPTS_HIP(hipMemset(out.p, 0, out.bytes));
    HIPTS_TRY(upload(A.p, a_bf16, (size_t)M * K * 2));
    HIPTS_TRY(upload(W.p, w_bf16, (size_t)N * K * 2));
    GemmArgs g{};
    g.A = A.as<bf16_t>(); g.W = W.as<bf16_t>(); g.M = M; g.N = N; g.K = K; g.bias = bias.as<float>(); g.out_f32 = out.as<float>();
    g.f16 = getenv("HIPTS_DBG_GEMM_F16") ? 1 : 0;       // the 16-bit patterns are IEEE half (tests/test_gpu_gemm.py: the 192-row tiles exist for half operands only)
    DevBuf skws;
    if (getenv("HIPTS_DBG_GEMM_SK")) {                  // the split-K tail as the forwards use it: a zeroed workspace, 256-row tiles (sub-batch streams)
        HIPTS_TRY(skws.alloc(GEMM_SK_WS_BYTES));
        HIPTS_HIP(hipMemset(skws.p, 0, skws.bytes));
        g.sk_ws = skws.p; g.sk_ws_bytes = GEMM_SK_WS_BYTES; g.shared_chip = 1;
    }
    HIPTS_TRY(launch_gemm(EPI_RESID, g, nullptr));
    if (skws.p) HIPTS_TRY(launch_gemm(EPI_RESID, g, nullptr));      // a second launch on the same workspace: the tickets must be back at zero
    if (skws.p) {                                                     // ... and are
        std::vector<unsigned> tk(1024);
        HIPTS_HIP(hipMemcpy(tk.data(), skws.p, 4096, hipMemcpyDeviceToHost));
        for (unsigned v : tk) HIPTS_REQUIRE(v == 0, "hiptsdbg_gemm_run: a split-K ticket was left at %u", v);
    }
    HIPTS_HIP(hipMemcpy(out_host, out.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}

// Test entry for the e4m3 operand path: quantises a (as is) and w (times the per-tensor power of two, returned in
// *w_exp) on the host, runs out = A W^T (kind 0: fp32 through EPI_RESID into zeros; kind 1: e4m3 bytes through the
// staged EPI_STAR epilogue with the identity activation) and returns the result.
extern "C" int hiptsdbg_gemm8_run(int M, int N, int K, const float* a_f32, const float* w_f32, int kind, void* out_host, int* w_exp) {
    HIPTS_TRY(use_device(0));
    HIPTS_REQUIRE(M >= 1 && N >= 16 && N % 16 == 0 && K >= 128 && K % 128 == 0 && w_exp, "hiptsdbg_gemm8_run: unsupported shape");
    const int Np = round_up(N, 256);
    DevBuf A, W, bias, out;
    HIPTS_TRY(A.alloc((size_t)M * K));
    std::vector<uint8_t> ha((size_t)M * K);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = f32_to_e4m3_rne(a_f32[i]);
    HIPTS_TRY(upload(A.p, ha.data(), ha.size()));
    HIPTS_TRY(upload_matrix8(W, w_f32, N, K, Np, w_exp));
    HIPTS_TRY(bias.alloc((size_t)Np * 4));
    HIPTS_TRY(out.alloc((size_t)M * N * 4));
    HIPTS_HIP(hipMemset(bias.p, 0, bias.bytes));
    HIPTS_HIP(hipMemset(out.p, 0, out.bytes));
    GemmArgs g{};
    g.A = A.as<bf16_t>(); g.W = W.as<bf16_t>(); g.M = M; g.N = N; g.K = K; g.bias = bias.as<float>();
    g.f16 = 1; g.op8 = 1; g.w_exp = *w_exp;
    if (kind == 0) {
        g.out_f32 = out.as<float>();
        HIPTS_TRY(launch_gemm(EPI_RESID, g, nullptr));
        HIPTS_HIP(hipMemcpy(out_host, out.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    } else {
        g.out_bf16 = out.as<bf16_t>(); g.out8 = 1; g.star_kind = 2;
        HIPTS_TRY(launch_gemm(EPI_STAR, g, nullptr));
        HIPTS_HIP(hipMemcpy(out_host, out.p, (size_t)M * N, hipMemcpyDeviceToHost));
    }
    return HIPTS_OK;
}

// Test entry (tests/test_gpu_attention.py): the attention kernel alone on caller-supplied operands.  q, k: 16-bit patterns
// [batch * heads][tokens_pad][head_dim] (q already scaled by head_dim^-0.5 * log2 e, padding rows zero), vT: [batch * heads][head_dim][tokens_pad];
// out: 16-bit patterns [batch][tokens][heads * head_dim].
extern "C" int hiptsdbg_attention_run(const uint16_t* q, const uint16_t* k, const uint16_t* vT, uint16_t* out_host, int batch, int heads, int tokens,
                                      int tokens_pad, int head_dim, int f16) {
    HIPTS_REQUIRE(q && k && vT && out_host && batch >= 1 && heads >= 1 && tokens >= 1, "hiptsdbg_attention_run: bad arguments");
    HIPTS_TRY(use_device(0));
    const size_t nqk = (size_t)batch * heads * tokens_pad * head_dim, nout = (size_t)batch * tokens * heads * head_dim;
    DevBuf dq, dk, dv, dout;
    HIPTS_TRY(dq.alloc(nqk * 2));
    HIPTS_TRY(dk.alloc(nqk * 2));
    HIPTS_TRY(dv.alloc(nqk * 2));
    HIPTS_TRY(dout.alloc(nout * 2));
    HIPTS_TRY(upload(dq.p, q, nqk * 2));
    HIPTS_TRY(upload(dk.p, k, nqk * 2));
    HIPTS_TRY(upload(dv.p, vT, nqk * 2));
    HIPTS_HIP(hipMemset(dout.p, 0, nout * 2));
    HIPTS_TRY(launch_attention(dq.as<bf16_t>(), dk.as<bf16_t>(), dv.as<bf16_t>(), dout.as<bf16_t>(), batch, heads, tokens, tokens_pad, f16 != 0, nullptr,
                               head_dim, 0));
    HIPTS_HIP(hipDeviceSynchronize());
    HIPTS_HIP(hipMemcpy(out_host, dout.p, nout * 2, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}

// Development aid: average device time of `iters` attention launches with the chip to itself (tools/attn_time.py).
extern "C" int hiptsdbg_attention_time(const uint16_t* q, const uint16_t* k, const uint16_t* vT, int batch, int heads, int tokens, int tokens_pad,
                                       int head_dim, int f16, int iters, double* avg_us) {
    HIPTS_REQUIRE(q && k && vT && avg_us && batch >= 1 && heads >= 1 && tokens >= 1 && iters >= 1, "hiptsdbg_attention_time: bad arguments");
    HIPTS_TRY(use_device(0));
    const size_t nqk = (size_t)batch * heads * tokens_pad * head_dim, nout = (size_t)batch * tokens * heads * head_dim;
    DevBuf dq, dk, dv, dout;
    HIPTS_TRY(dq.alloc(nqk * 2));
    HIPTS_TRY(dk.alloc(nqk * 2));
    HIPTS_TRY(dv.alloc(nqk * 2));
    HIPTS_TRY(dout.alloc(nout * 2));
    HIPTS_TRY(upload(dq.p, q, nqk * 2));
    HIPTS_TRY(upload(dk.p, k, nqk * 2));
    HIPTS_TRY(upload(dv.p, vT, nqk * 2));
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    int rc = HIPTS_OK;
    for (int i = 0; i < 3 && rc == HIPTS_OK; ++i)
        rc = launch_attention(dq.as<bf16_t>(), dk.as<bf16_t>(), dv.as<bf16_t>(), dout.as<bf16_t>(), batch, heads, tokens, tokens_pad, f16 != 0, nullptr, head_dim, 0);
    HIPTS_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters && rc == HIPTS_OK; ++i)
        rc = launch_attention(dq.as<bf16_t>(), dk.as<bf16_t>(), dv.as<bf16_t>(), dout.as<bf16_t>(), batch, heads, tokens, tokens_pad, f16 != 0, nullptr, head_dim, 0);
    HIPTS_HIP(hipEventRecord(e1, nullptr));
    HIPTS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = 1e3 * ms / iters;
    return rc;
}

// The head_dim-64 kernel of attn2.hip alone: q, k, v 16-bit patterns [batch * heads][tokens_pad][64] (v NOT transposed); iters == 0: one launch,
// result to out_host ([batch][tokens][heads * 64]); iters > 0: timing only (avg_us).  variant: launch_attention2's.
extern "C" int hiptsdbg_attention2(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out_host, int batch, int heads, int tokens,
                                   int tokens_pad, int f16, int variant, int iters, double* avg_us) {
    HIPTS_REQUIRE(q && k && v && batch >= 1 && heads >= 1 && tokens >= 1 && (iters > 0 ? avg_us != nullptr : out_host != nullptr), "hiptsdbg_attention2: bad arguments");
    HIPTS_TRY(use_device(0));
    const size_t nqk = (size_t)batch * heads * tokens_pad * 64, nout = (size_t)batch * tokens * heads * 64;
    DevBuf dq, dk, dv, dout;
    HIPTS_TRY(dq.alloc(nqk * 2));
    HIPTS_TRY(dk.alloc(nqk * 2));
    HIPTS_TRY(dv.alloc(nqk * 2));
    HIPTS_TRY(dout.alloc(nout * 2));
    HIPTS_TRY(upload(dq.p, q, nqk * 2));
    HIPTS_TRY(upload(dk.p, k, nqk * 2));
    HIPTS_TRY(upload(dv.p, v, nqk * 2));
    HIPTS_HIP(hipMemset(dout.p, 0, nout * 2));
    auto run = [&]() { return launch_attention2(dq.as<bf16_t>(), dk.as<bf16_t>(), dv.as<bf16_t>(), dout.as<bf16_t>(), batch, heads, tokens, tokens_pad, f16 != 0, nullptr, 0, variant); };
    if (iters <= 0) {
        HIPTS_TRY(run());
        HIPTS_HIP(hipDeviceSynchronize());
        HIPTS_HIP(hipMemcpy(out_host, dout.p, nout * 2, hipMemcpyDeviceToHost));
        return HIPTS_OK;
    }
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    int rc = HIPTS_OK;
    for (int i = 0; i < 3 && rc == HIPTS_OK; ++i) rc = run();
    HIPTS_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters && rc == HIPTS_OK; ++i) rc = run();
    HIPTS_HIP(hipEventRecord(e1, nullptr));
    HIPTS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = 1e3 * ms / iters;
    return rc;
}

extern "C" int hiptsdbg_attention2_stamps(unsigned long long* host, int n) { return attention2_read_stamps(host, n); }

// Development aid: copy one workspace buffer of the last forward to the host (tools/determinism.py).
extern "C" int hiptsdbg_vit_dump(hipts_vit_t* h, const char* name, void* out_host, size_t max_bytes, size_t* bytes) {
    HIPTS_REQUIRE(h && name && out_host && bytes, "null argument");
    HIPTS_TRY(use_device(h->device));
    const std::string n(name);
    const DevBuf* b = n == "a0" ? &h->a0 : n == "x" ? &h->x : n == "xn" ? &h->xn : n == "q" ? &h->q : n == "k" ? &h->k
                      : n == "v" ? &h->v : n == "att" ? &h->att : n == "hmid" ? &h->hmid : n == "pool_part" ? &h->pool_part
                      : n == "pooled2" ? &h->pooled2 : nullptr;
    HIPTS_REQUIRE(b, "unknown buffer %s", name);
    *bytes = b->bytes < max_bytes ? b->bytes : max_bytes;
    HIPTS_HIP(hipDeviceSynchronize());
    HIPTS_HIP(hipMemcpy(out_host, b->p, *bytes, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}
