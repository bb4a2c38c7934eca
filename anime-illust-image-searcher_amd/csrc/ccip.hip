// ccip.hip -- CCIP character-feature encoder (gen_cfeatures.py:112-118,133-159): a CAFormer (timm
// MetaFormer) forward behind hipts_ccip_*.
//
// Reference call being replaced: `self.embed_model.run(['output'], {'input': x})` with x float32
// [B,3,384,384] (gen_cfeatures.py:158).  The ONNX graph itself is not in /root/reference; the layer
// algebra below is timm 1.0.9 `models/metaformer.py` -- see oracle/ccip.py, which
// restates the same definition on the CPU and is what the parity tests compare against.
//
// Data layout: activations are token-major NHWC throughout -- the residual stream x is float32
// [B*H*W][C], GEMM operands are bf16 [rows][K] -- so every 1x1 convolution / Linear is the same
// persistent MFMA GEMM as in the ViT (gemm.hip) with a fused epilogue:
//   pwconv1 / fc1      -> EPI_STAR     bf16(s * relu(.)^2 + b)        (StarReLU)
//   pwconv2 / fc2 / proj -> EPI_RESID / EPI_RESCALE   x = rs * x + .  (float32 read-modify-write)
//   qkv                -> EPI_QK + EPI_VT with head_dim 32 layouts, then attn.hip<HD = 32>
//   stem 7x7 s4, downsample 3x3 s2 -> im2col kernel + GEMM with EPI_BIAS
// The stem's patch matrix is stored as bf16 hi | lo halves against [W | W] (K = 2 * 160), so the
// normalised pixels enter the MFMA with 16 mantissa bits (same device as the ViT float input path).
// Memory-bound pieces are plain HIP kernels here: depthwise 7x7 (NHWC, 8 channels = 16 B per thread),
// im2col gathers, bias-free LayerNorm, pooled LayerNorm head.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "vit_internal.h"

using namespace hipts;

namespace {

struct Block {
    bool attn = false;
    DevBuf n1, n2;                 // LayerNorm gammas (no beta)
    DevBuf w_in, w_out;            // SepConv: pwconv1 [2C,C], pwconv2 [C,2C]; attention: qkv [3C,C], proj [C,C]
    DevBuf dw;                     // SepConv: depthwise weights as [49][2C] float32
    DevBuf dwz;                    // ... and as the lane images of their Toeplitz operands, [2C][7][64] u32 (dwconv7_mfma_kernel)
    DevBuf fc1, fc2;               // [4C,C], [C,4C]
    DevBuf mlp_img;                // the two as the chunk images of the fused MLP kernel (mlp.hip; widths 128 / 256, half operands)
    std::vector<float> fc1_host, fc2_host;      // host copies (stages 0-1: 15 MB in all): the image is rebuilt whenever either tensor is set again
    DevBuf u_qk, u_v, u_fc1;       // attention blocks of the wide stages: W gamma per output column of q | k, v (norm1) and fc1 (norm2) -- the
                                   // LayerNorm folded into the consumer GEMM (round 5; EPI_RESID_XG prepares gamma * x and the row sums)
    DevBuf rs1, rs2;               // optional res_scale vectors
    bool has_rs1 = false, has_rs2 = false;
    float s1 = 1.f, b1 = 0.f;      // token-mixer StarReLU
    float s2 = 1.f, b2 = 0.f;      // MLP StarReLU
};

struct Stage {
    int C = 0, H = 0, T = 0, Tp = 0;       // width, spatial side, tokens = H*H, padded tokens
    DevBuf ds_norm, ds_w, ds_b;            // downsample (stage > 0): LN gamma [Cprev], conv as [C][9*Cprev], bias [C]
    std::vector<Block> blocks;
};

constexpr int STEM_KH = 160;               // 7*7*3 = 147 taps padded to 160; K = hi | lo = 320
constexpr int STEM_K = 2 * STEM_KH;

}  // namespace

struct hipts_ccip {
    int device = 0;
    hipts_ccip_config_t cfg{};
    Stage st[4];
    DevBuf stem_w, stem_b, stem_norm, head_g, head_b, zeros, lut;
    std::vector<std::string> missing;
    // workspace (sized for cfg.max_batch)
    DevBuf img_in, a0, x, xn, h1, h2, m1, col, q, k, vT, feat;
    DevBuf stat_part, fold_c;      // folded LayerNorms of the wide stages: per (256-column tile, row) partial (sum, sum of squares); scratch for W beta (unused: no beta)
    size_t pstat = 0;              // float2 per image of stat_part (largest wide stage)
    bool fold_dirty = true;        // a norm / qkv / fc1 tensor changed since the fold vectors were computed
    size_t px = 0, p2c = 0, p4c = 0, pcol = 0, pqk = 0;   // per-image element strides of the workspace buffers (largest stage)
    static constexpr int MAX_SUB = 4;
    hipStream_t sub[MAX_SUB] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SUB] = {};
    double flops_per_image = 0.0;
};

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Stem patch matrix.  A0[m][(ky*7 + kx)*3 + c] = hi, A0[m][160 + ...] = lo of the normalised pixel at
// (4 oy - 2 + ky, 4 ox - 2 + kx), zero outside the image (Conv2d padding = 2 pads the NORMALISED input).
// U8: images uint8 NHWC RGB; /255 in float32, (x - mean) / std in float64, cast (gen_cfeatures.py:100-110,156) --
// 3 x 256 possible values, tabulated once on the host with exactly that arithmetic.
// One thread per (token, ky): 21 values.
// ---------------------------------------------------------------------------------------------
template <bool U8, bool F16>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const void* __restrict__ img, const float* __restrict__ lut,
                                                          bf16_t* __restrict__ a0, int batch, int S, int H0) {
    // lut[c][u] = normalised value of byte u in channel c, built on the host with the reference's arithmetic
    __shared__ float slut[3 * 256];
    if constexpr (U8) {
        for (int i = threadIdx.x; i < 3 * 256; i += 256) slut[i] = lut[i];
        __syncthreads();
    }
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)batch * H0 * H0 * 7;
    if (idx >= total) return;
    const int ky = (int)(idx % 7);
    const int64_t m = idx / 7;
    const int ox = (int)(m % H0), oy = (int)((m / H0) % H0), b = (int)(m / ((int64_t)H0 * H0));
    const int iy = 4 * oy - 2 + ky;
    bf16_t* row = a0 + m * STEM_K + ky * 21;
    const bool iny = iy >= 0 && iy < S;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
        const int ix = 4 * ox - 2 + kx;
        const bool in = iny && ix >= 0 && ix < S;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = 0.f;
            if (in) {
                if constexpr (U8) v = slut[c * 256 + reinterpret_cast<const uint8_t*>(img)[(((int64_t)b * S + iy) * S + ix) * 3 + c]];
                else v = reinterpret_cast<const float*>(img)[(((int64_t)b * 3 + c) * S + iy) * S + ix];
            }
            const bf16_t hi = to_op<F16>(v);
            row[kx * 3 + c] = hi;
            row[STEM_KH + kx * 3 + c] = to_op<F16>(v - from_op<F16>(hi));
        }
    }
}

// The uint8 entry point's version: a workgroup owns 64 adjacent output pixels of one output row.  It loads
// the 7 x (64*4+3) x 3 input bytes they touch with coalesced byte loads into LDS, builds the 64 patch rows
// (hi | lo halves, zero pad columns included) in LDS through the normalisation table, and writes them out
// as ONE contiguous 40 KB block with 16 B stores (consecutive pixels are consecutive rows of A0).  The
// one-thread-per-kernel-row version above spends its time on scattered byte loads and 2-byte stores
// (628 us for 64 images).
template <bool F16>
__global__ __launch_bounds__(256) void stem_im2col_u8_kernel(const uint8_t* __restrict__ img, const float* __restrict__ lut,
                                                             bf16_t* __restrict__ a0, int S, int H0, int xtiles) {
    constexpr int TW = (64 * 4 + 3) * 3;        // 777 bytes per input row of the tile
    __shared__ float slut[3 * 256];
    __shared__ uint8_t tile[7][TW + 7];
    __shared__ __attribute__((aligned(16))) bf16_t orow[64][STEM_K];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int xt = bid % xtiles;
    bid /= xtiles;
    const int oy = bid % H0;
    const int64_t b = bid / H0;
    const int ox0 = xt * 64, ix0 = 4 * ox0 - 2, iy0 = 4 * oy - 2;
    for (int i = tid; i < 3 * 256; i += 256) slut[i] = lut[i];
    for (int i = tid; i < 7 * TW; i += 256) {
        const int ry = i / TW, c = i - ry * TW;
        const int iy = iy0 + ry, ix = ix0 + c / 3;
        tile[ry][c] = (iy >= 0 && iy < S && ix >= 0 && ix < S) ? img[((b * S + iy) * S + ix0) * 3 + c] : (uint8_t)0;
    }
    for (int i = tid; i < 64 * 2 * (STEM_KH - 147); i += 256) {        // the pad columns of both halves
        const int t = i / (2 * (STEM_KH - 147)), r = i - t * (2 * (STEM_KH - 147));
        const int half = r / (STEM_KH - 147), c = r - half * (STEM_KH - 147);
        orow[t][half * STEM_KH + 147 + c] = to_op<F16>(0.f);
    }
    __syncthreads();
    const int tok = tid & 63, part = tid >> 6;
    const int ox = ox0 + tok;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ky = part + 4 * kk;
        if (ky >= 7) break;
        const int iy = iy0 + ky;
        const bool iny = iy >= 0 && iy < S;
        bf16_t* row = &orow[tok][ky * 21];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const int ix = 4 * ox - 2 + kx;
            const bool in = iny && ix >= 0 && ix < S;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = in ? slut[c * 256 + tile[ky][(tok * 4 + kx) * 3 + c]] : 0.f;
                const bf16_t hi = to_op<F16>(v);
                row[kx * 3 + c] = hi;
                row[STEM_KH + kx * 3 + c] = to_op<F16>(v - from_op<F16>(hi));
            }
        }
    }
    __syncthreads();
    const int ntok = H0 - ox0 < 64 ? H0 - ox0 : 64;
    const int64_t m0 = (b * H0 + oy) * H0 + ox0;
    const uint4* src = reinterpret_cast<const uint4*>(&orow[0][0]);
    uint4* dst = reinterpret_cast<uint4*>(a0 + m0 * STEM_K);
    for (int i = tid; i < ntok * (STEM_K * 2 / 16); i += 256) dst[i] = src[i];
}

// Bias-free LayerNorm of float32 rows, in place (the stem's norm: its output IS the residual stream).
// float4 c of row `row` of the fp32 residual stream: row-major, or (blk) in 16 x 16 blocks of 1 KB (csrc/gemm_epi.h::x_off)
__device__ __forceinline__ float4* x_vec(float* x, int64_t row, int c, int D, int blk) {
    if (blk) return reinterpret_cast<float4*>(x + ((((row >> 4) * (int64_t)(D >> 4)) + (c >> 2)) << 8) + (row & 15) * 16 + (c & 3) * 4);
    return reinterpret_cast<float4*>(x + row * D) + c;
}

__global__ __launch_bounds__(256) void ln_inplace_kernel(float* __restrict__ x, const float* __restrict__ g, int64_t rows, int D,
                                                         float eps, int blk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = D >> 2;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nvec ? *x_vec(x, row, c, D, blk) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum_f(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    const float rstd = 1.0f / sqrtf(wave_sum_f(ss) / (float)D + eps);
    const float4* gr = reinterpret_cast<const float4*>(g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float4 gg = gr[c];
            *x_vec(x, row, c, D, blk) = make_float4((v[i].x - mean) * rstd * gg.x, (v[i].y - mean) * rstd * gg.y, (v[i].z - mean) * rstd * gg.z,
                                (v[i].w - mean) * rstd * gg.w);
        }
    }
}

// vit.hip's layernorm_kernel (bias-free, 16-bit output, same operations in the same order) reading the BLOCKED residual stream: the
// downsample norms of the stages whose stream is stored in 16 x 16 blocks (two launches per forward)
template <bool F16>
__global__ __launch_bounds__(256) void layernorm_blk_kernel(float* __restrict__ x, const float* __restrict__ g, bf16_t* __restrict__ out,
                                                            int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = D >> 2;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nvec ? *x_vec(x, row, c, D, 1) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum_f(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane + 64 * i < nvec) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum_f(ss) / (float)D + eps);
    const float4* gr = reinterpret_cast<const float4*>(g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            const float4 gg = gr[c];
            *reinterpret_cast<bf16x4*>(out + row * D + 4 * c) =
                pack4<F16>((v[i].x - mean) * rstd * gg.x + 0.f, (v[i].y - mean) * rstd * gg.y + 0.f, (v[i].z - mean) * rstd * gg.z + 0.f,
                           (v[i].w - mean) * rstd * gg.w + 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 7x7, padding 3, NHWC: out[b][y][x][c] = sum_{ky,kx} in[b][y+ky-3][x+kx-3][c] * w[ky*7+kx][c].
// 49 float32 FMAs per output element make this VALU-bound (not HBM-bound), so the kernel is built to
// spend its issue slots on FMAs: a workgroup stages an (8+6) x (16+6) pixel x 64 channel input tile and
// the 49 x 64 weights in LDS once (zero padded at the image border), and every thread produces 4
// horizontally adjacent pixels x 8 channels -- per kernel row it reads 10 input vectors for 4 x 7 taps
// (2.8x fewer LDS reads than one pixel per thread) and converts each bf16 input once.
// LDS pixel pitch 160 B: the 16 lanes of a ds_read_b128 phase (8 channel chunks x 2 pixel groups 4 pixels
// apart) fall on disjoint bank halves.  float32 accumulation in (ky, kx) order.
// ---------------------------------------------------------------------------------------------
constexpr int DW_TH = 8, DW_TW = 16, DW_CS = 64;
constexpr int DW_PH = DW_TH + 6, DW_PW = DW_TW + 6, DW_PITCH = 160;
constexpr int DW_IN_BYTES = DW_PH * DW_PW * DW_PITCH;          // 49,280 B
constexpr int DW_LDS_BYTES = DW_IN_BYTES + 49 * DW_CS * 4;     // + 12,544 B of weights

template <bool F16>
__global__ __launch_bounds__(256) void dwconv7_kernel(const bf16_t* __restrict__ in, const float* __restrict__ w,
                                                      bf16_t* __restrict__ out, int H, int C, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem + DW_IN_BYTES);
    const int tid = threadIdx.x;
    const int slabs = C / DW_CS;
    int bid = blockIdx.x;
    const int slab = bid % slabs;
    bid /= slabs;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int64_t img0 = (int64_t)(bid / tiles_y) * H * H;
    const int y0 = ty * DW_TH, x0t = tx * DW_TW, c0 = slab * DW_CS;

    // stage the input tile (zero outside the image) and the weight slab
    for (int i = tid; i < DW_PH * DW_PW * 8; i += 256) {
        const int g = i & 7, p = i >> 3;
        const int py = p / DW_PW, px = p - py * DW_PW;
        const int iy = y0 + py - 3, ix = x0t + px - 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (iy >= 0 && iy < H && ix >= 0 && ix < H) v = *reinterpret_cast<const uint4*>(in + ((img0 + (int64_t)iy * H + ix) * C + c0 + g * 8));
        *reinterpret_cast<uint4*>(smem + p * DW_PITCH + g * 16) = v;
    }
    for (int i = tid; i < 49 * DW_CS / 4; i += 256) {
        const int tap = i / (DW_CS / 4), q = i - tap * (DW_CS / 4);
        *reinterpret_cast<float4*>(wl + tap * DW_CS + q * 4) = *reinterpret_cast<const float4*>(w + (size_t)tap * C + c0 + q * 4);
    }
    __syncthreads();

    const int g = tid & 7, pt = tid >> 3;
    const int r = pt >> 2, xo = (pt & 3) * 4;            // output row in the tile, first of 4 output columns
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        float wv[7][8];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float4 a = *reinterpret_cast<const float4*>(wl + (ky * 7 + kx) * DW_CS + g * 8);
            const float4 b = *reinterpret_cast<const float4*>(wl + (ky * 7 + kx) * DW_CS + g * 8 + 4);
            wv[kx][0] = a.x; wv[kx][1] = a.y; wv[kx][2] = a.z; wv[kx][3] = a.w;
            wv[kx][4] = b.x; wv[kx][5] = b.y; wv[kx][6] = b.z; wv[kx][7] = b.w;
        }
        const char* rowp = smem + ((r + ky) * DW_PW + xo) * DW_PITCH + g * 16;
#pragma unroll
        for (int j = 0; j < 10; ++j) {                    // input column xo + j feeds output o = j - kx, kx = 0..6
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(rowp + j * DW_PITCH);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = from_op<F16>(v[e]);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int kx = j - o;
                if (kx < 0 || kx > 6) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(f[e], wv[kx][e], acc[o][e]);
            }
        }
    }
    const int oy = y0 + r;
    if (oy < H) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int ox = x0t + xo + o;
            if (ox >= H) continue;
            bf16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = to_op<F16>(acc[o][e]);
            *reinterpret_cast<bf16x8*>(out + ((img0 + (int64_t)oy * H + ox) * C + c0 + g * 8)) = ov;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same depthwise 7x7 on the matrix cores (round 4; IEEE-half operands, spatial side >= 16).
//
// For ONE channel and ONE kernel row ky, 16 output rows x 16 output columns are a matrix product:
//     out[y][x] += sum_k in[y + ky - 3][x0 - 4 + k] * T_ky[k][x],   T_ky[k][x] = w[ky][k - x - 1] (zero unless 0 <= k - x - 1 < 7)
// -- a banded Toeplitz matrix of the 7 weights of that kernel row, K = 32 input columns (23 of them under the band).  Seven
// v_mfma_f32_16x16x32_f16 per (channel, 16 x 16 outputs): 224 multiply-adds issued per output instead of 49, on units 32 times as
// fast as the fp32 FMAs the VALU kernel is bound by (229 us per stage-0 launch at batch 32 against 55 us of bytes).
//   * a workgroup owns CS channels and walks `tpw` tiles of 16 x 16 XT outputs.  The Toeplitz operands of its channels are built ONCE,
//     into registers: lane (x, q) needs w[ky][8 q + i - x - 1], i = 0..7 -- a window of the zero-padded weight row whose offset depends
//     on the lane only.  The padded row lies across the lanes of one register per (channel, ky), once with even and once with odd
//     alignment (table built at upload, 256 B per (channel, ky)), and four ds_bpermute_b32 gather a lane's window.  (Built per tile, the
//     gathers took as much of the LDS pipeline as the operand reads: 125 us per stage-0 launch.)
//   * input tile: 22 rows x (16 XT + 8) columns x CS channels, staged CHANNEL-MAJOR in LDS (a plane of [row][column] halves per
//     channel; global memory is pixel-major), two pixels of one channel per 32-bit write.  The loads of the NEXT tile are requested
//     before the products of this one and land in registers meanwhile.  Row pitch 96 / 160 bytes: the lane groups of a ds_read_b128
//     (MI355X_MICROARCH.md, LDS) fall on disjoint banks; the columns the band never reaches hold zeros (finite);
//   * the input operand of (channel, x-tile t, ky): lane (m, q) reads the 8 halves at row m + ky, column 16 t + 8 q -- one ds_read_b128;
//   * fp32 accumulators leave through LDS (the planes are dead by then) as pixel-major rows, 16-byte stores;
//   * workgroup ids equal mod 8 share an XCD and its L2: each XCD walks a contiguous eighth of (tile group, channel slab), so the
//     channel slabs that split a pixel's 128-byte lines and the tiles that share halo rows meet in ONE L2.
// The weights are rounded to half (the VALU kernel multiplies by the float32 weights): tests/test_gpu_ccip.py bounds the feature error of
// the whole encoder and checks this kernel alone against a float64 convolution; operand_f16 = 0 (bf16) keeps the VALU kernel.
// ---------------------------------------------------------------------------------------------
#ifndef HIPTS_DW_PREFETCH
#define HIPTS_DW_PREFETCH 0
#endif
// -DHIPTS_DW_STAMPS=<workgroup>: that workgroup's first wave leaves 100 MHz time stamps of its phases (tools/dwconv_stamps.py)
__device__ unsigned long long dw_stamps[32];
#ifdef HIPTS_DW_STAMPS
#define DW_STAMP(i) do { if (blockIdx.x == HIPTS_DW_STAMPS && threadIdx.x == 0 && (i) < 32) dw_stamps[i] = wall_clock64(); } while (0)
#else
#define DW_STAMP(i) do { } while (0)
#endif
template <int XT, int CS>
struct DwMfma {
    static constexpr int ROWS = 22;
    static constexpr int PB = XT == 3 ? 160 : 96;            // bytes per row of a channel plane (>= (16 XT + 16) halves)
    static constexpr int PLANE = ROWS * PB + 16;             // + 16: the eight-channel groups of the staging writes start on different banks
    static constexpr int PAIRS = 8 * XT + 4;                 // real column pairs per row (columns x0 - 4 .. x0 + 16 XT + 3)
    static constexpr int NP = PAIRS + 4;                     // + the zero columns the last x-tile's K range ends in
    static constexpr int OP = CS * 2 + 16;                   // pixel pitch of the output image
    static constexpr int LDS = CS * PLANE;
    static_assert(XT >= 1 && XT <= 3 && (16 * XT + 16) * 2 <= PB, "row pitch");
    static_assert(256 * XT * OP <= LDS, "the output image reuses the planes");
};

template <int XT, int CS>
__global__ __launch_bounds__(256, 2) void dwconv7_mfma_kernel(const bf16_t* __restrict__ in, const uint32_t* __restrict__ tz,
                                                           bf16_t* __restrict__ out, int H, int C, int tiles_x, int tiles_y, int ntiles,
                                                           int tpw, int xcd_chunk) {
    using L = DwMfma<XT, CS>;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slabs = C / CS;
    int bid = blockIdx.x;
    if (xcd_chunk) bid = (bid & 7) * xcd_chunk + (bid >> 3);
    const int c0 = (bid % slabs) * CS;
    const int t_begin = (bid / slabs) * tpw, t_end = min(t_begin + tpw, ntiles);
    constexpr int CW = CS / 4;                               // channels of a wave
    constexpr int G = CS / 8;                                // 16-byte channel groups of a pixel
    constexpr int ITEMS = L::ROWS * L::NP * G, NIT = (ITEMS + 255) / 256;
    const int m = lane & 15, kq = lane >> 4;

    DW_STAMP(0);
    // ---- Toeplitz operands of this wave's channels
    f16x8 band[CW][7];
    {
        const int shift = 8 * kq - m + 15;                       // first half of this lane's window in the padded weight row (0 .. 39)
        const int baddr = 4 * ((shift & 1) * 32 + (shift >> 1)); // lanes 0..23: even alignment, lanes 32..55: odd alignment
#pragma unroll
        for (int cc = 0; cc < CW; ++cc)
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const uint32_t sv = tz[((size_t)(c0 + wave * CW + cc) * 7 + ky) * 64 + lane];
                u32x4 d;
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(baddr + 4 * j, (int)sv);
                band[cc][ky] = __builtin_bit_cast(f16x8, d);
            }
    }

    // ---- staging: item = (row r, column pair pr, channel group g); every load of a tile is requested in one go, with clamped
    // addresses and no branches (a predicated load is a basic block of its own and the compiler drains the queue in front of it)
    uint4 v0[NIT], v1[NIT];
    uint32_t okmask = 0;
    auto request = [&](int tile) {
        const int tx = tile % tiles_x, q = tile / tiles_x;
        const int y0 = (q % tiles_y) * 16, x0 = tx * 16 * XT;
        const bf16_t* img = in + ((int64_t)(q / tiles_y) * H * H) * C + c0;
        okmask = 0;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = tid + u * 256;
            const int g = it % G, pr = (it / G) % L::NP, r = it / (G * L::NP);
            const int iy = y0 - 3 + r, ix = x0 - 4 + 2 * pr;
            const bool rowok = it < ITEMS && pr < L::PAIRS && iy >= 0 && iy < H;
            const bool ok0 = rowok && ix >= 0 && ix < H, ok1 = rowok && ix + 1 >= 0 && ix + 1 < H;
            const int iyc = min(max(iy, 0), H - 1), ix0 = min(max(ix, 0), H - 1), ix1 = min(max(ix + 1, 0), H - 1);
            // 32-bit byte offsets from the image's (scalar) base: an image is far below 4 GB, and the loads take the base from SGPRs
            // (the review's item 5c: the 64-bit address arithmetic was a fifth of the request phase's instructions)
            const char* imgb = reinterpret_cast<const char*>(img);
            v0[u] = *reinterpret_cast<const uint4*>(imgb + (unsigned)(((iyc * H + ix0) * C + g * 8) * 2));
            v1[u] = *reinterpret_cast<const uint4*>(imgb + (unsigned)(((iyc * H + ix1) * C + g * 8) * 2));
            okmask |= ((ok0 ? 1u : 0u) | (ok1 ? 2u : 0u)) << (2 * u);
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int it = tid + u * 256;
            if (it >= ITEMS) break;
            const int g = it % G, pr = (it / G) % L::NP, r = it / (G * L::NP);
            char* dst = smem + (g * 8) * L::PLANE + r * L::PB + pr * 4;
            const uint32_t k0 = (okmask >> (2 * u)) & 1u ? 0xffffffffu : 0u, k1 = (okmask >> (2 * u + 1)) & 1u ? 0xffffffffu : 0u;
            const uint32_t a[4] = {v0[u].x & k0, v0[u].y & k0, v0[u].z & k0, v0[u].w & k0};
            const uint32_t b[4] = {v1[u].x & k1, v1[u].y & k1, v1[u].z & k1, v1[u].w & k1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<uint32_t*>(dst + (2 * k) * L::PLANE) = (a[k] & 0xffffu) | (b[k] << 16);
                *reinterpret_cast<uint32_t*>(dst + (2 * k + 1) * L::PLANE) = (a[k] >> 16) | (b[k] & 0xffff0000u);
            }
        }
    };

#if HIPTS_DW_PREFETCH
    if (t_begin < t_end) request(t_begin);
#endif
    DW_STAMP(1);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int sb = 2 + 8 * (tile - t_begin);     // stamps of this tile
        DW_STAMP(sb);
#if !HIPTS_DW_PREFETCH
        request(tile);      // (requested a tile ahead the registers do not fit two waves per SIMD: 75 spilled; the other workgroup of the CU fills the wait)
#endif
        DW_STAMP(sb + 1);
        deposit();
        DW_STAMP(sb + 2);
        __syncthreads();
        DW_STAMP(sb + 3);
        // products: the seven operand reads of the NEXT (channel, x-tile) are in flight under the seven MFMAs of this one (read one at a
        // time in front of its MFMA, every MFMA waited ~100 cycles for the LDS: 10.7 us per tile)
        f32x4 acc[CW][XT];
        f16x8 abuf[2][7];
        const char* plane0 = smem + (wave * CW) * L::PLANE + m * L::PB + kq * 16;
        auto fetch = [&](int buf, int idx) {
            const char* p = plane0 + (idx / XT) * L::PLANE + (idx % XT) * 32;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) abuf[buf][ky] = *reinterpret_cast<const f16x8*>(p + ky * L::PB);
        };
        fetch(0, 0);
#pragma unroll
        for (int idx = 0; idx < CW * XT; ++idx) {
            if (idx + 1 < CW * XT) fetch((idx + 1) & 1, idx + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
                v = __builtin_amdgcn_mfma_f32_16x16x32_f16(abuf[idx & 1][ky], band[idx / XT][ky], v, 0, 0, 0);      // rows 4 kq + i = output row, lane & 15 = output column
            acc[idx / XT][idx % XT] = v;
            __builtin_amdgcn_sched_barrier(0);
        }
#if HIPTS_DW_PREFETCH
        if (tile + 1 < t_end) request(tile + 1);                  // lands while this tile's results leave
#endif
        DW_STAMP(sb + 4);
        __syncthreads();                                          // every wave is done with the planes
        DW_STAMP(sb + 5);
#pragma unroll
        for (int cp = 0; cp < CW / 2; ++cp)
#pragma unroll
            for (int t = 0; t < XT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pixel = (4 * kq + i) * (16 * XT) + 16 * t + m;
                    const uint32_t lo = __builtin_bit_cast(uint16_t, (_Float16)acc[2 * cp][t][i]), hi = __builtin_bit_cast(uint16_t, (_Float16)acc[2 * cp + 1][t][i]);
                    *reinterpret_cast<uint32_t*>(smem + pixel * L::OP + (wave * CW + 2 * cp) * 2) = lo | (hi << 16);
                }
        __syncthreads();
        DW_STAMP(sb + 6);
        {
            const int tx = tile % tiles_x, q = tile / tiles_x;
            const int y0 = (q % tiles_y) * 16, x0 = tx * 16 * XT;
            bf16_t* img = out + ((int64_t)(q / tiles_y) * H * H) * C + c0;
            static_assert((256 * XT * G) % 256 == 0, "whole passes");
            if (y0 + 16 <= H && x0 + 16 * XT <= H) {              // uniform: a tile inside the image stores without predicates
#pragma unroll
                for (int u = 0; u < XT * G; ++u) {
                    const int it = tid + u * 256;
                    const int g = it % G, pixel = it / G;
                    const int oy = y0 + pixel / (16 * XT), ox = x0 + pixel % (16 * XT);
                    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(img) + (unsigned)(((oy * H + ox) * C + g * 8) * 2)) = *reinterpret_cast<const uint4*>(smem + pixel * L::OP + g * 16);
                }
            } else {
                for (int it = tid; it < 256 * XT * G; it += 256) {
                    const int g = it % G, pixel = it / G;
                    const int oy = y0 + pixel / (16 * XT), ox = x0 + pixel % (16 * XT);
                    if (oy < H && ox < H)
                        *reinterpret_cast<uint4*>(img + ((int64_t)oy * H + ox) * C + g * 8) = *reinterpret_cast<const uint4*>(smem + pixel * L::OP + g * 16);
                }
            }
        }
        DW_STAMP(sb + 7);
        __syncthreads();                                          // the image is rewritten by the next tile's planes
    }
    DW_STAMP(31);
}

template <int XT, int CS>
int launch_dwconv7_mfma_as(const bf16_t* in, const uint32_t* tz, bf16_t* out, int batch, int H, int C, hipStream_t s) {
    using L = DwMfma<XT, CS>;
    static PerDevice once;              // the attribute is per device
    int dev = 0;
    HIPTS_HIP(hipGetDevice(&dev));
    HIPTS_REQUIRE(dev >= 0 && dev < 64, "depthwise 7x7: device %d", dev);
    {
        std::lock_guard<std::mutex> lk(once.mu);
        if (!once.done(dev)) {
            HIPTS_HIP(hipFuncSetAttribute((const void*)dwconv7_mfma_kernel<XT, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS));
            once.mark(dev);
        }
    }
    const int tiles_x = (H + 16 * XT - 1) / (16 * XT), tiles_y = (H + 15) / 16;
    const int ntiles = batch * tiles_y * tiles_x, slabs = C / CS;
    // tiles per workgroup: about four workgroups per CU over the launch, at most 8 tiles each (the Toeplitz operands are built per workgroup)
    static const int tpw_env = getenv("HIPTS_CCIP_DW_TPW") ? atoi(getenv("HIPTS_CCIP_DW_TPW")) : 0;
    const int tpw = tpw_env > 0 ? tpw_env : std::min(8, std::max(1, (int)(((int64_t)ntiles * slabs + 512) / 1024)));
    const int grid = ((ntiles + tpw - 1) / tpw) * slabs;
    static const bool plain_order = getenv("HIPTS_CCIP_DW_PLAIN_ORDER") != nullptr;
    dwconv7_mfma_kernel<XT, CS><<<grid, 256, L::LDS, s>>>(in, tz, out, H, C, tiles_x, tiles_y, ntiles, tpw, (grid % 8 == 0 && !plain_order) ? grid / 8 : 0);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

// mode 1: 48-column tiles where they waste no more of the side than 32-column ones; 2 / 3: forced
int launch_dwconv7_mfma(const bf16_t* in, const uint32_t* tz, bf16_t* out, int batch, int H, int C, int mode, hipStream_t s) {
    HIPTS_REQUIRE(C % 16 == 0, "depthwise 7x7 (matrix cores): %d channels, must be a multiple of 16", C);
    const int waste32 = (H + 31) / 32 * 32 - H, waste48 = (H + 47) / 48 * 48 - H;
    const bool wide = mode == 3 || (mode != 2 && waste48 <= waste32);
    return wide ? launch_dwconv7_mfma_as<3, 16>(in, tz, out, batch, H, C, s) : launch_dwconv7_mfma_as<2, 16>(in, tz, out, batch, H, C, s);
}

// The Toeplitz operands of dwconv7_mfma_kernel from weights [channels][49]: per (channel, kernel row) the weight row rounded to half inside
// 48 zero halves, Z[16 + kx] = w[ky][kx], as 32-bit pairs across 64 lanes -- lane L < 24: (Z[2L], Z[2L+1]); lane 32 + L, L < 24: (Z[2L+1], Z[2L+2])
std::vector<uint32_t> dw_toeplitz_lanes(const float* w, int channels) {
    std::vector<uint32_t> tzv((size_t)channels * 7 * 64, 0u);
    for (int cc = 0; cc < channels; ++cc)
        for (int ky = 0; ky < 7; ++ky) {
            uint16_t Z[50] = {};
            for (int kx = 0; kx < 7; ++kx) Z[16 + kx] = f32_to_f16_rne(w[(size_t)cc * 49 + ky * 7 + kx]);
            uint32_t* row = tzv.data() + ((size_t)cc * 7 + ky) * 64;
            for (int l = 0; l < 24; ++l) {
                row[l] = (uint32_t)Z[2 * l] | ((uint32_t)Z[2 * l + 1] << 16);
                row[32 + l] = (uint32_t)Z[2 * l + 1] | ((uint32_t)Z[2 * l + 2] << 16);
            }
        }
    return tzv;
}

// Downsampling patch matrix: col[m'][(ky*3 + kx)*C + c] = xn[b][2 oy - 1 + ky][2 ox - 1 + kx][c] (zero outside).
// One thread = one 16 B chunk.
__global__ __launch_bounds__(256) void ds_im2col_kernel(const bf16_t* __restrict__ xn, bf16_t* __restrict__ col, int batch, int H,
                                                        int C) {
    const int Ho = H >> 1, cg = C >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)batch * Ho * Ho * 9 * cg;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    const int tap = (int)((idx / cg) % 9);
    const int64_t m = idx / ((int64_t)cg * 9);
    const int ox = (int)(m % Ho), oy = (int)((m / Ho) % Ho);
    const int64_t b = m / ((int64_t)Ho * Ho);
    const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (iy >= 0 && iy < H && ix >= 0 && ix < H) v = *reinterpret_cast<const uint4*>(xn + (((b * H + iy) * H + ix) * C + g * 8));
    *reinterpret_cast<uint4*>(col + (m * 9 + tap) * C + g * 8) = v;
}

// Head: out[b][:] = LN(mean over the T tokens of x[b])  (with bias).  One 1024-thread workgroup per image:
// four row groups x 256 channel threads sum a quarter of the tokens each (the loop is load-latency bound,
// so more rows in flight is what matters), partial sums meet in LDS, the first 256 threads normalise.
__global__ __launch_bounds__(1024) void pool_ln_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ bta, float* __restrict__ out, int T, int C, float eps, int blk) {
    __shared__ float part[4][1024];
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x & 255, rg = threadIdx.x >> 8;
    const float* xb = x + (int64_t)b * T * C;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = rg; r < T; r += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;
            if (c < C) m[u] += blk ? xb[((((int64_t)(r >> 4) * (C >> 4)) + (c >> 4)) << 8) + (r & 15) * 16 + (c & 15)] : xb[(int64_t)r * C + c];
        }
#pragma unroll
    for (int u = 0; u < 4; ++u) part[rg][tid + 256 * u] = m[u];
    __syncthreads();
    // (every thread keeps walking to the barriers; only row group 0 does the arithmetic)
    float s = 0.f;
    if (rg == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = tid + 256 * u;
            m[u] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / (float)T;
            if (c < C) s += m[u];
        }
        s = wave_sum_f(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
    }
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    __syncthreads();
    if (rg == 0) {
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + 256 * u < C) ss += (m[u] - mean) * (m[u] - mean);
        ss = wave_sum_f(ss);
        if ((tid & 63) == 0) red[tid >> 6] = ss;
    }
    __syncthreads();
    if (rg != 0) return;
    const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)C + eps);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + 256 * u;
        if (c < C) out[(int64_t)b * C + c] = (m[u] - mean) * rstd * g[c] + bta[c];
    }
}

int upload_f32(DevBuf& buf, const float* data, size_t n) {
    HIPTS_TRY(buf.alloc(n * 4));
    return upload(buf.p, data, n * 4);
}

// y = act(A W^T) helpers over the shared persistent GEMM
int gemm(GemmEpilogue epi, GemmArgs& g, hipStream_t s) { return launch_gemm(epi, g, s); }

// The whole kernel sequence for images [i0, i0 + nb) on stream s.  Every workspace buffer is carved per image
// with the stride of the LARGEST stage, so that sub-batches on different streams never share bytes even when
// they are in different stages at the same time.
int ccip_run_images(hipts_ccip* h, const void* in_dev, bool is_u8, int i0, int batch, float* f_dev, hipStream_t s, bool shared_chip) {
    const auto& c = h->cfg;
    const int S = c.image_size;
    const bool f16 = c.operand_f16 != 0;
    const float* zeros = h->zeros.as<float>();
    const size_t img_bytes = (size_t)S * S * 3 * (is_u8 ? 1 : 4);
    in_dev = (const char*)in_dev + (size_t)i0 * img_bytes;
    float* x = h->x.as<float>() + (size_t)i0 * h->px;
    bf16_t* xn = h->xn.as<bf16_t>() + (size_t)i0 * h->px;
    bf16_t* h1 = h->h1.as<bf16_t>() + (size_t)i0 * h->p2c;
    bf16_t* h2 = h->h2.as<bf16_t>() + (size_t)i0 * h->p2c;
    bf16_t* m1 = h->m1.as<bf16_t>() + (size_t)i0 * h->p4c;
    bf16_t* col = h->col.as<bf16_t>() + (size_t)i0 * h->pcol;
    bf16_t* qb = h->q.as<bf16_t>() + (size_t)i0 * h->pqk;
    bf16_t* kb = h->k.as<bf16_t>() + (size_t)i0 * h->pqk;
    bf16_t* vb = h->vT.as<bf16_t>() + (size_t)i0 * h->pqk;
    bf16_t* a0 = h->a0.as<bf16_t>() + (size_t)i0 * h->st[0].T * STEM_K;
    GemmArgs g;
    bool xn_ready = false;          // xn already holds the LayerNorm the next consumer needs
    // x = rs * x + A W^T, optionally followed in the same epilogue by xn = LN(x) * gamma
    // (The e4m3 operand mode of rounds 1-3 -- operand_f16 = 2, BASELINE.json configs[4]'s "fp8 MFMA" -- was withdrawn in round 4:
    // hipts_ccip_create refuses it.  DESIGN.md section 6 has the numbers: cosine 0.968 against the float32 oracle, 1 % slower than half
    // operands, and no scaling scheme the MFMA offers lifts e4m3's three mantissa bits above 0.995 through 36 blocks.)
    // depthwise 7x7: 0 = the VALU kernel, 1 = matrix cores with the tile shape chosen by the spatial side, 2 / 3 = 32- / 48-column tiles forced
    static const int dw_mfma = getenv("HIPTS_CCIP_DW_MFMA") ? atoi(getenv("HIPTS_CCIP_DW_MFMA")) : 1;
    static const bool fused_mlp = !(getenv("HIPTS_CCIP_FUSED_MLP") && atoi(getenv("HIPTS_CCIP_FUSED_MLP")) == 0);      // A/B: 0 = fc1 and fc2 as two GEMM launches
    // The fp32 residual stream of the narrow stages (C <= 256: SepConv blocks, fused MLP) in 16 x 16 blocks of 1 KB (round 5,
    // csrc/gemm_epi.h::x_off; the ViT forward has the measurements): the loads / stores of the residual epilogues and of the fused MLP
    // are lane = (row, 4 columns of a 16-column block) -- one contiguous kilobyte per instruction instead of sixteen half lines.  Every
    // stage's stream is produced afresh by its stem / downsample GEMM, so the layout is a per-stage choice: blocked where the readers are
    // the epilogues and the fused MLP (stages 0-1; their three elementwise readers -- stem norm, two downsample norms -- read blocks),
    // row-major for the wide stages (LayerNorm kernels, pool).  HIPTS_CCIP_X_BLOCKED=0: row-major everywhere (A/B).
    static const bool xblk_env = !(getenv("HIPTS_CCIP_X_BLOCKED") && atoi(getenv("HIPTS_CCIP_X_BLOCKED")) == 0);
    auto stage_blocked = [&](int si) {
        const Stage& Sg = h->st[si];
        return xblk_env && Sg.C <= 256 && Sg.C % 16 == 0 && Sg.T % 16 == 0;
    };
    bool xblk = false;          // layout of x right now (the stage being processed)
    auto layernorm_xn = [&](const float* gamma, int64_t rows, int D) -> int {
        if (xblk) {
            if (f16) layernorm_blk_kernel<true><<<ceil_div(rows, 4), 256, 0, s>>>(x, gamma, xn, rows, D, c.ln_eps);
            else layernorm_blk_kernel<false><<<ceil_div(rows, 4), 256, 0, s>>>(x, gamma, xn, rows, D, c.ln_eps);
            HIPTS_LAUNCH_CHECK();
            return HIPTS_OK;
        }
        return launch_layernorm(x, gamma, nullptr, xn, rows, D, c.ln_eps, f16, s);
    };
    auto residual = [&](GemmArgs& ga, const float* rs, const float* gamma) -> int {
        ga.res_scale = rs;
        ga.x_blocked = xblk ? 1 : 0;
        if (gamma) {
            ga.ln_gamma = gamma;
            ga.ln_eps = c.ln_eps;
            ga.out_bf16 = xn;
            return launch_gemm(EPI_RESID_LN, ga, s);
        }
        return launch_gemm(rs ? EPI_RESCALE : EPI_RESID, ga, s);
    };

    // ---- stem: conv 7x7 s4 p2 (+bias) -> bias-free LN = residual stream of stage 0
    {
        const Stage& S0 = h->st[0];
        const int64_t M = (int64_t)batch * S0.T;
        const int blocks = ceil_div(M * 7, 256);
        if (is_u8) {
            const int xtiles = ceil_div(S0.H, 64);
            const int grid_u8 = batch * S0.H * xtiles;
            if (f16) stem_im2col_u8_kernel<true><<<grid_u8, 256, 0, s>>>((const uint8_t*)in_dev, h->lut.as<float>(), a0, S, S0.H, xtiles);
            else stem_im2col_u8_kernel<false><<<grid_u8, 256, 0, s>>>((const uint8_t*)in_dev, h->lut.as<float>(), a0, S, S0.H, xtiles);
        } else {
            if (f16) stem_im2col_kernel<false, true><<<blocks, 256, 0, s>>>(in_dev, nullptr, a0, batch, S, S0.H);
            else stem_im2col_kernel<false, false><<<blocks, 256, 0, s>>>(in_dev, nullptr, a0, batch, S, S0.H);
        }
        HIPTS_LAUNCH_CHECK();
        g = GemmArgs{};
        g.f16 = f16;
        g.shared_chip = shared_chip;
        g.A = a0; g.W = h->stem_w.as<bf16_t>(); g.M = (int)M; g.N = S0.C; g.K = STEM_K;
        g.bias = h->stem_b.as<float>(); g.out_f32 = x;
        xblk = stage_blocked(0);
        g.x_blocked = xblk ? 1 : 0;
        HIPTS_TRY(gemm(EPI_BIAS, g, s));
        ln_inplace_kernel<<<ceil_div(M, 4), 256, 0, s>>>(x, h->stem_norm.as<float>(), M, S0.C, c.ln_eps, xblk ? 1 : 0);
        HIPTS_LAUNCH_CHECK();
    }

    for (int si = 0; si < 4; ++si) {
        Stage& St = h->st[si];
        const int C = St.C, H = St.H, T = St.T, Tp = St.Tp;
        const int M = batch * T;
        if (si > 0) {
            // downsample: LN(x) -> 3x3 s2 p1 conv (+bias) -> x
            const Stage& Pv = h->st[si - 1];
            if (!xn_ready) HIPTS_TRY(layernorm_xn(St.ds_norm.as<float>(), (int64_t)batch * Pv.T, Pv.C));
            xn_ready = false;
            const int64_t chunks = (int64_t)M * 9 * (Pv.C / 8);
            ds_im2col_kernel<<<ceil_div(chunks, 256), 256, 0, s>>>(xn, col, batch, Pv.H, Pv.C);
            HIPTS_LAUNCH_CHECK();
            g = GemmArgs{};
            g.f16 = f16;
            g.shared_chip = shared_chip;
            g.A = col; g.W = St.ds_w.as<bf16_t>(); g.M = M; g.N = C; g.K = 9 * Pv.C;
            g.bias = St.ds_b.as<float>(); g.out_f32 = x;
            xblk = stage_blocked(si);           // (the norm above read the previous stage's layout)
            g.x_blocked = xblk ? 1 : 0;
            HIPTS_TRY(gemm(EPI_BIAS, g, s));
        }
        if (si >= c.attn_from_stage && Tp != T) {
            // the padded token rows of this stage's q / k / v^T layouts must be zero (finite at least): the
            // same bytes held another stage's values before
            const size_t bytes = (size_t)batch * Tp * C * 2;
            HIPTS_HIP(hipMemsetAsync(qb, 0, bytes, s));
            HIPTS_HIP(hipMemsetAsync(kb, 0, bytes, s));
            HIPTS_HIP(hipMemsetAsync(vb, 0, bytes, s));
        }
        // A stage whose rows fit one 256-wide GEMM tile gets its LayerNorms from the epilogue of the residual GEMM
        // that produces the row (EPI_RESID_LN): the separate pass over the fp32 stream is the largest HBM
        // consumer of the wide early stages.
        const bool fuse_ln = C <= 256 && !getenv("HIPTS_CCIP_NO_LN_FUSION");
        // Wide stages (rows of more than one 256-column tile: C = 512, 768; round 5): the LayerNorms are folded as in the ViT -- the residual
        // GEMM that finishes a row (EPI_RESID_XG, here with the CAFormer's res_scale) also writes gamma * x as the consumer's 16-bit operand
        // and the row's partial sums; q | k, v and fc1 apply rstd / mean in their epilogues (col_u = W gamma; the CAFormer's norms have no
        // beta).  40 of a forward's 45 layernorm_kernel launches (5.2 % of its kernel time, each a pass over the fp32 stream) go; a stage's
        // first norm1 (its rows come from the downsample GEMM) and the downsample norms (their consumer is the im2col) stay.  Only when the
        // stage's residual launches are not the two-workgroups-per-CU kernel's anyway (that loop has no statistics epilogue): more tiles
        // than half the CUs.  HIPTS_CCIP_LN_FOLD=0: off (A/B).
        static const bool fold_env = !(getenv("HIPTS_CCIP_LN_FOLD") && atoi(getenv("HIPTS_CCIP_LN_FOLD")) == 0);
        const int cus_dev = current_device_cus(nullptr);
        const int sblocks = (C + 255) / 256;
        const bool fold23 = fold_env && !fuse_ln && si >= c.attn_from_stage && C % 256 == 0 && M % 256 == 0 && h->stat_part.p &&
                            (long)((M + 255) / 256) * sblocks * 4 > (long)cus_dev * 2;
        float* stat_p = fold23 ? h->stat_part.as<float>() + 2 * (size_t)i0 * h->pstat : nullptr;
        bool xn_folded = false;         // xn holds gamma * x + stat_p the row sums (not the LayerNorm itself)
        auto folded = [&](GemmArgs& ga, const float* u) {
            ga.stat_in = stat_p; ga.stat_in_blocks = sblocks; ga.stat_in_stride = M; ga.ln_dim = C; ga.ln_eps = c.ln_eps;
            ga.col_u = u; ga.bias = zeros;
        };
        auto residual_xg = [&](GemmArgs& ga, const float* rs, const float* gamma) -> int {
            ga.res_scale = rs; ga.ln_gamma = gamma; ga.ln_eps = c.ln_eps; ga.out_bf16 = xn; ga.stat_part = stat_p; ga.stat_stride = M;
            ga.x_blocked = xblk ? 1 : 0;
            return launch_gemm(EPI_RESID_XG, ga, s);
        };
        for (size_t bi = 0; bi < St.blocks.size(); ++bi) {
            Block& B = St.blocks[bi];
            const bool in_folded = xn_ready && xn_folded;
            if (!xn_ready) HIPTS_TRY(layernorm_xn(B.n1.as<float>(), M, C));
            xn_ready = false;
            xn_folded = false;
            // LayerNorm that follows this block's MLP: the next block's norm1, or the next stage's downsample norm
            const float* next_gamma = bi + 1 < St.blocks.size() ? St.blocks[bi + 1].n1.as<float>()
                                      : (si < 3 ? h->st[si + 1].ds_norm.as<float>() : nullptr);
            if (!B.attn) {
                // SepConv: 1x1 (C -> 2C) + StarReLU -> depthwise 7x7 -> 1x1 (2C -> C) + residual
                g = GemmArgs{};
                g.f16 = f16;
                g.shared_chip = shared_chip;
                g.A = xn; g.W = B.w_in.as<bf16_t>(); g.M = M; g.N = 2 * C; g.K = C; g.bias = zeros;
                g.out_bf16 = h1; g.star_scale = B.s1; g.star_bias = B.b1;
                HIPTS_TRY(gemm(EPI_STAR, g, s));
                const int tiles_x = ceil_div(H, DW_TW), tiles_y = ceil_div(H, DW_TH);
                const int dw_grid = batch * tiles_y * tiles_x * (2 * C / DW_CS);
                if (f16 && H >= 16 && dw_mfma && B.dwz.p) HIPTS_TRY(launch_dwconv7_mfma(h1, B.dwz.as<uint32_t>(), h2, batch, H, 2 * C, dw_mfma, s));
                else if (f16) dwconv7_kernel<true><<<dw_grid, 256, DW_LDS_BYTES, s>>>(h1, B.dw.as<float>(), h2, H, 2 * C, tiles_x, tiles_y);
                else dwconv7_kernel<false><<<dw_grid, 256, DW_LDS_BYTES, s>>>(h1, B.dw.as<float>(), h2, H, 2 * C, tiles_x, tiles_y);
                HIPTS_LAUNCH_CHECK();
                g = GemmArgs{};
                g.f16 = f16;
                g.shared_chip = shared_chip;
                g.A = h2; g.W = B.w_out.as<bf16_t>(); g.M = M; g.N = C; g.K = 2 * C; g.bias = zeros; g.out_f32 = x;
                HIPTS_TRY(residual(g, B.has_rs1 ? B.rs1.as<float>() : nullptr, fuse_ln ? B.n2.as<float>() : nullptr));
            } else {
                const int heads = C / c.head_dim;
                g = GemmArgs{};
                g.f16 = f16;
                g.shared_chip = shared_chip;
                g.A = xn; g.W = B.w_in.as<bf16_t>(); g.M = M; g.N = 2 * C; g.K = C; g.bias = zeros;
                g.out_bf16 = qb; g.out2_bf16 = kb;
                g.tokens = T; g.tokens_pad = Tp; g.heads = heads; g.dim = C; g.hd_log2 = 5;
                g.qscale = 0.17677669529663687f * 1.4426950408889634f;      // 32^-0.5 * log2(e): attention works in base 2
                if (in_folded) folded(g, B.u_qk.as<float>());
                HIPTS_TRY(gemm(EPI_QK, g, s));
                g = GemmArgs{};
                g.f16 = f16;
                g.shared_chip = shared_chip;
                g.A = xn; g.W = B.w_in.as<bf16_t>() + (size_t)2 * C * C; g.M = M; g.N = C; g.K = C; g.bias = zeros;
                g.out_bf16 = vb;
                g.tokens = T; g.tokens_pad = Tp; g.heads = heads; g.dim = C; g.hd_log2 = 5;
                if (in_folded) folded(g, B.u_v.as<float>());
                HIPTS_TRY(gemm(EPI_VT, g, s));
                HIPTS_TRY(launch_attention(qb, kb, vb, h1, batch, heads, T, Tp,
                                           f16, s, 32));
                g = GemmArgs{};
                g.f16 = f16;
                g.shared_chip = shared_chip;
                g.A = h1; g.W = B.w_out.as<bf16_t>(); g.M = M; g.N = C; g.K = C; g.bias = zeros; g.out_f32 = x;
                if (fold23) HIPTS_TRY(residual_xg(g, B.has_rs1 ? B.rs1.as<float>() : nullptr, B.n2.as<float>()));
                else HIPTS_TRY(residual(g, B.has_rs1 ? B.rs1.as<float>() : nullptr, fuse_ln ? B.n2.as<float>() : nullptr));
            }
            // MLP: fc1 + StarReLU, fc2 + residual
            const bool mlp_folded = fold23 && B.attn;
            if (!fuse_ln && !mlp_folded) HIPTS_TRY(layernorm_xn(B.n2.as<float>(), M, C));
            if (fused_mlp && fuse_ln && f16 && B.mlp_img.p) {
                // stages 0-1 (rows of <= 256 columns): one kernel, the hidden tensor stays in registers (mlp.hip)
                const bool fuse_next = next_gamma != nullptr;
                HIPTS_TRY(launch_mlp_fused(xn, B.mlp_img.p, x, B.has_rs2 ? B.rs2.as<float>() : nullptr, next_gamma, xn, M, C, B.s2, B.b2, c.ln_eps, s, 0,
                                           xblk ? 1 : 0));
                xn_ready = fuse_next;
                continue;
            }
            g = GemmArgs{};
            g.f16 = f16;
            g.shared_chip = shared_chip;
            g.A = xn; g.W = B.fc1.as<bf16_t>(); g.M = M; g.N = 4 * C; g.K = C; g.bias = zeros;
            g.out_bf16 = m1; g.star_scale = B.s2; g.star_bias = B.b2;
            if (mlp_folded) folded(g, B.u_fc1.as<float>());
            HIPTS_TRY(gemm(EPI_STAR, g, s));
            g = GemmArgs{};
            g.f16 = f16;
            g.shared_chip = shared_chip;
            g.A = m1; g.W = B.fc2.as<bf16_t>(); g.M = M; g.N = C; g.K = 4 * C; g.bias = zeros; g.out_f32 = x;
            const bool fuse_next = fuse_ln && next_gamma != nullptr;
            // folded: the next block's norm1 (same stage: an attention block too) is prepared here
            const bool xg_next = fold23 && bi + 1 < St.blocks.size() && St.blocks[bi + 1].attn;
            if (xg_next) HIPTS_TRY(residual_xg(g, B.has_rs2 ? B.rs2.as<float>() : nullptr, St.blocks[bi + 1].n1.as<float>()));
            else HIPTS_TRY(residual(g, B.has_rs2 ? B.rs2.as<float>() : nullptr, fuse_next ? next_gamma : nullptr));
            xn_ready = fuse_next || xg_next;
            xn_folded = xg_next;
        }
    }
    // ---- head: global average pool -> LayerNorm
    const Stage& L = h->st[3];
    pool_ln_kernel<<<batch, 1024, 0, s>>>(x, h->head_g.as<float>(), h->head_b.as<float>(), f_dev + (size_t)i0 * L.C, L.T, L.C, c.ln_eps, xblk ? 1 : 0);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

int ccip_forward_impl(hipts_ccip* h, const void* input, int in_memspace, bool is_u8, int batch, float* out, int out_memspace,
                      hipStream_t s) {
    HIPTS_REQUIRE(h && input && out && batch >= 1, "hipts_ccip_forward: bad arguments");
    HIPTS_REQUIRE(batch <= h->cfg.max_batch, "batch %d exceeds max_batch %d", batch, h->cfg.max_batch);
    if (!h->missing.empty())
        return set_error(HIPTS_ERR_STATE, "hipts_ccip_forward: %zu checkpoint tensors not set (first: %s)", h->missing.size(),
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
    float* f_dev = dev_out ? out : h->feat.as<float>();
    if (h->fold_dirty) {        // W gamma of the wide stages' q | k, v and fc1 (folded LayerNorms): once per checkpoint
        const bool f16w = c.operand_f16 != 0;
        for (int si = c.attn_from_stage; si < 4; ++si) {
            Stage& St = h->st[si];
            const int C = St.C;
            if (C <= 256 || C % 256) continue;
            HIPTS_TRY(h->fold_c.reserve((size_t)4 * C * 4));
            for (Block& B : St.blocks) {
                if (!B.attn) continue;
                HIPTS_TRY(B.u_qk.reserve((size_t)2 * C * 4));
                HIPTS_TRY(B.u_v.reserve((size_t)C * 4));
                HIPTS_TRY(B.u_fc1.reserve((size_t)4 * C * 4));
                HIPTS_TRY(launch_fold_ln(B.w_in.as<bf16_t>(), f16w, B.n1.as<float>(), nullptr, nullptr, B.u_qk.as<float>(), h->fold_c.as<float>(), 2 * C, C, s));
                HIPTS_TRY(launch_fold_ln(B.w_in.as<bf16_t>() + (size_t)2 * C * C, f16w, B.n1.as<float>(), nullptr, nullptr, B.u_v.as<float>(),
                                         h->fold_c.as<float>(), C, C, s));
                HIPTS_TRY(launch_fold_ln(B.fc1.as<bf16_t>(), f16w, B.n2.as<float>(), nullptr, nullptr, B.u_fc1.as<float>(), h->fold_c.as<float>(), 4 * C, C, s));
            }
        }
        h->fold_dirty = false;
    }
    // Two sub-batches on two internal streams (as in the ViT forward): the late stages have fewer output
    // tiles than the chip has CUs, and a kernel of one half fills the CUs the other half leaves idle.
    static const int want_streams = getenv("HIPTS_CCIP_STREAMS") ? atoi(getenv("HIPTS_CCIP_STREAMS")) : 2;
    // (measured, B36 @384: batch 64 2567 -> 2809 images/s; at the reference's batch of 20 the halves are too small
    // to gain and the doubled launch count costs, so small batches stay on the caller's stream)
    static const int min_sub = getenv("HIPTS_CCIP_MINSUB") && atoi(getenv("HIPTS_CCIP_MINSUB")) > 0 ? atoi(getenv("HIPTS_CCIP_MINSUB")) : 16;      // images per sub-batch needed to split
    const int ns = std::min({want_streams, (int)hipts_ccip::MAX_SUB, batch / min_sub});      // (round 5, HIPTS_CCIP_STREAMS = 2 / 3 / 4 at batch 64: 3608 / 3458 / 2844 images/s -- two it stays)
    if (ns >= 2) {
        if (!h->ev_fork) {
            HIPTS_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            for (int i = 0; i < hipts_ccip::MAX_SUB; ++i) {
                HIPTS_HIP(hipStreamCreateWithFlags(&h->sub[i], hipStreamNonBlocking));
                HIPTS_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
            }
        }
        HIPTS_HIP(hipEventRecord(h->ev_fork, s));
        for (int i = 0; i < ns; ++i) {
            const int i0 = (int)((int64_t)batch * i / ns), i1 = (int)((int64_t)batch * (i + 1) / ns);      // (two streams: the halves as before, the larger one first)
            const int a0 = ns == 2 ? (i ? (batch + 1) / 2 : 0) : i0, a1 = ns == 2 ? (i ? batch : (batch + 1) / 2) : i1;
            HIPTS_HIP(hipStreamWaitEvent(h->sub[i], h->ev_fork, 0));
            HIPTS_TRY(ccip_run_images(h, in_dev, is_u8, a0, a1 - a0, f_dev, h->sub[i], true));
            HIPTS_HIP(hipEventRecord(h->ev_join[i], h->sub[i]));
            HIPTS_HIP(hipStreamWaitEvent(s, h->ev_join[i], 0));
        }
    } else {
        HIPTS_TRY(ccip_run_images(h, in_dev, is_u8, 0, batch, f_dev, s, false));
    }
    if (!dev_out) {
        HIPTS_HIP(hipMemcpyAsync(out, f_dev, (size_t)batch * h->st[3].C * 4, hipMemcpyDeviceToHost, s));
        HIPTS_HIP(hipStreamSynchronize(s));
    }
    return HIPTS_OK;
}

}  // namespace

extern "C" {

int hipts_ccip_create(const hipts_ccip_config_t* cfg, int device, hipts_ccip_t** out) {
    HIPTS_REQUIRE(cfg && out, "hipts_ccip_create: null argument");
    HIPTS_REQUIRE(cfg->image_size >= 32 && cfg->image_size % 32 == 0, "image_size %d must be a multiple of 32", cfg->image_size);
    HIPTS_REQUIRE(cfg->head_dim == 32, "head_dim %d: only 32 is built", cfg->head_dim);
    HIPTS_REQUIRE(cfg->max_batch >= 1, "max_batch must be >= 1");
    HIPTS_REQUIRE(cfg->operand_f16 == 0 || cfg->operand_f16 == 1,
                  "operand_f16 = %d: 0 (bf16) or 1 (IEEE half); the e4m3 mode (2) was withdrawn in round 4 -- cosine 0.968 against the float32 "
                  "forward and no faster than half operands (DESIGN.md section 6)", cfg->operand_f16);
    HIPTS_REQUIRE(cfg->attn_from_stage >= 0 && cfg->attn_from_stage <= 4, "attn_from_stage must be 0 .. 4");
    for (int s = 0; s < 4; ++s) {
        HIPTS_REQUIRE(cfg->dims[s] >= 64 && cfg->dims[s] % 64 == 0 && cfg->dims[s] <= 1024, "dims[%d] = %d must be a multiple of 64, at most 1024",
                      s, cfg->dims[s]);
        HIPTS_REQUIRE(cfg->depths[s] >= 1, "depths[%d] must be >= 1", s);
    }
    HIPTS_TRY(use_device(device));
    auto* h = new hipts_ccip();
    h->device = device;
    h->cfg = *cfg;
    const int B = cfg->max_batch;
    size_t max_x = 0, max_2c = 0, max_4c = 0, max_col = 16, max_qk = 16;
    double flops = 0.0;
    int H = cfg->image_size / 4;
    flops += 2.0 * H * H * cfg->dims[0] * 147.0;
    for (int s = 0; s < 4; ++s) {
        Stage& St = h->st[s];
        if (s > 0) H /= 2;
        St.C = cfg->dims[s];
        St.H = H;
        St.T = H * H;
        St.Tp = round_up(St.T, 64);
        St.blocks.resize(cfg->depths[s]);
        const double T = St.T, C = St.C;
        if (s > 0) flops += 2.0 * T * C * 9.0 * cfg->dims[s - 1];
        for (int i = 0; i < cfg->depths[s]; ++i) {
            Block& Bk = St.blocks[i];
            Bk.attn = s >= cfg->attn_from_stage;
            if (Bk.attn) flops += 2.0 * T * 3 * C * C + 4.0 * T * T * C + 2.0 * T * C * C;
            else flops += 2.0 * T * 2 * C * C * 2 + 2.0 * 49.0 * T * 2 * C;
            flops += 2.0 * T * 4 * C * C * 2;
        }
        h->px = std::max(h->px, (size_t)St.T * St.C);
        if (St.C > 256 && St.C % 256 == 0) h->pstat = std::max(h->pstat, (size_t)(St.C / 256) * St.T);
        h->p2c = std::max(h->p2c, (size_t)St.T * 2 * St.C);
        h->p4c = std::max(h->p4c, (size_t)St.T * 4 * St.C);
        if (s > 0) h->pcol = std::max(h->pcol, (size_t)St.T * 9 * cfg->dims[s - 1]);
        if (s >= cfg->attn_from_stage) h->pqk = std::max(h->pqk, (size_t)St.Tp * St.C);
    }
    max_x = (size_t)B * h->px;
    max_2c = (size_t)B * h->p2c;
    max_4c = (size_t)B * h->p4c;
    max_col = std::max(max_col, (size_t)B * h->pcol);
    max_qk = std::max(max_qk, (size_t)B * h->pqk);
    h->flops_per_image = flops;
    int st = 0;
    std::vector<float> z(4096, 0.f);
    std::vector<float> lut(3 * 256);
    {
        const double mean[3] = {0.48145466, 0.4578275, 0.40821073}, stdv[3] = {0.26862954, 0.26130258, 0.27577711};   // gen_cfeatures.py:103-104
        for (int cc = 0; cc < 3; ++cc)
            for (int u = 0; u < 256; ++u) lut[cc * 256 + u] = (float)(((double)((float)u / 255.0f) - mean[cc]) / stdv[cc]);
    }
    if ((st = upload_f32(h->zeros, z.data(), z.size())) || (st = upload_f32(h->lut, lut.data(), lut.size())) || (st = h->a0.alloc((size_t)B * h->st[0].T * STEM_K * 2)) ||
        (st = h->x.alloc(max_x * 4)) || (st = h->xn.alloc(max_x * 2)) || (st = h->h1.alloc(max_2c * 2)) || (st = h->h2.alloc(max_2c * 2)) ||
        (st = h->m1.alloc(max_4c * 2)) || (st = h->col.alloc(max_col * 2)) || (st = h->q.alloc(max_qk * 2)) || (st = h->k.alloc(max_qk * 2)) ||
        (st = h->vT.alloc(max_qk * 2)) || (st = h->feat.alloc((size_t)B * cfg->dims[3] * 4)) ||
        (h->pstat && (st = h->stat_part.alloc((size_t)B * h->pstat * 8)))) {
        delete h;
        return st;
    }
    // pad columns of the stem patch matrix and the padded token rows of q / k / v^T stay zero for ever
    hipError_t e = hipMemset(h->a0.p, 0, h->a0.bytes);
    if (e == hipSuccess) e = hipMemset(h->q.p, 0, h->q.bytes);
    if (e == hipSuccess) e = hipMemset(h->k.p, 0, h->k.bytes);
    if (e == hipSuccess) e = hipMemset(h->vT.p, 0, h->vT.bytes);
    if (e != hipSuccess) {
        delete h;
        return set_error(HIPTS_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dwconv7_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dwconv7_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS_BYTES);
    if (e != hipSuccess) {
        delete h;
        return set_error(HIPTS_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    auto need = [&](const std::string& k) { h->missing.push_back(k); };
    need("stem.conv.weight"); need("stem.conv.bias"); need("stem.norm.weight"); need("head.norm.weight"); need("head.norm.bias");
    for (int s = 0; s < 4; ++s) {
        const std::string sp = "stages." + std::to_string(s) + ".";
        if (s > 0) { need(sp + "downsample.norm.weight"); need(sp + "downsample.conv.weight"); need(sp + "downsample.conv.bias"); }
        for (int i = 0; i < cfg->depths[s]; ++i) {
            const std::string p = sp + "blocks." + std::to_string(i) + ".";
            need(p + "norm1.weight"); need(p + "norm2.weight");
            if (s >= cfg->attn_from_stage) { need(p + "token_mixer.qkv.weight"); need(p + "token_mixer.proj.weight"); }
            else {
                need(p + "token_mixer.pwconv1.weight"); need(p + "token_mixer.act1.scale"); need(p + "token_mixer.act1.bias");
                need(p + "token_mixer.dwconv.weight"); need(p + "token_mixer.pwconv2.weight");
            }
            need(p + "mlp.fc1.weight"); need(p + "mlp.act.scale"); need(p + "mlp.act.bias"); need(p + "mlp.fc2.weight");
        }
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_ccip_destroy(hipts_ccip_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        for (int i = 0; i < hipts_ccip::MAX_SUB; ++i) {
            if (h->sub[i]) (void)hipStreamDestroy(h->sub[i]);
            if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
        }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_ccip_set_tensor(hipts_ccip_t* h, const char* key_c, const float* data, int64_t numel) {
    HIPTS_REQUIRE(h && key_c && data, "hipts_ccip_set_tensor: null argument");
    HIPTS_TRY(use_device(h->device));
    const std::string key(key_c);
    const bool f16 = h->cfg.operand_f16 != 0;
    int st = HIPTS_OK;
#define EXPECT(n)                                                                                                         \
    do {                                                                                                                  \
        if (numel != (int64_t)(n)) return set_error(HIPTS_ERR_INVALID, "tensor %s: %lld elements, expected %lld", key_c, (long long)numel, (long long)(n)); \
    } while (0)
    const int C0 = h->cfg.dims[0], C3 = h->cfg.dims[3];
    if (key == "stem.conv.weight") {
        EXPECT((int64_t)C0 * 147);
        // [n][c][ky][kx] -> [n][(ky*7 + kx)*3 + c], duplicated for the hi | lo halves of the patch matrix
        std::vector<float> w2((size_t)C0 * STEM_K, 0.f);
        for (int n = 0; n < C0; ++n)
            for (int c = 0; c < 3; ++c)
                for (int t = 0; t < 49; ++t) {
                    const float v = data[((size_t)n * 3 + c) * 49 + t];
                    w2[(size_t)n * STEM_K + t * 3 + c] = v;
                    w2[(size_t)n * STEM_K + STEM_KH + t * 3 + c] = v;
                }
        st = upload_matrix16(h->stem_w, w2.data(), C0, STEM_K, round_up(C0, 256), f16);
    } else if (key == "stem.conv.bias") { EXPECT(C0); st = upload_f32(h->stem_b, data, C0); }
    else if (key == "stem.norm.weight") { EXPECT(C0); st = upload_f32(h->stem_norm, data, C0); }
    else if (key == "head.norm.weight") { EXPECT(C3); st = upload_f32(h->head_g, data, C3); }
    else if (key == "head.norm.bias") { EXPECT(C3); st = upload_f32(h->head_b, data, C3); }
    else if (key.rfind("stages.", 0) == 0) {
        const size_t d1 = key.find('.', 7);
        if (d1 == std::string::npos) return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
        const int s = atoi(key.substr(7, d1 - 7).c_str());
        if (s < 0 || s > 3) return set_error(HIPTS_ERR_INVALID, "tensor %s: stage out of range", key_c);
        Stage& St = h->st[s];
        const int C = St.C;
        std::string sub = key.substr(d1 + 1);
        if (sub.rfind("downsample.", 0) == 0) {
            if (s == 0) return set_error(HIPTS_ERR_INVALID, "tensor %s: stage 0 has no downsample", key_c);
            const int Cp = h->cfg.dims[s - 1];
            if (sub == "downsample.norm.weight") { EXPECT(Cp); st = upload_f32(St.ds_norm, data, Cp); }
            else if (sub == "downsample.conv.bias") { EXPECT(C); st = upload_f32(St.ds_b, data, C); }
            else if (sub == "downsample.conv.weight") {
                EXPECT((int64_t)C * Cp * 9);
                std::vector<float> w2((size_t)C * 9 * Cp);          // [n][c][tap] -> [n][tap*Cp + c]
                for (int n = 0; n < C; ++n)
                    for (int cc = 0; cc < Cp; ++cc)
                        for (int t = 0; t < 9; ++t) w2[((size_t)n * 9 + t) * Cp + cc] = data[((size_t)n * Cp + cc) * 9 + t];
                st = upload_matrix16(St.ds_w, w2.data(), C, 9 * Cp, round_up(C, 256), f16);
            } else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
        } else if (sub.rfind("blocks.", 0) == 0) {
            const size_t d2 = sub.find('.', 7);
            if (d2 == std::string::npos) return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
            const int bi = atoi(sub.substr(7, d2 - 7).c_str());
            if (bi < 0 || bi >= (int)St.blocks.size()) return set_error(HIPTS_ERR_INVALID, "tensor %s: block out of range", key_c);
            Block& B = St.blocks[bi];
            const std::string t = sub.substr(d2 + 1);
            auto up = [&](DevBuf& buf, int rows, int cols, int rows_pad) -> int { return upload_matrix16(buf, data, rows, cols, rows_pad, f16); };
            if (t == "norm1.weight") { EXPECT(C); st = upload_f32(B.n1, data, C); }
            else if (t == "norm2.weight") { EXPECT(C); st = upload_f32(B.n2, data, C); }
            else if (t == "res_scale1.scale") { EXPECT(C); st = upload_f32(B.rs1, data, C); B.has_rs1 = true; }
            else if (t == "res_scale2.scale") { EXPECT(C); st = upload_f32(B.rs2, data, C); B.has_rs2 = true; }
            else if (t == "mlp.fc1.weight" || t == "mlp.fc2.weight") {
                EXPECT((int64_t)4 * C * C);
                const bool first = t == "mlp.fc1.weight";
                st = first ? up(B.fc1, 4 * C, C, round_up(4 * C, 256)) : up(B.fc2, C, 4 * C, round_up(C, 256));
                if (st == HIPTS_OK && f16 && mlp_fused_supports(C)) {
                    (first ? B.fc1_host : B.fc2_host).assign(data, data + (size_t)4 * C * C);
                    if (!B.fc1_host.empty() && !B.fc2_host.empty()) {
                        const std::vector<uint16_t> img = mlp_weight_image(B.fc1_host.data(), B.fc2_host.data(), C);
                        st = B.mlp_img.alloc(img.size() * 2);
                        if (st == HIPTS_OK) st = upload(B.mlp_img.p, img.data(), img.size() * 2);
                    }
                }
            }
            else if (t == "mlp.act.scale") { EXPECT(1); B.s2 = data[0]; }
            else if (t == "mlp.act.bias") { EXPECT(1); B.b2 = data[0]; }
            else if (!B.attn && t == "token_mixer.pwconv1.weight") { EXPECT((int64_t)2 * C * C); st = upload_matrix16(B.w_in, data, 2 * C, C, round_up(2 * C, 256), f16); }
            else if (!B.attn && t == "token_mixer.pwconv2.weight") { EXPECT((int64_t)2 * C * C); st = up(B.w_out, C, 2 * C, round_up(C, 256)); }
            else if (!B.attn && t == "token_mixer.act1.scale") { EXPECT(1); B.s1 = data[0]; }
            else if (!B.attn && t == "token_mixer.act1.bias") { EXPECT(1); B.b1 = data[0]; }
            else if (!B.attn && t == "token_mixer.dwconv.weight") {
                EXPECT((int64_t)2 * C * 49);
                std::vector<float> w2((size_t)49 * 2 * C);          // [c][tap] -> [tap][c]
                for (int cc = 0; cc < 2 * C; ++cc)
                    for (int tp = 0; tp < 49; ++tp) w2[(size_t)tp * 2 * C + cc] = data[(size_t)cc * 49 + tp];
                st = upload_f32(B.dw, w2.data(), w2.size());
                if (st == HIPTS_OK && f16) {
                    const std::vector<uint32_t> tzv = dw_toeplitz_lanes(data, 2 * C);
                    st = upload_f32(B.dwz, reinterpret_cast<const float*>(tzv.data()), tzv.size());
                }
            }
            else if (B.attn && t == "token_mixer.qkv.weight") {
                EXPECT((int64_t)3 * C * C);
                // q|k rows and the v rows are read by separate launches whose last tile may run past its
                // own rows: pad behind the v rows as well
                st = upload_matrix16(B.w_in, data, 3 * C, C, round_up(2 * C, 256) + round_up(C, 256) + 256, f16);
            }
            else if (B.attn && t == "token_mixer.proj.weight") { EXPECT((int64_t)C * C); st = upload_matrix16(B.w_out, data, C, C, round_up(C, 256), f16); }
            else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
        } else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
    } else return set_error(HIPTS_ERR_INVALID, "unknown tensor key %s", key_c);
#undef EXPECT
    if (st) return st;
    h->fold_dirty = true;
    auto it = std::find(h->missing.begin(), h->missing.end(), key);
    if (it != h->missing.end()) h->missing.erase(it);
    return HIPTS_OK;
}

int hipts_ccip_forward_u8(hipts_ccip_t* h, const uint8_t* images, int images_memspace, int batch, float* features_out,
                          int out_memspace, void* stream) {
    return ccip_forward_impl(h, images, images_memspace, true, batch, features_out, out_memspace, (hipStream_t)stream);
}

int hipts_ccip_forward_f32(hipts_ccip_t* h, const float* x, int x_memspace, int batch, float* features_out, int out_memspace,
                           void* stream) {
    return ccip_forward_impl(h, x, x_memspace, false, batch, features_out, out_memspace, (hipStream_t)stream);
}

int hipts_ccip_flops_per_image(const hipts_ccip_t* h, double* flops) {
    HIPTS_REQUIRE(h && flops, "null argument");
    *flops = h->flops_per_image;
    return HIPTS_OK;
}

// Debug / test entry (include/hip_tagsearch_debug.h): the depthwise 7x7 of a SepConv block on its own.
int hiptsdbg_dwconv7(const uint16_t* in_f16, const float* w, uint16_t* out_f16, int batch, int H, int C, int mode, int iters, float* ms_out) {
    HIPTS_REQUIRE(in_f16 && w && out_f16 && batch >= 1 && H >= 1 && C >= 64 && C % 64 == 0 && mode >= 0 && mode <= 3, "hiptsdbg_dwconv7: bad argument");
    HIPTS_REQUIRE(mode == 0 || H >= 16, "hiptsdbg_dwconv7: the matrix-core kernel needs a side of 16 or more");
    const size_t n = (size_t)batch * H * H * C;
    DevBuf in, out, wv, tz;
    HIPTS_TRY(in.alloc(n * 2));
    HIPTS_TRY(out.alloc(n * 2));
    HIPTS_TRY(upload(in.p, in_f16, n * 2));
    std::vector<float> w2((size_t)49 * C);
    for (int cc = 0; cc < C; ++cc)
        for (int tp = 0; tp < 49; ++tp) w2[(size_t)tp * C + cc] = w[(size_t)cc * 49 + tp];
    HIPTS_TRY(upload_f32(wv, w2.data(), w2.size()));
    const std::vector<uint32_t> tzv = dw_toeplitz_lanes(w, C);
    HIPTS_TRY(upload_f32(tz, reinterpret_cast<const float*>(tzv.data()), tzv.size()));
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    const int tiles_x = ceil_div(H, DW_TW), tiles_y = ceil_div(H, DW_TH);
    for (int it = 0; it <= iters; ++it) {
        if (it == 1) HIPTS_HIP(hipEventRecord(e0, nullptr));
        if (mode == 0) {
            dwconv7_kernel<true><<<batch * tiles_y * tiles_x * (C / DW_CS), 256, DW_LDS_BYTES, nullptr>>>(in.as<bf16_t>(), wv.as<float>(), out.as<bf16_t>(), H, C, tiles_x, tiles_y);
            HIPTS_LAUNCH_CHECK();
        } else {
            HIPTS_TRY(launch_dwconv7_mfma(in.as<bf16_t>(), tz.as<uint32_t>(), out.as<bf16_t>(), batch, H, C, mode, nullptr));
        }
    }
    HIPTS_HIP(hipEventRecord(e1, nullptr));
    HIPTS_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    if (iters > 0) HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = iters > 0 ? ms / iters : 0.f;
    HIPTS_HIP(hipEventDestroy(e0));
    HIPTS_HIP(hipEventDestroy(e1));
    HIPTS_HIP(hipMemcpy(out_f16, out.p, n * 2, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}

// Host only (no GPU call): the lane images dw_toeplitz_lanes builds from weights [channels][49], u32 [channels][7][64]
// (tests/test_host_layouts.py, runs without a GPU).
int hiptsdbg_dw_toeplitz(const float* w, int channels, uint32_t* out) {
    HIPTS_REQUIRE(w && out && channels >= 1, "hiptsdbg_dw_toeplitz: bad argument");
    const std::vector<uint32_t> t = dw_toeplitz_lanes(w, channels);
    memcpy(out, t.data(), t.size() * 4);
    return HIPTS_OK;
}

// the stamps of a -DHIPTS_DW_STAMPS build (zeros otherwise): [0] start, [1] operands built, then per tile 8 stamps from [2]: start, loads
// requested, planes written, barrier, products done, barrier, results in LDS + barrier, stores issued; [31] end
int hiptsdbg_dwconv7_stamps(unsigned long long* host, int n) {
    HIPTS_REQUIRE(host && n >= 1 && n <= 32, "hiptsdbg_dwconv7_stamps: bad argument");
    HIPTS_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(dw_stamps), (size_t)n * 8));
    return HIPTS_OK;
}

}  // extern "C"
