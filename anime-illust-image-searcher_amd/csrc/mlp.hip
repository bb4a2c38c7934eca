// The MLP of a CAFormer block as ONE kernel (round 4; CCIP encoder stages 0-1, gen_cfeatures.py:158 -- the MetaFormer `Mlp`:
// fc1 -> StarReLU -> fc2, then the block's scaled residual and the LayerNorm that follows it):
//
//     x[m][:] = rs * x[m][:] + StarReLU(xn[m][:] W1^T) W2^T          xn_out[m][:] = LayerNorm(x[m][:]) * gamma      (C = 128 / 256, hidden = 4 C)
//
// As two GEMM launches the hidden tensor (M x 4C halves: 302 MB per stage-0 launch at batch 32) is written by the first and read by
// the second -- 57 % of the bytes of a stage-0 MLP, and both launches run at the HBM roofline.  Here it never leaves the
// registers.  The structure is the attention kernel's: the hidden dimension plays the key sequence, W1 the keys, W2 the values,
// StarReLU the softmax -- without a normaliser:
//   * a wave owns 32 rows: their xn fragments (second MFMA operand) stay in registers for the whole kernel, and so do the fp32
//     accumulators of the 32 x C output;
//   * the hidden units are walked in chunks of 32.  S^T = W1[chunk] . xn^T (first operand = weights: the accumulator's rows are hidden
//     units, its columns the wave's rows), so a lane holds 4 + 4 hidden values of ONE row -- after StarReLU and rounding to half exactly
//     the eight K-elements of the second operand of  O^T += W2[:, chunk] . P^T.  W2's hidden index is permuted at upload to the order
//     in which the accumulator hands them over (within a chunk: position 8 q + e <-> hidden 16 (e >> 2) + 4 q + (e & 3));
//   * the weights of a chunk are one contiguous, LDS-shaped image in global memory (row pitches 2 C + 32 and 96 bytes: conflict-free
//     ds_read_b128), copied by LDS-DMA two chunks deep; one barrier per chunk for the eight waves (256 rows) of a workgroup;
//   * epilogue: residual read-modify-write of the fp32 stream, LayerNorm within the four lanes that hold a row.
// Numerics: the same roundings as the two launches (half P, fp32 accumulation; the hidden units are summed in another order).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "vit_internal.h"

namespace hipts {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef HIPTS_MLP_RING
#define HIPTS_MLP_RING 4
#endif
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <int OFF>
__device__ __forceinline__ void lds_read16(f16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(f16x8& reg) {          // the register passes through: its users cannot move above the wait
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "n"(N));
}
__device__ __forceinline__ void issue_fence(f32x4& a, f32x4& b) { asm volatile("" : "+v"(a), "+v"(b)); }

template <int C>
struct MlpImg {
    static constexpr int HC = 32;                          // hidden units per chunk
    static constexpr int P1 = 2 * C + 32;                  // bytes per W1 row (C halves) in the image
    static constexpr int P2 = 96;                          // bytes per W2 row (32 halves)
    static constexpr int W2_OFF = HC * P1;
    static constexpr int BYTES = HC * P1 + C * P2;         // 21 504 (C = 128) / 41 984 (C = 256): whole KiB
    static_assert(BYTES % 1024 == 0, "the image is copied in 1 KiB pieces");
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int C>
__global__ __launch_bounds__(512) void mlp_fused_kernel(const bf16_t* xn, const char* __restrict__ wimg, float* __restrict__ x,
                                                        const float* __restrict__ res_scale, const float* __restrict__ gamma,
                                                        bf16_t* xn_out, int M, int chunks, float star_s, float star_b, float eps) {
    using I = MlpImg<C>;
    constexpr int KS = C / 32, CF = C / 16, RF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * 256 + wave * 32;
    static_assert(RF == 2, "the issue fences below name two accumulators");
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    auto copy_chunk = [&](int j) {          // wave w copies the 1 KiB pieces w, w + 8, ...
        const char* src = wimg + (size_t)j * I::BYTES;
        char* dst = smem + (j & 1) * I::BYTES;
        for (int p = wave; p < I::BYTES / 1024; p += 8) glds16(src + p * 1024 + lane * 16, dst + p * 1024);
    };
    copy_chunk(0);

    // the wave's rows as second operands: lane (m, q) holds xn[row][32 ks + 8 q ..+7]
    f16x8 xf[RF][KS];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        const int row = min(row0 + rf * 16 + lr, M - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[rf][ks] = *reinterpret_cast<const f16x8*>(xn + (size_t)row * C + ks * 32 + kq * 8);
    }
    f32x4 O[CF][RF];
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) O[cf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < chunks; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of chunk j (and, the first time, its rows)
        __syncthreads();                                           // everybody's pieces; everybody is done with chunk j - 1's buffer
        if (j + 1 < chunks) copy_chunk(j + 1);
        // Operand reads run RING - 1 fragments ahead of the MFMAs that use them, through inline asm with counted waits: written as plain
        // loads the compiler put s_waitcnt lgkmcnt(0) in front of every second or fourth pair of MFMAs (420 TFLOP/s).  Step t < 2 KS:
        // W1 fragment (hf = t & 1, ks = t >> 1); step 2 KS + cf: W2 fragment cf; slot t % RING.  Nothing else in the loop touches
        // LGKM (no scalar loads, no other LDS operation), so "RING - 1 younger reads outstanding" is exact.
        constexpr int RING = HIPTS_MLP_RING, T = 2 * KS + CF;
        const uint32_t a1 = lds_base + (j & 1) * I::BYTES + lr * I::P1 + kq * 16;
        const uint32_t a2 = lds_base + (j & 1) * I::BYTES + I::W2_OFF + lr * I::P2 + kq * 16;
        f16x8 w[RING];
        auto read = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < 2 * KS) lds_read16<(t & 1) * 16 * I::P1 + (t >> 1) * 64>(w[t % RING], a1);
            else if constexpr (t < T) lds_read16<(t - 2 * KS) * 16 * I::P2>(w[t % RING], a2);
        };
        auto wait = [&](auto tc) {          // until the read of step t has landed: min(RING - 1, T - 1 - t) younger reads may be outstanding
            constexpr int t = decltype(tc)::value, n = (T - 1 - t) < (RING - 1) ? (T - 1 - t) : (RING - 1);
            lds_wait<n>(w[t % RING]);
        };
        f32x4 S[2][RF];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) S[hf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<0, RING>([&](auto tc) { read(tc); });
        static_for<0, 2 * KS>([&](auto tc) {
            constexpr int t = decltype(tc)::value, hf = t & 1, ks = t >> 1;
            wait(tc);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) S[hf][rf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t % RING], xf[rf][ks], S[hf][rf], 0, 0, 0);
            // the slot is rewritten by the next read: the MFMAs above must have been issued (they read their operands at issue)
            issue_fence(S[hf][0], S[hf][1]);
            read(std::integral_constant<int, t + RING>{});
        });
        f16x8 P[RF];
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x4 r = __builtin_elementwise_max(S[hf][rf], f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 v = r * r * star_s + star_b;
#pragma unroll
                for (int i = 0; i < 4; ++i) P[rf][hf * 4 + i] = (_Float16)v[i];
            }
        }
        static_for<0, CF>([&](auto cc) {
            constexpr int cf = decltype(cc)::value, t = 2 * KS + cf;
            wait(std::integral_constant<int, t>{});
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) O[cf][rf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t % RING], P[rf], O[cf][rf], 0, 0, 0);
            issue_fence(O[cf][0], O[cf][1]);
            read(std::integral_constant<int, t + RING>{});
        });
    }

    // ---- epilogue: lane (m = lr, q) holds columns 16 cf + 4 q ..+3 of row rf * 16 + m.  A wave whose 32 rows all exist (uniform) stores
    // without predicates: a predicated store is a basic block of its own, entered through s_waitcnt vmcnt(0)
    auto finish = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) {
            const int row = row0 + rf * 16 + lr;
            const bool ok = FULL || row < M;
            float* xr = x + (size_t)(ok ? row : M - 1) * C + 4 * kq;
            f32x4 xo[CF];
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) xo[cf] = *reinterpret_cast<const f32x4*>(xr + cf * 16);
            float s1 = 0.f;
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                f32x4 v;
                if (res_scale) v = xo[cf] * *reinterpret_cast<const f32x4*>(res_scale + cf * 16 + 4 * kq) + O[cf][rf];
                else v = xo[cf] + O[cf][rf];
                if (FULL || ok) *reinterpret_cast<f32x4*>(xr + cf * 16) = v;
                O[cf][rf] = v;
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
            }
            if (gamma) {
                s1 += __shfl_xor(s1, 16);
                s1 += __shfl_xor(s1, 32);
                const float mean = s1 / (float)C;
                float s2 = 0.f;
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const f32x4 d = O[cf][rf] - mean;
                    s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                }
                s2 += __shfl_xor(s2, 16);
                s2 += __shfl_xor(s2, 32);
                const float rstd = 1.0f / sqrtf(s2 / (float)C + eps);
                bf16_t* orow = xn_out + (size_t)(ok ? row : M - 1) * C + 4 * kq;
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cf * 16 + 4 * kq);
                    const f32x4 o = (O[cf][rf] - mean) * rstd * g;
                    f16x4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = (_Float16)o[i];
                    if (FULL || ok) *reinterpret_cast<f16x4*>(orow + cf * 16) = h;
                }
            }
        }
    };
    if (row0 + 32 <= M) finish(std::true_type{});
    else finish(std::false_type{});
}

}  // namespace

// The weights of a fused MLP as the kernel's chunk images (half): w1 [4C][C], w2 [C][4C] float32 on the host.
std::vector<uint16_t> mlp_weight_image(const float* w1, const float* w2, int C) {
    const int hid = 4 * C, P1 = 2 * C + 32, P2 = 96, bytes = 32 * P1 + C * P2;
    std::vector<uint16_t> img((size_t)(hid / 32) * bytes / 2, 0);
    for (int j = 0; j < hid / 32; ++j) {
        uint16_t* base = img.data() + (size_t)j * bytes / 2;
        for (int h = 0; h < 32; ++h)
            for (int c = 0; c < C; ++c) base[(size_t)h * (P1 / 2) + c] = f32_to_f16_rne(w1[(size_t)(j * 32 + h) * C + c]);
        uint16_t* w2b = base + 32 * P1 / 2;
        for (int n = 0; n < C; ++n)
            for (int q = 0; q < 4; ++q)
                for (int e = 0; e < 8; ++e)
                    w2b[(size_t)n * (P2 / 2) + 8 * q + e] = f32_to_f16_rne(w2[(size_t)n * hid + j * 32 + 16 * (e >> 2) + 4 * q + (e & 3)]);
    }
    return img;
}

bool mlp_fused_supports(int C) { return C == 128 || C == 256; }

int launch_mlp_fused(const bf16_t* xn, const void* wimg, float* x, const float* res_scale, const float* gamma, bf16_t* xn_out, int M, int C,
                     float star_s, float star_b, float eps, hipStream_t s) {
    HIPTS_REQUIRE(mlp_fused_supports(C), "fused MLP: width %d is not built (128, 256)", C);
    HIPTS_REQUIRE(xn && wimg && x && M >= 1 && (!gamma || xn_out), "fused MLP: bad argument");
    const int grid = (M + 255) / 256, chunks = 4 * C / 32;
    static bool once = false;
    if (!once) {
        HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MlpImg<128>::BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MlpImg<256>::BYTES));
        once = true;
    }
    if (C == 128) mlp_fused_kernel<128><<<grid, 512, 2 * MlpImg<128>::BYTES, s>>>(xn, (const char*)wimg, x, res_scale, gamma, xn_out, M, chunks, star_s, star_b, eps);
    else mlp_fused_kernel<256><<<grid, 512, 2 * MlpImg<256>::BYTES, s>>>(xn, (const char*)wimg, x, res_scale, gamma, xn_out, M, chunks, star_s, star_b, eps);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace hipts

// Debug / test entry (include/hip_tagsearch_debug.h): the fused MLP on its own.  Host arrays: xn IEEE-half bits [M][C], w1 [4C][C], w2 [C][4C],
// x [M][C] (in / out), res_scale [C] or null, gamma [C] or null (then xn_out is not written).
extern "C" int hiptsdbg_mlp_fused(const uint16_t* xn, const float* w1, const float* w2, float* x, const float* res_scale, const float* gamma,
                                  uint16_t* xn_out, int M, int C, float star_s, float star_b, float eps, int iters, float* ms_out) {
    using namespace hipts;
    HIPTS_REQUIRE(xn && w1 && w2 && x && M >= 1 && mlp_fused_supports(C), "hiptsdbg_mlp_fused: bad argument");
    const std::vector<uint16_t> img = mlp_weight_image(w1, w2, C);
    DevBuf dxn, dimg, dx, drs, dg, dout;
    const size_t n = (size_t)M * C;
    HIPTS_TRY(dxn.alloc(n * 2));
    HIPTS_TRY(dimg.alloc(img.size() * 2));
    HIPTS_TRY(dx.alloc(n * 4));
    HIPTS_TRY(dout.alloc(n * 2));
    HIPTS_TRY(drs.alloc(C * 4));
    HIPTS_TRY(dg.alloc(C * 4));
    HIPTS_TRY(upload(dxn.p, xn, n * 2));
    HIPTS_TRY(upload(dimg.p, img.data(), img.size() * 2));
    if (res_scale) HIPTS_TRY(upload(drs.p, res_scale, C * 4));
    if (gamma) HIPTS_TRY(upload(dg.p, gamma, C * 4));
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    HIPTS_TRY(upload(dx.p, x, n * 4));
    for (int it = 0; it <= iters; ++it) {
        if (it == 1) HIPTS_HIP(hipEventRecord(e0, nullptr));
        if (it == iters && iters > 0) {          // the last launch is the one whose result is returned: on the caller's x again
            HIPTS_HIP(hipEventRecord(e1, nullptr));
            HIPTS_TRY(upload(dx.p, x, n * 4));
        }
        HIPTS_TRY(launch_mlp_fused(dxn.as<bf16_t>(), dimg.p, dx.as<float>(), res_scale ? drs.as<float>() : nullptr, gamma ? dg.as<float>() : nullptr,
                                   dout.as<bf16_t>(), M, C, star_s, star_b, eps, nullptr));
    }
    HIPTS_HIP(hipDeviceSynchronize());
    float ms = 0.f;
    if (iters > 1) HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = iters > 1 ? ms / (iters - 1) : 0.f;
    HIPTS_HIP(hipEventDestroy(e0));
    HIPTS_HIP(hipEventDestroy(e1));
    HIPTS_HIP(hipMemcpy(x, dx.p, n * 4, hipMemcpyDeviceToHost));
    if (gamma && xn_out) HIPTS_HIP(hipMemcpy(xn_out, dout.p, n * 2, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}
