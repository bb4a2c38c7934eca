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
//     ds_read_b128), copied by LDS-DMA into a ring of four buffers, a piece at a time between the MFMAs, two chunks ahead of its barrier; one barrier per chunk for the eight waves (256 rows) of a workgroup;
//   * the accumulators START as rs * x (the wave's rows of the fp32 stream, requested at kernel start); epilogue: store them, LayerNorm
//     within the four lanes that hold a row.
// Numerics: the same roundings as the two launches (half P, fp32 accumulation; the hidden units are summed in another order).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "vit_internal.h"

namespace hipts {
namespace {


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

// -DHIPTS_MLP_STAMPS=<workgroup>: wave 0 of that workgroup leaves 100 MHz time stamps of chunk 8's phases (tools/mlp_stamps.py)
__device__ unsigned long long mlp_stamps[16];
#ifdef HIPTS_MLP_STAMPS
#define MLP_STAMP(i) do { if (blockIdx.x == HIPTS_MLP_STAMPS && threadIdx.x == 0 && j == 8) mlp_stamps[i] = wall_clock64(); } while (0)
#define MLP_STAMP_AT(i) do { if (blockIdx.x == HIPTS_MLP_STAMPS && threadIdx.x == 0) mlp_stamps[i] = wall_clock64(); } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#define MLP_STAMP_AT(i) do { } while (0)
#endif
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
    static constexpr int P2 = 80;                          // bytes per W2 row (32 halves): two-way conflicts on 16 reads per chunk, and FOUR chunk buffers fit (96: none, three)
    static constexpr int W2_OFF = HC * P1;
    static constexpr int BYTES = HC * P1 + C * P2;         // 19 456 (C = 128) / 37 888 (C = 256): whole KiB
    static_assert(BYTES % 1024 == 0, "the image is copied in 1 KiB pieces");
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int C, int WAVES, int NBUF = 4>
__global__ __launch_bounds__(WAVES * 64, NBUF == 2 ? 2 : 1) void mlp_fused_kernel(const bf16_t* xn, const char* __restrict__ wimg, float* __restrict__ x,
                                                        const float* __restrict__ res_scale, const float* __restrict__ gamma,
                                                        bf16_t* xn_out, int M, int chunks, float star_s, float star_b, float eps, int xblk) {
    using I = MlpImg<C>;
    constexpr int KS = C / 32, CF = C / 16, RF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * (WAVES * 32) + wave * 32;
    static_assert(RF == 2, "the issue fences below name two accumulators");
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    MLP_STAMP_AT(8);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);          // in a scalar register: no divergent control flow around the copies
    // NBUF = 4: the scheme described above.  NBUF = 2 (four-wave workgroups at C = 256: 74 KB of LDS, TWO workgroups per CU, each other's
    // waits filled): chunk j + 1 is copied during chunk j, published by the barrier that ends chunk j, and the read ring is refilled behind it.
    static_assert(NBUF == 2 || NBUF == 4, "chunk buffers");
    constexpr bool CROSS = NBUF == 4;
    constexpr int NP = I::BYTES / 1024, PER = (NP + WAVES - 1) / WAVES;
    // piece i of this wave for chunk j: the 1 KiB pieces w, w + WAVES, ...; every wave issues PER of them (the last ones twice: the same bytes),
    // so that "PER copies outstanding" means the same in every wave
    auto copy_piece = [&](int j, int slot, int i) {
        const int p = min(wave_u + WAVES * i, NP - 1);
        glds16(wimg + (size_t)j * I::BYTES + p * 1024 + lane * 16, smem + slot * I::BYTES + p * 1024);
    };
    for (int j = 0; j < (CROSS ? 3 : 1) && j < chunks; ++j)
#pragma unroll
        for (int i = 0; i < PER; ++i) copy_piece(j, j, i);

    // the wave's rows as second operands: lane (m, q) holds xn[row][32 ks + 8 q ..+7]
    f16x8 xf[RF][KS];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        const int row = min(row0 + rf * 16 + lr, M - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[rf][ks] = *reinterpret_cast<const f16x8*>(xn + (size_t)row * C + ks * 32 + kq * 8);
    }
    // the accumulators start as the wave's rows of the residual stream (requested now, needed at the first second product): read in the
    // epilogue, every workgroup of the chip asked for its 256 KB at the same moment and waited 28 us for them (stage 1)
    f32x4 O[CF][RF];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        // (xblk: the stream as 16 x 16 blocks of 1 KB, gemm_epi.h::x_off -- a load instruction here is lane (row lr, columns 4 kq ..+3) of a
        // 16-column block: one contiguous kilobyte of the blocked stream, sixteen half lines of the row-major one)
        const int rr = min(row0 + rf * 16 + lr, M - 1);
        const float* xr = xblk ? x + (((size_t)(rr >> 4) * (C >> 4)) << 8) + (rr & 15) * 16 + 4 * kq : x + (size_t)rr * C + 4 * kq;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) O[cf][rf] = *reinterpret_cast<const f32x4*>(xr + cf * (xblk ? 256 : 16));
    }

    // Operand reads run RING - 1 fragments ahead of the MFMAs that use them, through inline asm with counted waits: written as plain
    // loads the compiler put s_waitcnt lgkmcnt(0) in front of every second or fourth pair of MFMAs (420 TFLOP/s).  Step t < 2 KS: W1
    // fragment (hf = t & 1, ks = t >> 1); step 2 KS + cf: W2 fragment cf; slot t % RING; steps past T are the first of the NEXT chunk --
    // four chunk buffers, and the barrier that ends chunk j has every wave's pieces of chunk j + 2, so the ring never drains between
    // chunks.  Nothing else in the loop touches LGKM (no scalar loads, no other LDS operation): "RING - 1 younger reads outstanding" is exact.
    constexpr int RING = HIPTS_MLP_RING, T = 2 * KS + CF;
    static_assert(T % RING == 0 && RING <= 2 * KS && PER <= 2 * KS, "ring slots continue across chunks; the copies fit the first product");
    const uint32_t lane_w1 = lr * I::P1 + kq * 16, lane_w2 = I::W2_OFF + lr * I::P2 + kq * 16;
    f16x8 w[RING];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of chunks 0 .. 2, and its rows
    __builtin_amdgcn_s_barrier();
    static_for<0, RING>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        lds_read16<(t & 1) * 16 * I::P1 + (t >> 1) * 64>(w[t], lds_base + lane_w1);
    });
    auto chunk = [&](auto first_tag, auto last_tag, int j, int slot) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const int slot1 = (slot + 1) & (NBUF - 1), slot3 = (slot + 3) & (NBUF - 1);          // NBUF = 2: both are the other buffer
        constexpr int AHEAD = CROSS ? 3 : 1;
        const bool copying = !LAST && j + AHEAD < chunks;          // chunk j + AHEAD into the buffer that held chunk j - 1 (every wave is past the barrier that ended it)
        MLP_STAMP(0);
        const uint32_t a1 = lds_base + slot * I::BYTES + lane_w1, a2 = lds_base + slot * I::BYTES + lane_w2, n1 = lds_base + slot1 * I::BYTES + lane_w1;
        auto read = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t < 2 * KS) lds_read16<(t & 1) * 16 * I::P1 + (t >> 1) * 64>(w[t % RING], a1);
            else if constexpr (t < T) lds_read16<(t - 2 * KS) * 16 * I::P2>(w[t % RING], a2);
            else if constexpr (!LAST && CROSS) lds_read16<((t - T) & 1) * 16 * I::P1 + ((t - T) >> 1) * 64>(w[t % RING], n1);
        };
        auto wait = [&](auto tc) {          // until the read of step t has landed
            constexpr int t = decltype(tc)::value, n = ((LAST || !CROSS) && (T - 1 - t) < (RING - 1)) ? (T - 1 - t) : (RING - 1);
            lds_wait<n>(w[t % RING]);
        };
        MLP_STAMP(1);
        f32x4 S[2][RF];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) S[hf][rf] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<0, 2 * KS>([&](auto tc) {
            constexpr int t = decltype(tc)::value, hf = t & 1, ks = t >> 1;
            wait(tc);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) S[hf][rf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t % RING], xf[rf][ks], S[hf][rf], 0, 0, 0);
            // the slot is rewritten by the next read: the MFMAs above must have been issued (they read their operands at issue)
            issue_fence(S[hf][0], S[hf][1]);
            read(std::integral_constant<int, t + RING>{});
            if constexpr (!LAST && t < PER) {          // one piece of the copy behind a pair of MFMAs
                if (copying) copy_piece(j + AHEAD, slot3, t);
            }
        });
        MLP_STAMP(2);
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
        MLP_STAMP(3);
        if constexpr (FIRST) {          // x * res_scale before the first accumulation (the compiler waits for the rows here)
            if (res_scale) {
#pragma unroll
                for (int cf = 0; cf < CF; ++cf) {
                    const f32x4 rsv = *reinterpret_cast<const f32x4*>(res_scale + cf * 16 + 4 * kq);
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) O[cf][rf] = O[cf][rf] * rsv;
                }
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
        MLP_STAMP(4);
        if constexpr (!LAST) {
            // this wave's pieces of chunk j + 2 (requested a chunk ago); those of chunk j + 3 stay in flight
            if (CROSS && copying) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MLP_STAMP(5);
            __builtin_amdgcn_s_barrier();                              // everybody's; and everybody is done with chunk j's buffer
            MLP_STAMP(6);
            if constexpr (!CROSS) {          // the ring starts again on the chunk just published
                static_for<0, RING>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    lds_read16<(t & 1) * 16 * I::P1 + (t >> 1) * 64>(w[t], n1);
                });
            }
        }
    };
    chunk(std::true_type{}, std::false_type{}, 0, 0);          // chunks >= 16
    int slot = 1;
    for (int j = 1; j + 1 < chunks; ++j) {
        chunk(std::false_type{}, std::false_type{}, j, slot);
        slot = (slot + 1) & (NBUF - 1);
    }
    chunk(std::false_type{}, std::true_type{}, chunks - 1, slot);
    MLP_STAMP_AT(9);

    // ---- epilogue: lane (m = lr, q) holds columns 16 cf + 4 q ..+3 of row rf * 16 + m.  A wave whose 32 rows all exist (uniform) stores
    // without predicates: a predicated store is a basic block of its own, entered through s_waitcnt vmcnt(0)
    auto finish = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) {
            const int row = row0 + rf * 16 + lr;
            const bool ok = FULL || row < M;
            const int rr = ok ? row : M - 1;
            float* xr = xblk ? x + (((size_t)(rr >> 4) * (C >> 4)) << 8) + (rr & 15) * 16 + 4 * kq : x + (size_t)rr * C + 4 * kq;
            float s1 = 0.f;
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                const f32x4 v = O[cf][rf];          // rs * x + the products
                if (FULL || ok) *reinterpret_cast<f32x4*>(xr + cf * (xblk ? 256 : 16)) = v;
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
    MLP_STAMP_AT(10);
}

}  // namespace

// The weights of a fused MLP as the kernel's chunk images (half): w1 [4C][C], w2 [C][4C] float32 on the host.
std::vector<uint16_t> mlp_weight_image(const float* w1, const float* w2, int C) {
    const int hid = 4 * C, P1 = 2 * C + 32, P2 = 80, bytes = 32 * P1 + C * P2;
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
                     float star_s, float star_b, float eps, hipStream_t s, int waves, int xblk) {
    HIPTS_REQUIRE(!xblk || M % 16 == 0, "fused MLP: the blocked residual stream needs whole 16-row blocks (M = %d)", M);
    HIPTS_REQUIRE(mlp_fused_supports(C), "fused MLP: width %d is not built (128, 256)", C);
    HIPTS_REQUIRE(xn && wimg && x && M >= 1 && (!gamma || xn_out), "fused MLP: bad argument");
    const int chunks = 4 * C / 32;
    // eight waves (256 rows) per workgroup share a chunk's weights; a launch that would leave CUs without a workgroup, or whose last round
    // is nearly empty, takes four-wave workgroups (128 rows: twice the weight traffic, half the lifetime).  HIPTS_MLP_WAVES=4 / 8 forces.
    static const int waves_env = getenv("HIPTS_MLP_WAVES") ? atoi(getenv("HIPTS_MLP_WAVES")) : 0;
    static int cus_of[64] = {};          // CUs per device, asked once -- together with the kernels' LDS attribute, which is per device
    static PerDevice once;
    int dev = 0;
    HIPTS_HIP(hipGetDevice(&dev));
    HIPTS_REQUIRE(dev >= 0 && dev < 64, "fused MLP: device %d", dev);
    {
        std::lock_guard<std::mutex> lk(once.mu);
        if (!once.done(dev)) {
            int n = 0;
            HIPTS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * MlpImg<128>::BYTES));
            HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * MlpImg<256>::BYTES));
            HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * MlpImg<128>::BYTES));
            HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * MlpImg<256>::BYTES));
            HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<256, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MlpImg<256>::BYTES));
            HIPTS_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<128, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MlpImg<128>::BYTES));
            cus_of[dev] = n > 0 ? n : 256;
            once.mark(dev);
        }
    }
    const int cus = cus_of[dev];
    const int g8 = (M + 255) / 256, g4 = (M + 127) / 128;
    static const bool two_buffers = !(getenv("HIPTS_MLP_NBUF") && atoi(getenv("HIPTS_MLP_NBUF")) == 4);      // C = 256, four waves: two chunk buffers = two workgroups per CU (A/B: HIPTS_MLP_NBUF=4, one per CU)
    // rounds x relative lifetime.  C = 256: one workgroup per CU either way, a four-wave one lives 50 us against 85; C = 128: two four-wave
    // workgroups per CU (LDS 76 KB, 150 registers), each about as long as the eight-wave one, out of step with each other (152 against 175 us)
    // C = 256 with two chunk buffers (late round 4): two four-wave workgroups per CU as well (M = 73728: 160 -> 127 us, M = 23040: 77 -> 49 us);
    // only a launch that fills more than half the chip with ONE round of eight-wave workgroups stays with those (M = 46080: 82 against 84 us)
    const int slots4 = C == 128 ? 2 * cus : cus;
    const double t8 = (double)((g8 + cus - 1) / cus), t4 = (C == 128 ? 0.9 : 0.6) * (double)((g4 + slots4 - 1) / slots4);
    const int want = waves ? waves : waves_env;
    const bool four = want == 4 || (want != 8 && ((C == 256 && two_buffers) ? !(g8 <= cus && g8 * 2 > cus) : t4 < t8));
    const int grid = four ? g4 : g8;
#define HIPTS_MLP_LAUNCH(CC, WW) mlp_fused_kernel<CC, WW><<<grid, WW * 64, 4 * MlpImg<CC>::BYTES, s>>>(xn, (const char*)wimg, x, res_scale, gamma, xn_out, M, chunks, star_s, star_b, eps, xblk)
    static const bool two_buffers128 = !(getenv("HIPTS_MLP_NBUF128") && atoi(getenv("HIPTS_MLP_NBUF128")) == 4);      // C = 128 four waves with two chunk buffers: three workgroups per CU (148 -> 134 us alone, the encoder the same; A/B: HIPTS_MLP_NBUF128=4)
    if (C == 128 && four && two_buffers128) mlp_fused_kernel<128, 4, 2><<<grid, 256, 2 * MlpImg<128>::BYTES, s>>>(xn, (const char*)wimg, x, res_scale, gamma, xn_out, M, chunks, star_s, star_b, eps, xblk);
    else if (C == 128) { if (four) HIPTS_MLP_LAUNCH(128, 4); else HIPTS_MLP_LAUNCH(128, 8); }
    else if (four && two_buffers) mlp_fused_kernel<256, 4, 2><<<grid, 256, 2 * MlpImg<256>::BYTES, s>>>(xn, (const char*)wimg, x, res_scale, gamma, xn_out, M, chunks, star_s, star_b, eps, xblk);
    else { if (four) HIPTS_MLP_LAUNCH(256, 4); else HIPTS_MLP_LAUNCH(256, 8); }
#undef HIPTS_MLP_LAUNCH
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace hipts

// Debug / test entry (include/hip_tagsearch_debug.h): the fused MLP on its own.  Host arrays: xn IEEE-half bits [M][C], w1 [4C][C], w2 [C][4C],
// x [M][C] (in / out), res_scale [C] or null, gamma [C] or null (then xn_out is not written).
// Host only (no GPU call): the chunk images mlp_weight_image builds, for the layout test that runs without a GPU (tests/test_host_layouts.py).
extern "C" int hiptsdbg_mlp_weight_image(const float* w1, const float* w2, int C, uint16_t* out, long long out_halves) {
    using namespace hipts;
    HIPTS_REQUIRE(w1 && w2 && out && mlp_fused_supports(C), "hiptsdbg_mlp_weight_image: bad argument");
    const std::vector<uint16_t> img = mlp_weight_image(w1, w2, C);
    HIPTS_REQUIRE((long long)img.size() == out_halves, "hiptsdbg_mlp_weight_image: the image has %lld halves, the buffer %lld", (long long)img.size(), out_halves);
    memcpy(out, img.data(), img.size() * 2);
    return HIPTS_OK;
}

// the stamps of a -DHIPTS_MLP_STAMPS build (zeros otherwise): [0..6] chunk 8 of wave 0: start, copy of chunk 10 requested, first product done,
// StarReLU done, second product done, own copies landed, barrier passed; [8] kernel start, [9] last chunk done, [10] epilogue done
extern "C" int hiptsdbg_mlp_stamps(unsigned long long* host, int n) {
    HIPTS_REQUIRE(host && n >= 1 && n <= 16, "hiptsdbg_mlp_stamps: bad argument");
    HIPTS_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(hipts::mlp_stamps), (size_t)n * 8));
    return HIPTS_OK;
}

extern "C" int hiptsdbg_mlp_fused(const uint16_t* xn, const float* w1, const float* w2, float* x, const float* res_scale, const float* gamma,
                                  uint16_t* xn_out, int M, int C, float star_s, float star_b, float eps, int iters, float* ms_out, int waves) {
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
                                   dout.as<bf16_t>(), M, C, star_s, star_b, eps, nullptr, waves));
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
