// attn2.hip -- fused softmax(Q K^T) V, head_dim 64, round 3 (ViT-B/16 and EVA02; timm Attention inside tagging.py:174).
//
// What changed against attn.hip (which still serves head_dim 32, the CAFormer):
//   * K and V tiles of 64 keys reach LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a ring of three K and
//     three V slots (K two tiles ahead, V one): no staging registers, no ds_write pass, one counted vmcnt + one raw s_barrier per tile.
//   * V stays in its natural [token][d] layout (the QKV GEMM needs no transposing epilogue any more); the V^T fragment of
//     O^T = V^T P^T comes out of the row-major tile by ds_read_b64_tr_b16.
//   * a wave owns QB = 2 blocks of 32 query rows: one K / V fragment read feeds two MFMAs (half the LDS instructions per flop).
//   * four waves per workgroup, two workgroups per CU: the two waves that share a SIMD belong to DIFFERENT workgroups, so no
//     barrier keeps them in phase and one's softmax runs under the other's MFMAs.
//   * the query blocks of an (image, head) are dealt to its workgroups evenly (784 rows = 13 blocks of 64 -> 4 + 3 + 3 + 3)
//     instead of leaving a last workgroup with a single busy wave.
//   * O leaves through v_permlane32_swap pairs as 16 B pieces (cdna_hip_programming.md T21).
//
// Arithmetic is attn.hip's: S^T = K Q^T on v_mfma_f32_32x32x16 (query on the lane, so the softmax is lane-local), the S^T accumulator
// converted to 16 bit is directly the B operand of O^T = V^T P^T, scores in the base-2 domain (q pre-scaled by head_dim^-0.5 log2 e),
// softmax with a FIXED reference exponent (see attn.hip: bf16 operands P = 2^S, half operands P = 2^(S - m_ref) with m_ref the row's
// maximum over the first key tile) and the classic per-tile-maximum body as the fallback of a workgroup whose row sum leaves the window.
//
// LDS images (both 128 B per key row, filled lane-linear by the DMA with the swizzle on the SOURCE chunk):
//   K  chunk c of row r at position c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group read 16 different 16 B slots.
//   V  chunk c of row r at position c ^ (((r >> 1) & 1) << 2): the 4 rows x 64 B a half-wave of ds_read_b64_tr_b16 covers
//      lie in the four different 64 B quarters of the 64 banks.
#include <mutex>
#include <type_traits>

#include "vit_internal.h"

// Head room of the half-operand fast path's reference exponent (round 4): m_ref = (the row's maximum over its first 32 keys) + this many
// bits, so P = 2^(S - m_ref) overflows half only when a later score exceeds that maximum by 15 + MARGIN bits, and the row sum's window is
// [2^(-1 - MARGIN), 2^15); keys more than 14 - MARGIN bits below the reference become half subnormals (their P keeps an absolute
// precision of 2^-24, against a row sum of at least 2^-MARGIN).  With 0 a padded picture (prepare_image's white bars are the first keys of every row, tagging.py:100-120) sent most
// query blocks through the fast pass AND the classic one: 4.6 k images/s where noise runs at 5.3 k (tools/vit_content_bench.py).
#ifndef HIPTS_ATTN_REF_MARGIN
#define HIPTS_ATTN_REF_MARGIN 10     // measured (tools/gpurun/r4_margin.sh), padded picture: 0 / 4 / 8 / 10 / 12 bits -> 4667 / 4979 / 5199-5298 / 5261 / 5311 images/s (noise 5340-5435 at every setting); EVA02-L 1129 / 1158 / 1164 at 8 / 10 / 12 (noise 1176); at 12 an attention test fails (the reference keys' P sits two bits above the half subnormals), at 10 the oracle check of the structured images is unchanged
#endif
namespace hipts {
namespace {

#ifndef HIPTS_ATTN2_ASM_ADD
#define HIPTS_ATTN2_ASM_ADD 0       // 1: the row sums as single v_add_f32 in (non-volatile) inline asm -- what first kept hipcc from pairing them into v_pk_add_f32; the Makefile's -fno-slp-vectorize does the same without asm (with the sampled reference below the asm form came out WRONG: negative row sums, every workgroup on the fallback pass)
#endif
#ifndef HIPTS_ATTN_SAMPLED_REF
#define HIPTS_ATTN_SAMPLED_REF 0     // attn2_seq_body, half operands: the reference exponent from 32 keys sampled across the sequence instead of the first 32 (then 4 bits of head room do: -DHIPTS_ATTN_REF_MARGIN=4).  Correct; alone the same 84-88 us; forward 5369 against 5400-5407 images/s on noise, 5367-5372 against 5333-5338 on the padded test picture: costs what it gains, off
#endif
#ifndef HIPTS_ATTN2_PV_SNAKE
#define HIPTS_ATTN2_PV_SNAKE 0
#endif
#ifndef HIPTS_ATTN2_MFMA_SUM
#define HIPTS_ATTN2_MFMA_SUM 0      // 1: the row sums as two more MFMAs per 32 keys (ones x P^T) instead of 32 v_add_f32 -- the loop is bound by vector issue, the matrix pipe ~40 % busy.  Measured and NOT usable with half operands: the MFMA reads half subnormals as zero.  In the P V product that loss is invisible (0.1-0.2 % of a peaked row's mass times the averaged V of its small keys; the default path's error is the same 1.902e-3 for 0 .. 10 bits of head room, profiles/r05_attn_margin_error.txt), but in the SUM it is a pure scale error of the row's output: a few rows of 784 come out 1.0013-1.0017 x too large = 5.9e-3 (profiles/r05_attn_mfma_sum_debug.txt; 3 = both sums side by side, they agree to 4e-5 elsewhere).  P once more as bf16 for the sum MFMAs alone fails too (9e-3: a peaked row's sum is only as good as its largest term's 8 mantissa bits)
#endif
#ifndef HIPTS_ATTN2_SEQ_WAVES
#define HIPTS_ATTN2_SEQ_WAVES 3          // waves per SIMD the sequential body (MODE 1) is compiled for: 4 -> 128 registers, 3 -> 168
#endif
constexpr int KV = 64;                   // keys per tile
constexpr int HD = 64;
constexpr int TILE = KV * HD * 2;        // 8 KiB
#ifndef HIPTS_ATTN2_SEQ_SLOTS
#define HIPTS_ATTN2_SEQ_SLOTS 2
#endif
constexpr int NSK = 3, NSV = 3;          // ring slots: K two tiles ahead, V one (its tile t - 1 is still being multiplied during step t)
constexpr int V_BASE = NSK * TILE;
constexpr int LDS_BYTES = (NSK + NSV) * TILE;      // 48 KiB: three workgroups per CU

typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifdef HIPTS_X_STAMPS                               // measurement-only build: cycle stamps of one wave (tools/gpurun/r3_attn_x.sh)
__device__ unsigned long long g_attn2_stamps[4096];
#define HIPTS_STAMP(slot)                                                                                 \
    do {                                                                                                  \
        if (stamp_on) {                                                                                   \
            const unsigned long long ts_ = __builtin_amdgcn_s_memtime();                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
            if (lane == 0) g_attn2_stamps[(slot)] = ts_;                                                  \
        }                                                                                                 \
    } while (0)
#else
#define HIPTS_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int crow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

// One workgroup's pass over its query blocks with the fixed-reference softmax.  Returns true when a row sum left its window
// (nothing is stored then; the workgroup repeats the blocks with attn2_classic).
// The low halves of four values whose high halves are `hi`: value - (float)hi, rounded once (the hi | lo operand split: two 16-bit
// halves carry 22 significant bits of an IEEE-half operand, 16 of a bf16 one; the consumer multiplies them against [W | W]).
// lo_scale (a power of two; the consumer's weights for the low half carry its inverse): keeps the low halves of IEEE-half operands out
// of the subnormal range -- value 2^-11 is below 2^-14 for every |value| < 1/8.
template <bool F16>
__device__ __forceinline__ bf16x4 pack4_lo(float a, float b, float c, float d, bf16x4 hi, float lo_scale) {
    return pack4<F16>((a - from_op<F16>(hi[0])) * lo_scale, (b - from_op<F16>(hi[1])) * lo_scale, (c - from_op<F16>(hi[2])) * lo_scale,
                      (d - from_op<F16>(hi[3])) * lo_scale);
}

// Output quads 2 kp (even) and 2 kp + 1 (odd) of one query row, normalised, as one 16 B store per lane (attn2_body's comment); with
// lo_off != 0 the low halves follow at op + lo_off (hipts_vit_config_t.operand_f16 bit 4: the attention output as a hi | lo pair).
template <bool F16>
__device__ __forceinline__ void store_o_pair(const f32x16& oe, const f32x16& oo, int kp, float inv, int h, bool valid, bf16_t* op, int lo_off,
                                             float lo_scale) {
    const int ke = 2 * kp, ko = 2 * kp + 1;
    const int ie = 4 * (ke & 3), io = 4 * (ko & 3);
    const float e0 = oe[ie] * inv, e1 = oe[ie + 1] * inv, e2 = oe[ie + 2] * inv, e3 = oe[ie + 3] * inv;
    const float f0 = oo[io] * inv, f1 = oo[io + 1] * inv, f2 = oo[io + 2] * inv, f3 = oo[io + 3] * inv;
    const bf16x4 we = pack4<F16>(e0, e1, e2, e3), wo = pack4<F16>(f0, f1, f2, f3);
    const int kq = h ? ko : ke;              // the chunk this lane owns after the swap: d = 32 (kq >> 2) + 8 (kq & 3) .. + 7
    bf16_t* dst = op + 32 * (kq >> 2) + 8 * (kq & 3);
    {
        const uint2 ue = __builtin_bit_cast(uint2, we), uo = __builtin_bit_cast(uint2, wo);
        const auto r0 = __builtin_amdgcn_permlane32_swap(ue.x, uo.x, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(ue.y, uo.y, false, false);
        if (valid) *reinterpret_cast<uint4*>(dst) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
    if (lo_off) {                            // uniform over the launch
        const uint2 ue = __builtin_bit_cast(uint2, pack4_lo<F16>(e0, e1, e2, e3, we, lo_scale)),
                    uo = __builtin_bit_cast(uint2, pack4_lo<F16>(f0, f1, f2, f3, wo, lo_scale));
        const auto r0 = __builtin_amdgcn_permlane32_swap(ue.x, uo.x, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(ue.y, uo.y, false, false);
        if (valid) *reinterpret_cast<uint4*>(dst + lo_off) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

template <bool F16, int QB, int NW>
__device__ __forceinline__ bool attn2_body(char* __restrict__ smem, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                           const bf16_t* __restrict__ v, bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad,
                                           int bh, int blk0, int nblk, int out_stride, int out_ld, int lo_off, float lo_scale) {
    constexpr int PCS = 8 / NW;                      // 1 KiB pieces of a K (and of a V) tile per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const bool active = wave < nblk;
    const int q0 = (blk0 + wave) * (32 * QB);
    const int b = bh / heads, head = bh - b * heads;

    // ---- LDS-DMA sources: piece = 8 key rows x 128 B; lane l fills position l & 7 of row l >> 3 with the chunk the image wants there
    const bf16_t* kg[PCS];
    const bf16_t* vg[PCS];
    int dst[PCS];
#pragma unroll
    for (int pc = 0; pc < PCS; ++pc) {
        const int piece = wave + NW * pc;
        const int row = piece * 8 + (lane >> 3);
        kg[pc] = k + ((size_t)bh * tokens_pad + row) * HD + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
        vg[pc] = v + ((size_t)bh * tokens_pad + row) * HD + (((lane & 7) ^ (((row >> 1) & 1) << 2)) * 8);
        dst[pc] = piece * 1024;
    }
    auto stage_k = [&](int t) __attribute__((always_inline)) {                      // tile t -> K slot t % 3
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) glds16(kg[pc] + (size_t)t * (KV * HD), smem + (t % NSK) * TILE + dst[pc]);
    };
    auto stage_v = [&](int t) __attribute__((always_inline)) {                      // tile t -> V slot t % 3
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) glds16(vg[pc] + (size_t)t * (KV * HD), smem + V_BASE + (t % NSV) * TILE + dst[pc]);
    };

    // ---- Q^T fragments (B operand): lane (query r, half h), k-step s: d = 16 s + 8 h .. + 7
    bf16x8 qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qrow = q0 + 32 * qb + r;
        qrow = qrow < tokens_pad ? qrow : tokens_pad - 1;
        const bf16_t* qp = q + ((size_t)bh * tokens_pad + qrow) * HD + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    const int nkv = tokens_pad / KV;
    stage_k(0);                                      // issue order K0 V0 K1 | step 0: V1 K2 | step 1: V2 K3 | ...: at the top of step t the
    stage_v(0);                                      // youngest requests are K(t + 1)'s, everything older -- K(t), V(t) -- is what vmcnt(PCS) waits for
    if (nkv > 1) stage_k(1);

    // ---- fragment addresses inside a tile
    int ka[4];                                       // K row r, logical chunk 2 s + h
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) * 16);
    int va[2];                                       // V: lane 4 qd + p of a 16-lane group -> row 4 h + qd, columns 16 dgrp + 4 p .. + 3 (+ 32 blk)
    {
        const int l16 = lane & 15, qd = l16 >> 2, p = l16 & 3, dgrp = (lane >> 4) & 1;
        const int x = qd >> 1;                       // ((row >> 1) & 1) for row = 4 h + qd (+ multiples of 8)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) va[blk] = (4 * h + qd) * 128 + (2 * dgrp + (p >> 1) + 4 * (blk ^ x)) * 16 + (p & 1) * 8;
    }

    f32x16 o[QB][2], sacc[QB][2];
    bf16x8 pf[QB][2][2];
    float l_run[QB], m_ref[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        l_run[qb] = 0.f;
        m_ref[qb] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][blk][i] = 0.f;
    }

    const int tail_keys = tokens - (nkv - 1) * KV;                 // valid keys of the last tile, 1 .. 64
    const bool half_tail = tail_keys <= 32;                         // the last tile's second 32-key half holds no valid key: skipped altogether
    using T_ = std::true_type;
    using F_ = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // S^T of tile t (K slot t % 3), 32-key half g.  LAST: keys >= tokens start at -inf (the MFMA carries it through)
    auto s_half = [&](const char* kt, int t, auto g_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        constexpr int g = decltype(g_c)::value;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (LAST) sacc[qb][g][i] = (t * KV + g * 32 + crow(i, h) >= tokens) ? -INFINITY : 0.f;
                else sacc[qb][g][i] = 0.f;
            }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + g * 4096 + ka[s]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) sacc[qb][g] = mfma_32x32x16<F16>(kf, qf[qb][s], sacc[qb][g]);
        }
    };
    // O^T += V^T P^T for the 32-key half g of the V tile
    auto pv_half = [&](const char* vt, auto g_c) __attribute__((always_inline)) {
        constexpr int g = decltype(g_c)::value;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const char* p0 = vt + va[blk] + (32 * g + 16 * s2) * 128;
                const bf16x4 lo = tr_read(p0);
                const bf16x4 hi = tr_read(p0 + 8 * 128);
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) o[qb][blk] = mfma_32x32x16<F16>(vf, pf[qb][g][s2], o[qb][blk]);
            }
    };
    // softmax of the 32-key halves [0, GH) of sacc -> pf
    auto softmax = [&](auto first_c, auto gh_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int GH = decltype(gh_c)::value;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            {
                if constexpr (F16 && FIRST) {
                    float mx = sacc[qb][0][0];
#pragma unroll
                    for (int g = 0; g < GH; ++g)
#pragma unroll
                        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[qb][g][i]);
                    m_ref[qb] = fmaxf(mx, __shfl_xor(mx, 32)) + (float)HIPTS_ATTN_REF_MARGIN;      // finite: tile 0 holds at least one unmasked key
                }
                float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
                for (int g = 0; g < GH; ++g) {
                    bf16x8 w0, w1;                    // whole-vector writes into pf: element-wise writes sent the array to scratch memory (QB = 2)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float p0 = __builtin_amdgcn_exp2f(F16 ? sacc[qb][g][i] - m_ref[qb] : sacc[qb][g][i]);
                        const float p1 = __builtin_amdgcn_exp2f(F16 ? sacc[qb][g][8 + i] - m_ref[qb] : sacc[qb][g][8 + i]);
                        if (i & 1) ls1 += p0 + p1; else ls0 += p0 + p1;
                        w0[i] = to_op<F16>(p0);
                        w1[i] = to_op<F16>(p1);
                    }
                    pf[qb][g][0] = w0;
                    pf[qb][g][1] = w1;
                }
                l_run[qb] += ls0 + ls1;
            }
        }
    };
    // One tile step.  K(t) and V(t) have landed (every wave waited for its own pieces, then the barrier); V(t + 1) and K(t + 2) are
    // requested into the slots that V(t - 2) and K(t - 1) were last read from before this barrier; then S(t), P V of tile t - 1, softmax(t).
    auto step = [&](int t, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        if constexpr (LAST) wait_vm<0>();
        else wait_vm<PCS>();                         // leaves K(t + 1) in flight
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nkv) stage_v(t + 1);
        if (t + 2 < nkv) stage_k(t + 2);
        if (!active) return;
        const char* kt = smem + (t % NSK) * TILE;
        const char* vt = smem + V_BASE + ((t + NSV - 1) % NSV) * TILE;
        s_half(kt, t, I0{}, last_c);
        if constexpr (QB > 1) __builtin_amdgcn_sched_barrier(0);
        if (!(LAST && half_tail)) s_half(kt, t, I1{}, last_c);
        if constexpr (QB > 1) __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FIRST) {
            pv_half(vt, I0{});
            if constexpr (QB > 1) __builtin_amdgcn_sched_barrier(0);
            pv_half(vt, I1{});
            if constexpr (QB > 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (LAST && half_tail) softmax(first_c, std::integral_constant<int, 1>{});
        else softmax(first_c, std::integral_constant<int, 2>{});
    };

    if (nkv == 1) step(0, T_{}, T_{});
    else {
        step(0, T_{}, F_{});
        for (int t = 1; t + 1 < nkv; ++t) step(t, F_{}, F_{});
        step(nkv - 1, F_{}, T_{});
    }
    if (!active) return false;
    {
        const char* vt = smem + V_BASE + ((nkv - 1) % NSV) * TILE;
        pv_half(vt, I0{});
        if (!half_tail) pv_half(vt, I1{});
    }

    // ---- normalise and store: out[(b * out_stride + query)][head * 64 + d]; o[qb][blk][i] = (query q0 + 32 qb + r, d = 32 blk + crow(i, h))
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        {   // 0.5 <= l < 2^15 (half) / 2^-100 <= l < 2^100 (bf16) on the bit pattern: a NaN or a negative value fails whatever -fno-honor-nans assumes
            const uint32_t lb = __float_as_uint(l_tot);
            constexpr uint32_t LO = F16 ? 0x3f000000u - ((uint32_t)HIPTS_ATTN_REF_MARGIN << 23) : 0x0d800000u, HI = F16 ? 0x47000000u : 0x71800000u;
            if (!(lb >= LO && lb < HI)) bad = true;
        }
        l_run[qb] = 1.0f / l_tot;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0) return true;      // wave-uniform: the lane swaps below need every lane
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float inv = l_run[qb];
        const int qi = q0 + 32 * qb + r;
        bf16_t* op = out + ((size_t)b * out_stride + qi) * out_ld + head * HD;
        // quad kq (0 .. 7) of this lane = registers 4 (kq & 3) .. + 3 of blk = kq >> 2 -> d = 32 blk + 8 (kq & 3) + 4 h + (0 .. 3): the two
        // halves of a query's 16 B chunk sit in lanes r and r + 32.  Swap so that lane h = 0 holds whole chunks of the even quads and
        // lane h = 1 of the odd ones (cdna_hip_programming.md T21), then 4 stores of 16 B per lane instead of 8 of 8 B.
        // v_permlane32_swap a, b: a.lanes[32..63] <-> b.lanes[0..31].  With a = the even quad's word and b = the odd quad's:
        // afterwards lane h = 0 holds (a = its own even word, b = its partner's even word) = d + 0..3 | d + 4..7 of the even chunk,
        // lane h = 1 holds (a = its partner's odd word, b = its own odd word) = the odd chunk, in that order as well.
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) store_o_pair<F16>(o[qb][(2 * kp) >> 2], o[qb][(2 * kp + 1) >> 2], kp, inv, h, qi < tokens, op, lo_off, lo_scale);
    }
    return false;
}

// The same pass with the least state per wave (MODE 1): the two 32-key halves of a tile go S -> softmax -> P V one after the other, so a
// wave holds one 32 x 32 score block instead of two plus a lagging P (about 100 registers instead of 150-160): four waves per SIMD, and
// the overlap of one wave's softmax with another's MFMAs is left entirely to the hardware.  Ring: two K and two V slots (one tile ahead).
template <bool F16, int NW>
__device__ __forceinline__ bool attn2_seq_body(char* __restrict__ smem, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                               const bf16_t* __restrict__ v, bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad,
                                               int bh, int blk0, int nblk, int out_stride, int out_ld, int lo_off, float lo_scale) {
    constexpr int PCS = 8 / NW;
    constexpr int SNS = HIPTS_ATTN2_SEQ_SLOTS;       // ring slots per operand: 2 = one tile in flight, 3 = two
    constexpr int VB = SNS * TILE;                   // V slots behind the K slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: LDS-DMA bases and the active test stay scalar
    const int r = lane & 31, h = lane >> 5;
    const bool active = wave < nblk;
    const int q0 = (blk0 + wave) * 32;
    const int b = bh / heads, head = bh - b * heads;
    const int nkv = tokens_pad / KV;

#ifdef HIPTS_X_STAMPS
    const bool stamp_on = blockIdx.x == HIPTS_X_STAMPS && wave == 0;
#endif
    unsigned kgo[PCS], vgo[PCS];                     // element offsets of this lane's DMA sources inside a tile
#pragma unroll
    for (int pc = 0; pc < PCS; ++pc) {
        const int row = (wave + NW * pc) * 8 + (lane >> 3);
        kgo[pc] = row * HD + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
        vgo[pc] = row * HD + (((lane & 7) ^ (((row >> 1) & 1) << 2)) * 8);
    }
    const bf16_t* kb = k + (size_t)bh * tokens_pad * HD;
    const bf16_t* vb = v + (size_t)bh * tokens_pad * HD;
    auto stage = [&](int t) __attribute__((always_inline)) {
        const int sl = (t % SNS) * TILE;
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) glds16(kb + (size_t)t * (KV * HD) + kgo[pc], smem + sl + (wave + NW * pc) * 1024);
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) glds16(vb + (size_t)t * (KV * HD) + vgo[pc], smem + VB + sl + (wave + NW * pc) * 1024);
    };
    bf16x8 qf[4];
    {
        int qrow = q0 + r;
        qrow = qrow < tokens_pad ? qrow : tokens_pad - 1;
        const bf16_t* qp = q + ((size_t)bh * tokens_pad + qrow) * HD + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    stage(0);
    if (SNS == 3 && nkv > 1) stage(1);
    int ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) * 16);
    int va[2];
    {
        const int l16 = lane & 15, qd = l16 >> 2, p = l16 & 3, dgrp = (lane >> 4) & 1;
        const int x = qd >> 1;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) va[blk] = VB + (4 * h + qd) * 128 + (2 * dgrp + (p >> 1) + 4 * (blk ^ x)) * 16 + (p & 1) * 8;
    }
    f32x16 o[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[blk][i] = 0.f;
    float l_run = 0.f, m_ref = 0.f;
#if HIPTS_ATTN_SAMPLED_REF
    if constexpr (F16) {
        // The reference exponent of the half-operand fast path: this row's maximum over 32 keys SAMPLED across the sequence (one extra
        // 32 x 32 score block, the K rows straight from memory in the A-operand layout) + the head room.  The first keys' maximum was
        // the bar of a padded picture; a sample sees what the row will meet, so the head room can be smaller and fewer P fall below
        // half's normal range (the MFMA reads half subnormals as zero).
        const int stride = tokens >= 32 ? tokens / 32 : 1;
        int krow = r * stride + (stride >> 1);
        krow = krow < tokens ? krow : tokens - 1;
        const bf16_t* kp = kb + (size_t)krow * HD + 8 * h;
        f32x16 sref;
#pragma unroll
        for (int i = 0; i < 16; ++i) sref[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) sref = mfma_32x32x16<F16>(*reinterpret_cast<const bf16x8*>(kp + 16 * s), qf[s], sref);
        float mx = sref[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sref[i]);
        m_ref = fmaxf(mx, __shfl_xor(mx, 32)) + (float)HIPTS_ATTN_REF_MARGIN;
    }
#endif
#if HIPTS_ATTN2_MFMA_SUM
    f32x16 lacc;                                     // every row: the sum over the keys of P[key][query r]
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = to_op<F16>(1.0f);
#endif
    const int tail_keys = tokens - (nkv - 1) * KV;
    using T_ = std::true_type;
    using F_ = std::false_type;

    // A tile's two 32-key halves go S -> softmax -> P V one after the other.  LDS reads are requested in groups, ahead of their use
    // (left to itself hipcc reads one fragment, waits, multiplies, reads the next: the matrix pipe idles ~100 cycles per MFMA): the four
    // K fragments of a half together; the eight V^T pieces of a half right after its S MFMAs, so that they land under the exponentials;
    // and the K fragments of the SECOND half before the softmax of the first, so that S of the second half starts without an LDS wait.
    auto k_reads = [&](int sl, int g, bf16x8 (&kf)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(smem + sl + g * 4096 + ka[s]);
    };
    auto v_reads = [&](int sl, int g, bf16x8 (&vf)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const char* p0 = smem + sl + va[blk] + (32 * g) * 128;
            const bf16x4 a0 = tr_read(p0), a1 = tr_read(p0 + 8 * 128), b0 = tr_read(p0 + 16 * 128), b1 = tr_read(p0 + 24 * 128);
            vf[blk][0] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            vf[blk][1] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    // half operands: the accumulator starts at -m_ref, so S - m_ref comes out of the MFMAs (hipcc keeps the 16-register tuple across the
    // loop; 32 v_sub per tile otherwise).  The very first half (FIRST) defines m_ref and subtracts explicitly.
    auto s_mfma = [&](const bf16x8 (&kf)[4], int t, int g, f32x16& sacc, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        constexpr bool PRESUB = F16 && (!FIRST || HIPTS_ATTN_SAMPLED_REF);
        const float c0 = PRESUB ? -m_ref : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (LAST) sacc[i] = (t * KV + g * 32 + crow(i, h) >= tokens) ? -INFINITY : c0;
            else sacc[i] = c0;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc = mfma_32x32x16<F16>(kf[s], qf[s], sacc);
    };
    auto soft = [&](const f32x16& sacc, bf16x8& w0, bf16x8& w1, auto first_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr bool PRESUB = F16 && (!FIRST || HIPTS_ATTN_SAMPLED_REF);
        if constexpr (F16 && FIRST && !HIPTS_ATTN_SAMPLED_REF) {                // the reference: this row's maximum over the first 32 keys
            float mx = sacc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sacc[i]);
            m_ref = fmaxf(mx, __shfl_xor(mx, 32)) + (float)HIPTS_ATTN_REF_MARGIN;
        }
        float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#ifdef HIPTS_X_NOEXP
            const float p0 = (F16 && !PRESUB) ? sacc[i] - m_ref : sacc[i], p1 = (F16 && !PRESUB) ? sacc[8 + i] - m_ref : sacc[8 + i];
#else
            const float p0 = __builtin_amdgcn_exp2f((F16 && !PRESUB) ? sacc[i] - m_ref : sacc[i]);
            const float p1 = __builtin_amdgcn_exp2f((F16 && !PRESUB) ? sacc[8 + i] - m_ref : sacc[8 + i]);
#endif
#if HIPTS_ATTN2_MFMA_SUM == 1
#elif !defined(HIPTS_X_NOSUM)
#if HIPTS_ATTN2_ASM_ADD
            asm("v_add_f32 %0, %0, %1" : "+v"(ls0) : "v"(p0));      // single adds: hipcc pairs plain ones into v_pk_add_f32, which costs more issue cycles than two v_add_f32
            asm("v_add_f32 %0, %0, %1" : "+v"(ls1) : "v"(p1));
#else
            if (i & 1) ls1 += p0 + p1; else ls0 += p0 + p1;
#endif
#endif
            w0[i] = to_op<F16>(p0);
            w1[i] = to_op<F16>(p1);
        }
#ifdef HIPTS_X_NOSUM
        ls0 = sacc[0];
#endif
        l_run += ls0 + ls1;
    };
    auto pv = [&](const bf16x8 (&vf)[2][2], const bf16x8& w0, const bf16x8& w1) __attribute__((always_inline)) {
#if HIPTS_ATTN2_PV_SNAKE      // every MFMA shares an operand with its predecessor (gemm.hip: HIPTS_MFMA_ORDER); same chain per accumulator, same bits
        o[0] = mfma_32x32x16<F16>(vf[0][0], w0, o[0]);
        o[1] = mfma_32x32x16<F16>(vf[1][0], w0, o[1]);
        o[1] = mfma_32x32x16<F16>(vf[1][1], w1, o[1]);
        o[0] = mfma_32x32x16<F16>(vf[0][1], w1, o[0]);
#else
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            o[blk] = mfma_32x32x16<F16>(vf[blk][0], w0, o[blk]);
            o[blk] = mfma_32x32x16<F16>(vf[blk][1], w1, o[blk]);
        }
#endif
#if HIPTS_ATTN2_MFMA_SUM
        lacc = mfma_32x32x16<F16>(ones, w0, lacc);
        lacc = mfma_32x32x16<F16>(ones, w1, lacc);
#endif
    };
    auto step = [&](int t, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_c)::value;
        HIPTS_STAMP(t * 16 + 0);
        // tile t (requested SNS - 1 steps ago) has landed for this wave; with three slots tile t + 1's pieces stay in flight ...
        if (SNS == 3 && t + 1 < nkv) wait_vm<2 * PCS>();
        else wait_vm<0>();
        HIPTS_STAMP(t * 16 + 5);
#ifndef HIPTS_X_NOBARRIER                            // (HIPTS_X_*: measurement-only builds, tools/gpurun/r3_attn_x.sh -- results are wrong with them)
        __builtin_amdgcn_s_barrier();                // ... and for every wave; tile t - 1's slots are free
#endif
        HIPTS_STAMP(t * 16 + 6);
#ifdef HIPTS_X_NODMA
        if (t == 0)
#endif
        if (t + SNS - 1 < nkv) stage(t + SNS - 1);
        if (!active) return;
        HIPTS_STAMP(t * 16 + 1);
        int sl = (t % SNS) * TILE;
        if (SNS == 3) asm volatile("" : "+s"(sl));      // a per-step scalar: three hoisted sets of LDS addresses would not fit the register budget
        const bool two = !(LAST && tail_keys <= 32);
        bf16x8 kf0[4], kf1[4], vf[2][2], w0, w1;
        f32x16 sacc;
        k_reads(sl, 0, kf0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // 4 DS reads
        s_mfma(kf0, t, 0, sacc, first_c, last_c);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);       // 4 MFMA
        HIPTS_STAMP(t * 16 + 2);
        v_reads(sl, 0, vf);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);       // 8 DS reads (transposed)
#ifndef HIPTS_ATTN2_NO_KPREFETCH
        if (two) {
            k_reads(sl, 1, kf1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
#endif
        soft(sacc, w0, w1, first_c);
        HIPTS_STAMP(t * 16 + 3);
        pv(vf, w0, w1);
        HIPTS_STAMP(t * 16 + 4);
        if (two) {
#ifdef HIPTS_ATTN2_NO_KPREFETCH
            k_reads(sl, 1, kf1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#endif
            s_mfma(kf1, t, 1, sacc, F_{}, last_c);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            HIPTS_STAMP(t * 16 + 10);
            v_reads(sl, 1, vf);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            soft(sacc, w0, w1, F_{});
            HIPTS_STAMP(t * 16 + 11);
            pv(vf, w0, w1);
            HIPTS_STAMP(t * 16 + 12);
        }
    };
    if (nkv == 1) step(0, T_{}, T_{});
    else {
        step(0, T_{}, F_{});
        for (int t = 1; t + 1 < nkv; ++t) step(t, F_{}, F_{});
        step(nkv - 1, F_{}, T_{});
    }
    if (!active) return false;

#if HIPTS_ATTN2_MFMA_SUM == 3
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (blockIdx.x < 2 && wave == 0 && (lane == 0 || lane == 1 || lane == 33)) printf("blk %d lane %d: adds %.9g  mfma %.9g %.9g %.9g\n", (int)blockIdx.x, lane, l_tot, lacc[0], lacc[5], lacc[15]);
#elif HIPTS_ATTN2_MFMA_SUM
    const float l_tot = lacc[0];                     // over all keys already (the MFMA sums the two lane halves' keys)
#else
    const float l_tot = l_run + __shfl_xor(l_run, 32);
#endif
    bool bad;
    {
        const uint32_t lb = __float_as_uint(l_tot);
        constexpr uint32_t LO = F16 ? 0x3f000000u - ((uint32_t)HIPTS_ATTN_REF_MARGIN << 23) : 0x0d800000u, HI = F16 ? 0x47000000u : 0x71800000u;
        bad = !(lb >= LO && lb < HI);
    }
#ifndef HIPTS_X_NOFALLBACK
    if (__builtin_amdgcn_ballot_w64(bad) != 0) return true;
#endif
    const float inv = 1.0f / l_tot;
    const int qi = q0 + r;
    bf16_t* op = out + ((size_t)b * out_stride + qi) * out_ld + head * HD;
#pragma unroll
    for (int kp = 0; kp < 4; ++kp) store_o_pair<F16>(o[(2 * kp) >> 2], o[(2 * kp + 1) >> 2], kp, inv, h, qi < tokens, op, lo_off, lo_scale);      // see attn2_body
    return false;
}

// The fallback: online softmax with the per-tile running maximum (cannot overflow), for the query blocks of a workgroup whose fast
// pass reported a row sum outside its window.  Rare, so it is written for few registers, not speed -- the kernel's register
// allocation is the larger of the two bodies': one block of 32 query rows per wave and pass (QB passes over the keys), one tile
// in flight, S -> softmax -> P V in order.  Same fragment layouts, LDS images and store path as attn2_body.
template <bool F16, int QB, int NW>
__device__ __forceinline__ void attn2_classic(char* __restrict__ smem, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                           const bf16_t* __restrict__ v, bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad, int bh,
                                           int blk0, int nblk, int out_stride, int out_ld, int lo_off, float lo_scale, int wave_blk0 = -1,
                                           int wave_nq = 0) {
    // wave_blk0 >= 0 (attn3_kernel): this wave's blocks are wave_blk0 .. wave_blk0 + wave_nq - 1, one per pass
    constexpr int PCS = 8 / NW;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));      // opaque: the lane constants of this rare pass are computed here, not hoisted to the kernel's entry and kept alive
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = bh / heads, head = bh - b * heads;
    const int nkv = tokens_pad / KV;
    int ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) * 16);
    int va[2];
    {
        const int l16 = lane & 15, qd = l16 >> 2, p = l16 & 3, dgrp = (lane >> 4) & 1;
        const int x = qd >> 1;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) va[blk] = (4 * h + qd) * 128 + (2 * dgrp + (p >> 1) + 4 * (blk ^ x)) * 16 + (p & 1) * 8;
    }
    for (int qb = 0; qb < QB; ++qb) {
        const int q0 = wave_blk0 >= 0 ? (wave_blk0 + qb) * 32 : (blk0 + wave) * (32 * QB) + 32 * qb;
        const bool active = wave_blk0 >= 0 ? qb < wave_nq : wave < nblk;
        int qrow = q0 + r;
        qrow = qrow < tokens_pad ? qrow : tokens_pad - 1;
        const bf16_t* qp = q + ((size_t)bh * tokens_pad + qrow) * HD + 8 * h;
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
        f32x16 o[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[blk][i] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        for (int t = 0; t < nkv; ++t) {
            __syncthreads();                          // every wave is done with the slot (also: with the fast pass's ring)
#pragma unroll
            for (int pc = 0; pc < PCS; ++pc) {
                const int piece = wave + NW * pc;
                const int row = piece * 8 + (lane >> 3);
                const size_t g0 = ((size_t)bh * tokens_pad + (size_t)t * KV + row) * HD;
                glds16(k + g0 + (((lane & 7) ^ ((row >> 1) & 7)) * 8), smem + piece * 1024);
                glds16(v + g0 + (((lane & 7) ^ (((row >> 1) & 1) << 2)) * 8), smem + V_BASE + piece * 1024);
            }
            wait_vm<0>();
            __syncthreads();
            if (!active) continue;
            f32x16 sacc[2];
            bf16x8 pf[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc[g][i] = (t * KV + g * 32 + crow(i, h) >= tokens) ? -INFINITY : 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + g * 4096 + ka[s]);
                    sacc[g] = mfma_32x32x16<F16>(kf, qf[s], sacc[g]);
                }
            }
            float mx = m_run;                         // every tile holds at least one unmasked key: the maximum is finite from tile 0 on
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[g][i]);
            const float m_new = fmaxf(mx, __shfl_xor(mx, 32));
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ls = 0.f;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(sacc[g][i] - m_new);
                    ls += p;
                    pf[g][i >> 3][i & 7] = to_op<F16>(p);
                }
            l_run = l_run * alpha + ls;
            m_run = m_new;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int i = 0; i < 16; ++i) o[blk][i] *= alpha;
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const char* p0 = smem + V_BASE + va[blk] + (32 * g + 16 * s2) * 128;
                        const bf16x4 lo = tr_read(p0);
                        const bf16x4 hi = tr_read(p0 + 8 * 128);
                        const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[blk] = mfma_32x32x16<F16>(vf, pf[g][s2], o[blk]);
                    }
            }
        }
        if (!active) continue;
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + r;
        bf16_t* op = out + ((size_t)b * out_stride + qi) * out_ld + head * HD;
        if (qi < tokens) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float x0 = o[blk][4 * g4] * inv, x1 = o[blk][4 * g4 + 1] * inv, x2 = o[blk][4 * g4 + 2] * inv, x3 = o[blk][4 * g4 + 3] * inv;
                    const bf16x4 hi = pack4<F16>(x0, x1, x2, x3);
                    *reinterpret_cast<bf16x4*>(op + blk * 32 + 8 * g4 + 4 * h) = hi;
                    if (lo_off) *reinterpret_cast<bf16x4*>(op + lo_off + blk * 32 + 8 * g4 + 4 * h) = pack4_lo<F16>(x0, x1, x2, x3, hi, lo_scale);
                }
        }
    }
}

template <bool F16, int QB, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, MODE == 1 ? HIPTS_ATTN2_SEQ_WAVES : 2) void attn2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                        bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad, int chunks, int out_stride,
                                                        int classic, int out_ld, int lo_off, float lo_scale) {
    // ALL of the kernel's LDS is this one array (ring + the fallback flag word): with a second LDS object in the kernel -- __syncthreads_or()
    // brings one -- hipcc puts an s_waitcnt vmcnt(0) in front of the first ds_read of every tile step and the DMA ring never runs ahead
    // (cdna_hip_programming.md section 5, "three .s-level traps" (a); seen in this kernel's .s).
    constexpr int RING = MODE == 1 ? 2 * HIPTS_ATTN2_SEQ_SLOTS * TILE : LDS_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[RING + 16];
    int* redo = reinterpret_cast<int*>(smem + RING);
    if (threadIdx.x == 0) *redo = classic;           // ordered before every reader by the barriers of the tile loop
    // XCD-aware work id (attn.hip): the workgroups of one (image, head) stream the same K / V -- keep them on one XCD's L2
    int wid = blockIdx.x;
    {
        const int nwg = gridDim.x, qd = nwg >> 3, rm = nwg & 7, xcd = wid & 7, loc = wid >> 3;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    const int bh = wid / chunks, c = wid - bh * chunks;
    const int nb = (tokens + 32 * QB - 1) / (32 * QB);              // query blocks of the (image, head), dealt evenly to its workgroups
    const int blk0 = (c * nb) / chunks, nblk = ((c + 1) * nb) / chunks - blk0;
    // classic != 0 (HIPTS_ATTN_CLASSIC=1): the fallback everywhere (A/B runs, its own test)
    if (!classic) {
        bool bad;
        if constexpr (MODE == 1) bad = attn2_seq_body<F16, NW>(smem, q, k, v, out, heads, tokens, tokens_pad, bh, blk0, nblk, out_stride, out_ld, lo_off, lo_scale);
        else bad = attn2_body<F16, QB, NW>(smem, q, k, v, out, heads, tokens, tokens_pad, bh, blk0, nblk, out_stride, out_ld, lo_off, lo_scale);
        if (bad) *redo = 1;
    }
    __syncthreads();
    // a wave whose row sum left the window stored nothing; the workgroup (its waves stage K / V together) repeats its blocks classically
    if (*redo) attn2_classic<F16, QB, NW>(smem, q, k, v, out, heads, tokens, tokens_pad, bh, blk0, nblk, out_stride, out_ld, lo_off, lo_scale);
}

#include "attn3.h"

// The fallback as a function of its own: inlined into the persistent kernel its registers took part in the item loop's allocation and the
// two hand-placed streams were allocated around them (AGPR spills to scratch in front of each stream).
template <bool F16>
__device__ __noinline__ void attn3_classic(char* smem, const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int heads, int tokens, int tokens_pad,
                                           int bh, int blk0, int nblk, int out_stride, int out_ld, int lo_off, float lo_scale, int wb0, int nq) {
    attn2_classic<F16, A3_MAXQB, 4>(smem, q, k, v, out, heads, tokens, tokens_pad, bh, blk0, nblk, out_stride, out_ld, lo_off, lo_scale, wb0, nq);
}

template <bool F16>
__global__ __launch_bounds__(256, 1) void attn3_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                       bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad, int chunks, int nitems,
                                                       int out_stride, int classic, int out_ld, int lo_off, float lo_scale) {
    // Persistent: one workgroup per CU walks the items (an item = one chunk of query blocks of one (image, head)); the K / V ring runs on
    // across the items -- the last tile steps of an item stage the first tiles of the next.
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    int* redo = reinterpret_cast<int*>(smem3 + A3_LDS);
    if (threadIdx.x == 0) *redo = classic;
    const unsigned lds0 = (unsigned)(uintptr_t)smem3;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb = (tokens + 31) / 32, nkv = tokens_pad / KV;
    // XCD-aware item -> work id (attn.hip): the chunks of one (image, head) stream the same K / V -- same XCD, same round.  The chunk roles
    // rotate with the round, so that the workgroup that had the larger chunk (a wave with four blocks) gets a smaller one next.
    auto item_bh = [&](int item, int round, int& c) __attribute__((always_inline)) -> int {
        const int qd = nitems >> 3, rm = nitems & 7, xcd = item & 7, loc = item >> 3;
        const int wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
        const int bh = wid / chunks;
        c = (wid - bh * chunks + round) % chunks;
        return bh;
    };
    int gbase = 0;
    bool fresh = true;
    int round = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++round) {
        int c, c2;
        const int bh = item_bh(item, round, c);
        const int bh2 = item + (int)gridDim.x < nitems ? item_bh(item + gridDim.x, round + 1, c2) : -1;
        const int blk0 = (c * nb) / chunks, nblk = ((c + 1) * nb) / chunks - blk0;       // <= A3_WG_BLOCKS
        const int wb0 = blk0 + (wave * nblk) / 4, nq = blk0 + ((wave + 1) * nblk) / 4 - wb0;      // 13 -> 3 3 3 4
        if (!classic) {
            bool bad;
            if (nq == A3_MAXQB)
                bad = attn3_body<F16, 4, false>(lds0, q, k, v, out, heads, tokens, tokens_pad, bh, wb0, nq, bh2, gbase, fresh, out_stride, out_ld, lo_off, lo_scale);
            else
                bad = attn3_body<F16, 3, F16>(lds0, q, k, v, out, heads, tokens, tokens_pad, bh, wb0, nq, bh2, gbase, fresh, out_stride, out_ld, lo_off, lo_scale);
            if (bad) *redo = 1;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(*redo) != 0) {             // a wave whose row sum left the window stored nothing; the workgroup repeats the item classically
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the staging of the next item may still be in flight: the fallback reuses the ring
            __syncthreads();
            attn3_classic<F16>(smem3, q, k, v, out, heads, tokens, tokens_pad, bh, blk0, nblk, out_stride, out_ld, lo_off, lo_scale, wb0, nq);
            __syncthreads();
            if (threadIdx.x == 0) *redo = classic;
            fresh = true;
        } else {
            fresh = false;
        }
        gbase += nkv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool F16>
int launch_attn3(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int batch, int heads, int tokens, int tokens_pad, int ost, int classic,
                 int out_ld, int lo_off, float lo_scale, hipStream_t s) {
    static PerDevice attr;
    int dev = 0;
    const int cus = current_device_cus(&dev);
    {
        std::lock_guard<std::mutex> lk(attr.mu);
        if (!attr.done(dev)) {
            HIPTS_HIP(hipFuncSetAttribute((const void*)attn3_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, A3_LDS + 16));
            attr.mark(dev);
        }
    }
    const int nb = (tokens + 31) / 32;
    const int chunks = (nb + A3_WG_BLOCKS - 1) / A3_WG_BLOCKS;
    const int nitems = batch * heads * chunks;
    // a grid of whole rounds: a multiple of 8 * chunks workgroups (the chunks of an (image, head) sit 8 items apart, on one XCD)
    static const int env_grid = getenv("HIPTS_ATTN3_GRID") ? atoi(getenv("HIPTS_ATTN3_GRID")) : 0;
    int grid = env_grid > 0 ? env_grid : cus / (8 * chunks) * (8 * chunks);
    if (grid <= 0 || grid > nitems) grid = nitems;
    attn3_kernel<F16><<<grid, 256, A3_LDS + 16, s>>>(q, k, v, out, heads, tokens, tokens_pad, chunks, nitems, ost, classic, out_ld, lo_off, lo_scale);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

template <bool F16, int QB, int NW, int MODE>
int launch_cfg(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int batch, int heads, int tokens, int tokens_pad, int ost,
               int classic, int out_ld, int lo_off, float lo_scale, hipStream_t s) {
    const int nb = (tokens + 32 * QB - 1) / (32 * QB);
    const int chunks = (nb + NW - 1) / NW;
    attn2_kernel<F16, QB, NW, MODE><<<batch * heads * chunks, NW * 64, 0, s>>>(q, k, v, out, heads, tokens, tokens_pad, chunks, ost, classic, out_ld, lo_off, lo_scale);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace

// q, k, v: [batch * heads][tokens_pad][64] 16-bit (q pre-scaled by head_dim^-0.5 log2 e; rows past `tokens` zero), out [batch * out_stride][heads * 64].
// variant: 0 = default; 1: 4 waves x 64 rows, 2: 8 waves x 32 rows, 3: 4 waves x 32 rows (A/B, hiptsdbg_attention2_*).
int launch_attention2(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* out, int batch, int heads, int tokens, int tokens_pad, bool f16,
                      hipStream_t s, int out_tokens_stride, int variant, int split_lo, float lo_scale) {
    const int ost = out_tokens_stride > 0 ? out_tokens_stride : tokens;
    // split_lo: rows of `out` are [hi (heads * 64) | lo (heads * 64)] -- the output as a hi | lo pair of 16-bit halves
    const int out_ld = split_lo ? 2 * heads * HD : heads * HD, lo_off = split_lo ? heads * HD : 0;
    HIPTS_REQUIRE(tokens_pad % KV == 0 && tokens_pad >= tokens && tokens >= 1, "attention: tokens_pad must be a multiple of %d", KV);
    static const int classic = (getenv("HIPTS_ATTN_CLASSIC") && atoi(getenv("HIPTS_ATTN_CLASSIC"))) ? 1 : 0;
    static const int env_variant = getenv("HIPTS_ATTN2") ? atoi(getenv("HIPTS_ATTN2")) : 0;
    if (variant == 0) variant = env_variant ? env_variant : 5;
#define HIPTS_ATTN2_CASE(QB_, NW_, MODE_)                                                                                        \
    return f16 ? launch_cfg<true, QB_, NW_, MODE_>(q, k, v, out, batch, heads, tokens, tokens_pad, ost, classic, out_ld, lo_off, lo_scale, s)                      \
               : launch_cfg<false, QB_, NW_, MODE_>(q, k, v, out, batch, heads, tokens, tokens_pad, ost, classic, out_ld, lo_off, lo_scale, s)
    if (variant == 6 && tokens_pad >= 2 * KV)
        return f16 ? launch_attn3<true>(q, k, v, out, batch, heads, tokens, tokens_pad, ost, classic, out_ld, lo_off, lo_scale, s)
                   : launch_attn3<false>(q, k, v, out, batch, heads, tokens, tokens_pad, ost, classic, out_ld, lo_off, lo_scale, s);
    if (variant == 6) variant = 5;
    switch (variant) {
        case 1: HIPTS_ATTN2_CASE(2, 4, 0);
        case 2: HIPTS_ATTN2_CASE(1, 8, 0);
        case 4: HIPTS_ATTN2_CASE(1, 8, 1);
        case 5: HIPTS_ATTN2_CASE(1, 4, 1);
        default: HIPTS_ATTN2_CASE(1, 4, 0);
    }
#undef HIPTS_ATTN2_CASE
}

#ifdef HIPTS_X_STAMPS
int attention2_read_stamps(unsigned long long* host, int n) {
    HIPTS_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn2_stamps), (size_t)n * 8));
    return HIPTS_OK;
}
#elif defined(HIPTS_A3_STAMPS)
int attention2_read_stamps(unsigned long long* host, int n) {
    HIPTS_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn3_stamps), (size_t)n * 8));
    return HIPTS_OK;
}
#else
int attention2_read_stamps(unsigned long long*, int) { return set_error(HIPTS_ERR_STATE, "built without HIPTS_X_STAMPS"); }
#endif

}  // namespace hipts
