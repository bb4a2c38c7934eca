// attn.hip -- fused softmax(Q K^T) V for the ViT blocks (timm Attention inside tagging.py:174).
//
// Non-causal, no mask, head_dim 64, a few hundred tokens (784 for ViT-B/16 @448).
// One workgroup = 4 waves = 128 query rows of one (image, head); a wave owns 32 query rows.
// Flash-style: K / V^T tiles of 64 keys are staged through LDS (registers -> ds_write, next
// tile's global loads issued before the current tile's math), scores never leave registers.
//
// (q arrives pre-scaled by head_dim^-0.5 * log2 e, so exp2 of the raw scores is the softmax numerator.)
//   S^T = K Q^T      v_mfma_f32_32x32x16_bf16, A = K tile rows (keys), B = Q^T held in registers.
//                    The accumulator then has the QUERY on the lane (column) and 16 of the 32 keys in
//                    its registers, so the online softmax is lane-local (one cross-half exchange).
//   O^T = V^T P^T    the S^T accumulator, converted to bf16, is directly the B operand (no lane
//                    movement); its k order inside a 16-key step is  16s + 8(j>>2) + 4h + (j&3),
//                    and the V^T A-fragment is read from LDS in that same order (two ds_read_b64).
//                    O^T keeps the query on the lane too: rescale and final 1/l are lane-local.
// V arrives already transposed ([head][d][token]) from the QKV GEMM epilogue.
// LDS rows are padded (K: 144 B, V^T: 136 B) so the b128 / b64 fragment reads are conflict free.
// Keys beyond `tokens` (padding up to a multiple of 64) are masked to -inf; padded K / V^T entries
// are zero (buffers are cleared once and the epilogues never write there).
#include "vit_internal.h"

// head room of the half-operand reference exponent: see attn2.hip (a row's first key tile is not representative of the row -- padding,
// background -- and a workgroup that leaves the window runs both passes)
#ifndef HIPTS_ATTN_REF_MARGIN
#define HIPTS_ATTN_REF_MARGIN 10
#endif
namespace hipts {
namespace {

constexpr int KV = 64;               // keys per tile
constexpr int VS = 136;              // LDS bytes per V^T row (64 key* 2 B + 8)
// head_dim HD = 64 (ViT) or 32 (CAFormer): K rows are HD * 2 B + 16 (144 / 80 B: both conflict free for the
// b128 fragment reads), the V^T tile has HD rows.
template <int HD> struct Geo {
    static constexpr int KS = HD * 2 + 16;          // LDS bytes per K row
    static constexpr int K_BYTES = KV * KS;         // 9216 / 5120
    static constexpr int V_BYTES = HD * VS;         // 8704 / 4352
    static constexpr int V_BASE = 2 * K_BYTES;      // LDS: two K slots, then three V^T slots
    static constexpr int LDS_BYTES = 2 * K_BYTES + 3 * V_BYTES;   // 44.5 KB (HD 64): three workgroups per CU
};

// Written with plain fmaxf so the compiler sees the MFMA -> VALU dependency and inserts the required
// wait states itself.  (An inline-asm v_max3_f32 here read accumulator registers before the MFMA had
// retired them: hipcc pads nothing around asm operands -- results differed run to run by a few bf16
// ulps.)  This file is compiled with -fno-honor-nans, which drops the canonicalising v_max that fmaxf
// on MFMA results otherwise costs (45 v_max + 9 v_max3 per tile instead of 17 v_max3); no NaN can
// occur: every tile has at least one unmasked key, so the running maximum is finite.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ int crow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// The per-tile work of one wave (32 queries x 64 keys), split in three so that the loop can be
// software-pipelined:   S(t) MFMAs | P(t-1) V(t-1) MFMAs | softmax(t) on the VALU.
// The second MFMA group keeps the matrix pipe busy while the VALU chews on S(t); by the time the
// softmax wants to rescale O, the P V product it must include has retired.
// q is pre-scaled by head_dim^-0.5 * log2(e) in the QK GEMM epilogue: scores are in the base-2 domain.
// MASK (last tile only): keys >= tokens start their accumulator at -inf, which the MFMA carries through.
// GH: 32-key halves of the tile that hold a valid key (1: the last tile ends inside its first half -- the second half's MFMAs,
// exponentials and P V steps are skipped altogether).
template <bool MASK, bool F16, int HD, int GH = 2>
__device__ __forceinline__ void s_tile(const char* __restrict__ kt, int kv0, int tokens, int r, int h, const bf16x8 (&qf)[HD / 16],
                                       f32x16 (&sacc)[2]) {
    constexpr int KS = Geo<HD>::KS;
#pragma unroll
    for (int g = 0; g < GH; ++g) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (MASK) sacc[g][i] = (kv0 + g * 32 + crow(i, h) >= tokens) ? -INFINITY : 0.f;
            else sacc[g][i] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (g * 32 + r) * KS + (16 * s + 8 * h) * 2);
            sacc[g] = mfma_32x32x16<F16>(kf, qf[s], sacc[g]);
        }
    }
}

template <bool F16, int HD, int GH = 2>
__device__ __forceinline__ void pv_tile(const char* __restrict__ vt, int r, int h, const bf16x8 (&pf)[2][2], f32x16 (&o)[HD / 32]) {
#pragma unroll
    for (int blk = 0; blk < HD / 32; ++blk) {
        const char* vrow = vt + (blk * 32 + r) * VS;
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int key = g * 32 + 16 * s2 + 4 * h;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + key * 2);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + (key + 8) * 2);
                const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[blk] = mfma_32x32x16<F16>(vf, pf[g][s2], o[blk]);
            }
    }
}

// (Moving the row sums onto the matrix pipe -- a fifth 32 x 32 block with the constant A operand "row 0 = ones",
// 4 more MFMAs per tile for 34 fewer v_add_f32 -- was measured slightly SLOWER, 209-212 vs 205 us, and costs
// 20 VGPRs: the loop is not simply VALU-throughput bound.)
// (A lazy reference -- rescale O and l only when some row's tile maximum exceeds the reference by 2^8,
// behind a wave-uniform branch -- was measured SLOWER, 217 vs 196 us: the branch stops the scheduler from
// running this VALU work under the P V MFMAs.)
template <bool F16, int HD>
__device__ __forceinline__ void softmax_tile(const f32x16 (&sacc)[2], bf16x8 (&pf)[2][2], f32x16 (&o)[HD / 32], float& m_run,
                                             float& l_run) {
    float mx = max3f(sacc[0][0], sacc[1][0], m_run);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = max3f(mx, sacc[0][i], sacc[1][i]);
    const float m_new = fmaxf(mx, __shfl_xor(mx, 32));
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(sacc[g][i] - m_new);
            if (i & 1) lsum1 += p; else lsum0 += p;
            pf[g][i >> 3][i & 7] = to_op<F16>(p);
        }
    l_run = l_run * alpha + (lsum0 + lsum1);
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 16; ++i) {       // O already holds every tile before this one (incl. the P V just issued)
#pragma unroll
        for (int blk = 0; blk < HD / 32; ++blk) o[blk][i] *= alpha;
    }
}

// Softmax with a FIXED reference exponent (the default path).  softmax(S) V = (sum_j 2^(S_j - m) V_j) / (sum_j 2^(S_j - m)) for any m,
// so no running maximum is needed as long as nothing leaves the number format: no maximum (17 v_max3), no rescale of O (32 v_mul + an
// exp) per tile.
//  * bf16 operands: m = 0, P = 2^S directly.  bf16 / fp32 carry the same 8 exponent bits; while the row's unnormalised sum l stays
//    inside [2^-100, 2^100] nothing overflowed (P, l, and O = sum P V for |V| < 2^27), and everything that underflowed (S < -126)
//    weighs less than 2^-16 of the row -- below the 2^-9 rounding of P.  A tile is 32 v_exp + 32 v_add + 16 v_cvt_pk per wave:
//    ~460 VALU issue cycles per 64-key tile against 512 of MFMA, where the classic step above costs ~880.
//  * IEEE-half operands (5 exponent bits: 2^S is +inf from S = 16 on, i.e. natural-log scores of 11): m = m_ref, the row's maximum
//    over the FIRST key tile, fixed afterwards; P = 2^(S - m_ref) costs one v_sub per score.  The largest P of tile 0 is exactly 1,
//    so l >= 1 and whatever flushes to zero (below 2^-24 of a row whose sum is >= 1) is under the 2^-11 rounding of P itself.  A
//    later key may exceed the reference, P > 1: while l < 2^15 every P <= l is finite in half.  (Starting the S accumulator at
//    -m_ref would give S - m_ref for free, but the 16-register C tuple it needs takes the kernel from 156 to 172+ VGPRs and the
//    third workgroup off the CU; measured by compiling it.)
// A row whose sum leaves its window -- for half operands a key more than ~15 binary orders above everything among the first 64 keys
// -- is detected at the end (attn_kernel, on the BITS of l so that a NaN cannot slip through -fno-honor-nans) and the workgroup
// repeats the block with the classic per-tile maximum, which cannot overflow.
// (Round 2 ran P = 2^S for half operands too, and its window check on l could not see a P that had already become +inf.)
template <bool F16, int HD, int GH = 2>
__device__ __forceinline__ void softmax_first(const f32x16 (&sacc)[2], bf16x8 (&pf)[2][2], float& l_run, float& m_ref) {
    if constexpr (F16) {
        float mx = sacc[0][0];
#pragma unroll
        for (int g = 0; g < GH; ++g)
#pragma unroll
            for (int i = (g == 0 ? 1 : 0); i < 16; ++i) mx = fmaxf(mx, sacc[g][i]);
        m_ref = fmaxf(mx, __shfl_xor(mx, 32)) + (float)HIPTS_ATTN_REF_MARGIN;          // finite: tile 0 holds at least one unmasked key
    }
    float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
    for (int g = 0; g < GH; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(F16 ? sacc[g][i] - m_ref : sacc[g][i]);
            if (i & 1) lsum1 += p; else lsum0 += p;
            pf[g][i >> 3][i & 7] = to_op<F16>(p);
        }
    l_run += lsum0 + lsum1;
}

template <bool F16, int HD, int GH = 2>
__device__ __forceinline__ void softmax_nomax(const f32x16 (&sacc)[2], bf16x8 (&pf)[2][2], float& l_run, float m_ref) {
    float lsum0 = 0.f, lsum1 = 0.f;
#pragma unroll
    for (int g = 0; g < GH; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(F16 ? sacc[g][i] - m_ref : sacc[g][i]);
            if (i & 1) lsum1 += p; else lsum0 += p;
            pf[g][i >> 3][i & 7] = to_op<F16>(p);
        }
    l_run += lsum0 + lsum1;
}

// CLASSIC: online softmax with the per-tile running maximum (softmax_tile); otherwise softmax_nomax.  Returns true when the
// fast path's row sum left its safe window (nothing is stored then).
template <bool F16, int HD, bool CLASSIC>
__device__ __forceinline__ bool attn_body(char* __restrict__ smem, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                          const bf16_t* __restrict__ vT, bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad,
                                          int qblocks, int out_stride, int trim) {
    using G = Geo<HD>;
    constexpr int KS = G::KS, K_BYTES = G::K_BYTES, V_BYTES = G::V_BYTES, V_BASE = G::V_BASE;
    // iteration t multiplies K(t) (slot t & 1) and V(t-1) (slot (t-1) % 3) while tile t+1 is written: K(t+1)
    // over K(t-1), V(t+1) over V(t-2), both last read before the previous barrier.  Two K and three V^T
    // slots = 44.5 KB, so three workgroups (three waves per SIMD at 148 VGPRs) share a CU and one wave's
    // softmax (VALU) runs under the others' MFMAs.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware work id: workgroups are dealt round-robin over the 8 XCDs (ids equal mod 8 share an
    // L2).  Remap so that consecutive work items -- the query blocks of one (image, head), which all
    // stream the same K / V^T -- run on ONE XCD and hit its L2 instead of each pulling its own copy
    // through the fabric (measured: 1.23 GB fetched per launch vs 0.23 GB algorithmic before this).
    int wid = blockIdx.x;
    {
        const int nwg = gridDim.x, qd = nwg >> 3, rm = nwg & 7, xcd = wid & 7, loc = wid >> 3;
        wid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
    }
    const int bh = wid / qblocks, qb = wid - bh * qblocks;
    const int b = bh / heads, head = bh - b * heads;
    const int q0 = (qb * 4 + wave) * 32;
    // A wave whose 32 query rows all lie past the last token (ViT @448: 784 rows = 24.5 blocks of 32, so three of the 28 waves of an
    // (image, head)) takes its share of the K / V^T staging and every barrier, and nothing else.  (trim = 0: the A/B switch
    // HIPTS_ATTN_TRIM=0 -- every wave and every key half computes, as before.)
    const bool active = !trim || q0 < tokens;
    int qrow = q0 + r;
    qrow = qrow < tokens_pad ? qrow : tokens_pad - 1;

    // Q^T fragments (B operand): lane (q = r, half h), k-step s: d = 16 s + 8 h .. + 7
    bf16x8 qf[HD / 16];
    {
        const bf16_t* qp = q + ((size_t)bh * tokens_pad + qrow) * HD + 8 * h;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }

    // staging: K tile = 64 rows x HD/8 chunks of 16 B, V^T tile = HD rows x 8 chunks; one (HD 32) or two (HD 64)
    // chunks of each per thread.  Plain scalars, no arrays: a per-thread array here is "promoted" to LDS by
    // the compiler (8 KB) and costs the third workgroup per CU.
    constexpr int KCH = HD / 8;
    constexpr bool TWO = HD == 64;
    const int kc0 = tid, kc1 = tid + 256;
    const bf16_t* kp0 = k + (size_t)bh * tokens_pad * HD + (size_t)(kc0 / KCH) * HD + (kc0 % KCH) * 8;
    const bf16_t* kp1 = k + (size_t)bh * tokens_pad * HD + (size_t)(kc1 / KCH) * HD + (kc1 % KCH) * 8;
    const bf16_t* vp0 = vT + (size_t)bh * HD * tokens_pad + (size_t)(kc0 >> 3) * tokens_pad + (kc0 & 7) * 8;
    const bf16_t* vp1 = vT + (size_t)bh * HD * tokens_pad + (size_t)(kc1 >> 3) * tokens_pad + (kc1 & 7) * 8;
    const int kdst0 = (kc0 / KCH) * KS + (kc0 % KCH) * 16, kdst1 = (kc1 / KCH) * KS + (kc1 % KCH) * 16;
    const int vdst0 = V_BASE + (kc0 >> 3) * VS + (kc0 & 7) * 16, vdst1 = V_BASE + (kc1 >> 3) * VS + (kc1 & 7) * 16;
    uint4 kreg0, kreg1 = {}, vreg0, vreg1 = {};
    auto attn_load = [&](int kv0) {
        kreg0 = *reinterpret_cast<const uint4*>(kp0 + (size_t)kv0 * HD);
        vreg0 = *reinterpret_cast<const uint4*>(vp0 + kv0);
        if constexpr (TWO) {
            kreg1 = *reinterpret_cast<const uint4*>(kp1 + (size_t)kv0 * HD);
            vreg1 = *reinterpret_cast<const uint4*>(vp1 + kv0);
        }
    };
    auto attn_write = [&](int ks, int vs) {
        *reinterpret_cast<uint4*>(smem + ks * K_BYTES + kdst0) = kreg0;
        *reinterpret_cast<uint2*>(smem + vs * V_BYTES + vdst0) = make_uint2(vreg0.x, vreg0.y);
        *reinterpret_cast<uint2*>(smem + vs * V_BYTES + vdst0 + 8) = make_uint2(vreg0.z, vreg0.w);
        if constexpr (TWO) {
            *reinterpret_cast<uint4*>(smem + ks * K_BYTES + kdst1) = kreg1;
            *reinterpret_cast<uint2*>(smem + vs * V_BYTES + vdst1) = make_uint2(vreg1.x, vreg1.y);
            *reinterpret_cast<uint2*>(smem + vs * V_BYTES + vdst1 + 8) = make_uint2(vreg1.z, vreg1.w);
        }
    };

    f32x16 o[HD / 32];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int blk = 0; blk < HD / 32; ++blk) o[blk][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    float m_ref = 0.f;                  // fast path: the row's reference exponent, fixed by tile 0 (softmax_first)
    f32x16 sacc[2];
    bf16x8 pf[2][2];

    const int nkv = tokens_pad / KV;
    const int tail_keys = tokens - (nkv - 1) * KV;                 // valid keys of the last tile, 1 .. 64
    const bool masked_tail = tail_keys < KV;
    const bool half_tail = !CLASSIC && trim && tail_keys <= 32;     // (the classic fallback keeps one code path: -inf scores weigh 0)
    // the last tile: unmasked, masked, or masked with only its first 32 keys computed
    auto s_last = [&](const char* kt, int kv0) {
        if (half_tail) s_tile<true, F16, HD, 1>(kt, kv0, tokens, r, h, qf, sacc);
        else if (masked_tail) s_tile<true, F16, HD>(kt, kv0, tokens, r, h, qf, sacc);
        else s_tile<false, F16, HD>(kt, kv0, tokens, r, h, qf, sacc);
    };
    auto softmax_last = [&](bool first) {
        if constexpr (CLASSIC) softmax_tile<F16, HD>(sacc, pf, o, m_run, l_run);
        else if (first) {
            if (half_tail) softmax_first<F16, HD, 1>(sacc, pf, l_run, m_ref);
            else softmax_first<F16, HD>(sacc, pf, l_run, m_ref);
        } else if (half_tail) softmax_nomax<F16, HD, 1>(sacc, pf, l_run, m_ref);
        else softmax_nomax<F16, HD>(sacc, pf, l_run, m_ref);
    };
    attn_load(0);
    attn_write(0, 0);
    __syncthreads();
    // tile 0: scores and softmax only (its P V is issued with the next tile's scores)
    if (nkv > 1) attn_load(KV);
    if (active) {
        if (nkv == 1) {
            s_last(smem, 0);
            softmax_last(true);
        } else {
            s_tile<false, F16, HD>(smem, 0, tokens, r, h, qf, sacc);
            if constexpr (CLASSIC) softmax_tile<F16, HD>(sacc, pf, o, m_run, l_run);
            else softmax_first<F16, HD>(sacc, pf, l_run, m_ref);
        }
    }
    if (nkv > 1) attn_write(1, 1);
    __syncthreads();
    int vprev = 0, vcur = 1;                  // V^T slot of tile t-1 / of tile t
    if (active) {
        // full tiles 1 .. nkv-2: every one of them stages its successor
        for (int t = 1; t + 1 < nkv; ++t) {
            const int vnxt = vcur == 2 ? 0 : vcur + 1;
            attn_load((t + 1) * KV);
            s_tile<false, F16, HD>(smem + (t & 1) * K_BYTES, t * KV, tokens, r, h, qf, sacc);
            pv_tile<F16, HD>(smem + V_BASE + vprev * V_BYTES, r, h, pf, o);
            if constexpr (CLASSIC) softmax_tile<F16, HD>(sacc, pf, o, m_run, l_run);
            else softmax_nomax<F16, HD>(sacc, pf, l_run, m_ref);
            attn_write((t + 1) & 1, vnxt);
            __syncthreads();
            vprev = vcur;
            vcur = vnxt;
        }
    } else {
        for (int t = 1; t + 1 < nkv; ++t) {
            const int vnxt = vcur == 2 ? 0 : vcur + 1;
            attn_load((t + 1) * KV);
            attn_write((t + 1) & 1, vnxt);
            __syncthreads();
            vprev = vcur;
            vcur = vnxt;
        }
        return false;
    }
    if (nkv > 1) {
        // the last tile (nothing left to stage, no barrier: its K and V^T slots are not written again)
        const int t = nkv - 1;
        s_last(smem + (t & 1) * K_BYTES, t * KV);
        pv_tile<F16, HD>(smem + V_BASE + vprev * V_BYTES, r, h, pf, o);
        softmax_last(false);
        vprev = vcur;
    }
    if (half_tail) pv_tile<F16, HD, 1>(smem + V_BASE + vprev * V_BYTES, r, h, pf, o);
    else pv_tile<F16, HD>(smem + V_BASE + vprev * V_BYTES, r, h, pf, o);

    // ---- normalise and store: out[(b*tokens + q)][head*HD + d], 4 consecutive d per register group
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + r;
    if constexpr (!CLASSIC) {
        // 0.5 <= l < 2^15 (half) / 2^-100 <= l < 2^100 (bf16), tested on the bit pattern: positive floats order like their bits, a
        // negative value or a NaN (>= 0x7f800001, or sign bit set) falls outside whatever the compiler assumes about NaNs
        const uint32_t lb = __float_as_uint(l_tot);
        constexpr uint32_t LO = F16 ? 0x3f000000u - ((uint32_t)HIPTS_ATTN_REF_MARGIN << 23) : 0x0d800000u, HI = F16 ? 0x47000000u : 0x71800000u;
        if (!(lb >= LO && lb < HI)) return true;
    }
    if (qi < tokens) {
        bf16_t* op = out + ((size_t)b * out_stride + qi) * (heads * HD) + head * HD;
#pragma unroll
        for (int blk = 0; blk < HD / 32; ++blk)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *reinterpret_cast<bf16x4*>(op + blk * 32 + 8 * g4 + 4 * h) =
                    pack4<F16>(o[blk][4 * g4] * inv, o[blk][4 * g4 + 1] * inv, o[blk][4 * g4 + 2] * inv, o[blk][4 * g4 + 3] * inv);
            }
    }
    return false;
}

template <bool F16, int HD>
__global__ __launch_bounds__(256) void attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                   const bf16_t* __restrict__ vT, bf16_t* __restrict__ out, int heads,
                                                   int tokens, int tokens_pad, int qblocks, int out_stride, int classic, int trim) {
    __shared__ __attribute__((aligned(16))) char smem[Geo<HD>::LDS_BYTES];
    if (classic) {              // HIPTS_ATTN_CLASSIC=1: the per-tile maximum everywhere (A/B runs; the fallback's own test)
        attn_body<F16, HD, true>(smem, q, k, vT, out, heads, tokens, tokens_pad, qblocks, out_stride, trim);
        return;
    }
    const bool bad = attn_body<F16, HD, false>(smem, q, k, vT, out, heads, tokens, tokens_pad, qblocks, out_stride, trim);
    // a wave whose row sum left the window stored nothing; the workgroup (its waves stage K / V^T together) repeats the block classically
    if (__syncthreads_or(bad ? 1 : 0)) attn_body<F16, HD, true>(smem, q, k, vT, out, heads, tokens, tokens_pad, qblocks, out_stride, trim);
}

}  // namespace

int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* vT, bf16_t* out, int batch, int heads, int tokens,
                     int tokens_pad, bool f16, hipStream_t s, int head_dim, int out_tokens_stride) {
    const int ost = out_tokens_stride > 0 ? out_tokens_stride : tokens;
    HIPTS_REQUIRE(tokens_pad % KV == 0 && tokens_pad >= tokens, "attention: tokens_pad must be a multiple of %d", KV);
    HIPTS_REQUIRE(head_dim == 64 || head_dim == 32, "attention: head_dim must be 64 or 32");
    const int qtiles = (tokens + 31) / 32;
    const int qblocks = (qtiles + 3) / 4;
    const int grid = batch * heads * qblocks;
    static const int classic = (getenv("HIPTS_ATTN_CLASSIC") && atoi(getenv("HIPTS_ATTN_CLASSIC"))) ? 1 : 0;
    static const int trim = (getenv("HIPTS_ATTN_TRIM") && atoi(getenv("HIPTS_ATTN_TRIM")) == 0) ? 0 : 1;
    if (head_dim == 64) {
        if (f16) attn_kernel<true, 64><<<grid, 256, 0, s>>>(q, k, vT, out, heads, tokens, tokens_pad, qblocks, ost, classic, trim);
        else attn_kernel<false, 64><<<grid, 256, 0, s>>>(q, k, vT, out, heads, tokens, tokens_pad, qblocks, ost, classic, trim);
    } else {
        if (f16) attn_kernel<true, 32><<<grid, 256, 0, s>>>(q, k, vT, out, heads, tokens, tokens_pad, qblocks, ost, classic, trim);
        else attn_kernel<false, 32><<<grid, 256, 0, s>>>(q, k, vT, out, heads, tokens, tokens_pad, qblocks, ost, classic, trim);
    }
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace hipts
