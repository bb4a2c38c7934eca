// attn3.h -- the head_dim-64 attention with ONE wave per SIMD and a hand-placed instruction stream (round 5; timm Attention inside
// tagging.py:174).  Included by attn2.hip inside namespace hipts { namespace { -- it shares attn2's LDS images, fragment layouts, fixed
// reference softmax, output path and fallback.
//
// What differs from attn2_seq_body (three waves per SIMD, 32 query rows per wave, the hardware left to overlap one wave's softmax with
// another's MFMAs: the matrix pipe ~40 % busy):
//   * a workgroup is 4 waves = one per SIMD, each with the whole 512-register file: QB = 3 (one wave of an (image, head): 4) blocks of 32
//     query rows per wave.  O^T (QB x 32 registers), Q^T (QB x 16) and the K / V^T fragments live in the AGPR half, the scores S^T (two
//     buffers), P and the softmax's temporaries in the VGPR half.  A K or V^T fragment read feeds QB MFMAs.
//   * the 32-key half tiles form a software pipeline written out by hand: half-step j is two phases of 4 QB MFMAs --
//         phase A:  S(j) = K Q^T              ||  second half of softmax(j - 1)  ||  V^T fragment reads of half j - 1
//         phase B:  O^T += V^T P^T (j - 1)    ||  first half of softmax(j)       ||  K fragment reads of half j + 1, LDS-DMA of later tiles
//     every MFMA is followed by its share of the vector work (2 v_exp, 2 v_add, 1 v_cvt_pk: 34 issue cycles measured, beside the 8 of the MFMA itself) and at
//     most one LDS / DMA instruction.  All of it is `asm volatile` statements in program order (hipcc keeps their order and allocates the
//     registers); waits are counted by hand.
//   * K / V tiles of 64 keys by LDS-DMA into rings of four slots, two tiles ahead; one vmcnt + one s_barrier per tile.  The kernel is
//     persistent (one workgroup per CU walks several items = chunks of query blocks of (image, head) pairs) and the ring runs on across
//     the items: the DMA of the tiles past an item's last one stages the workgroup's NEXT item (without one: this item's first tiles again,
//     into slots nobody reads) -- no branch in the stream, a constant vmcnt.
//   * the 25 query blocks of ViT-B/16@448 go to two workgroups per (image, head): 13 = 3 + 3 + 3 + 4 and 12 = 3 + 3 + 3 + 3, the roles
//     rotating per round; a wave with four blocks runs the QB = 4 stream (no register room for the -m_ref accumulator images: it
//     subtracts explicitly).
//   * the V^T fragments live in FIXED registers a[240:255] (see A3_VREAD_P), and the rare fallback pass is a __noinline__ function
//     (attn2.hip::attn3_classic): both keep hipcc from copying or spilling registers that an asynchronous LDS read is still filling.
// Measured (LABNOTES.md, round 5): correct, 90 / 96 us alone against 82-85 for attn2_seq_body, slower still inside the forward: the
// loop is bound by vector issue (42 cycles per MFMA slot at best at head_dim 64) and half of an item is outside the loop.  A/B variant.
// Arithmetic per element as attn2_seq_body (same MFMA chains, same conversions); only the order of the row sums' additions differs.
// Requires tokens_pad >= 128 (two key tiles); launch_attention2 keeps shorter sequences on attn2_seq_body.

constexpr int A3_SLOTS = 4;
constexpr int A3_VBASE = A3_SLOTS * TILE;
constexpr int A3_LDS = 2 * A3_SLOTS * TILE;          // 64 KiB (+ 16 B: the fallback flag)
constexpr int A3_MAXQB = 4;
constexpr int A3_WG_BLOCKS = 13;                     // query blocks one workgroup takes at most (3 + 3 + 3 + 4)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifdef HIPTS_A3_STAMPS                  // measurement build: cycle stamps of wave 0 of workgroup HIPTS_A3_STAMPS (tools/gpurun/r5_attn3_stamps.sh)
__device__ unsigned long long g_attn3_stamps[4096];
#define A3_STAMP(slot)                                                                      \
    do {                                                                                    \
        unsigned long long ts_;                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");         \
        if (stamp_on && stamp_base + (slot) < 4096) g_attn3_stamps[stamp_base + (slot)] = ts_; \
    } while (0)
#else
#define A3_STAMP(slot) do { } while (0)
#endif

template <int QB, bool CI>
struct A3Regs {
    f32x16 S[2][QB];                 // S^T of half-step j in S[j & 1]                                      (VGPR)
    f32x16 O[QB][2];                 // O^T: [block][d / 32]                                                (AGPR)
    bf16x8 Q[QB][4];                 // Q^T fragments                                                       (AGPR)
    bf16x8 Kf[4];                    // K fragments of one half tile                                        (AGPR)
    bf16x8 Vf[2][2];                 // V^T fragments of one half tile: [d / 32][16-key step], each two b64 transposing reads (AGPR)
    bf16x4 Vlo[2][2];                // (the first of the two, until the second is requested)
    uint32_t Pa[2][QB][4];           // P of the keys 0-15 of half-step j in Pa[j & 1] (written one half-step before it is multiplied)
    uint32_t Pb[QB][4];              // P of the keys 16-31
    f32x16 C[CI ? QB : 1];           // -m_ref in every register: the accumulator S^T starts from
    float l0[QB], l1[QB], mref[QB];
    float T[2][2];                   // the exponentials of the pair in flight
};

// ---- the instructions ----
template <bool F16>
__device__ __forceinline__ void a3_mfma_s_first_c(f32x16& s, const bf16x8& kf, const bf16x8& qf, const f32x16& c) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=v"(s) : "a"(kf), "a"(qf), "v"(c));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=v"(s) : "a"(kf), "a"(qf), "v"(c));
}
template <bool F16>
__device__ __forceinline__ void a3_mfma_s_first_0(f32x16& s, const bf16x8& kf, const bf16x8& qf) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(s) : "a"(kf), "a"(qf));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(s) : "a"(kf), "a"(qf));
}
template <bool F16>
__device__ __forceinline__ void a3_mfma_s(f32x16& s, const bf16x8& kf, const bf16x8& qf) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "a"(kf), "a"(qf));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "a"(kf), "a"(qf));
}
#define A3_MFMA_O(OP, REGS) asm volatile(OP " %0, " REGS ", %2, %0" : "+a"(o) : "{" REGS "}"(vf), "v"(p))
template <bool F16, int N>
__device__ __forceinline__ void a3_mfma_o(f32x16& o, const bf16x8& vf, const u32x4& p) {       // vf = V^T fragment N, in its fixed registers
    if constexpr (F16) {
        if constexpr (N == 0) A3_MFMA_O("v_mfma_f32_32x32x16_f16", "a[240:243]");
        else if constexpr (N == 1) A3_MFMA_O("v_mfma_f32_32x32x16_f16", "a[244:247]");
        else if constexpr (N == 2) A3_MFMA_O("v_mfma_f32_32x32x16_f16", "a[248:251]");
        else A3_MFMA_O("v_mfma_f32_32x32x16_f16", "a[252:255]");
    } else {
        if constexpr (N == 0) A3_MFMA_O("v_mfma_f32_32x32x16_bf16", "a[240:243]");
        else if constexpr (N == 1) A3_MFMA_O("v_mfma_f32_32x32x16_bf16", "a[244:247]");
        else if constexpr (N == 2) A3_MFMA_O("v_mfma_f32_32x32x16_bf16", "a[248:251]");
        else A3_MFMA_O("v_mfma_f32_32x32x16_bf16", "a[252:255]");
    }
}
__device__ __forceinline__ void a3_exp(float& d, float s) { asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s)); }
__device__ __forceinline__ void a3_sub(float& d, float s, float m) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(s), "v"(m)); }
__device__ __forceinline__ void a3_acc(float& l, float p) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(l) : "v"(p)); }
template <bool F16>
__device__ __forceinline__ void a3_cvt(uint32_t& w, float a, float b) {
    if constexpr (F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(a), "v"(b));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(a), "v"(b));
}
#define A3_KREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(DST) : "v"(ADDR), "i"(OFF))
// The V^T fragments sit in FIXED registers a[240:255]: an MFMA operand is four registers that two ds_read_b64_tr_b16 fill, and hipcc does
// not coalesce two asm-defined register pairs into one tuple (it copied them -- through VGPRs, in front of the wait).  With the pair and
// the tuple constrained to the same physical registers every copy is an identity.
#define A3_VREAD_P(REGS, DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 " REGS ", %1 offset:%2" : "={" REGS "}"(DST) : "v"(ADDR), "i"(OFF))
template <int N, int HI, int OFF>
__device__ __forceinline__ void a3_vread(bf16x4& dst, unsigned addr) {
    if constexpr (N == 0 && HI == 0) A3_VREAD_P("a[240:241]", dst, addr, OFF);
    else if constexpr (N == 0) A3_VREAD_P("a[242:243]", dst, addr, OFF);
    else if constexpr (N == 1 && HI == 0) A3_VREAD_P("a[244:245]", dst, addr, OFF);
    else if constexpr (N == 1) A3_VREAD_P("a[246:247]", dst, addr, OFF);
    else if constexpr (N == 2 && HI == 0) A3_VREAD_P("a[248:249]", dst, addr, OFF);
    else if constexpr (N == 2) A3_VREAD_P("a[250:251]", dst, addr, OFF);
    else if constexpr (N == 3 && HI == 0) A3_VREAD_P("a[252:253]", dst, addr, OFF);
    else A3_VREAD_P("a[254:255]", dst, addr, OFF);
}
// LDS-DMA of one 1 KiB piece: M0 = the wave-uniform LDS byte address, per-lane 32-bit byte offset from a scalar base pointer
#define A3_GLDS(M0ADDR, VOFF, SPTR) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(M0ADDR), "v"(VOFF), "s"(SPTR) : "memory")

template <int QB, bool CI>
__device__ __forceinline__ void a3_wait_k(A3Regs<QB, CI>& R) {          // the fragments named "+a": no consumer (or copy) moves above the wait
    asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(R.Kf[0]), "+a"(R.Kf[1]), "+a"(R.Kf[2]), "+a"(R.Kf[3]));
}
template <int QB, bool CI>
__device__ __forceinline__ void a3_wait_v(A3Regs<QB, CI>& R) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+{a[240:243]}"(R.Vf[0][0]), "+{a[244:247]}"(R.Vf[0][1]), "+{a[248:251]}"(R.Vf[1][0]), "+{a[252:255]}"(R.Vf[1][1]));
}

// what a tile's DMA needs: scalar bases of the K / V rows of this (image, head), per-lane byte offsets, the wave's first LDS piece
struct A3Dma {
    const char* kb;
    const char* vb;
    const char* kb2;                 // the workgroup's NEXT (image, head): the tiles past this one's last are its first ones (the ring runs on
    const char* vb2;                 // across the seam); without a next one they are the last tile again, into a slot nobody reads any more
    unsigned kvo[2], vvo[2];         // per-lane byte offsets inside a tile, pieces wave and wave + 4
    unsigned m0k;                    // lds0 + wave * 1024
    int nkv;
    int gbase;                       // tiles this workgroup has been through before this item: tile t sits in ring slot (gbase + t) & 3
};
// one of the four pieces (K pieces 0, 1, then V pieces 0, 1) of tile tt of this item's numbering (tt >= nkv: the next item's tile tt - nkv)
template <int PC>
__device__ __forceinline__ void a3_dma_tile_piece(const A3Dma& d, int tt) {
    constexpr bool ISV = PC >= 2;
    constexpr int pc = PC & 1;
    const int slot = (d.gbase + tt) & (A3_SLOTS - 1);
    const bool next = tt >= d.nkv;
    int tn = next ? tt - d.nkv : tt;
    tn = tn < d.nkv ? tn : d.nkv - 1;
    const char* src = (next ? (ISV ? d.vb2 : d.kb2) : (ISV ? d.vb : d.kb)) + (size_t)tn * TILE;
    const unsigned m0 = d.m0k + (ISV ? A3_VBASE : 0) + slot * TILE + pc * 4096;
    A3_GLDS(m0, ISV ? d.vvo[pc] : d.kvo[pc], src);
}
// piece PC of the DMA that follows barrier B_t: K tile t + 3 and V tile t + 2
template <int PC>
__device__ __forceinline__ void a3_dma_piece(const A3Dma& d, int t) {
    a3_dma_tile_piece<PC>(d, t + (PC >= 2 ? 2 : 3));
}

// ---- one MFMA slot of phase A: S(j) chain element, softmax(j - 1) second half, V^T reads of half j - 1, (DMA) ----
// CUR = j & 1.  MF / VA / LD switch the three streams (prologue and drain run single streams).
template <bool F16, int QB, bool CI, int CUR, bool MF, bool VA, bool LD, bool DMA, int M>
__device__ __forceinline__ void a3_slot_a(A3Regs<QB, CI>& R, const unsigned (&vad)[2], const A3Dma& d, int t) {
    constexpr int PREV = CUR ^ 1;
    if constexpr (MF) {
        constexpr int s = M / QB, qb = M % QB;
        if constexpr (s == 0) {
            if constexpr (CI) a3_mfma_s_first_c<F16>(R.S[CUR][qb], R.Kf[0], R.Q[qb][0], R.C[qb]);
            else a3_mfma_s_first_0<F16>(R.S[CUR][qb], R.Kf[0], R.Q[qb][0]);
        } else {
            a3_mfma_s<F16>(R.S[CUR][qb], R.Kf[s], R.Q[qb][s]);
        }
    }
    if constexpr (VA) {
        constexpr int uqb = M / 4, k = M % 4, i0 = 8 + 2 * k;
        if constexpr (F16 && !CI) {
            a3_sub(R.T[M & 1][0], R.S[PREV][uqb][i0], R.mref[uqb]);
            a3_sub(R.T[M & 1][1], R.S[PREV][uqb][i0 + 1], R.mref[uqb]);
            a3_exp(R.T[M & 1][0], R.T[M & 1][0]);
            a3_exp(R.T[M & 1][1], R.T[M & 1][1]);
        } else {
            a3_exp(R.T[M & 1][0], R.S[PREV][uqb][i0]);
            a3_exp(R.T[M & 1][1], R.S[PREV][uqb][i0 + 1]);
        }
        if constexpr (M > 0) {
            constexpr int pqb = (M - 1) / 4, pk = (M - 1) % 4;
            a3_acc(R.l0[pqb], R.T[(M - 1) & 1][0]);
            a3_acc(R.l1[pqb], R.T[(M - 1) & 1][1]);
            a3_cvt<F16>(R.Pb[pqb][pk], R.T[(M - 1) & 1][0], R.T[(M - 1) & 1][1]);
        }
    }
    if constexpr (LD && M < 8) {
        constexpr int s2 = M >> 2, dblk = (M >> 1) & 1, hi = M & 1;
        constexpr int imm = PREV * 4096 + s2 * 2048 + hi * 1024;
        if constexpr (hi) {
            bf16x4 vhi;
            a3_vread<dblk * 2 + s2, 1, imm>(vhi, vad[dblk]);
            R.Vf[dblk][s2] = __builtin_shufflevector(R.Vlo[dblk][s2], vhi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            a3_vread<dblk * 2 + s2, 0, imm>(R.Vlo[dblk][s2], vad[dblk]);
        }
    }
    if constexpr (DMA && M >= 8 && M < 12) a3_dma_piece<M - 8>(d, t);
}
template <bool F16, int QB, bool CI, int CUR, bool MF, bool VA, bool LD, bool DMA, int M = 0>
__device__ __forceinline__ void a3_phase_a(A3Regs<QB, CI>& R, const unsigned (&vad)[2], const A3Dma& d, int t) {
    a3_slot_a<F16, QB, CI, CUR, MF, VA, LD, DMA, M>(R, vad, d, t);
    if constexpr (M + 1 < 4 * QB) {
        a3_phase_a<F16, QB, CI, CUR, MF, VA, LD, DMA, M + 1>(R, vad, d, t);
    } else if constexpr (VA) {                  // the last pair's sums and word
        constexpr int pqb = M / 4, pk = M % 4;
        a3_acc(R.l0[pqb], R.T[M & 1][0]);
        a3_acc(R.l1[pqb], R.T[M & 1][1]);
        a3_cvt<F16>(R.Pb[pqb][pk], R.T[M & 1][0], R.T[M & 1][1]);
    }
}

// ---- one MFMA slot of phase B: O^T += V^T P^T of half j - 1, softmax(j) first half, K reads of half j + 1 ----
template <bool F16, int QB, bool CI, int CUR, bool MF, bool VA, bool LD, int M>
__device__ __forceinline__ void a3_slot_b(A3Regs<QB, CI>& R, const unsigned (&kad)[4]) {
    constexpr int PREV = CUR ^ 1;
    if constexpr (MF) {
        constexpr int s2 = M / (2 * QB), dblk = (M / QB) % 2, qb = M % QB;
        u32x4 p;
        if constexpr (s2 == 0) {
            p[0] = R.Pa[PREV][qb][0]; p[1] = R.Pa[PREV][qb][1]; p[2] = R.Pa[PREV][qb][2]; p[3] = R.Pa[PREV][qb][3];
        } else {
            p[0] = R.Pb[qb][0]; p[1] = R.Pb[qb][1]; p[2] = R.Pb[qb][2]; p[3] = R.Pb[qb][3];
        }
        a3_mfma_o<F16, dblk * 2 + s2>(R.O[qb][dblk], R.Vf[dblk][s2], p);
    }
    if constexpr (VA) {
        constexpr int uqb = M / 4, k = M % 4, i0 = 2 * k;
        if constexpr (F16 && !CI) {
            a3_sub(R.T[M & 1][0], R.S[CUR][uqb][i0], R.mref[uqb]);
            a3_sub(R.T[M & 1][1], R.S[CUR][uqb][i0 + 1], R.mref[uqb]);
            a3_exp(R.T[M & 1][0], R.T[M & 1][0]);
            a3_exp(R.T[M & 1][1], R.T[M & 1][1]);
        } else {
            a3_exp(R.T[M & 1][0], R.S[CUR][uqb][i0]);
            a3_exp(R.T[M & 1][1], R.S[CUR][uqb][i0 + 1]);
        }
        if constexpr (M > 0) {
            constexpr int pqb = (M - 1) / 4, pk = (M - 1) % 4;
            a3_acc(R.l0[pqb], R.T[(M - 1) & 1][0]);
            a3_acc(R.l1[pqb], R.T[(M - 1) & 1][1]);
            a3_cvt<F16>(R.Pa[CUR][pqb][pk], R.T[(M - 1) & 1][0], R.T[(M - 1) & 1][1]);
        }
    }
    if constexpr (LD && M < 4) A3_KREAD(R.Kf[M], kad[M], PREV * 4096);
}
template <bool F16, int QB, bool CI, int CUR, bool MF, bool VA, bool LD, int M = 0>
__device__ __forceinline__ void a3_phase_b(A3Regs<QB, CI>& R, const unsigned (&kad)[4]) {
    a3_slot_b<F16, QB, CI, CUR, MF, VA, LD, M>(R, kad);
    if constexpr (M + 1 < 4 * QB) {
        a3_phase_b<F16, QB, CI, CUR, MF, VA, LD, M + 1>(R, kad);
    } else if constexpr (VA) {
        constexpr int pqb = M / 4, pk = M % 4;
        a3_acc(R.l0[pqb], R.T[M & 1][0]);
        a3_acc(R.l1[pqb], R.T[M & 1][1]);
        a3_cvt<F16>(R.Pa[CUR][pqb][pk], R.T[M & 1][0], R.T[M & 1][1]);
    }
}

// One wave's pass over its nq <= QB query blocks (first block wb0) of one (image, head); every wave of the workgroup runs one instance (the
// barriers).  bh2: the workgroup's next (image, head), -1 = none; gbase: key tiles the workgroup has been through (the ring position);
// fresh: the ring holds nothing of this item yet (the workgroup's first item, or the one after a fallback pass).
// Returns true when a row sum left the fast path's window (nothing stored).
#define A3_QLOAD(DST, ADDR, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(DST) : "v"(ADDR), "i"(OFF) : "memory")
template <bool F16, int QB, bool CI>
__device__ __forceinline__ bool attn3_body(unsigned lds0, const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                           bf16_t* __restrict__ out, int heads, int tokens, int tokens_pad, int bh, int wb0, int nq, int bh2, int gbase,
                                           bool fresh, int out_stride, int out_ld, int lo_off, float lo_scale) {
    static_assert(!CI || F16, "CI: the -m_ref accumulator images of the half-operand path");
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));      // opaque per item: the lane constants below are recomputed per item instead of being hoisted out of the
                                       // workgroup's item loop and kept (spilled) across the OTHER stream's registers
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int b = bh / heads, head = bh - b * heads;
    const int nkv = tokens_pad / KV;
    const int q0 = wb0 * 32;

    A3Dma d;
    d.kb = reinterpret_cast<const char*>(k + (size_t)bh * tokens_pad * HD);
    d.vb = reinterpret_cast<const char*>(v + (size_t)bh * tokens_pad * HD);
    d.kb2 = reinterpret_cast<const char*>(k + (size_t)(bh2 >= 0 ? bh2 : bh) * tokens_pad * HD);      // no next item: this one's first tiles again (never read)
    d.vb2 = reinterpret_cast<const char*>(v + (size_t)(bh2 >= 0 ? bh2 : bh) * tokens_pad * HD);
    d.m0k = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    d.nkv = nkv;
    d.gbase = gbase;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        const int row = (wave + 4 * pc) * 8 + (lane >> 3);
        d.kvo[pc] = (unsigned)(row * HD + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
        d.vvo[pc] = (unsigned)(row * HD + (((lane & 7) ^ (((row >> 1) & 1) << 2)) * 8)) * 2u;
    }
#ifdef HIPTS_A3_STAMPS
#ifndef HIPTS_A3_STAMP_WAVE
#define HIPTS_A3_STAMP_WAVE 0
#endif
    const bool stamp_on = blockIdx.x == HIPTS_A3_STAMPS && wave == HIPTS_A3_STAMP_WAVE && lane == 0;
    const int stamp_base = (gbase / nkv) * 128;          // per item of the workgroup
#endif
    A3_STAMP(4);
    if (fresh) {
        // in the order the counted waits assume: K0 V0 K1 | K2 V1 (what "after B_-1" would have issued) -- then after B_t: K(t + 3) V(t + 2).
        // Otherwise the previous item's last tile steps have issued them.
        a3_dma_tile_piece<0>(d, 0); a3_dma_tile_piece<1>(d, 0);
        a3_dma_tile_piece<2>(d, 0); a3_dma_tile_piece<3>(d, 0);
        a3_dma_tile_piece<0>(d, 1); a3_dma_tile_piece<1>(d, 1);
        a3_dma_tile_piece<0>(d, 2); a3_dma_tile_piece<1>(d, 2);
        a3_dma_tile_piece<2>(d, 1); a3_dma_tile_piece<3>(d, 1);
    }
    if (nq == 0) {                                   // a wave without query blocks only stages
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        a3_dma_piece<0>(d, 0); a3_dma_piece<1>(d, 0); a3_dma_piece<2>(d, 0); a3_dma_piece<3>(d, 0);
        for (int t = 1; t < nkv; ++t) {
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            a3_dma_piece<0>(d, t); a3_dma_piece<1>(d, t); a3_dma_piece<2>(d, t); a3_dma_piece<3>(d, t);
        }
        return false;
    }

    A3Regs<QB, CI> R;
    // Q^T fragments (B operand): lane (query r, half h), k-step s: d = 16 s + 8 h .. + 7.  Requested here, straight into the AGPRs; they are
    // waited for together with the epilogue stores of the previous item in front of B_0 (the compiler knows nothing of these loads)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qrow = q0 + 32 * qb + r;
        qrow = qrow < tokens_pad ? qrow : tokens_pad - 1;
        const bf16_t* qp = q + ((size_t)bh * tokens_pad + qrow) * HD + 8 * h;
        A3_QLOAD(R.Q[qb][0], qp, 0);
        A3_QLOAD(R.Q[qb][1], qp, 32);
        A3_QLOAD(R.Q[qb][2], qp, 64);
        A3_QLOAD(R.Q[qb][3], qp, 96);
        R.l0[qb] = 0.f;
        R.l1[qb] = 0.f;
        R.mref[qb] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) R.O[qb][blk][i] = 0.f;
    }
    // fragment addresses inside a tile (attn2_seq_body's)
    unsigned ka[4], va[2];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = lds0 + r * 128 + (((2 * s + h) ^ ((r >> 1) & 7)) * 16);
    {
        const int l16 = lane & 15, qd = l16 >> 2, p = l16 & 3, dgrp = (lane >> 4) & 1;
        const int x = qd >> 1;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) va[blk] = lds0 + A3_VBASE + (4 * h + qd) * 128 + (2 * dgrp + (p >> 1) + 4 * (blk ^ x)) * 16 + (p & 1) * 8;
    }
    auto k_addr = [&](int t, unsigned (&kad)[4]) __attribute__((always_inline)) {
        const unsigned o = (unsigned)(((gbase + t) & (A3_SLOTS - 1)) * TILE);
#pragma unroll
        for (int s = 0; s < 4; ++s) kad[s] = ka[s] + o;
    };
    auto v_addr = [&](int t, unsigned (&vad)[2]) __attribute__((always_inline)) {
        const unsigned o = (unsigned)(((gbase + t) & (A3_SLOTS - 1)) * TILE);
        vad[0] = va[0] + o;
        vad[1] = va[1] + o;
    };
    // keys >= tokens of half `g` of the last tile -> -inf (the exponential makes them 0)
    auto mask_last = [&](f32x16 (&S)[QB], int g) __attribute__((always_inline)) {
        const int key0 = (nkv - 1) * KV + g * 32;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) S[qb][i] = (key0 + crow(i, h) >= tokens) ? -INFINITY : S[qb][i];
    };

    unsigned kad[4], vad[2];
    A3_STAMP(0);
    // ---- half-step 0: B_0, S(0), the reference exponent, the first half of softmax(0), the K fragments of half-step 1
    // (vmcnt(0): Q, and with it everything older -- K0 V0 K1 K2 V1 of this item and the previous item's stores)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) asm volatile("s_waitcnt vmcnt(0)" : "+a"(R.Q[qb][0]), "+a"(R.Q[qb][1]), "+a"(R.Q[qb][2]), "+a"(R.Q[qb][3])::"memory");
    asm volatile("s_barrier" ::: "memory");
    a3_dma_piece<0>(d, 0); a3_dma_piece<1>(d, 0); a3_dma_piece<2>(d, 0); a3_dma_piece<3>(d, 0);
    k_addr(0, kad);
    v_addr(0, vad);
    a3_phase_b<F16, QB, CI, 1, false, false, true>(R, kad);      // the K reads of half 0 alone (CUR = 1 reads half 0)
    a3_wait_k(R);
    {
        // S(0) from C = 0 whatever CI says: the reference exponent comes out of it
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (s == 0) a3_mfma_s_first_0<F16>(R.S[0][qb], R.Kf[0], R.Q[qb][0]);
                else a3_mfma_s<F16>(R.S[0][qb], R.Kf[s], R.Q[qb][s]);
            }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the MFMAs are invisible to hipcc's hazard recogniser: results read by plain code below
    }
    if constexpr (F16) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = R.S[0][qb][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, R.S[0][qb][i]);
            const float m = fmaxf(mx, __shfl_xor(mx, 32)) + (float)HIPTS_ATTN_REF_MARGIN;
            R.mref[qb] = m;
            if constexpr (CI) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    R.S[0][qb][i] -= m;             // half-step 0 subtracts explicitly; later ones start from C = -m_ref
                    R.C[qb][i] = -m;
                }
            }
        }
    }
    a3_phase_b<F16, QB, CI, 0, false, true, true>(R, kad);       // softmax(0) first half -> Pa[0]; K reads of half 1 of tile 0
    A3_STAMP(1);

    // ---- tiles: body(t) = half-step 2 t + 1 | B_(t+1) | half-step 2 t + 2
    auto body = [&](int t, auto last_c) __attribute__((always_inline)) {
        constexpr bool LASTB = decltype(last_c)::value;
        v_addr(t, vad);
        k_addr(t + 1, kad);
        a3_wait_k(R);
        A3_STAMP(8 + t * 8 + 0);
        a3_phase_a<F16, QB, CI, 1, true, true, true, false>(R, vad, d, t);
        a3_wait_v(R);
        A3_STAMP(8 + t * 8 + 1);
        a3_phase_b<F16, QB, CI, 1, true, true, true>(R, kad);
        A3_STAMP(8 + t * 8 + 2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        A3_STAMP(8 + t * 8 + 3);
        asm volatile("s_barrier" ::: "memory");      // B_(t+1): K(t + 2) and V(t + 1) have landed for every wave
        a3_wait_k(R);
        A3_STAMP(8 + t * 8 + 4);
        a3_phase_a<F16, QB, CI, 0, true, true, true, true>(R, vad, d, t + 1);
        A3_STAMP(8 + t * 8 + 5);
        if constexpr (LASTB) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // S(j) is read by plain code
            mask_last(R.S[0], 0);
        }
        a3_wait_v(R);
        A3_STAMP(8 + t * 8 + 6);
        a3_phase_b<F16, QB, CI, 0, true, true, true>(R, kad);
        A3_STAMP(8 + t * 8 + 7);
    };
    for (int t = 0; t + 2 < nkv; ++t) body(t, std::false_type{});
    body(nkv - 2, std::true_type{});

    // ---- the tail: (half 1 of the last tile,) the second half of the last softmax, the last P V, normalise, store
    auto finish = [&](auto lc_c) __attribute__((always_inline)) -> bool {
        constexpr int LC = decltype(lc_c)::value;                // parity of the last half-step
        v_addr(nkv - 1, vad);
        a3_phase_a<F16, QB, CI, LC ^ 1, false, true, true, false>(R, vad, d, 0);
        a3_wait_v(R);
        a3_phase_b<F16, QB, CI, LC ^ 1, true, false, false>(R, kad);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // O is read by plain code
        bool bad = false;
        float inv[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float ls = R.l0[qb] + R.l1[qb];
            const float l_tot = ls + __shfl_xor(ls, 32);
            const uint32_t lb = __float_as_uint(l_tot);
            constexpr uint32_t LO = F16 ? 0x3f000000u - ((uint32_t)HIPTS_ATTN_REF_MARGIN << 23) : 0x0d800000u, HI = F16 ? 0x47000000u : 0x71800000u;
            if (qb < nq && !(lb >= LO && lb < HI)) bad = true;
            inv[qb] = 1.0f / l_tot;
        }
        A3_STAMP(2);
        if (__builtin_amdgcn_ballot_w64(bad) != 0) return true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qi = q0 + 32 * qb + r;
            bf16_t* op = out + ((size_t)b * out_stride + qi) * out_ld + head * HD;
            const bool valid = qb < nq && qi < tokens;
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) store_o_pair<F16>(R.O[qb][(2 * kp) >> 2], R.O[qb][(2 * kp + 1) >> 2], kp, inv[qb], h, valid, op, lo_off, lo_scale);
        }
        A3_STAMP(3);
        return false;
    };
    const int tail_keys = tokens - (nkv - 1) * KV;
    if (tail_keys <= 32) return finish(std::integral_constant<int, 0>{});
    // half 1 of the last tile (half-step 2 nkv - 1)
    v_addr(nkv - 1, vad);
    k_addr(nkv, kad);
    a3_wait_k(R);
    a3_phase_a<F16, QB, CI, 1, true, true, true, false>(R, vad, d, 0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    mask_last(R.S[1], 1);
    a3_wait_v(R);
    a3_phase_b<F16, QB, CI, 1, true, true, true>(R, kad);
    return finish(std::integral_constant<int, 1>{});
}
