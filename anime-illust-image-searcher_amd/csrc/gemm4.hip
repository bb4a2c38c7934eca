// gemm4.hip -- the GEMM main loop with FOUR waves per workgroup, one per SIMD (round 5; tagging.py:174).
//
//   C[M,N] = A[M,K] x W[N,K]^T as gemm.hip, same 256 x 256 x 64 tile, same LDS images (1 KB sub-tiles of 8 rows x 128 B, 16 B chunk
//   XOR-swizzled with the row), same epilogues (gemm_epi.h) and -- the MFMA chain of an output element runs over K in the same order --
//   the same bits.  What differs is who multiplies:
//
//   * 256 threads = 4 waves as 2 (M) x 2 (N); a wave owns 128 x 128 of the tile = 8 x 8 MFMA tiles of 16 x 16 = 256 accumulator registers,
//     which live in the AGPR half of the wave's 512-register file (one wave per SIMD has the whole file).  Per MAC a wave reads a third
//     fewer LDS bytes and operand registers than the 128 x 64 waves of gemm_pp_kernel (16 fragment reads per 64 MFMAs instead of 12 per 32).
//   * No partner wave hides anything, so nothing may stall the stream: a K-tile is two phases of 64 MFMAs (k-halves 0 / 1) whose operand
//     fragments (2 x 16 ds_read_b128 = 128 registers, double-buffered across the phases) are read one phase ahead, one read per MFMA gap;
//     the next-but-one K-tile's 16 LDS-DMA loads per wave ride in the second phase's gaps; ONE barrier per K-tile (between the phases:
//     "stage t is read, K-tile t + 1 has landed").  The whole loop is inline asm statements in program order (hipcc orders volatile asm
//     statements but schedules nothing across them): MFMAs on "+a" accumulators, ds_read / LDS-DMA with hand-counted waits.
//   * The K-tile stream runs ACROSS output tiles: the load slot of K-tile nt - 2 takes K-tile 0 of the workgroup's next tile, so a new
//     tile starts with its first operands already in LDS (stage 0); stage 1 is the epilogue's scratch in between.
//   * After the loop a wave hands its accumulators to the epilogues of gemm_epi.h as two 128 x 64 wave-parts (virtual wave_n = 2 wn + h).
//
// Built for the launches whose tiles all lie inside the matrix (M, N multiples of 256) with an even number of K-tiles: every GEMM of the
// ViT-B/16 forward but the tag head.  Everything else stays on gemm_pp_kernel (launch_gemm_q4 says "not handled").
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "vit_internal.h"

namespace hipts {
namespace {

#include "gemm_epi.h"

template <bool F16>
__device__ __forceinline__ void q4_mfma(f32x4& acc, const bf16x8& w, const bf16x8& x) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
}
// first product of an output tile: C = 0 (no zeroing pass over 256 registers)
template <bool F16>
__device__ __forceinline__ void q4_mfma0(f32x4& acc, const bf16x8& w, const bf16x8& x) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(x));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(x));
}

#ifndef HIPTS_Q4_EXECMASK
#define HIPTS_Q4_EXECMASK 0       // 1: bit-equal (MFMAs do ignore EXEC) and 1.4-1.7 x SLOWER: phase 1 goes from 1700 to 3760 cycles per K-tile -- an EXEC write, the dropped request and the restore cost ~59 cycles per gap in all four waves
#endif
#ifndef HIPTS_Q4_STAGGER
#define HIPTS_Q4_STAGGER 0         // 1: measured 1.7 x SLOWER and wrong -- the wave-uniform branches make hipcc copy the asm-owned accumulators at every merge (2452 v_accvgpr moves in the loop, read before the MFMAs have written them)
#endif
#define Q4_DSREAD(DST, BASE, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE), "i"(OFF))
// every fragment of a set named "+v": no consumer is scheduled above the wait (cdna_hip_programming.md section 5.7, form (ii))
#define Q4_WAIT_FRAGS(FA, FW)                                                                                                                    \
    do {                                                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FA[4]), "+v"(FA[5]), "+v"(FA[6]), "+v"(FA[7])); \
        asm volatile("" : "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3]), "+v"(FW[4]), "+v"(FW[5]), "+v"(FW[6]), "+v"(FW[7]));               \
    } while (0)
// LDS-DMA of one 1 KB sub-tile: M0 = wave-uniform LDS byte address (written in the statement that uses it), per-lane 32-bit offset from a
// scalar base pointer
#define Q4_GLDS(SB, IMM, VOFF, SPTR)                                                                                                             \
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(SB), "i"(IMM), "v"(VOFF), "s"(SPTR) : "memory", "scc")
// The same under an EXEC mask that is all ones for ONE of the four waves and zero for the others (HIPTS_Q4_EXECMASK): every wave runs the
// same instruction stream -- no branch, so hipcc keeps the asm-owned accumulators where they are -- but a request with EXEC = 0 never
// reaches the address path.  MFMAs ignore EXEC; nothing else sits between the two s_mov.
#define Q4_GLDS_X(WAVE, G, SB, IMM, VOFF, SPTR)                                                                                                  \
    asm volatile("s_cmp_eq_u32 %4, %5\n\ts_cselect_b64 exec, -1, 0\n\ts_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1" ::"s"(SB), \
                 "i"(IMM), "v"(VOFF), "s"(SPTR), "s"(WAVE), "i"(G)                                                                               \
                 : "memory", "scc")

// Measurement builds only (STAMP instantiations, tools/gemm_bench.py with HIPTS_GEMM_STAMPS=1 HIPTS_GEMM_Q4=...): the shader clock, with the
// wait inside the statement (s_memtime returns through the scalar cache's counter)
template <bool STAMP>
__device__ __forceinline__ unsigned long long q4_now() {
    if constexpr (STAMP) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    } else {
        return 0ull;
    }
}

// gaps 16 G .. 16 G + 15 of a K-tile's second phase; LOADS: this wave's 16 LDS-DMA requests ride in them
template <bool F16, int S, bool RD, bool LOADS, int G>
__device__ __forceinline__ void q4_p1_seg(f32x4 (&acc)[8][8], bf16x8 (&fa0)[8], bf16x8 (&fw0)[8], bf16x8 (&fa1)[8], bf16x8 (&fw1)[8],
                                          const unsigned (&bA)[2][2], const unsigned (&bW)[2][2], unsigned sb, const unsigned (&voff)[16],
                                          const char* asrc, const char* wsrc) {
#pragma unroll
    for (int s = 16 * G; s < 16 * G + 16; ++s) {
        const int i = s >> 3, j = s & 7;
        q4_mfma<F16>(acc[i][j], fw1[j], fa1[i]);
        if constexpr (RD) {
            if (s < 8) Q4_DSREAD(fa0[s], bA[S ^ 1][0], s * 2048);
            else if (s < 16) Q4_DSREAD(fw0[s - 8], bW[S ^ 1][0], (s - 8) * 2048);
        }
        if constexpr (LOADS) {
            const int u = s - 16 * G;
            if (u < 8) Q4_GLDS(sb, S * STAGE_BYTES + u * 4096, voff[u], asrc);
            else Q4_GLDS(sb, S * STAGE_BYTES + TILE_BYTES + (u - 8) * 4096, voff[u], wsrc);
        }
    }
}

// One K-tile of the stream.  S: its LDS stage; FIRST: first K-tile of an output tile (k-half 0 starts the accumulators); RD: a next
// K-tile exists in this output tile (its k-half-0 fragments are read in phase 1); IS: phase 1 carries the 16 loads of K-tile t + 2 (or of
// the next output tile's K-tile 0) into THIS stage, free since the barrier.
//   phase 0: 64 MFMAs on set 0 | gaps 0..15: reads of set 1 (this stage, k-half 1)
//   s_waitcnt lgkmcnt(0) (set 1 is in registers, this wave is through with the stage), vmcnt(0) (this wave's loads of K-tile t + 1
//   landed), s_barrier (... every wave's)
//   phase 1: 64 MFMAs on set 1 | gaps 0..15: reads of set 0 (other stage, k-half 0) | gaps 16..31: the loads
template <bool F16, int S, bool FIRST, bool RD, bool IS, bool STAMP = false>
__device__ __forceinline__ void q4_ktile(f32x4 (&acc)[8][8], bf16x8 (&fa0)[8], bf16x8 (&fw0)[8], bf16x8 (&fa1)[8], bf16x8 (&fw1)[8],
                                         const unsigned (&bA)[2][2], const unsigned (&bW)[2][2], unsigned sb, const unsigned (&voff)[16],
                                         const char* asrc, const char* wsrc, unsigned long long (&tsum)[3], int wave) {
    const unsigned long long t0 = q4_now<STAMP>();
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int i = s >> 3, j = s & 7;
        if constexpr (FIRST) q4_mfma0<F16>(acc[i][j], fw0[j], fa0[i]);
        else q4_mfma<F16>(acc[i][j], fw0[j], fa0[i]);
        if (s < 8) Q4_DSREAD(fa1[s], bA[S][1], s * 2048);
        else if (s < 16) Q4_DSREAD(fw1[s - 8], bW[S][1], (s - 8) * 2048);
    }
    const unsigned long long t1 = q4_now<STAMP>();
    Q4_WAIT_FRAGS(fa1, fw1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned long long t2 = q4_now<STAMP>();
    // The four waves run in step, and four LDS-DMA requests at once queue on the CU's one address path (measured: 44 cycles of a wave's
    // issue per request when all four ask in the same gap, phase 1 at 1700 cycles for 1024 of MFMAs).  HIPTS_Q4_STAGGER: wave w asks in
    // gaps 16 w .. 16 w + 15 -- one request in flight per gap over the whole phase; the price is a wave-uniform branch per 16 gaps.
#if HIPTS_Q4_EXECMASK
    // wave w's 16 requests ride in gaps 16 w .. 16 w + 15; the other three waves execute the same statements with EXEC = 0
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int i = s >> 3, j = s & 7;
        q4_mfma<F16>(acc[i][j], fw1[j], fa1[i]);
        if constexpr (RD) {
            if (s < 8) Q4_DSREAD(fa0[s], bA[S ^ 1][0], s * 2048);
            else if (s < 16) Q4_DSREAD(fw0[s - 8], bW[S ^ 1][0], (s - 8) * 2048);
        }
        if constexpr (IS) {
            const int u = s & 15;
            if (u < 8) Q4_GLDS_X(wave, s >> 4, sb, S * STAGE_BYTES + u * 4096, voff[u], asrc);
            else Q4_GLDS_X(wave, s >> 4, sb, S * STAGE_BYTES + TILE_BYTES + (u - 8) * 4096, voff[u], wsrc);
        }
    }
#elif HIPTS_Q4_STAGGER
    if (IS && wave == 0) q4_p1_seg<F16, S, RD, true, 0>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    else q4_p1_seg<F16, S, RD, false, 0>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    if (IS && wave == 1) q4_p1_seg<F16, S, RD, true, 1>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    else q4_p1_seg<F16, S, RD, false, 1>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    if (IS && wave == 2) q4_p1_seg<F16, S, RD, true, 2>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    else q4_p1_seg<F16, S, RD, false, 2>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    if (IS && wave == 3) q4_p1_seg<F16, S, RD, true, 3>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
    else q4_p1_seg<F16, S, RD, false, 3>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, asrc, wsrc);
#else
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const int i = s >> 3, j = s & 7;
        q4_mfma<F16>(acc[i][j], fw1[j], fa1[i]);
        if constexpr (RD) {
            if (s < 8) Q4_DSREAD(fa0[s], bA[S ^ 1][0], s * 2048);
            else if (s < 16) Q4_DSREAD(fw0[s - 8], bW[S ^ 1][0], (s - 8) * 2048);
        }
        if constexpr (IS) {
            if (s >= 16 && s < 32) {
                const int u = s - 16;
                if (u < 8) Q4_GLDS(sb, S * STAGE_BYTES + u * 4096, voff[u], asrc);
                else Q4_GLDS(sb, S * STAGE_BYTES + TILE_BYTES + (u - 8) * 4096, voff[u], wsrc);
            }
        }
    }
#endif
    if constexpr (RD) Q4_WAIT_FRAGS(fa0, fw0);
    if constexpr (STAMP) {
        const unsigned long long t3 = q4_now<STAMP>();
        tsum[0] += t1 - t0;
        tsum[1] += t2 - t1;
        tsum[2] += t3 - t2;
    }
}

// EPI: GELU, QK (incl. the fused q | k | v launch), RESID_XG (INT: every tile inside the matrix, 16-bit copy and row sums wanted)
template <int EPI, bool F16, bool INT, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void gemm_q4_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    static_assert(EPI == EPI_GELU || EPI == EPI_QK || EPI == EPI_RESID_XG, "gemm_q4_kernel: epilogues of the ViT forward");
    static_assert(!INT || EPI == EPI_RESID_XG, "INT: the residual epilogue's interior form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float2 st_table[256];                // folded LayerNorm: (rstd, rstd * mean) of the current tile's rows
    __shared__ f32x4 colvec[INT ? 512 : 1];         // INT: ln_gamma | bias of the launch's N <= 1024 columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = tiles_m * tiles_n;
    auto tile_origin = [&](int id, int& m0, int& n0) {      // as gemm_pp_kernel: XCD-aware bijective remap, then the launcher's raster
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        if (a.raster_gn > 0) {
            const int per = tiles_m * a.raster_gn, grp = bid / per, rem = bid - grp * per;
            const int left = tiles_n - grp * a.raster_gn, gn = left < a.raster_gn ? left : a.raster_gn;
            m0 = (rem / gn) * BM;
            n0 = (grp * a.raster_gn + rem % gn) * BN;
            return;
        }
        if (a.raster_gm > 0) {
            const int per = a.raster_gm * tiles_n, grp = bid / per, rem = bid - grp * per;
            const int left = tiles_m - grp * a.raster_gm, gm = left < a.raster_gm ? left : a.raster_gm;
            m0 = (grp * a.raster_gm + rem % gm) * BM;
            n0 = (rem / gm) * BN;
            return;
        }
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    };
    const int K = a.K, nt = K / BK;
    const size_t row_bytes = (size_t)K * 2;

    // ---- lane constants (kept live across the epilogues: 24 registers of the 256 the epilogue has)
    // fragment reads: lane (r = lane & 15, q = lane >> 4) takes row r of a 16-row block, logical chunk 4 kk + q at physical chunk
    // (4 kk + q) ^ (r & 7); block b of an operand tile starts at b * 2048
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    unsigned bA[2][2], bW[2][2];
    {
        const int r = lane & 15, q = lane >> 4;
        const unsigned row_off = (unsigned)((r >> 3) * 1024 + (r & 7) * 128);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned c = (unsigned)(((kk * 4 + q) ^ (r & 7)) * 16);
                bA[st][kk] = lds0 + st * STAGE_BYTES + wm * 16384 + row_off + c;
                bW[st][kk] = lds0 + st * STAGE_BYTES + TILE_BYTES + wn * 16384 + row_off + c;
            }
    }
    // staging: a K-tile is 32 + 32 sub-tiles of 8 rows; wave w takes A sub-tiles w + 4 u and W sub-tiles w + 4 u (u = 0..7).  Lane l
    // carries 16 B of row l >> 3, source chunk (l & 7) ^ (l >> 3): offsets from the tile's first row, the same for every tile
    const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
    unsigned voff[16];
    {
        const int row_in = lane >> 3, chunk = (lane & 7) ^ row_in;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            voff[u] = (unsigned)(((wave + 4 * u) * 8 + row_in) * row_bytes + chunk * 16);
            voff[8 + u] = voff[u];
        }
    }

    int tile = blockIdx.x, m0, n0;
    tile_origin(tile, m0, n0);
    const char* a0 = reinterpret_cast<const char*>(a.A) + (size_t)m0 * row_bytes;
    const char* w0 = reinterpret_cast<const char*>(a.W) + (size_t)n0 * row_bytes;

    // prologue of the first tile: K-tile 0 into stage 0
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u < 8) Q4_GLDS(sb, u * 4096, voff[u], a0);
        else Q4_GLDS(sb, TILE_BYTES + (u - 8) * 4096, voff[u], w0);
    }
    if constexpr (INT) {
        for (int i = tid; i < (a.N >> 2); i += 256) {
            colvec[i] = reinterpret_cast<const f32x4*>(a.ln_gamma)[i];
            colvec[256 + i] = reinterpret_cast<const f32x4*>(a.bias)[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    f32x4 acc[8][8];
    int stamp_tile = 0;
    for (;;) {
        unsigned long long tsum[3] = {0ull, 0ull, 0ull};
        const unsigned long long ts_top = q4_now<STAMP>();
        const int next = tile + gridDim.x;
        const bool has_next = next < nwg;
        int m0n = m0, n0n = n0;
        if (has_next) tile_origin(next, m0n, n0n);
        const char* an = reinterpret_cast<const char*>(a.A) + (size_t)m0n * row_bytes;      // !has_next: a harmless reload of this tile's K-tile 0
        const char* wn_ = reinterpret_cast<const char*>(a.W) + (size_t)n0n * row_bytes;

        bf16x8 fa0[8], fw0[8], fa1[8], fw1[8];
        // ---- tile start: fragments of (K-tile 0, k-half 0); K-tile 1 into stage 1 (the previous epilogue's scratch, free since its barrier)
#pragma unroll
        for (int i = 0; i < 8; ++i) Q4_DSREAD(fa0[i], bA[0][0], i * 2048);
#pragma unroll
        for (int i = 0; i < 8; ++i) Q4_DSREAD(fw0[i], bW[0][0], i * 2048);
        {
            const char* a1 = a0 + BK * 2;
            const char* w1 = w0 + BK * 2;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u < 8) Q4_GLDS(sb, STAGE_BYTES + u * 4096, voff[u], a1);
                else Q4_GLDS(sb, STAGE_BYTES + TILE_BYTES + (u - 8) * 4096, voff[u], w1);
            }
        }
        Q4_WAIT_FRAGS(fa0, fw0);
        const unsigned long long ts_loop = q4_now<STAMP>();
        const unsigned long long wall0 = STAMP ? wall_clock64() : 0ull;

        // ---- the K-tiles (nt even): 0 | 1, 2 | ... | nt - 1.  Load slot of K-tile t: K-tile t + 2 of this tile, or K-tile 0 of the next
        {
            const char* as = nt > 2 ? a0 + 2 * BK * 2 : an;
            const char* ws = nt > 2 ? w0 + 2 * BK * 2 : wn_;
            q4_ktile<F16, 0, true, true, true, STAMP>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, as, ws, tsum, wave);
        }
        for (int t = 1; t + 1 < nt; t += 2) {
            const char* as1 = a0 + (size_t)(t + 2) * (BK * 2);
            const char* ws1 = w0 + (size_t)(t + 2) * (BK * 2);
            q4_ktile<F16, 1, false, true, true, STAMP>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, as1, ws1, tsum, wave);
            const bool last_pair = t + 3 >= nt;
            const char* as2 = last_pair ? an : a0 + (size_t)(t + 3) * (BK * 2);
            const char* ws2 = last_pair ? wn_ : w0 + (size_t)(t + 3) * (BK * 2);
            q4_ktile<F16, 0, false, true, true, STAMP>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, as2, ws2, tsum, wave);
        }
        q4_ktile<F16, 1, false, false, false, STAMP>(acc, fa0, fw0, fa1, fw1, bA, bW, sb, voff, a0, w0, tsum, wave);
        const unsigned long long ts_end = q4_now<STAMP>();
        const unsigned long long wall1 = STAMP ? wall_clock64() : 0ull;

        // ---- the next tile's K-tile 0 has landed (this wave's loads; the barrier after the epilogue covers everybody's); the last MFMAs
        // have written their accumulators before the epilogue's first v_accvgpr_read (16 passes: two s_nop 15 are more than enough)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

        // ---- epilogue: the wave's 128 x 128 as two 128 x 64 wave-parts.  Stage 1 is the scratch (last read in phase 0 of the last K-tile,
        // every wave is past that K-tile's barrier)
        char* scratch = smem + STAGE_BYTES;
        constexpr bool FOLDABLE = EPI == EPI_GELU || EPI == EPI_QK;
        const bool fold_st = FOLDABLE && a.stat_in != nullptr;
        if (fold_st) {      // uniform over the workgroup: thread t finishes row m0 + t
            if (a.stat_in_blocks > 4) st_table[tid] = row_stat(a, m0 + tid);
            else {
                float2 st_pv[4];
                row_stat_request(a, m0 + tid, st_pv);
                st_table[tid] = row_stat_finish(a, st_pv);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int vwn = 2 * wn + h;
            f32x4 part[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[i][j] = acc[i][4 * h + j];
            f32x4 bias_pre[4];
            if constexpr (INT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bias_pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                load_bias<EPI>(a, n0, vwn, lane, bias_pre);
            }
            if constexpr (EPI == EPI_GELU || EPI == EPI_QK) {
                gemm_epilogue_staged<EPI, 8, F16>(a, part, m0, n0, wm, vwn, lane, scratch + wave * 8192, bias_pre, fold_st ? st_table : nullptr, nullptr);
            } else {
                gemm_epilogue<EPI, 8, F16, INT>(a, part, m0, n0, wm, vwn, lane, bias_pre, scratch, nullptr, INT ? colvec : nullptr, false);
            }
        }
        if constexpr (EPI == EPI_RESID_XG) {
            if (a.out_bf16 && a.stat_part) {      // uniform: the four wave-parts of a row have left their sums in `red`; thread t adds row t
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const float2* red = reinterpret_cast<const float2*>(scratch + 8 * 4096);
                const float2 p0 = red[tid * 4], p1 = red[tid * 4 + 1], p2 = red[tid * 4 + 2], p3 = red[tid * 4 + 3];
                const int m = m0 + tid;
                if (m < a.M)
                    *reinterpret_cast<float2*>(a.stat_part + 2 * ((size_t)(n0 >> 8) * a.stat_stride + m)) =
                        make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
            }
        }
        if constexpr (STAMP) {
            const unsigned long long ts_epi = q4_now<STAMP>();
            if (a.stamps && blockIdx.x == 8 && stamp_tile < 8 && lane == 0) {
                unsigned long long* o = a.stamps + wave * 64 + stamp_tile * 8;
                o[0] = ts_top; o[1] = ts_loop; o[2] = ts_end; o[3] = ts_epi; o[4] = tsum[0]; o[5] = tsum[1]; o[6] = tsum[2]; o[7] = wall1 - wall0;
            }
            ++stamp_tile;
        }
        if (!has_next) break;
        // every wave is through with the scratch (its LDS reads have returned: lgkmcnt) and has seen its K-tile-0 loads land
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        tile = next;
        m0 = m0n;
        n0 = n0n;
        a0 = an;
        w0 = wn_;
    }
}

template <int EPI, bool F16, bool INT>
int launch_q4_t(const GemmArgs& a, hipStream_t s, int tiles_m, int tiles_n, int cus, int dev) {
    static PerDevice attr;
    {
        std::lock_guard<std::mutex> lk(attr.mu);
        if (!attr.done(dev)) {
            HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_q4_kernel<EPI, F16, INT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            if constexpr (F16 && (EPI == EPI_GELU || INT))
                HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_q4_kernel<EPI, F16, INT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
            attr.mark(dev);
        }
    }
    const int ntile = tiles_m * tiles_n;
    const int slots = cus >= 8 ? cus / 8 * 8 : cus;
    if constexpr (F16 && (EPI == EPI_GELU || INT)) {
        if (a.stamps) {      // measurement build: in-kernel cycle stamps of workgroup 8 (tools/gemm_bench.py)
            gemm_q4_kernel<EPI, F16, INT, true><<<ntile > slots ? slots : ntile, 256, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
            HIPTS_LAUNCH_CHECK();
            return HIPTS_OK;
        }
    }
    gemm_q4_kernel<EPI, F16, INT><<<ntile > slots ? slots : ntile, 256, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace

static std::atomic<long long> g_q4_launches{0};
long long gemm_q4_launch_count() { return g_q4_launches.load(std::memory_order_relaxed); }

// The 4-wave loop for the launches it is built for; *handled = false leaves the launch to gemm_pp_kernel.  `a` carries the launcher's
// raster fields.
int launch_gemm_q4(GemmEpilogue epi, const GemmArgs& a, hipStream_t s, bool* handled) {
    *handled = false;
    if (a.op8 || a.sk_slices > 1) return HIPTS_OK;
    if (a.stamps && !(a.f16 && (epi == EPI_GELU || epi == EPI_RESID_XG))) return HIPTS_OK;
    if (a.M % BM || a.N % BN || a.K % (2 * BK) || a.K < 2 * BK) return HIPTS_OK;
    if (epi != EPI_GELU && epi != EPI_QK && epi != EPI_RESID_XG) return HIPTS_OK;
    const int ld = a.ld_out ? a.ld_out : a.N;
    if (epi == EPI_GELU && ld % 8) return HIPTS_OK;
    if (epi == EPI_QK && a.dim % 64) return HIPTS_OK;
    // per-lane source offsets are 32-bit: 256 rows of a tile
    if ((size_t)256 * a.K * 2 >= ((size_t)1 << 31)) return HIPTS_OK;
    int dev = 0;
    const int cus = current_device_cus(&dev);
    const int tiles_m = a.M / BM, tiles_n = a.N / BN;
    *handled = true;
    g_q4_launches.fetch_add(1, std::memory_order_relaxed);
    const bool f16 = a.f16 != 0;
    switch (epi) {
        case EPI_GELU:
            return f16 ? launch_q4_t<EPI_GELU, true, false>(a, s, tiles_m, tiles_n, cus, dev) : launch_q4_t<EPI_GELU, false, false>(a, s, tiles_m, tiles_n, cus, dev);
        case EPI_QK:
            return f16 ? launch_q4_t<EPI_QK, true, false>(a, s, tiles_m, tiles_n, cus, dev) : launch_q4_t<EPI_QK, false, false>(a, s, tiles_m, tiles_n, cus, dev);
        case EPI_RESID_XG: {
            const bool interior = a.N <= 1024 && !a.pos && !a.res_scale && a.out_bf16 && a.stat_part && !a.rowstat && !a.stat_in;
            if (interior)
                return f16 ? launch_q4_t<EPI_RESID_XG, true, true>(a, s, tiles_m, tiles_n, cus, dev) : launch_q4_t<EPI_RESID_XG, false, true>(a, s, tiles_m, tiles_n, cus, dev);
            return f16 ? launch_q4_t<EPI_RESID_XG, true, false>(a, s, tiles_m, tiles_n, cus, dev) : launch_q4_t<EPI_RESID_XG, false, false>(a, s, tiles_m, tiles_n, cus, dev);
        }
        default: break;
    }
    *handled = false;
    return HIPTS_OK;
}

}  // namespace hipts
