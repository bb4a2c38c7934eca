// gemm.hip -- bf16 MFMA GEMM with fused epilogues for the ViT forward (tagging.py:174).
//
//   C[M,N] = A[M,K] x W[N,K]^T, both operands K-contiguous bf16, fp32 accumulation on
//   v_mfma_f32_16x16x32_bf16.
//
// Geometry (CDNA4: 64-wide waves, 160 KiB LDS/CU, 512-register file per SIMD):
//   block tile 256 x 256 x 64, 512 threads = 8 waves arranged 2 (M) x 4 (N), each wave owns a
//   128 x 64 output tile = 8 x 4 MFMA tiles of 16 x 16 (128 accumulator registers);
//   LDS: 2 stages x (A 32 KiB + W 32 KiB) = 128 KiB -> one workgroup per CU;
//   staging: global_load_lds_dwordx4 (no VGPR round trip).  An operand tile is stored as 32
//   sub-tiles of 8 rows x 64 k = 8 FULL 128-byte lines (1 KiB = one wave instruction; half-line,
//   "fragment shaped" pieces double the texture-addresser work per byte); inside a sub-tile the
//   16-byte chunk index is XORed with the row index (applied on the per-lane global SOURCE address
//   and again on the ds_read_b128 address), which makes the fragment reads bank-conflict free;
//   the next K-step is staged while the current one is multiplied (2-stage pipeline).
// Operands are swapped in the MFMA (W fragment as "A", activation fragment as "B") so that a lane
// ends up with 4 consecutive output COLUMNS of one row: epilogue loads/stores are 16 B (fp32) or
// 8 B (bf16) per lane.  EPI_VT uses the natural order to get 4 consecutive ROWS (tokens) per lane,
// which is what the transposed V layout wants.
// Workgroup ids are remapped so that the workgroups sharing an XCD (ids equal mod 8) walk
// neighbouring tiles and reuse operand panels in that XCD's L2.
// The default kernel (gemm_pp_kernel) is persistent -- one workgroup per CU walks its tiles and requests the
// next tile's first K-tile before its epilogue -- and the same loop serves the CAFormer of the CCIP encoder
// (ccip.hip) through the StarReLU / scaled-residual / bias / residual+LayerNorm epilogues.  Half-precision
// outputs leave through per-wave LDS images as whole 128 B lines (gemm_epilogue_staged, staged_store_rows).
// gemm_dw_kernel (256 x 128 x 32, two workgroups per CU) takes launches with fewer tiles than CUs.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "vit_internal.h"

namespace hipts {
namespace {

#ifndef HIPTS_DW_SNAKE
#define HIPTS_DW_SNAKE 1       // the same snake order in gemm_dw_kernel's two 4 x 4 product blocks per step
#endif
#if HIPTS_DW_SNAKE
#define HIPTS_SNAKE_J(i, j) (((i) & 1) ? 3 - (j) : (j))
#else
#define HIPTS_SNAKE_J(i, j) (j)
#endif
#ifndef HIPTS_MFMA_ORDER
#define HIPTS_MFMA_ORDER 1      // order of a phase's 16 MFMAs (below): 1 = snake, every MFMA shares an operand with its predecessor: forward +0.5 % (5461 / 5460 against 5426 / 5435 images/s, the chip is power-bound), same bits
#endif
#include "gemm_epi.h"


template <int EPI>
__global__ __launch_bounds__(512) void gemm_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 2, wave_n = wave & 3;

    // XCD-aware, bijective remap of the workgroup id (8 XCDs, round-robin dispatch)
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = a.K, nt = K / BK;
    const int w_rows = tiles_n * BN;   // W is allocated zero-padded to a multiple of 256 rows

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_tile(a.A, a.M, K, m0, 0, smem, wave, lane);
    stage_tile(a.W, w_rows, K, n0, 0, smem + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        char* cur = smem + (t & 1) * STAGE_BYTES;
        if (t + 1 < nt) {
            char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
            stage_tile(a.A, a.M, K, m0, t + 1, nxt, wave, lane);
            stage_tile(a.W, w_rows, K, n0, t + 1, nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[8], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = read_frag(cur + TILE_BYTES, wave_n * 4 + j, kk, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = read_frag(cur, wave_m * 8 + i, kk, lane);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (EPI == EPI_VT)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    gemm_epilogue<EPI>(a, acc, m0, n0, wave_m, wave_n, lane);
}

// ---------------------------------------------------------------------------------------------
// Ping-pong main loop (default).  Same tile geometry and LDS images as above, but the 8 waves run
// as two groups of four (wave_m = 0 / 1; waves w and w+4 share a SIMD) staggered by one barrier:
//
//     group 0:          R(0) | C(0) | R(1) | C(1) | ...            | = s_barrier (all 8 waves)
//     group 1:   (idle) |    | R(0) | C(0) | R(1) | ...
//
// R(P) = read segment: the ds_read_b128 fragment loads of phase P, two global_load_lds of the NEXT
// K-tile, the counted waits; C(P) = compute segment: 16 MFMAs.  In every interval one wave of each
// SIMD is in its compute segment, so the matrix pipe never waits for LDS or for a barrier.
// A K-tile is 4 phases: (m-half, k-half) = (0,0) (0,1) (1,0) (1,1); the W fragments of a k-half are
// read once (phases 0/1) and kept in registers for phases 2/3.
// Staging order inside a K-tile follows the order of first use in the next one -- phase 0 and 1: the
// two halves of W; 2: A-low rows (first used by phase 0); 3: A-high rows (first used by phase 2) --
// so every load has >= 2 phases (4 barrier intervals) to land, is retired by a counted
// `s_waitcnt vmcnt(N)` at the end of the R segment one phase before its first reader (RAW: wait,
// then a barrier every reader passes; N = 2 before a new K-tile, 4 before phase 2), and no region is
// restaged sooner than 3 phases after its last ds_read (WAR).
// Barriers are raw s_barrier: __syncthreads() would drain the LDS-DMA queue (vmcnt(0)).
// ---------------------------------------------------------------------------------------------
// MR = 16-row blocks per wave: 8 -> 256-row tiles, 7 -> 224-row tiles (50176 = 224 * 224: for N = 768 the
// grid becomes 672 tiles = 3 rounds of 0.875-size tiles instead of 588 = 3 rounds (2.3 needed) of full ones).
// s_waitcnt vmcnt(n) for a run-time n in 0..8 (the counts of the ping-pong loop differ in its first and last K-tiles)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

// OP8: e4m3 operands.  A K-tile is the same 256 rows x 128 B (now 128 elements), staged by the same code with the
// matrices seen as 16-bit ones of K / 2 columns; one 16x16x128 MFMA (32 cycles) replaces the two 16x16x32 (16 each)
// of a (row block, column block), so the four phases split the column blocks instead of the K halves --
// phase = (m-half, j-half): 4 x 2 MFMAs, the same 256 matrix-pipe cycles -- and barriers, staging order and counted
// waits are unchanged.  W fragments are read in phases 0/1 and kept; A fragments in phases 0 and 2.
// SK (round 4): the split-K tail of GemmArgs::sk_* -- work items past sk_first are (tile, K slice) pairs.  A separate instantiation, so
// the default kernels' code is untouched.
template <int EPI, int MR = 8, bool F16 = false, bool OP8 = false, bool SK = false, bool INT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(HIPTS_INT_DEPTH == 3 ? 256 : 240))) void gemm_pp_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    static_assert(!INT || (EPI == EPI_RESID_XG && MR == 8 && !OP8 && !SK), "INT: the residual epilogue's interior form");
    static_assert(!OP8 || (F16 && MR == 8), "e4m3 operands: half 16-bit outputs, full tiles");
    static_assert(!SK || (!OP8 && MR == 8), "split-K: 16-bit operands, full tiles");
    constexpr int TBM = 2 * MR * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float2 st_table[256];        // folded LayerNorm: (rstd, rstd * mean) of the current tile's rows
    __shared__ float2 sw_red[EPI == EPI_SWIGLU ? 1024 : 1];      // SWIGLU: [256 rows][4 waves] partial row sums of the product
    __shared__ unsigned sk_ticket[SK ? 4 : 1];                  // SK: the ticket lane 0 drew, for the whole workgroup
    __shared__ f32x4 colvec[INT ? (EPI == EPI_RESID_XGI ? 768 : 512) : 1];      // INT: ln_gamma | bias | col_u of the launch's N <= 1024 columns
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;

    // Persistent: workgroup w walks tiles w, w + grid, w + 2 grid, ...  The grid is a multiple of 8 (or the
    // whole problem), so every tile of a workgroup keeps its XCD and the bijective remap below still
    // hands neighbouring tiles to the workgroups that share an L2.
    const int nwg = tiles_m * tiles_n;
    auto tile_origin = [&](int id, int& m0, int& n0) {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        if (a.raster_gn > 0) {
            // Column groups outermost: all row panels x raster_gn column tiles, row-major inside the group, then the next group.
            // An XCD's contiguous chunk of the order then lies inside one group (or two): its slice of W (raster_gn panels) is all the
            // W it ever reads and stays in its L2, a round is 32 / raster_gn row panels x raster_gn column tiles, and a row panel of A
            // is fetched once per column group.
            const int per = tiles_m * a.raster_gn, grp = bid / per, rem = bid - grp * per;
            const int left = tiles_n - grp * a.raster_gn, gn = left < a.raster_gn ? left : a.raster_gn;
            m0 = (rem / gn) * TBM;
            n0 = (grp * a.raster_gn + rem % gn) * BN;
            return;
        }
        if (a.raster_gm > 0) {
            // Wide outputs (fc1: 12 column tiles): row-major order hands the 32 workgroups of an XCD 2.7 row panels x ALL column
            // tiles per round -- the whole W (4.7 MB) plus 1 MB of A against a 4 MB L2, so W is pulled through the fabric again every
            // round.  Groups of raster_gm row panels, walked column-major inside the group: a round is raster_gm row panels x
            // 32 / raster_gm column tiles, and the next rounds keep the same row panels.
            const int per = a.raster_gm * tiles_n, grp = bid / per, rem = bid - grp * per;
            const int left = tiles_m - grp * a.raster_gm, gm = left < a.raster_gm ? left : a.raster_gm;
            m0 = (grp * a.raster_gm + rem % gm) * TBM;
            n0 = (rem / gm) * BN;
            return;
        }
        m0 = (bid / tiles_n) * TBM;
        n0 = (bid % tiles_n) * BN;
    };
    const int K = OP8 ? a.K / 2 : a.K, nt_all = K / BK;      // row stride in 16-bit units
    const int w_rows = tiles_n * BN;
    // Work items: tiles [0, sk_first) whole, then (tile, slice) pairs; without SK an item is a tile.
    const int S = SK ? a.sk_slices : 1;
    const int n_items = SK ? a.sk_first + (nwg - a.sk_first) * S : nwg;
    auto decode = [&](int item, int& tid_, int& slice_, int& kt0_, int& ntl_) {
        if (!SK || item < a.sk_first) {
            tid_ = item; slice_ = -1; kt0_ = 0; ntl_ = nt_all;
        } else {
            const int j = item - a.sk_first;
            tid_ = a.sk_first + j / S;
            slice_ = j - (j / S) * S;
            kt0_ = slice_ * nt_all / S;
            ntl_ = (slice_ + 1) * nt_all / S - kt0_;
        }
    };
    const int scale_w = (127 - a.w_exp) * 0x01010101, scale_1 = 127 * 0x01010101;

    int stamp_tile = 0;
#define PPSTAMP(idx)                                                                                     \
    do {                                                                                                 \
        if (a.stamps && blockIdx.x == 8 && stamp_tile < 8) {                                             \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            const unsigned long long ts_ = __builtin_readcyclecounter();                                 \
            if (lane == 0) a.stamps[wave * 64 + stamp_tile * 8 + (idx)] = ts_;                           \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    } while (0)
    int tile = blockIdx.x, m0, n0;      // `tile`: the work item
    int tix, slice, kt0, nt;             // its tile, K slice (-1: whole K), first K-tile and K-tile count
    decode(tile, tix, slice, kt0, nt);
    tile_origin(tix, m0, n0);
    int par = 0;        // K-tile t of the current output tile lives in LDS stage (par + t) & 1

    // prologue of the first tile: K-tile 0 complete
    if (wave * 4 < 4 * MR) stage_tile(a.A, a.M, K, m0, kt0, smem, wave, lane);     // 4 MR sub-tiles of 8 rows (wave 7 idle for MR = 7)
    stage_tile(a.W, w_rows, K, n0, kt0, smem + TILE_BYTES, wave, lane);
    if constexpr (INT) {
        // the per-column vectors of the residual epilogue, once per workgroup (behind the first K-tile's requests; the barrier that opens
        // the first main loop publishes them)
        for (int i = tid; i < (a.N >> 2); i += 512) {
            colvec[i] = reinterpret_cast<const f32x4*>(a.ln_gamma)[i];
            colvec[256 + i] = reinterpret_cast<const f32x4*>(a.bias)[i];
            if constexpr (EPI == EPI_RESID_XGI) colvec[512 + i] = reinterpret_cast<const f32x4*>(a.col_u)[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    for (;;) {
        f32x4 acc[MR][4];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // This wave's eight staging slots of a K-tile (two per phase), as (global source pointer for
        // K-tile 1, LDS offset inside a stage).  Phase lists of 16 sub-tiles (8 rows x 128 B each), wave w
        // takes entries 2w, 2w+1:  0: W rows 0..127 | 1: W rows 128..255 | 2: A-low | 3: A-high, where
        // A-low = the rows the two wave groups multiply in phases 0/1 (8-row blocks 0..7, 16..23).
        const bf16_t* src[8];
        int dst[8];
        {
            const int row_in = lane >> 3;
            const int chunk = (lane & 7) ^ row_in;
            auto slot = [&](int idx, bool isW, int rb8) {
                int grow = (isW ? n0 : m0) + rb8 * 8 + row_in;
                const int lim = isW ? w_rows : a.M;
                grow = grow < lim ? grow : lim - 1;
                src[idx] = (isW ? a.W : a.A) + (size_t)grow * K + (size_t)(kt0 + 1) * BK + chunk * 8;
                dst[idx] = (isW ? TILE_BYTES : 0) + rb8 * 1024;
            };
            // A-low = the first 64 rows of each wave group (8-row blocks g*2*MR + 0..7): 16 sub-tiles, 2 per wave.
            // A-high = the remaining (2 MR - 8) blocks of each group: 16 (MR = 8) or 12 (MR = 7) sub-tiles; with
            // MR = 7 waves 4..7 have one (n_hi = 1) and their counted waits are one lower.
            for (int u = 0; u < 2; ++u) {
                const int e = 2 * wave + u;
                slot(0 + u, true, e);
                slot(2 + u, true, 16 + e);
                slot(4 + u, false, e < 8 ? e : 2 * MR + (e - 8));
            }
            if constexpr (MR == 8) {
                for (int u = 0; u < 2; ++u) {
                    const int e = 2 * wave + u;
                    slot(6 + u, false, e < 8 ? 8 + e : 2 * MR + 8 + (e - 8));
                }
            } else if constexpr (MR == 6) {
                // 8 sub-tiles: group 0 blocks 8..11, group 1 blocks 2MR+8 .. 2MR+11; one per wave
                slot(6, false, wave < 4 ? 8 + wave : 2 * MR + 8 + (wave - 4));
                src[7] = src[6];
                dst[7] = dst[6];
            } else {
                static_assert(MR == 7, "tile heights: 256, 224 or 192 rows");
                // 12 sub-tiles: group 0 blocks 8..13, group 1 blocks 2MR+8 .. 2MR+13; waves 0..3 take two, 4..7 one
                auto hi = [](int f) { return f < 6 ? 8 + f : 2 * MR + 8 + (f - 6); };
                if (wave < 4) {
                    slot(6, false, hi(2 * wave));
                    slot(7, false, hi(2 * wave + 1));
                } else {
                    slot(6, false, hi(8 + (wave - 4)));
                    src[7] = src[6];
                    dst[7] = dst[6];
                }
            }
        }
        const int n_hi = (MR == 8 || (MR == 7 && wave < 4)) ? 2 : 1;
        // K-tile 0 landed; for every tile but the first this also retires the previous epilogue's stores,
        // which had that epilogue and these address computations to drain
        PPSTAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PPSTAMP(1);
        __builtin_amdgcn_s_barrier();
        if (wave_m == 1) __builtin_amdgcn_s_barrier();      // stagger group 1 by one interval
        PPSTAMP(2);
        if (a.stamps && blockIdx.x == 8 && stamp_tile < 8 && lane == 0) a.stamps[wave * 64 + stamp_tile * 8 + 6] = wall_clock64();
        if (HIPTS_STAGE_AHEAD && !HIPTS_STAGE_ORDER_OLD && nt > 1) {
            // "phase 3 of K-tile -1": A-low and W-low of K-tile 1 into the other stage (the previous epilogue's scratch, free now)
            char* st1 = smem + ((par + 1) & 1) * STAGE_BYTES;
            for (int k : {4, 5, 0, 1}) {
                glds16(src[k], st1 + dst[k]);
                src[k] += BK;
            }
        }

        bf16x8 wf[2][4];
        i32x8 wf8[4], af8[4];
        unsigned xpf = 0;       // destination of the epilogue prefetch: stays allocated until the epilogue's own loads have been waited for
        for (int t = 0; t < nt; ++t) {
            const char* cur = smem + ((par + t) & 1) * STAGE_BYTES;
            char* nxt = smem + ((par + t + 1) & 1) * STAGE_BYTES;
            const bool more = t + 1 < nt;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mh = p >> 1, kk = p & 1;
                // ---------------- R(P): issue only -- fragment reads of this phase, two loads of the next K-tile
                bf16x8 af[4];
                if constexpr (OP8) {
                    if (p < 2) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) wf8[2 * kk + jj] = read_frag8(cur + TILE_BYTES, wave_n * 4 + 2 * kk + jj, lane);
                    }
                    if (kk == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) af8[i] = read_frag8(cur, wave_m * MR + mh * 4 + i, lane);
                    }
                } else {
                if (p < 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[kk][j] = read_frag(cur + TILE_BYTES, wave_n * 4 + j, kk, lane);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (mh * 4 + i < MR) af[i] = read_frag(cur, wave_m * MR + mh * 4 + i, kk, lane);
                }
                if (more) {
                    // Issue order of the next K-tile's eight sub-tile loads (slots: 0,1 W-low  2,3 W-high  4,5 A-low  6,7 A-high):
                    // what the next tile reads first goes first -- A-low and W-low in phase 0, W-high in phase 1 (all read in the
                    // next phase 0: 4, 4 and 3 phases to land), A-high in phase 2 (read in the next phase 2: 4 phases; its region
                    // was last read in the previous tile's phase 3, restaged 3 phases later).  With the former order (W, W, A-low,
                    // A-high over phases 0..3) A-low had only 2 phases = ~300 ns, less than an L2 round trip.
                    auto issue = [&](int k) {
                        glds16(src[k], nxt + dst[k]);
                        src[k] += BK;
                    };
                    if (HIPTS_STAGE_ORDER_OLD) {
                        issue(2 * p);
                        if (p < 3 || n_hi == 2) issue(2 * p + 1);
                        if (p == 3) {           // W and A-low of tile t+1 landed; only this phase's A-high may be in flight
                            if (n_hi == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                        } else if (p == 1) {
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A-high of this tile landed
                        }
                    } else if (HIPTS_STAGE_AHEAD) {
                        // A region is restaged as soon as it is free (>= 2 phases after its last ds_read, >= 3 where the reader
                        // may be the staggered group): W-high and A-high of K-tile t+1 go to the other stage in phases 0 and 1, and
                        // A-low / W-low of K-tile t+2 go into THIS stage in phase 3 (last read in phase 1).  Every load then has
                        // >= 3 phases (W-high) or 4 (the rest) before the counted wait that retires it:
                        //   end of R(1): A-high of this tile -- issued since: A-low/W-low(t+1) 4, W-high(t+1) 2, A-high(t+1) n_hi
                        //   end of R(3): all of tile t+1 but its A-high -- issued since: A-high(t+1) n_hi, A-low/W-low(t+2) 4
                        if (p == 0) {
                            issue(2); issue(3);
                        } else if (p == 1) {
                            issue(6);
                            if (n_hi == 2) issue(7);
                            wait_vmcnt(6 + n_hi);
                        } else if (p == 3) {
                            if (t + 2 < nt) {
                                for (int k : {4, 5, 0, 1}) {
                                    glds16(src[k], const_cast<char*>(cur) + dst[k]);
                                    src[k] += BK;
                                }
                                wait_vmcnt(n_hi + 4);
                            } else {
                                wait_vmcnt(n_hi);
                            }
                        }
                    } else {
                        if (p == 0) {
                            issue(4); issue(5); issue(0); issue(1);
                            if (HIPTS_STAGE_W_EARLY) { issue(2); issue(3); }
                        } else if (p == 1) {
                            if (!HIPTS_STAGE_W_EARLY) { issue(2); issue(3); }
                            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A-high of this tile (issued in its predecessor's phase 2) landed
                        } else if (p == 2) {
                            issue(6);
                            if (n_hi == 2) issue(7);
                        } else {                // W and A-low of tile t+1 landed; only its A-high may be in flight
                            if (n_hi == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                        }
                    }
                } else if (p == 1) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if constexpr (INT) {
                    // HIPTS_EPI_PREFETCH (A/B): behind the loop's last counted wait, one dword of each of the 256 lines of this wave's part of
                    // the fp32 stream is requested into a register nobody reads -- the epilogue's four load steps then find the tile in L2
                    // instead of paying the trip beyond it four times in sequence.  (Earlier in the loop it cannot go: vmcnt retires in order,
                    // every later wait for a K-tile would wait for these loads too.)
                    if (!more && p == 2 && a.epi_prefetch) {
                        const int ldx = a.ld_out ? a.ld_out : a.N;
                        const char* xb = reinterpret_cast<const char*>(a.out_f32 + (size_t)(m0 + wave_m * 128) * ldx + n0 + wave_n * 64);
                        const unsigned o0 = (unsigned)(((lane >> 1) * ldx + (lane & 1) * 32) * 4), ost = (unsigned)(32 * ldx * 4);
                        asm volatile("global_load_dword %0, %1, %5\n\tglobal_load_dword %0, %2, %5\n\tglobal_load_dword %0, %3, %5\n\tglobal_load_dword %0, %4, %5"
                                     : "=&v"(xpf)
                                     : "v"(o0), "v"(o0 + ost), "v"(o0 + 2 * ost), "v"(o0 + 3 * ost), "s"(xb)
                                     : "memory");
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- C(P): the reads were issued a whole interval ago
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if constexpr (OP8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * kk + jj;
                            if constexpr (EPI == EPI_VT)
                                acc[mh * 4 + i][j] = mfma_16x16x128_e4m3(af8[i], wf8[j], acc[mh * 4 + i][j], scale_1, scale_w);
                            else
                                acc[mh * 4 + i][j] = mfma_16x16x128_e4m3(wf8[j], af8[i], acc[mh * 4 + i][j], scale_w, scale_1);
                        }
                } else {
#pragma unroll
                for (int ij = 0; ij < 16; ++ij) {
                    // HIPTS_MFMA_ORDER (measurement): 0 = rows outer (the A fragment stays for four MFMAs), 1 = snake (every MFMA shares one
                    // operand with its predecessor), 2 = columns outer (the W fragment stays)
#if HIPTS_MFMA_ORDER == 2
                    const int j = ij >> 2, i = ij & 3;
#elif HIPTS_MFMA_ORDER == 1
                    const int i = ij >> 2, j = (i & 1) ? 3 - (ij & 3) : (ij & 3);
#else
                    const int i = ij >> 2, j = ij & 3;
#endif
                    if (mh * 4 + i >= MR) continue;
                    if constexpr (EPI == EPI_VT)
                        acc[mh * 4 + i][j] = mfma_16x16x32<F16>(af[i], wf[kk][j], acc[mh * 4 + i][j]);
                    else
                        acc[mh * 4 + i][j] = mfma_16x16x32<F16>(wf[kk][j], af[i], acc[mh * 4 + i][j]);
                }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wave_m == 0) __builtin_amdgcn_s_barrier();      // balance the stagger
        PPSTAMP(3);
        if (a.stamps && blockIdx.x == 8 && stamp_tile < 8 && lane == 0) a.stamps[wave * 64 + stamp_tile * 8 + 7] = wall_clock64();

        // Every wave is past its last fragment read and no LDS-DMA is in flight.  Stage (par + nt) & 1 (last
        // read in K-tile nt - 2) receives K-tile 0 of this workgroup's next tile now, so that its HBM
        // latency is covered by the epilogue; the stage of the last K-tile is the epilogue's scratch.
        f32x4 bias_pre[4];
        if constexpr (INT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bias_pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};      // INT reads bias (and gamma, col_u) from colvec
        } else {
            load_bias<EPI>(a, n0, wave_n, lane, bias_pre);      // in flight while the next prologue is issued
        }
        constexpr bool FOLDABLE = EPI == EPI_VT || EPI == EPI_GELU || EPI == EPI_STAR || EPI == EPI_QK || EPI == EPI_QK_ROPE || EPI == EPI_SWIGLU ||
                                  EPI == EPI_RESID_ROWSTAT || EPI == EPI_RESID_XGI;
        const bool fold_st = FOLDABLE && a.stat_in != nullptr;
        const bool fold_many = fold_st && a.stat_in_blocks > 4;      // EVA02's fc2: 22 partial pairs per row (one per 256-column tile of the SwiGLU launch)
        float2 st_pv[4];
        if (fold_st && !fold_many && tid < TBM) row_stat_request(a, m0 + tid, st_pv);
        const int next = tile + gridDim.x;
        const bool has_next = next < n_items;
        int m0n = 0, n0n = 0, tixn = 0, slicen = -1, kt0n = 0, ntn = nt_all;
        const int par_next = (par + nt) & 1;
        if (has_next) {
            decode(next, tixn, slicen, kt0n, ntn);
            tile_origin(tixn, m0n, n0n);
            char* st = smem + par_next * STAGE_BYTES;
            if (wave * 4 < 4 * MR) stage_tile(a.A, a.M, K, m0n, kt0n, st, wave, lane);
            stage_tile(a.W, w_rows, K, n0n, kt0n, st + TILE_BYTES, wave, lane);
        }
        PPSTAMP(4);
        bool run_epilogue = true;       // uniform over the workgroup
        if constexpr (SK) {
            if (slice >= 0) {
                constexpr size_t SLAB = (size_t)TBM * BN * 4;
                const int tl = tix - a.sk_first;
                unsigned* tickets = reinterpret_cast<unsigned*>(a.sk_ws);
                char* slabs = reinterpret_cast<char*>(a.sk_ws) + 4096 + (size_t)tl * S * SLAB;
                // (1) this slice's accumulators into its slab, write-through (sc1: no release fence needed, the bytes are at the memory
                // side when vmcnt says so): [wave][i][j][lane] f32x4, 1 KB per store instruction
                {
                    const auto rs = __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)slice * SLAB, 0, (int)SLAB, 0x00020000);
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, (((wave * MR + i) * 4 + j) * 64 + lane) * 16, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains (this also lands the next item's first K-tile)
                __builtin_amdgcn_s_barrier();
                if (threadIdx.x == 0)
                    sk_ticket[0] = __hip_atomic_fetch_add(tickets + tl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const unsigned ticket = sk_ticket[0];
                if (ticket != (unsigned)(S - 1)) {
                    run_epilogue = false;                              // somebody else finishes this tile
                } else {
                    // (2) the last arriver: every slab is complete.  One agent-scope acquire (this CU's L1), then plain loads; the
                    // ticket word goes back to zero for the next launch (nobody else touches it any more).
                    if (threadIdx.x == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        __hip_atomic_store(tickets + tl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    // slabs in slice order: the sum does not depend on which slice this workgroup computed
                    const f32x4* sl = reinterpret_cast<const f32x4*>(slabs) + (size_t)(wave * MR * 4) * 64 + lane;
#pragma unroll
                    for (int i = 0; i < MR; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = sl[(size_t)(i * 4 + j) * 64];
                        for (int sx = 1; sx < S; ++sx) {
                            const f32x4* sp = sl + (size_t)sx * (SLAB / 16);
                            f32x4 v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = sp[(size_t)(i * 4 + j) * 64];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] = acc[i][j] + v[j];
                        }
                    }
                }
            }
        }
        if (run_epilogue) {
        if (fold_st) {      // uniform over the workgroup
            if (fold_many) {
                if (tid < TBM) st_table[tid] = row_stat(a, m0 + tid);       // sums the partials in index order (deterministic), 8 loads in flight
            } else if (tid < TBM) st_table[tid] = row_stat_finish(a, st_pv);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        bool staged = false;
        if constexpr (EPI == EPI_VT || EPI == EPI_GELU || EPI == EPI_STAR || EPI == EPI_QK || EPI == EPI_QK_ROPE || EPI == EPI_SWIGLU) {
            const bool ok = EPI == EPI_VT                            ? (a.tokens % 8 == 0 && a.tokens_pad % 8 == 0)
                            : (EPI == EPI_QK || EPI == EPI_QK_ROPE) ? (a.dim % 64 == 0)
                                                                     : ((a.ld_out ? a.ld_out : a.N) % 8 == 0);
            if (ok) {
                float2* red = (EPI == EPI_SWIGLU && a.stat_part) ? sw_red : nullptr;
                gemm_epilogue_staged<EPI, MR, F16>(a, acc, m0, n0, wave_m, wave_n, lane,
                                                   smem + ((par + nt + 1) & 1) * STAGE_BYTES + wave * 8192, bias_pre, fold_st ? st_table : nullptr, red);
                staged = true;
                if constexpr (EPI == EPI_SWIGLU) {
                    if (red) {      // uniform: the four waves that hold a row meet; sw_red is written again only after the next tile's main loop
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (tid < TBM) {
                            const float2 p0 = red[tid * 4], p1 = red[tid * 4 + 1], p2 = red[tid * 4 + 2], p3 = red[tid * 4 + 3];
                            const int m = m0 + tid;
                            if (m < a.M)
                                *reinterpret_cast<float2*>(a.stat_part + 2 * ((size_t)(n0 >> 8) * a.stat_stride + m)) =
                                    make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
                        }
                    }
                }
            }
        }
        if (!staged)
            gemm_epilogue<EPI, MR, F16, INT>(a, acc, m0, n0, wave_m, wave_n, lane, bias_pre, smem + ((par + nt + 1) & 1) * STAGE_BYTES,
                                             fold_st ? st_table : nullptr, INT ? colvec : nullptr);
        }       // run_epilogue
        if (a.epi_prio) __builtin_amdgcn_s_setprio(0);
        if constexpr (INT) asm volatile("" ::"v"(xpf));
        PPSTAMP(5);
        ++stamp_tile;
        if (!has_next) break;
        tile = next;
        tix = tixn;
        slice = slicen;
        kt0 = kt0n;
        nt = ntn;
        m0 = m0n;
        n0 = n0n;
        par = par_next;
    }
}

// ---------------------------------------------------------------------------------------------
// Ping-pong with 32-MFMA segments (HIPTS_GEMM=pp2): two phases per K-tile (phase = m-half, both
// k-halves: 4 x 4 x 2 = 32 MFMAs per segment), half the barriers per flop.  In-kernel s_memtime stamps
// (template parameter STAMP, tools/gemm_bench.py with HIPTS_GEMM_STAMPS=1) show where a K-tile's
// ~4.6 k cycles go at 4096^3: a read segment costs 300-650 cycles to ISSUE (16 ds_read_b128 + 6
// LDS-DMA loads at ~60-100 cycles each), its counted vmcnt wait another 350-640 (LDS-DMA issue ->
// landed ~2.5 k cycles under load, more than the one K-tile of lead a 2 x 64 KiB ring allows), the
// 32 MFMAs 610-650.  So the read segment, not the barrier count, bounds the interval.
//   R(0): 8 W + 8 A fragment reads, 6 of the wave's 8 loads of the next K-tile (all of W and the
//         A-low rows: what phase 0 of the next tile reads);  wait vmcnt(6) -> A-high of THIS tile landed
//   R(1): 8 A fragment reads, the 2 A-high loads of the next tile;  wait vmcnt(2) -> W/A-low of next tile landed
// RAW: every wait sits one phase before the first reader with a barrier in between; WAR: a region is
// restaged two phases after its last ds_read.
// ---------------------------------------------------------------------------------------------
#define PP2_STAMP(idx)                                                                                   \
    if constexpr (STAMP) {                                                                               \
        if (blockIdx.x == 0 && (t == 10 || t == 11)) {                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            unsigned long long ts_;                                                                      \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");                 \
            if (lane == 0) a.stamps[wave * 64 + (t - 10) * 16 + mh * 8 + (idx)] = ts_;                   \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    }

template <int EPI, bool STAMP = false>
__global__ __launch_bounds__(512) void gemm_pp2_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int K = a.K, nt = K / BK;
    const int w_rows = tiles_n * BN;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_tile(a.A, a.M, K, m0, 0, smem, wave, lane);
    stage_tile(a.W, w_rows, K, n0, 0, smem + TILE_BYTES, wave, lane);

    // staging slots of this wave: [0..3] W sub-tiles 4w..4w+3, [4,5] A-low, [6,7] A-high
    const bf16_t* src[8];
    int dst[8];
    {
        const int row_in = lane >> 3;
        const int chunk = (lane & 7) ^ row_in;
        auto slot = [&](int idx, bool isW, int rb8) {
            int grow = (isW ? n0 : m0) + rb8 * 8 + row_in;
            const int lim = isW ? w_rows : a.M;
            grow = grow < lim ? grow : lim - 1;
            src[idx] = (isW ? a.W : a.A) + (size_t)grow * K + BK + chunk * 8;
            dst[idx] = (isW ? TILE_BYTES : 0) + rb8 * 1024;
        };
        for (int u = 0; u < 4; ++u) slot(u, true, 4 * wave + u);
        for (int u = 0; u < 2; ++u) {
            const int e = 2 * wave + u;
            slot(4 + u, false, e < 8 ? e : e + 8);          // A-low : 8-row blocks 0..7, 16..23
            slot(6 + u, false, e < 8 ? 8 + e : 16 + e);     // A-high: 8-row blocks 8..15, 24..31
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave_m == 1) __builtin_amdgcn_s_barrier();      // stagger group 1 by one interval

    bf16x8 wf[2][4];
    for (int t = 0; t < nt; ++t) {
        const char* cur = smem + (t & 1) * STAGE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
        const bool more = t + 1 < nt;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            // ---------------- R: issue only
            PP2_STAMP(0)
            bf16x8 af[2][4];
            if (mh == 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[kk][j] = read_frag(cur + TILE_BYTES, wave_n * 4 + j, kk, lane);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i) af[kk][i] = read_frag(cur, wave_m * 8 + mh * 4 + i, kk, lane);
            if (more) {
                if (mh == 0) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        glds16(src[u], nxt + dst[u]);
                        src[u] += BK;
                    }
                    PP2_STAMP(1)
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                } else {
#pragma unroll
                    for (int u = 6; u < 8; ++u) {
                        glds16(src[u], nxt + dst[u]);
                        src[u] += BK;
                    }
                    PP2_STAMP(1)
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
            } else if (mh == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            PP2_STAMP(2)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PP2_STAMP(3)
            // ---------------- C: 32 MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            PP2_STAMP(4)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (EPI == EPI_VT)
                            acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], wf[kk][j], acc[mh * 4 + i][j], 0, 0, 0);
                        else
                            acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], af[kk][i], acc[mh * 4 + i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
            PP2_STAMP(5)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PP2_STAMP(6)
        }
    }
    if (wave_m == 0) __builtin_amdgcn_s_barrier();      // balance the stagger
    gemm_epilogue<EPI>(a, acc, m0, n0, wave_m, wave_n, lane);
}

// ---------------------------------------------------------------------------------------------
// Three-stage variant for epilogue-heavy shapes: block tile 256 x 128 x 32, 256 threads = 4 waves
// (2 x 2, each wave still 128 x 64), LDS ring of 3 x 24 KiB = 72 KiB -> TWO workgroups per CU.
// With K = 768 the fp32 residual read-modify-write (or the bf16 store of a 3072-wide hidden) of a
// tile costs about as much time as its 12..24 K-steps; with one workgroup per CU all CUs hit their
// epilogues together and the matrix pipe idles during that HBM burst.  Two independent workgroups
// per CU drift apart, so one streams its epilogue while the other multiplies.
// One raw barrier per K-step: wait (counted) for stage t, barrier, issue stage t+2 into the slot
// everybody just finished reading, multiply stage t.
// ---------------------------------------------------------------------------------------------
constexpr int S3_BN = 128, S3_BK = 32, S3_STAGE = (BM + S3_BN) * S3_BK * 2, S3_LDS = 3 * S3_STAGE;   // 24 KiB, 72 KiB

__device__ __forceinline__ void s3_stage(const GemmArgs& a, int m0, int n0, int w_rows, int kt, char* st, int wave, int lane) {
    const int row_in = lane >> 2;
    const int chunk = (lane & 3) ^ (((row_in >> 3) & 1) << 1);
    const size_t koff = (size_t)kt * S3_BK + chunk * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rb = wave * 4 + i;
        int grow = m0 + rb * 16 + row_in;
        grow = grow < a.M ? grow : a.M - 1;
        glds16(a.A + (size_t)grow * a.K + koff, st + rb * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rb = wave * 2 + i;
        int grow = n0 + rb * 16 + row_in;
        grow = grow < w_rows ? grow : w_rows - 1;
        glds16(a.W + (size_t)grow * a.K + koff, st + BM * S3_BK * 2 + rb * 1024);
    }
}

__device__ __forceinline__ bf16x8 s3_frag(const char* base, int rowblk, int lane) {
    const int r = lane & 15;
    const int c = (lane >> 4) ^ (((r >> 3) & 1) << 1);
    return *reinterpret_cast<const bf16x8*>(base + rowblk * 1024 + r * 64 + c * 16);
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_s3_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * S3_BN;
    const int nt = a.K / S3_BK;
    const int w_rows = ((a.N + 255) / 256) * 256;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    s3_stage(a, m0, n0, w_rows, 0, smem, wave, lane);
    if (nt > 1) s3_stage(a, m0, n0, w_rows, 1, smem + S3_STAGE, wave, lane);
    int slot = 0;
    for (int t = 0; t < nt; ++t) {
        // stage t complete (6 loads of stage t+1 may stay in flight), visible to all after the barrier
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) {
            int s2 = slot + 2;
            s2 = s2 >= 3 ? s2 - 3 : s2;
            s3_stage(a, m0, n0, w_rows, t + 2, smem + s2 * S3_STAGE, wave, lane);
        }
        const char* cur = smem + slot * S3_STAGE;
        bf16x8 af[8], wf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = s3_frag(cur + BM * S3_BK * 2, wave_n * 4 + j, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = s3_frag(cur, wave_m * 8 + i, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (EPI == EPI_VT)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
            }
        slot = slot == 2 ? 0 : slot + 1;
    }
    gemm_epilogue<EPI>(a, acc, m0, n0, wave_m, wave_n, lane);
}

// ---------------------------------------------------------------------------------------------
// Dual-workgroup loop ("dw"): 256 x 128 x 32 tiles, 4 waves (2 x 2, 128 x 64 each), three 24 KB LDS
// stages = 72 KB, <= 256 VGPRs: TWO workgroups per CU, one wave of each per SIMD.  The point is the
// epilogue: with one 512-thread workgroup per CU the matrix pipe idles while that workgroup adds bias /
// GELU / residual and drains its stores (4..20 us on a 18 us K = 768 main loop); here the other resident
// workgroup keeps multiplying meanwhile.  A wave therefore has no lock-step partner on its SIMD and
// software-pipelines its own fragments in registers:
//
//   step s:   16 MFMA A-low(s) x W(s), A-high(s) read in their shadow | waits | barrier B_s |
//             16 MFMA A-high(s) x W(s), W(s+1) / A-low(s+1) read and stage s+3 issued in their shadow
//
// B_s is the only barrier of a step.  Before it a wave has retired its own reads of stage s
// (lgkmcnt(0)) and its own loads of stage s+1 (vmcnt(6): stage s+2 stays in flight); after it slot
// s % 3 is free for stage s+3 and stage s+1 is visible.  Every load has two steps to land.
// LDS image: a 16-row x 32-k block is 1 KB of 64 B rows; the 16 B chunk c of row r is stored at chunk
// c ^ ((r >> 2) & 3) (swizzle applied to the global source address, the LDS-DMA writes lane-linear),
// which makes the 16 lanes of a ds_read_b128 phase (one k-chunk, rows 0..15) hit 16 different bank groups.
// ---------------------------------------------------------------------------------------------
constexpr int DW_BM = 256, DW_BN = 128, DW_BK = 32;
constexpr int DW_A_BYTES = DW_BM * DW_BK * 2;                    // 16 KiB
constexpr int DW_STAGE = (DW_BM + DW_BN) * DW_BK * 2;            // 24 KiB
constexpr int DW_LDS = 3 * DW_STAGE;                             // 72 KiB

template <int EPI, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_dw_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * DW_BM, n0 = tn * DW_BN;
    const int K = a.K, nt = K / DW_BK;
    const int w_rows = ((a.N + 255) / 256) * 256;

    // six staging slots per wave and stage: A blocks 4 wave .. 4 wave + 3, W blocks 2 wave, 2 wave + 1
    const bf16_t* src[6];
    int dst[6];
    {
        const int r = lane >> 2, q = (lane & 3) ^ ((r >> 2) & 3);     // four lanes fetch one 64 B row segment
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rb = wave * 4 + i;
            int grow = m0 + rb * 16 + r;
            grow = grow < a.M ? grow : a.M - 1;
            src[i] = a.A + (size_t)grow * K + q * 8;
            dst[i] = rb * 1024;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = wave * 2 + i;
            int grow = n0 + rb * 16 + r;
            grow = grow < w_rows ? grow : w_rows - 1;
            src[4 + i] = a.W + (size_t)grow * K + q * 8;
            dst[4 + i] = DW_A_BYTES + rb * 1024;
        }
    }
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ (((lane & 15) >> 2) & 3)) * 16);
    auto issue = [&](int i, char* st) {
        glds16(src[i], st + dst[i]);
        src[i] += DW_BK;
    };
    auto frag = [&](const char* base, int rb) { return *reinterpret_cast<const bf16x8*>(base + rb * 1024 + frag_off); };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool trace = a.stamps && a.trace && blockIdx.x < 4096;
    if (trace && tid == 0) {
        a.stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_getreg(63492);
        a.stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg(63508);
        a.stamps[blockIdx.x * 8 + 2] = wall_clock64();
    }
    const int npro = nt < 3 ? nt : 3;
    for (int s0 = 0; s0 < npro; ++s0)
#pragma unroll
        for (int i = 0; i < 6; ++i) issue(i, smem + s0 * DW_STAGE);
    if (npro == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (npro == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 wfr[2][4], alo[2][4], ahi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wfr[0][j] = frag(smem + DW_A_BYTES, wave_n * 4 + j);
#pragma unroll
    for (int i = 0; i < 4; ++i) alo[0][i] = frag(smem, wave_m * 8 + i);

    int slot = 0;       // LDS slot of stage s
    unsigned long long wait_cyc = 0, bar_cyc = 0;
    const unsigned long long loop_t0 = a.stamps ? __builtin_readcyclecounter() : 0;
    const unsigned long long wall_t0 = a.stamps ? wall_clock64() : 0;
    auto mm = [&](f32x4& c, const bf16x8& av, const bf16x8& wv) {
        if constexpr (EPI == EPI_VT) c = mfma_16x16x32<F16>(av, wv, c);
        else c = mfma_16x16x32<F16>(wv, av, c);
    };
    // One step; H1/H2/H3 = stage st+1 / st+2 / st+3 exists (compile time: straight-line code, exact
    // compiler-generated lgkmcnt), U = register set of this step.
    auto step = [&](auto H1, auto H2, auto H3, auto U) {
        constexpr bool has1 = decltype(H1)::value, has2 = decltype(H2)::value, has3 = decltype(H3)::value;
        constexpr int u = decltype(U)::value;
        const char* cur = smem + slot * DW_STAGE;
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const char* nx = smem + slot1 * DW_STAGE;
        // 16 MFMA A-low x W (set u was read during the previous step) with the A-high reads in their shadow
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mm(acc[i][HIPTS_SNAKE_J(i, j)], alo[u][i], wfr[u][HIPTS_SNAKE_J(i, j)]);
                if (i == 0) ahi[j] = frag(cur, wave_m * 8 + 4 + j);
                __builtin_amdgcn_sched_barrier(0);
            }
        if constexpr (has1) {
            // own reads of stage st retired, own loads of stage st + 1 landed (stage st + 2 may fly)
            unsigned long long q0 = 0, q1 = 0;
            if (a.stamps) q0 = __builtin_readcyclecounter();
            if constexpr (has2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (a.stamps) q1 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();
            if (a.stamps) { const unsigned long long q2 = __builtin_readcyclecounter(); wait_cyc += q1 - q0; bar_cyc += q2 - q1; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // 16 MFMA A-high x W with the eight reads of set u ^ 1 and the six loads of stage st + 3 in their shadow
        char* fill = smem + slot * DW_STAGE;        // the slot of stage st is free now
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mm(acc[4 + i][HIPTS_SNAKE_J(i, j)], ahi[i], wfr[u][HIPTS_SNAKE_J(i, j)]);
                const int idx = i * 4 + j;
                if constexpr (has1) {
                    if (idx < 4) wfr[u ^ 1][idx] = frag(nx + DW_A_BYTES, wave_n * 4 + idx);
                    else if (idx < 8) alo[u ^ 1][idx - 4] = frag(nx, wave_m * 8 + (idx - 4));
                }
                if constexpr (has3) {
                    if (idx >= 8 && idx < 14) issue(idx - 8, fill);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        slot = slot1;
    };
    using T = std::true_type;
    using F = std::false_type;
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    for (int s = 0; s + 6 <= nt; s += 2) {      // nt is even (K % 64 == 0)
        step(T{}, T{}, T{}, U0{});
        step(T{}, T{}, T{}, U1{});
    }
    if (nt >= 4) {
        step(T{}, T{}, T{}, U0{});
        step(T{}, T{}, F{}, U1{});
    }
    step(T{}, F{}, F{}, U0{});
    step(F{}, F{}, F{}, U1{});
    if (trace && tid == 0) a.stamps[blockIdx.x * 8 + 3] = wall_clock64();
    if (a.stamps && !trace && blockIdx.x == 8 && lane == 0) {
        a.stamps[wave * 64 + 0] = __builtin_readcyclecounter() - loop_t0;
        a.stamps[wave * 64 + 1] = wait_cyc;
        a.stamps[wave * 64 + 2] = bar_cyc;
        a.stamps[wave * 64 + 3] = wall_clock64() - wall_t0;
    }
    __builtin_amdgcn_s_barrier();       // every wave is past its last fragment read; all loads have landed
    if constexpr (EPI == EPI_VT || EPI == EPI_GELU || EPI == EPI_STAR || EPI == EPI_QK || EPI == EPI_QK_ROPE || EPI == EPI_SWIGLU) {
        const bool ok = EPI == EPI_VT                            ? (a.tokens % 8 == 0 && a.tokens_pad % 8 == 0)
                        : (EPI == EPI_QK || EPI == EPI_QK_ROPE) ? (a.dim % 64 == 0)
                                                                 : ((a.ld_out ? a.ld_out : a.N) % 8 == 0);
        if (ok) {
            const float2* st_tab = nullptr;
            if (a.stat_in) {        // folded LayerNorm: finish this tile's 256 row statistics (one row per thread) into LDS past the staging images
                float2 pv[4];
                row_stat_request(a, m0 + tid, pv);
                float2* tab = reinterpret_cast<float2*>(smem + 4 * 8192 * 2);
                tab[tid] = row_stat_finish(a, pv);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                st_tab = tab;
            }
            gemm_epilogue_staged<EPI, 8, F16>(a, acc, m0, n0, wave_m, wave_n, lane, smem + wave * 8192, nullptr, st_tab);
            if (trace && tid == 0) a.stamps[blockIdx.x * 8 + 4] = wall_clock64();
            return;
        }
    }
    gemm_epilogue<EPI, 8, F16>(a, acc, m0, n0, wave_m, wave_n, lane, nullptr, smem);      // all of LDS is free: scratch of the RESID_XG copy
    if (trace && tid == 0) a.stamps[blockIdx.x * 8 + 4] = wall_clock64();
}

// HIPTS_GEMM selects the main loop for A/B runs: "pp" (default) ping-pong with 16-MFMA segments;
// "pp2" 32-MFMA segments (better at K >= 4096, slightly worse on the ViT's K = 768 shapes);
// "s3" three-stage 256x128 tile, two workgroups per CU; "v1" simple two-barrier loop.
// Bit mask over epilogue numbers whose launches take the 4-wave loop of gemm4.hip where it is built (HIPTS_GEMM_Q4; hiptsdbg_set_gemm_q4
// changes it at run time for A/B runs and the comparison tests)
std::atomic<long long> g_q4_mask{-1};
unsigned gemm_q4_mask() {
    long long v = g_q4_mask.load(std::memory_order_relaxed);
    if (v < 0) {
        v = getenv("HIPTS_GEMM_Q4") ? (long long)(unsigned)strtoul(getenv("HIPTS_GEMM_Q4"), nullptr, 0) : (long long)GEMM_Q4_DEFAULT_MASK;
        g_q4_mask.store(v, std::memory_order_relaxed);
    }
    return (unsigned)v;
}

int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("HIPTS_GEMM");
        v = (e && strcmp(e, "v1") == 0) ? 0 : (e && strcmp(e, "pp2") == 0) ? 3 : (e && strcmp(e, "s3") == 0) ? 2 : (e && strcmp(e, "dw") == 0) ? 4 : 1;
    }
    return v;
}

template <int EPI>
int launch_t(const GemmArgs& a, hipStream_t s) {
    int dev = 0;
    const int cus_dev = current_device_cus(&dev);
    static PerDevice attr;
    {
      std::lock_guard<std::mutex> lk(attr.mu);
      if (!attr.done(dev)) {
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp2_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_s3_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_dw_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS));
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_dw_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS));
        attr.mark(dev);
      }
    }
    const int tiles_m = (a.M + BM - 1) / BM;
    if (a.op8) {
        // e4m3 operands: the persistent ping-pong loop with full tiles only
        if constexpr (EPI == EPI_STAR || EPI == EPI_RESID || EPI == EPI_RESCALE || EPI == EPI_RESID_LN || EPI == EPI_QK || EPI == EPI_VT ||
                      EPI == EPI_BIAS) {
            static PerDevice attr8;
            {
                std::lock_guard<std::mutex> lk(attr8.mu);
                if (!attr8.done(dev)) {
                    HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                    attr8.mark(dev);
                }
            }
            const int cus8 = cus_dev;
            const int tiles_n = (a.N + BN - 1) / BN;
            const int ntile = tiles_m * tiles_n;
            const int slots = cus8 >= 8 ? cus8 / 8 * 8 : cus8;
            gemm_pp_kernel<EPI, 8, true, true><<<ntile > slots ? slots : ntile, 512, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
            HIPTS_LAUNCH_CHECK();
            return HIPTS_OK;
        } else {
            return set_error(HIPTS_ERR_INVALID, "gemm: e4m3 operands are not built for epilogue %d", (int)EPI);
        }
    }
    int variant = gemm_variant();
    if (EPI == EPI_QK_ROPE || EPI == EPI_SWIGLU || EPI == EPI_RESID_XG || EPI == EPI_RESID_XGI) variant = 1;      // staged epilogue only (pp, or dw below)
    if (EPI == EPI_RESID_LN) variant = 1;      // the row reduction across waves uses the persistent loop's LDS scratch stage
    if (variant == 1 && EPI != EPI_HEAD && EPI != EPI_RESID_LN && (!getenv("HIPTS_GEMM") || EPI == EPI_QK_ROPE || EPI == EPI_SWIGLU || EPI == EPI_RESID_XG || EPI == EPI_RESID_XGI)) {
        // A launch with fewer 256 x 256 tiles than CUs (the CAFormer's late stages: 11 520 tokens x 512 columns
        // = 90 tiles) leaves most of the chip idle; the 256 x 128 two-per-CU kernel has 2 x the tiles and
        // 2 x the slots (measured, CCIP B36 @384 batch 20: 9.3 -> 8.8 ms; no difference at batch 64).
        const int cus0 = cus_dev;
        static const bool auto_dw = !(getenv("HIPTS_GEMM_AUTO_DW") && atoi(getenv("HIPTS_GEMM_AUTO_DW")) == 0);      // A/B
        // (round 3: only below 3/4 of the CUs -- EVA02-L's q|k|v at batch 10 is 252 tiles on 256 CUs and runs 1 % faster on the persistent kernel.
        // Late round 4: only up to HALF the CUs, where every 256 x 128 tile gets a CU of its own; between a half and the whole chip the
        // persistent kernel with 192-row tiles -- one round of 3/4 the length -- is faster: CCIP batch 64, whose stage-2 launches are 144
        // tiles, 3244 -> 3412 images/s, batch 20 (90 tiles) stays on this kernel: 2586 against 2553.  HIPTS_GEMM_DW_LIMIT = n/4 of the CUs.)
        static const int dw_limit4 = getenv("HIPTS_GEMM_DW_LIMIT") ? atoi(getenv("HIPTS_GEMM_DW_LIMIT")) : 2;
        if (auto_dw && (long)tiles_m * ((a.N + BN - 1) / BN) * 4 <= (long)cus0 * dw_limit4 && a.M > BM && !((EPI == EPI_RESID_XG || EPI == EPI_RESID_XGI || EPI == EPI_SWIGLU) && a.stat_part)) variant = 4;
    }
    {   // A/B: HIPTS_GEMM_DW_MASK = bit mask over epilogue numbers whose launches take the two-workgroups-per-CU 256 x 128 kernel (its
        // residents run out of phase, so one's epilogue overlaps the other's main loop; it pays only where the epilogue is long and K short)
        static const unsigned dw_mask = getenv("HIPTS_GEMM_DW_MASK") ? (unsigned)strtoul(getenv("HIPTS_GEMM_DW_MASK"), nullptr, 0) : 0u;
#ifdef HIPTS_X_DW_STAT      // timing probe only (the statistics come out wrong): lets the masked epilogues take the dw kernel even with stat_part
        if (variant == 1 && ((dw_mask >> (int)EPI) & 1u) && a.M > BM) variant = 4;
#else
        if (variant == 1 && ((dw_mask >> (int)EPI) & 1u) && a.M > BM && !((EPI == EPI_RESID_XG || EPI == EPI_RESID_XGI) && a.stat_part)) variant = 4;
#endif
    }
    HIPTS_REQUIRE(!a.f16 || variant == 1 || variant == 4, "half-precision operands are only built for the pp and dw GEMM loops");
    if (variant == 4) {
        const int tiles_n = (a.N + DW_BN - 1) / DW_BN;
        if (a.f16) gemm_dw_kernel<EPI, true><<<tiles_m * tiles_n, 256, DW_LDS, s>>>(a, tiles_m, tiles_n);
        else gemm_dw_kernel<EPI, false><<<tiles_m * tiles_n, 256, DW_LDS, s>>>(a, tiles_m, tiles_n);
    } else if (variant == 2) {
        const int tiles_n = (a.N + S3_BN - 1) / S3_BN;
        gemm_s3_kernel<EPI><<<tiles_m * tiles_n, 256, S3_LDS, s>>>(a, tiles_m, tiles_n);
    } else {
        const int tiles_n = (a.N + BN - 1) / BN;
        if (variant == 3) {
            if constexpr (EPI == EPI_GELU) {
                if (a.stamps) {
                    HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp2_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                    gemm_pp2_kernel<EPI, true><<<tiles_m * tiles_n, 512, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
                    HIPTS_LAUNCH_CHECK();
                    return HIPTS_OK;
                }
            }
            gemm_pp2_kernel<EPI><<<tiles_m * tiles_n, 512, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
        }
        else if (variant == 1) {
            // 256-, 224- or 192-row tiles, whichever needs fewer (size-weighted) rounds over the CUs
            const int cus = cus_dev;
            static const int min_mr = !getenv("HIPTS_GEMM_BM") ? 6 : strcmp(getenv("HIPTS_GEMM_BM"), "256") == 0 ? 8 : strcmp(getenv("HIPTS_GEMM_BM"), "224") == 0 ? 7 : 6;
            auto tiles_of = [&](int mr) { return (a.M + 32 * mr - 1) / (32 * mr); };
            auto cost_of = [&](int mr) { return ((long)tiles_of(mr) * tiles_n + cus - 1) / cus * (32 * mr); };
            // measured (r01): the switch pays when the predicted saving is large (N = 768: 2.625 vs 3 rounds,
            // -4..6 %) and costs 3 % when it is marginal (N = 3072: 9.6 vs 10) -- smaller tiles re-read W more.
            // ... and only when the launch has the chip to itself: with sub-batches on several streams the
            // partial last round is filled by the other stream's kernel and full tiles win (4.50 -> 4.59 k img/s).
            // 192 rows (round 3, half operands only): EVA02-L at the reference's batch of 10 -- 10 250 rows x 1024 columns are 164
            // tiles of 256 rows on 256 CUs (one round, 64 % of the chip) but 216 tiles of 192 rows (one round of 3/4 the length).
            int mr = 8;
            // ... unless the whole launch is smaller than the chip (late round 4): then there is no last round for the other stream to fill,
            // and shorter tiles end the launch sooner (HIPTS_GEMM_MR_SHARED=0: as before, 1: by cost for every shared launch)
            static const int mr_shared = getenv("HIPTS_GEMM_MR_SHARED") ? atoi(getenv("HIPTS_GEMM_MR_SHARED")) : -1;
            if (!a.shared_chip || mr_shared == 1 || (mr_shared != 0 && (long)tiles_of(8) * tiles_n < cus)) {
                long best = cost_of(8) * 93;
                for (int c = 7; c >= (a.f16 ? min_mr : (min_mr > 7 ? min_mr : 7)); --c)
                    if (cost_of(c) * 100 < best) {
                        best = cost_of(c) * 100;
                        mr = c;
                    }
            }
            if (EPI == EPI_HEAD && a.sk_ws && getenv("HIPTS_GEMM_SPLITK_HEAD") && atoi(getenv("HIPTS_GEMM_SPLITK_HEAD")) >= 2) mr = 8;       // the split-K instantiation is built for 256-row tiles
            const int tiles_mr = tiles_of(mr);
            // persistent grid: one workgroup per CU (a multiple of 8 so that a workgroup's tiles keep their XCD)
            static const bool persist = !(getenv("HIPTS_GEMM_PERSIST") && strcmp(getenv("HIPTS_GEMM_PERSIST"), "0") == 0);
            const int ntile = tiles_mr * tiles_n;
            const int slots = cus >= 8 ? cus / 8 * 8 : cus;
            const int grid = (persist && ntile > slots) ? slots : ntile;
            static const int raster = getenv("HIPTS_GEMM_RASTER") ? atoi(getenv("HIPTS_GEMM_RASTER")) : 8;      // measured: 8 +0.4..0.8 % on the ViT forward, 4 / 16 +-0
            static const int raster_gn = getenv("HIPTS_GEMM_RASTER_GN") ? atoi(getenv("HIPTS_GEMM_RASTER_GN")) : 6;      // measured (r03): fc1 fetches 251 -> 207 MB, q|k|v 194 -> 167 MB per launch, images/s +-0; 0 = off
            GemmArgs ar = a;
            static const int epi_prio = getenv("HIPTS_EPI_PRIO") ? atoi(getenv("HIPTS_EPI_PRIO")) : 0;
            ar.epi_prio = epi_prio;
            static const int epi_prefetch = getenv("HIPTS_EPI_PREFETCH") ? atoi(getenv("HIPTS_EPI_PREFETCH")) : 0;
            ar.epi_prefetch = epi_prefetch;
            ar.raster_gm = (raster > 0 && tiles_n >= 8) ? raster : 0;
            ar.raster_gn = (raster_gn > 0 && tiles_n >= 8 && tiles_n > raster_gn) ? raster_gn : 0;
            {   // the 4-wave, one-wave-per-SIMD loop (gemm4.hip) where it is built: HIPTS_GEMM_Q4 = bit mask over epilogue numbers (A/B)
                const unsigned q4_mask = gemm_q4_mask();
                if (mr == 8 && ((q4_mask >> (int)EPI) & 1u)) {
                    bool handled = false;
                    const int st = launch_gemm_q4((GemmEpilogue)EPI, ar, s, &handled);
                    if (handled) return st;
                }
            }
            static const bool resid_general = getenv("HIPTS_RESID_GENERAL") && atoi(getenv("HIPTS_RESID_GENERAL")) != 0;      // A/B: the predicated residual epilogue on interior tiles too
            // Split-K tail (GemmArgs::sk_*), an EXPERIMENT that lost (round 4) and stays off: the residual GEMMs with a long K whose last round
            // fills less than half of the chip -- EVA02-L's proj / fc2 at the reference's batch of 10 (84 tiles per sub-batch on 256 CUs), the
            // ViT's fc2 per 32-image sub-batch (294 tiles: 38 in the second round) -- with S <= HIPTS_GEMM_SPLITK slices of at least
            // HIPTS_GEMM_SPLITK_MINKT K-tiles.  Measured (tools/gpurun/r4_splitk.sh, one box): ViT-B/16 5093-5103 -> 4894-4898 images/s
            // (S = 4; S = 2: 4982), EVA02-L batch 10 1044 -> 829-832 images/s (S = 3), batch 32 unchanged.  A 256 x 256 fp32 slab is 256 KB:
            // 84 tiles x 3 slices write 64 MB through to memory and their last arrivers read it back, ~50 us per launch, more than the
            // under-filled main loop costs (cdna_hip_programming.md says as much: combine in-launch only when the slabs of a tile are tens of
            // KB).  It also gives up batch invariance -- a tile summed as S partial chains has other low bits than the same tile summed as one
            // chain, and WHICH tiles are split depends on the launch's size (tests/test_gpu_vit.py::test_folded_layernorm_path..., the sharded
            // CLIs' byte-equal output files) -- though run to run it is deterministic (slabs are added in slice order).  The tag head's launch
            // (EPI_HEAD: one row panel of 43 column tiles whatever the batch) would keep the invariance; HIPTS_GEMM_SPLITK_HEAD=4 enables it.
            if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_XG || EPI == EPI_RESID_XGI || EPI == EPI_RESID_ROWSTAT || EPI == EPI_HEAD) {
                static const int sk_env = getenv("HIPTS_GEMM_SPLITK") ? atoi(getenv("HIPTS_GEMM_SPLITK")) : 0;
                static const int sk_head = getenv("HIPTS_GEMM_SPLITK_HEAD") ? atoi(getenv("HIPTS_GEMM_SPLITK_HEAD")) : 0;
                const int sk_max = EPI == EPI_HEAD ? (tiles_mr == 1 ? sk_head : 0) : sk_env;
                static const int sk_minkt = getenv("HIPTS_GEMM_SPLITK_MINKT") ? atoi(getenv("HIPTS_GEMM_SPLITK_MINKT")) : 5;
                const int nkt = a.K / BK;
                const int rem = ntile % slots;      // tiles of the partial last round (the whole launch when it is smaller than the chip)
                if (a.sk_ws && mr == 8 && sk_max >= 2 && rem > 0 && rem * 2 <= slots && rem <= 1024) {
                    int sl = std::min(sk_max, std::min(slots / rem, nkt / (sk_minkt > 0 ? sk_minkt : 1)));
                    while (sl >= 2 && 4096 + (size_t)rem * sl * ((size_t)BM * BN * 4) > a.sk_ws_bytes) --sl;
                    if (sl >= 2) {
                        static PerDevice attr_sk;
                        {
                            std::lock_guard<std::mutex> lk(attr_sk.mu);
                            if (!attr_sk.done(dev)) {
                                HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                                HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                                attr_sk.mark(dev);
                            }
                        }
                        ar.sk_first = ntile - rem;
                        ar.sk_slices = sl;
                        const int items = ar.sk_first + rem * sl;
                        const int grid_sk = (persist && items > slots) ? slots : items;
                        if (a.f16) gemm_pp_kernel<EPI, 8, true, false, true><<<grid_sk, 512, LDS_BYTES, s>>>(ar, tiles_m, tiles_n);
                        else gemm_pp_kernel<EPI, 8, false, false, true><<<grid_sk, 512, LDS_BYTES, s>>>(ar, tiles_m, tiles_n);
                        HIPTS_LAUNCH_CHECK();
                        return HIPTS_OK;
                    }
                }
            }
            if constexpr (EPI == EPI_RESID_XG) {
                // every tile inside the matrix: the instantiation without per-lane predication (HIPTS_RESID_GENERAL=1: the general one, A/B).
                // (Not instantiated for RESID_XGI: its only user, EVA02, has 1025 tokens per image -- no launch of whole tiles -- and with the
                // input fold's extra column vector the interior form compiled to 60 spilled registers.)
                if (mr == 8 && a.M % 256 == 0 && a.N % 256 == 0 && a.N <= 1024 && !a.pos && !a.res_scale && a.out_bf16 && a.stat_part && !resid_general) {
                    static PerDevice attr_int;
                    {
                        std::lock_guard<std::mutex> lk(attr_int.mu);
                        if (!attr_int.done(dev)) {
                            HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                            HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, 8, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
                            attr_int.mark(dev);
                        }
                    }
                    if (a.f16) gemm_pp_kernel<EPI, 8, true, false, false, true><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_m, tiles_n);
                    else gemm_pp_kernel<EPI, 8, false, false, false, true><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_m, tiles_n);
                    HIPTS_LAUNCH_CHECK();
                    return HIPTS_OK;
                }
            }
            if (a.f16) {
                if (mr == 6) gemm_pp_kernel<EPI, 6, true><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_mr, tiles_n);
                else if (mr == 7) gemm_pp_kernel<EPI, 7, true><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_mr, tiles_n);
                else gemm_pp_kernel<EPI, 8, true><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_m, tiles_n);
            } else {
                if (mr == 7) gemm_pp_kernel<EPI, 7, false><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_mr, tiles_n);
                else gemm_pp_kernel<EPI, 8, false><<<grid, 512, LDS_BYTES, s>>>(ar, tiles_m, tiles_n);
            }
        }
        else
            gemm_kernel<EPI><<<tiles_m * tiles_n, 512, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
    }
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace

int launch_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
    if (a.op8) HIPTS_REQUIRE(a.K % 128 == 0 && a.K >= 128, "gemm: K=%d must be a positive multiple of 128 with e4m3 operands", a.K);
    if (a.out8)
        HIPTS_REQUIRE((epi == EPI_STAR || epi == EPI_RESID_LN) && (a.ld_out ? a.ld_out : a.N) % 16 == 0,
                      "gemm: e4m3 output needs the STAR / RESID_LN epilogue and a row stride that is a multiple of 16");
    HIPTS_REQUIRE(a.K % BK == 0 && a.K >= BK, "gemm: K=%d must be a positive multiple of %d", a.K, BK);
    HIPTS_REQUIRE(a.M >= 1 && a.N >= 1, "gemm: empty problem");
    if (epi != EPI_HEAD) HIPTS_REQUIRE(a.N % 16 == 0, "gemm: N=%d must be a multiple of 16", a.N);
    if (epi == EPI_VT) HIPTS_REQUIRE(a.M % 4 == 0 && a.tokens % 4 == 0, "gemm: V^T epilogue needs tokens %% 4 == 0");
    if (epi == EPI_QK || epi == EPI_VT) HIPTS_REQUIRE(a.hd_log2 == 5 || a.hd_log2 == 6, "gemm: head_dim must be 32 or 64");
    if (epi == EPI_QK || epi == EPI_QK_ROPE)
        HIPTS_REQUIRE(a.out_bf16 && a.out2_bf16 && a.dim > 0 && a.N <= 3 * a.dim && (a.N <= 2 * a.dim || a.out3_bf16), "gemm: q / k / v outputs missing for N = %d, dim = %d", a.N, a.dim);
    if (epi == EPI_QK_ROPE)
        HIPTS_REQUIRE(a.hd_log2 == 6 && a.dim % 64 == 0 && a.rope && a.rope_tokens >= 0 && a.tokens >= 1, "gemm: QK_ROPE needs head_dim 64 and the rotary table");
    if (epi == EPI_RESID_XG) HIPTS_REQUIRE(!a.rowstat && !a.stat_in, "gemm: RESID_XG ignores an input fold; use RESID_XGI");
    if (epi == EPI_RESID_XGI) HIPTS_REQUIRE((a.rowstat || a.stat_in) && a.col_u, "gemm: RESID_XGI needs rowstat / stat_in and col_u");
    if (epi == EPI_RESID_XG || epi == EPI_RESID_XGI)
        HIPTS_REQUIRE(a.out_f32 && (!a.out_bf16 || (a.ln_gamma && a.N % 8 == 0 && (!a.stat_part || a.stat_stride >= a.M))),
                      "gemm: RESID_XG needs the fp32 stream, col_u with rowstat, and gamma (N %% 8 == 0) with the 16-bit copy");
    if (a.stat_in) HIPTS_REQUIRE(a.stat_in_blocks <= 64 && a.stat_in_blocks >= 1 && a.stat_in_stride >= a.M && a.ln_dim >= 1, "gemm: stat_in needs its block count (<= 64), stride and ln_dim");
    if ((a.rowstat || a.stat_in) && epi != EPI_RESID_ROWSTAT && epi != EPI_RESID_XGI)
        HIPTS_REQUIRE(a.col_u && (epi == EPI_QK || epi == EPI_QK_ROPE || epi == EPI_VT || epi == EPI_GELU || epi == EPI_SWIGLU || epi == EPI_STAR),
                      "gemm: a folded LayerNorm (rowstat) is built for the QK, QK_ROPE, VT, GELU, STAR and SWIGLU epilogues");
    if ((a.rowstat || a.stat_in) && epi != EPI_RESID_ROWSTAT && epi != EPI_RESID_XGI) {
        // the fold lives in the LDS-staged epilogues only
        const int ldo = a.ld_out ? a.ld_out : (epi == EPI_SWIGLU ? a.N / 2 : a.N);
        const bool staged = (epi == EPI_QK || epi == EPI_QK_ROPE) ? a.dim % 64 == 0
                            : epi == EPI_VT                       ? (a.tokens % 8 == 0 && a.tokens_pad % 8 == 0)
                                                                  : ldo % 8 == 0;
        HIPTS_REQUIRE(staged && (gemm_variant() == 1 || gemm_variant() == 4), "gemm: a folded LayerNorm needs the staged epilogue (pp / dw loops, aligned outputs)");
    }
    if (epi == EPI_RESID_ROWSTAT) HIPTS_REQUIRE((a.rowstat || a.stat_in) && a.col_u && a.out_f32, "gemm: RESID_ROWSTAT needs rowstat / stat_in, col_u and the fp32 stream");
    if (epi == EPI_SWIGLU) HIPTS_REQUIRE(!a.stat_part || a.stat_stride >= a.M, "gemm: SWIGLU stat_stride must cover M rows");
    if (epi == EPI_SWIGLU)
        HIPTS_REQUIRE(a.N % 64 == 0 && (a.ld_out ? a.ld_out : a.N / 2) % 8 == 0 && a.out_bf16, "gemm: SWIGLU needs N %% 64 == 0 and an output stride that is a multiple of 8");
    if (epi == EPI_RESCALE) HIPTS_REQUIRE(a.res_scale != nullptr, "gemm: RESCALE epilogue needs res_scale");
    if (epi == EPI_RESID_LN)
        HIPTS_REQUIRE(a.N <= BN && a.ln_gamma && a.out_bf16 && a.out_f32, "gemm: RESID_LN needs the whole row in one tile (N <= %d), gamma and both outputs", BN);
    switch (epi) {
        case EPI_PATCH: return launch_t<EPI_PATCH>(a, s);
        case EPI_QK: return launch_t<EPI_QK>(a, s);
        case EPI_VT: return launch_t<EPI_VT>(a, s);
        case EPI_RESID: return launch_t<EPI_RESID>(a, s);
        case EPI_GELU: return launch_t<EPI_GELU>(a, s);
        case EPI_HEAD: return launch_t<EPI_HEAD>(a, s);
        case EPI_STAR: return launch_t<EPI_STAR>(a, s);
        case EPI_RESCALE: return launch_t<EPI_RESCALE>(a, s);
        case EPI_BIAS: return launch_t<EPI_BIAS>(a, s);
        case EPI_RESID_LN: return launch_t<EPI_RESID_LN>(a, s);
        case EPI_QK_ROPE: return launch_t<EPI_QK_ROPE>(a, s);
        case EPI_SWIGLU: return launch_t<EPI_SWIGLU>(a, s);
        case EPI_RESID_ROWSTAT: return launch_t<EPI_RESID_ROWSTAT>(a, s);
        case EPI_RESID_XG: return launch_t<EPI_RESID_XG>(a, s);
        case EPI_RESID_XGI: return launch_t<EPI_RESID_XGI>(a, s);
    }
    return set_error(HIPTS_ERR_INVALID, "gemm: unknown epilogue");
}

void set_gemm_q4_mask(unsigned mask) { g_q4_mask.store((long long)mask, std::memory_order_relaxed); }

namespace {
__global__ void gelu_probe_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int n4, int tanh_form) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) y[i] = gelu_f4(x[i], tanh_form);
}
}  // namespace

}  // namespace hipts

// Development/test aid (not part of the public ABI): the GELU of the fc1 epilogue (gelu_f4) on n host floats, n % 4 == 0.
extern "C" int hiptsdbg_gelu(const float* x_host, int n, int tanh_form, float* y_host) {
    using namespace hipts;
    HIPTS_TRY(use_device(0));
    HIPTS_REQUIRE(x_host && y_host && n > 0 && n % 4 == 0, "hiptsdbg_gelu: n must be a positive multiple of 4");
    DevBuf x, y;
    HIPTS_TRY(x.alloc((size_t)n * 4));
    HIPTS_TRY(y.alloc((size_t)n * 4));
    HIPTS_TRY(upload(x.p, x_host, (size_t)n * 4));
    gelu_probe_kernel<<<(n / 4 + 255) / 256, 256>>>(x.as<f32x4>(), y.as<f32x4>(), n / 4, tanh_form);
    HIPTS_LAUNCH_CHECK();
    HIPTS_HIP(hipMemcpy(y_host, y.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}
