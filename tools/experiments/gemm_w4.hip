// gemm_w4.hip -- EXPERIMENT (not on any product path): a 256 x 256 x 64 GEMM main loop with FOUR waves per workgroup,
// each owning a 128 x 128 block of the tile (256 accumulator registers in the AGPR half of the unified file, one wave per
// SIMD), instead of the eight 128 x 64 waves of gemm_pp_kernel.  Per MAC it reads a third fewer LDS bytes and operand
// registers -- the lever a power-limited main loop has left ("Where a GEMM launch's time goes").  With one
// wave per SIMD there is no partner wave to hide LDS latency: the fragments of the next k-half are register
// double-buffered and their ds_reads, like the next K-tile's global_load_lds, are interleaved with the MFMAs by
// sched_group_barrier.  Built WITHOUT -amdgpu-mfma-vgpr-form (the accumulators must live in AGPRs).
//
// RESULT (round 1): correct (max error 2.6e-6 against a float64 product) but 418-585 TFLOP/s where gemm_pp_kernel reaches
// 680-1000 on the same shapes (50176x768x768 ... 8192^3).  The compiled loop is full of v_accvgpr_read/write and has 112 B
// of scratch: with 256 accumulators in AGPRs, 128 fragment registers, 16 64-bit source pointers and the address math the
// allocator shuffles fragments through the AGPR half inside the loop.  Next steps if this is picked up again: 32-bit
// offsets from two base pointers instead of 16 pointers, single-buffered W fragments, and checking the allocation with
// -save-temps after every edit.  Not part of the library build: to try it, copy it into csrc/ and give gemm_w4.o the
// CXXFLAGS without `-mllvm -amdgpu-mfma-vgpr-form`.
//
// Entry: hiptsdbg_gemm_w4(M, N, K, a_bits, w_bits, out, iters, &ms): out[m][n] = sum_k A[m][k] W[n][k] (bf16 operands,
// fp32 out), timing over `iters` launches; M, N multiples of 256 and K of 64 only (interior tiles, no masking).
#include <type_traits>

#include "vit_internal.h"

namespace hipts {
namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // 128 KiB

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// same LDS image as gemm.hip: 1 KB sub-tiles of 8 rows x 128 B, 16 B chunk XOR-swizzled with the row
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rowblk, int kk, int lane) {
    const int r = lane & 15;
    const int c = (kk * 4 + (lane >> 4)) ^ (r & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + (rowblk * 2 + (r >> 3)) * 1024 + (r & 7) * 128 + c * 16);
}

__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, float* __restrict__ out,
                                                        int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int nt = K / BK;

    // staging: a K-tile is 64 sub-tiles (32 of A, 32 of W); wave w stages sub-tiles w, w + 4, ...: 16 loads per wave
    const int row_in = lane >> 3, chunk = (lane & 7) ^ row_in;
    const bf16_t* src[16];
    int dst[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int e = wave + 4 * u;              // 0..63
        const bool isW = e >= 32;
        const int rb8 = e & 31;
        const int grow = (isW ? n0 : m0) + rb8 * 8 + row_in;
        src[u] = (isW ? W : A) + (size_t)grow * K + chunk * 8;
        dst[u] = (isW ? TILE_BYTES : 0) + rb8 * 1024;
    }
    auto issue = [&](int u, char* stage) {
        glds16(src[u], stage + dst[u]);
        src[u] += BK;
    };

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: K-tile 0 complete, fragments of its first k-half
#pragma unroll
    for (int u = 0; u < 16; ++u) issue(u, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 fa[2][8], fw[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[0][i] = read_frag(smem, wm * 8 + i, 0, lane);
        fw[0][i] = read_frag(smem + TILE_BYTES, wn * 8 + i, 0, lane);
    }

    // one K-tile; MORE = a next K-tile exists (compile-time, so that each block below is one basic block for the scheduler)
    auto k_tile = [&](int t, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        const char* cur = smem + (t & 1) * STAGE_BYTES;
        char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
        // ---- block A: 64 MFMAs on k-half 0; the 16 fragment reads of k-half 1 and the 16 loads of K-tile t+1 ride along
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int i = g >> 1, jb = (g & 1) * 4;
            if (g < 8) fa[1][g] = read_frag(cur, wm * 8 + g, 1, lane);
            else fw[1][g - 8] = read_frag(cur + TILE_BYTES, wn * 8 + (g - 8), 1, lane);
            if constexpr (MORE) issue(g, nxt);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][jb + j], fa[0][i], acc[i][jb + j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 ds_read
            if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);       // 1 vmem read (the LDS-DMA load)
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);        // 4 MFMA
        }
        // ---- block B, first half: 32 MFMAs on k-half 1
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][j], fa[1][i], acc[i][j], 0, 0, 0);
        // K-tile t+1 landed (its loads had 96 MFMAs = ~1500 cycles); every wave is past its reads of K-tile t
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- block B, second half: 32 MFMAs; the 16 fragment reads of K-tile t+1's k-half 0 ride along
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int i = 4 + (g >> 1), jb = (g & 1) * 4;
            if constexpr (MORE) {
                fa[0][g] = read_frag(nxt, wm * 8 + g, 0, lane);
                fw[0][g] = read_frag(nxt + TILE_BYTES, wn * 8 + g, 0, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][jb + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][jb + j], fa[1][i], acc[i][jb + j], 0, 0, 0);
            if constexpr (MORE) __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            __builtin_amdgcn_sched_group_barrier(0x8, 4, 1);
        }
    };
    for (int t = 0; t + 1 < nt; ++t) k_tile(t, std::true_type{});
    k_tile(nt - 1, std::false_type{});

    // epilogue (experiment): plain fp32 store; acc[i][j][e] = (row 16 i + lane & 15, column 16 j + 4 (lane >> 4) + e) (operands swapped)
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wm * 128 + i * 16 + lr;
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(out + (size_t)m * N + n0 + wn * 128 + j * 16 + 4 * lq) = acc[i][j];
    }
}

}  // namespace
}  // namespace hipts

using namespace hipts;

extern "C" int hiptsdbg_gemm_w4(int M, int N, int K, const uint16_t* a_bits, const uint16_t* w_bits, float* out_host, int iters, float* ms_out) {
    HIPTS_TRY(use_device(0));
    HIPTS_REQUIRE(M >= 256 && M % 256 == 0 && N >= 256 && N % 256 == 0 && K >= 64 && K % 64 == 0 && a_bits && w_bits && ms_out, "hiptsdbg_gemm_w4: M, N %% 256, K %% 64");
    DevBuf A, W, out;
    HIPTS_TRY(A.alloc((size_t)M * K * 2));
    HIPTS_TRY(W.alloc((size_t)N * K * 2));
    HIPTS_TRY(out.alloc((size_t)M * N * 4));
    HIPTS_TRY(upload(A.p, a_bits, (size_t)M * K * 2));
    HIPTS_TRY(upload(W.p, w_bits, (size_t)N * K * 2));
    static bool attr = false;
    if (!attr) {
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr = true;
    }
    const int tiles_n = N / 256, grid = (M / 256) * tiles_n;
    hipEvent_t e0, e1;
    HIPTS_HIP(hipEventCreate(&e0));
    HIPTS_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) gemm_w4_kernel<<<grid, 256, LDS_BYTES, nullptr>>>(A.as<bf16_t>(), W.as<bf16_t>(), out.as<float>(), M, N, K, tiles_n);
    HIPTS_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) gemm_w4_kernel<<<grid, 256, LDS_BYTES, nullptr>>>(A.as<bf16_t>(), W.as<bf16_t>(), out.as<float>(), M, N, K, tiles_n);
    HIPTS_HIP(hipEventRecord(e1, nullptr));
    HIPTS_HIP(hipEventSynchronize(e1));
    HIPTS_LAUNCH_CHECK();
    float ms = 0.f;
    HIPTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (iters > 0 ? iters : 1);
    if (out_host) HIPTS_HIP(hipMemcpy(out_host, out.p, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return HIPTS_OK;
}
