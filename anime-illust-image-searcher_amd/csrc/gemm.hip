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
//   subtiles of 16 rows x 32 k (1 KiB = one wave instruction); inside a subtile the 16-byte chunk
//   index is XORed with 2 for rows 8..15 (applied on the per-lane global SOURCE address and again
//   on the ds_read_b128 address), which makes the fragment reads bank-conflict free;
//   the next K-step is staged while the current one is multiplied (2-stage pipeline).
// Operands are swapped in the MFMA (W fragment as "A", activation fragment as "B") so that a lane
// ends up with 4 consecutive output COLUMNS of one row: epilogue loads/stores are 16 B (fp32) or
// 8 B (bf16) per lane.  EPI_VT uses the natural order to get 4 consecutive ROWS (tokens) per lane,
// which is what the transposed V layout wants.
// Workgroup ids are remapped so that the workgroups sharing an XCD (ids equal mod 8) walk
// neighbouring tiles and reuse operand panels in that XCD's L2.
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

// Stage one 256 x 64 operand tile: 32 subtiles, 4 per wave.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ X, int rows_total, int K, int row0, int kt,
                                           char* lds_tile, int wave, int lane) {
    const int row_in = lane >> 2;
    const int chunk = (lane & 3) ^ (((row_in >> 3) & 1) << 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = wave * 4 + i;
        const int rowblk = s >> 1, kblk = s & 1;
        int grow = row0 + rowblk * 16 + row_in;
        grow = grow < rows_total ? grow : rows_total - 1;
        const bf16_t* g = X + (size_t)grow * K + (size_t)kt * BK + kblk * 32 + chunk * 8;
        glds16(g, lds_tile + s * 1024);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rowblk, int kk, int lane) {
    const int r = lane & 15;
    const int c = (lane >> 4) ^ (((r >> 3) & 1) << 1);
    return *reinterpret_cast<const bf16x8*>(lds_tile + (rowblk * 2 + kk) * 1024 + r * 64 + c * 16);
}

__device__ __forceinline__ float gelu_f(float x, int tanh_form) {
    if (tanh_form) {
        // torch gelu(approximate='tanh'): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
        const float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
        const float inner = kBeta * (x + kKappa * x * x * x);
        return 0.5f * x * (1.0f + tanhf(inner));
    }
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}

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

    // ------------------------------------------------------------------ epilogue
    const int lr = lane & 15, lq = lane >> 4;
    const int ld = a.ld_out ? a.ld_out : a.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 c = acc[i][j];
            if constexpr (EPI == EPI_VT) {
                // natural order: lane = column n, registers = 4 consecutive rows (tokens)
                const int n = n0 + wave_n * 64 + j * 16 + lr;
                const int m = m0 + wave_m * 128 + i * 16 + 4 * lq;
                if (m < a.M && n < a.N) {
                    const float bv = a.bias[n];
                    const int b = m / a.tokens, t = m - b * a.tokens;
                    const int head = n >> 6, d = n & 63;
                    bf16x4 o;
                    o[0] = (bf16_t)(c[0] + bv);
                    o[1] = (bf16_t)(c[1] + bv);
                    o[2] = (bf16_t)(c[2] + bv);
                    o[3] = (bf16_t)(c[3] + bv);
                    bf16_t* dst = a.out_bf16 + ((size_t)(b * a.heads + head) * 64 + d) * a.tokens_pad + t;
                    *reinterpret_cast<bf16x4*>(dst) = o;
                }
            } else {
                // swapped order: lane = row m, registers = 4 consecutive columns n
                const int m = m0 + wave_m * 128 + i * 16 + lr;
                const int n = n0 + wave_n * 64 + j * 16 + 4 * lq;
                if (m >= a.M || n >= a.N) continue;
                if constexpr (EPI == EPI_HEAD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < a.N) {
                            const float v = c[e] + a.bias[n + e];
                            if (a.out_f32) a.out_f32[(size_t)m * ld + n + e] = v;
                            if (a.out2_f32) a.out2_f32[(size_t)m * ld + n + e] = 1.0f / (1.0f + expf(-v));
                        }
                    }
                } else {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
                    f32x4 v = c + bv;
                    if constexpr (EPI == EPI_PATCH) {
                        v = c * a.qscale + bv;
                        const int t = m % a.tokens;
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(a.pos + (size_t)t * a.N + n);
                        *reinterpret_cast<f32x4*>(a.out_f32 + (size_t)m * ld + n) = v + pv;
                    } else if constexpr (EPI == EPI_RESID) {
                        float* p = a.out_f32 + (size_t)m * ld + n;
                        const f32x4 x = *reinterpret_cast<const f32x4*>(p);
                        *reinterpret_cast<f32x4*>(p) = x + v;
                    } else if constexpr (EPI == EPI_GELU) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)gelu_f(v[e], a.gelu_tanh);
                        *reinterpret_cast<bf16x4*>(a.out_bf16 + (size_t)m * ld + n) = o;
                    } else if constexpr (EPI == EPI_QK) {
                        const int which = n >= a.dim ? 1 : 0;
                        const int nn = n - which * a.dim;
                        const int head = nn >> 6, d = nn & 63;
                        const int b = m / a.tokens, t = m - b * a.tokens;
                        const float sc = which ? 1.0f : a.qscale;
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(v[e] * sc);
                        bf16_t* base = which ? a.out2_bf16 : a.out_bf16;
                        *reinterpret_cast<bf16x4*>(base + ((size_t)(b * a.heads + head) * a.tokens_pad + t) * 64 + d) = o;
                    }
                }
            }
        }
    }
}

template <int EPI>
int launch_t(const GemmArgs& a, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        HIPTS_HIP(hipFuncSetAttribute((const void*)gemm_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    gemm_kernel<EPI><<<tiles_m * tiles_n, 512, LDS_BYTES, s>>>(a, tiles_m, tiles_n);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace

int launch_gemm(GemmEpilogue epi, const GemmArgs& a, hipStream_t s) {
    HIPTS_REQUIRE(a.K % BK == 0 && a.K >= BK, "gemm: K=%d must be a positive multiple of %d", a.K, BK);
    HIPTS_REQUIRE(a.M >= 1 && a.N >= 1, "gemm: empty problem");
    if (epi != EPI_HEAD) HIPTS_REQUIRE(a.N % 16 == 0, "gemm: N=%d must be a multiple of 16", a.N);
    if (epi == EPI_VT) HIPTS_REQUIRE(a.M % 4 == 0 && a.tokens % 4 == 0, "gemm: V^T epilogue needs tokens %% 4 == 0");
    switch (epi) {
        case EPI_PATCH: return launch_t<EPI_PATCH>(a, s);
        case EPI_QK: return launch_t<EPI_QK>(a, s);
        case EPI_VT: return launch_t<EPI_VT>(a, s);
        case EPI_RESID: return launch_t<EPI_RESID>(a, s);
        case EPI_GELU: return launch_t<EPI_GELU>(a, s);
        case EPI_HEAD: return launch_t<EPI_HEAD>(a, s);
    }
    return set_error(HIPTS_ERR_INVALID, "gemm: unknown epilogue");
}

}  // namespace hipts
