// gemm_epi.h -- tile constants, LDS staging helpers and the fused epilogues shared by the GEMM main loops (gemm.hip: the 8-wave
// ping-pong loops; gemm4.hip: the 4-wave, one-wave-per-SIMD loop).  Included INSIDE `namespace hipts { namespace {` of each file:
// every function is a forceinline device function of that translation unit.  An epilogue sees a wave-part of 16 MR rows x 64
// columns (acc[MR][4] of 16 x 16 tiles) with its (wave_m, wave_n) coordinates in the 256 x 256 tile, whoever computed it.

constexpr int BM = 256, BN = 256, BK = 64;
#ifndef HIPTS_STAGE_AHEAD
#define HIPTS_STAGE_AHEAD 1           // regions are restaged as soon as they are free, up to two K-tiles ahead (see the loop): one more phase
                                      // for every load.  Measured against 0 (one K-tile ahead): 8192^3 952 -> 1074 TFLOP/s (operands from HBM), the
                                      // isolated K = 768 shapes 2-4 % slower, but the forwards faster: ViT +2.2 % (4841 -> 4946 images/s, ABA
                                      // runs on one box), EVA02-L +3 % at batch 32, CCIP +1.5 %.  (Moving W-high into the phase-3 group as well
                                      // -- six loads in one phase, four phases for everything -- measured 1.5 % slower.)
#endif
#ifndef HIPTS_STAGE_W_EARLY
#define HIPTS_STAGE_W_EARLY 0         // 1: W-high in phase 0 as well (3 phases to land instead of 2): 8192^3 943 -> 1074 TFLOP/s but K = 3072 / 4096 shapes 5 % slower
#endif
#ifndef HIPTS_STAGE_ORDER_OLD
#define HIPTS_STAGE_ORDER_OLD 0       // 1: the previous staging order of the ping-pong loop (A/B builds)
#endif
#ifndef HIPTS_STAGED_INTERIOR
#define HIPTS_STAGED_INTERIOR 1         // 0: the staged 16-bit epilogues store under per-lane predicates everywhere (A/B builds)
#endif
#ifndef HIPTS_INT_DEPTH
#define HIPTS_INT_DEPTH 2
#endif
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // 128 KiB

__device__ __forceinline__ void glds16(const void* g, void* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// Stage one 256 x 64 operand tile: 32 sub-tiles of 8 rows, 4 per wave.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ X, int rows_total, int K, int row0, int kt,
                                           char* lds_tile, int wave, int lane) {
    const int row_in = lane >> 3;
    const int chunk = (lane & 7) ^ row_in;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rb8 = wave * 4 + i;
        int grow = row0 + rb8 * 8 + row_in;
        grow = grow < rows_total ? grow : rows_total - 1;
        const bf16_t* g = X + (size_t)grow * K + (size_t)kt * BK + chunk * 8;
        glds16(g, lds_tile + rb8 * 1024);
    }
}

// MFMA 16x16x32 fragment of 16-row block `rowblk`, k-half kk: lane (r = lane & 15, q = lane >> 4)
// takes row r, logical chunk 4 kk + q, stored at physical chunk (4 kk + q) ^ (r & 7).
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rowblk, int kk, int lane) {
    const int r = lane & 15;
    const int c = (kk * 4 + (lane >> 4)) ^ (r & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + (rowblk * 2 + (r >> 3)) * 1024 + (r & 7) * 128 + c * 16);
}

// e4m3 operands: a K-tile is 128 elements = the same 128 B row, and the 16x16x128 fragment of lane (r, q) is bytes
// 32 q .. 32 q + 31 of row r: logical chunks 2 q and 2 q + 1.
__device__ __forceinline__ i32x8 read_frag8(const char* lds_tile, int rowblk, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const char* row = lds_tile + (rowblk * 2 + (r >> 3)) * 1024 + (r & 7) * 128;
    const int4 lo = *reinterpret_cast<const int4*>(row + (((2 * q) ^ (r & 7)) * 16));
    const int4 hi = *reinterpret_cast<const int4*>(row + (((2 * q + 1) ^ (r & 7)) * 16));
    return i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}

// GELU of four values at once, in packed fp32 (v_pk_mul/fma/add_f32 process two floats per lane and issue slot -- the GELU epilogue
// is VALU-bound: 128 values per lane, two waves per SIMD).  Every epilogue form calls this one function, so a value does not depend on
// the path its tile took.
__device__ __forceinline__ f32x4 gelu_f4(f32x4 x, int tanh_form) {
    if (tanh_form) {
        // torch gelu(approximate='tanh'): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).
        // 0.5 (1 + tanh(u)) = sigmoid(2u) = 1 / (1 + 2^(-2 u log2 e)): one v_exp_f32 + one v_rcp_f32 per value.
        // exponent of 2: -2 log2(e) sqrt(2/pi) (x + 0.044715 x^3) = x (c1 + c2 x^2), explicit fma (the file is built with -ffp-contract=off)
        const float c1 = -2.885390081777927f * 0.7978845608028654f, c2 = c1 * 0.044715f;
        const f32x4 t = x * x;
        const f32x4 g = x * __builtin_elementwise_fma(f32x4{c2, c2, c2, c2}, t, f32x4{c1, c1, c1, c1});
        f32x4 e{__builtin_amdgcn_exp2f(g[0]), __builtin_amdgcn_exp2f(g[1]), __builtin_amdgcn_exp2f(g[2]), __builtin_amdgcn_exp2f(g[3])};
        e = e + f32x4{1.f, 1.f, 1.f, 1.f};
        return x * f32x4{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1]), __builtin_amdgcn_rcpf(e[2]), __builtin_amdgcn_rcpf(e[3])};
    }
    // erf form, x Phi(x) with Phi(x) - 1/2 = xc P(xc^2) / Q(xc^2), xc = x clamped to +-5.5 (Phi(-5.5) = 1.9e-8), P and Q of degree 5: a
    // weighted minimax fit (tools/fit_gelu_erf.py) whose fp32 evaluation is within 2.5e-7 max(1, |x|) of the exact value -- the
    // fp32 expression 0.5 x (1 + erff(x / sqrt 2)) is within 1.1e-7 max(1, |x|), and what is stored is rounded to 16 bits (4.9e-4 relative).
    // Everything but the clamp and the reciprocal is packed fp32: about 12 issue slots a value, like the tanh form, where libm's
    // branchy erff (rounds 1-3) took about 55 -- more than the tile's main loop.
    const f32x4 xc{__builtin_amdgcn_fmed3f(x[0], -5.5f, 5.5f), __builtin_amdgcn_fmed3f(x[1], -5.5f, 5.5f), __builtin_amdgcn_fmed3f(x[2], -5.5f, 5.5f),
                   __builtin_amdgcn_fmed3f(x[3], -5.5f, 5.5f)};
    const f32x4 t = xc * xc;
    auto k4 = [](float c) { return f32x4{c, c, c, c}; };
    f32x4 p = __builtin_elementwise_fma(k4(2.2240455115528255e-08f), t, k4(6.478198381570408e-06f));
    p = __builtin_elementwise_fma(p, t, k4(0.00017989516452868775f));
    p = __builtin_elementwise_fma(p, t, k4(0.00474442647621276f));
    p = __builtin_elementwise_fma(p, t, k4(0.03488362160853281f));
    p = __builtin_elementwise_fma(p, t, k4(0.39894214428114516f));
    f32x4 q = __builtin_elementwise_fma(k4(1.1920686967418627e-06f), t, k4(7.74708620425572e-05f));
    q = __builtin_elementwise_fma(q, t, k4(0.0019464965294343475f));
    q = __builtin_elementwise_fma(q, t, k4(0.02924650260921815f));
    q = __builtin_elementwise_fma(q, t, k4(0.25410501090673415f));
    q = __builtin_elementwise_fma(q, t, k4(1.0f));
    const f32x4 rq{__builtin_amdgcn_rcpf(q[0]), __builtin_amdgcn_rcpf(q[1]), __builtin_amdgcn_rcpf(q[2]), __builtin_amdgcn_rcpf(q[3])};
    const f32x4 r = (xc * p) * rq;
    return x * (r + k4(0.5f));
}

// (rstd, rstd * mean) of row m for a folded LayerNorm: finished by a kernel (rowstat) or from the producer's per-tile partial
// sums (stat_in; summed in index order, so the result does not depend on who reads it)
__device__ __forceinline__ float2 row_stat(const GemmArgs& a, int m) {
    m = m < a.M ? m : a.M - 1;
    if (a.stat_in) {
        float s1 = 0.f, s2 = 0.f;
        int b = 0;
        for (; b + 8 <= a.stat_in_blocks; b += 8) {      // eight loads in flight, added in index order
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2*>(a.stat_in + 2 * ((size_t)(b + u) * a.stat_in_stride + m));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s1 += v[u].x;
                s2 += v[u].y;
            }
        }
        for (; b < a.stat_in_blocks; ++b) {
            const float2 v = *reinterpret_cast<const float2*>(a.stat_in + 2 * ((size_t)b * a.stat_in_stride + m));
            s1 += v.x;
            s2 += v.y;
        }
        const float inv = 1.0f / (float)a.ln_dim;
        const float mean = s1 * inv;
        const float var = fmaxf(s2 * inv - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + a.ln_eps);
        return make_float2(rstd, rstd * mean);
    }
    return *reinterpret_cast<const float2*>(a.rowstat + 2 * (size_t)m);
}

// The same in two steps for the GEMM kernels: thread t of a workgroup requests the (at most four) partial pairs of row m0 + t
// when the main loop ends, and finishes them into an LDS table once the next tile's prologue has been issued.
__device__ __forceinline__ void row_stat_request(const GemmArgs& a, int m, float2 (&pv)[4]) {
    m = m < a.M ? m : a.M - 1;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        pv[b] = b < a.stat_in_blocks ? *reinterpret_cast<const float2*>(a.stat_in + 2 * ((size_t)b * a.stat_in_stride + m)) : make_float2(0.f, 0.f);
}
__device__ __forceinline__ float2 row_stat_finish(const GemmArgs& a, const float2 (&pv)[4]) {
    const float s1 = ((pv[0].x + pv[1].x) + pv[2].x) + pv[3].x, s2 = ((pv[0].y + pv[1].y) + pv[2].y) + pv[3].y;
    const float inv = 1.0f / (float)a.ln_dim;
    const float mean = s1 * inv;
    const float var = fmaxf(s2 * inv - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + a.ln_eps);
    return make_float2(rstd, rstd * mean);
}

// HIPTS_EPI_PRIO (GemmArgs::epi_prio, A/B): the two waves of a SIMD take turns at s_setprio 1 from one epilogue step to the next.  Both are
// VALU- or latency-bound there and arbitration is by priority, then AGE: the younger wave (wave_m = 1) otherwise gets the leftover issue
// slots for the whole epilogue and finishes 5-8 k cycles after its partner, who then waits for it at the next tile's first barrier
// (profiles/r04_gemm_tile_stamps.txt: epilogue 10.5 k / 15.4 k and 21.8 k / 29.8 k cycles).
__device__ __forceinline__ void epi_turn(const GemmArgs& a, int wave_m, int step) {
    if (a.epi_prio) {
        if ((step ^ wave_m) & 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}

// Element offset of x[m][nc .. nc + 3] in the fp32 residual stream (round 5).  Row-major, or -- GemmArgs::x_blocked -- 16 x 16 blocks of
// 1 KB: [m / 16][nc / 16][m % 16][nc % 16].  An epilogue load / store instruction of the accumulator layout (lane = row lr, four lanes x 16 B
// per row of a 16-column block) touches 16 rows x 64 B of a row-major stream -- sixteen half lines, twice the address work per byte of
// whole lines, on the launch's 512 such instructions per tile and CU -- and ONE contiguous kilobyte (eight whole lines) of the blocked one.
// The stream is a workspace nobody else reads: only residual epilogues touch it (ld % 16 == 0, rows allocated up to a multiple of 16).
__device__ __forceinline__ size_t x_off(const GemmArgs& a, int m, int nc, int ld) {
    if (a.x_blocked) return (((size_t)(m >> 4) * (size_t)(ld >> 4) + (size_t)(nc >> 4)) << 8) + (size_t)((m & 15) * 16 + (nc & 15));
    return (size_t)m * ld + nc;
}

// The bias values an epilogue needs, in its accumulator layout (V^T: one column per lane and 16-column
// block, in [j][0]; otherwise four consecutive columns).  The persistent loop issues these loads before
// the next tile's prologue so that their latency is not the first thing the epilogue waits for.
template <int EPI>
__device__ __forceinline__ void load_cols(const float* __restrict__ vec, const GemmArgs& a, int n0, int wave_n, int lane, f32x4 (&bv)[4]);
template <int EPI>
__device__ __forceinline__ void load_bias(const GemmArgs& a, int n0, int wave_n, int lane, f32x4 (&bv)[4]) {
    load_cols<EPI>(a.bias, a, n0, wave_n, lane, bv);
}
// a per-column vector (bias, col_u) in the epilogue's accumulator layout
template <int EPI>
__device__ __forceinline__ void load_cols(const float* __restrict__ vec, const GemmArgs& a, int n0, int wave_n, int lane, f32x4 (&bv)[4]) {
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (EPI == EPI_VT) {
            const int n = n0 + wave_n * 64 + j * 16 + lr;
            bv[j] = f32x4{n < a.N ? vec[n] : 0.f, 0.f, 0.f, 0.f};
        } else if constexpr (EPI == EPI_HEAD) {
            bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const int nc = n0 + wave_n * 64 + j * 16 + 4 * lq;
            bv[j] = nc < a.N ? *reinterpret_cast<const f32x4*>(vec + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// StarReLU (MetaFormer): s * relu(x)^2 + b with scalar s, b.
__device__ __forceinline__ f32x4 star_relu4(f32x4 x, float s, float b, int kind = 0) {
    if (kind == 1) {        // SiLU: x / (1 + 2^(-x log2 e))
        const f32x4 g = x * -1.4426950408889634f;
        const f32x4 e{__builtin_amdgcn_exp2f(g[0]), __builtin_amdgcn_exp2f(g[1]), __builtin_amdgcn_exp2f(g[2]), __builtin_amdgcn_exp2f(g[3])};
        const f32x4 d = e + 1.0f;
        return x * f32x4{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    }
    if (kind == 2) return x;
    const f32x4 r = __builtin_elementwise_max(x, f32x4{0.f, 0.f, 0.f, 0.f});
    return r * r * s + b;
}

// Row-contiguous store of a wave's (16 MR) x 64 block of 16-bit values through a private 8 KB LDS image (the
// mechanism of gemm_epilogue_staged, for epilogues that produce their values from a callback): lane layout
// in = (row 16 i + lr, columns 16 j + 4 lq ..+3), out = 8 rows x 128 B per store instruction.
template <int MR, bool F16, int P0 = 0, int P1 = 2, typename ValueOf>
__device__ __forceinline__ void staged_store_rows(char* region, int lane, int mrow0, int M, bf16_t* out, int ld, int ncol0, int N,
                                                  ValueOf value_of) {
    const int lr = lane & 15, lq = lane >> 4, lc = lane & 7, lrow = lane >> 3;
    const bool nvl = ncol0 + lc * 8 < N;
#pragma unroll
    for (int pass = P0; pass < P1; ++pass) {
        if (pass) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = pass * 4 + ii;
            if (i >= MR) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = ii * 16 + lr;
                const int pc = (j * 2 + (lq >> 1)) ^ ((row >> 1) & 7);
                const f32x4 v = value_of(i, j);
                *reinterpret_cast<bf16x4*>(region + row * 128 + pc * 16 + (lq & 1) * 8) = pack4<F16>(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (HIPTS_STAGED_INTERIOR && mrow0 + pass * 64 + 64 <= M && pass * 64 + 64 <= MR * 16 && ncol0 + 64 <= N) {
            // all 64 rows x 64 columns of this pass exist (wave-uniform: every tile but the last row / column of tiles): eight LDS reads,
            // then eight stores from one base address -- with the per-lane test each store is an exec-masked block with its own LDS read,
            // lgkmcnt(0) and 64-bit address arithmetic
            bf16_t* base = out + (size_t)(mrow0 + pass * 64 + lrow) * ld + ncol0 + lc * 8;
            uint4 v[8];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int row = r8 * 8 + lrow;
                v[r8] = *reinterpret_cast<const uint4*>(region + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) *reinterpret_cast<uint4*>(base + (size_t)r8 * 8 * ld) = v[r8];
        } else {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int row = r8 * 8 + lrow;
                const int m = mrow0 + pass * 64 + row;
                const uint4 v = *reinterpret_cast<const uint4*>(region + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
                if (pass * 64 + row >= MR * 16 || m >= M || !nvl) continue;
                *reinterpret_cast<uint4*>(out + (size_t)m * ld + ncol0 + lc * 8) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// The same for e4m3 outputs: a (16 MR) x 64 block is 64 B per row; image rows of 64 B with the 16 B chunk XOR-swizzled by
// (row >> 2) & 3 (conflict-free 4 B writes: bank = 16 (row & 3) + 4 (chunk ^ (row >> 2) & 3) + lq), read back as
// 16 rows x 64 B per store instruction.
template <int MR, typename ValueOf>
__device__ __forceinline__ void staged_store_rows8(char* region, int lane, int mrow0, int M, uint8_t* out, int ld, int ncol0, int N,
                                                   ValueOf value_of) {
    const int lr = lane & 15, lq = lane >> 4, lc = lane & 3, lrow = lane >> 2;
    const bool nvl = ncol0 + lc * 16 < N;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = pass * 4 + ii;
            if (i >= MR) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = ii * 16 + lr;
                const int pc = j ^ ((row >> 2) & 3);
                const f32x4 v = value_of(i, j);
                *reinterpret_cast<uint32_t*>(region + row * 64 + pc * 16 + lq * 4) = pack4_e4m3(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r16 = 0; r16 < 4; ++r16) {
            const int row = r16 * 16 + lrow;
            const int m = mrow0 + pass * 64 + row;
            const uint4 v = *reinterpret_cast<const uint4*>(region + row * 64 + ((lc ^ ((row >> 2) & 3)) * 16));
            if (pass * 64 + row >= MR * 16 || m >= M || !nvl) continue;
            *reinterpret_cast<uint4*>(out + (size_t)m * ld + ncol0 + lc * 16) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// One or two 16-row blocks (32 rows x 64 columns) of 16-bit values through a 4 KB wave-private image: the per-block form of
// staged_store_rows for epilogues that finish their rows two blocks at a time (EPI_RESID_XG).
template <bool F16, typename ValueOf>
__device__ __forceinline__ void staged_store_2blocks(char* image, int lane, int mrow_first, int nblk, int M, bf16_t* out, int ld, int ncol0, int N,
                                                     ValueOf value_of) {
    const int lr = lane & 15, lq = lane >> 4, lc = lane & 7, lrow = lane >> 3;
    const bool nvl = ncol0 + lc * 8 < N;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nblk) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = u * 16 + lr;
            const int pc = (j * 2 + (lq >> 1)) ^ ((row >> 1) & 7);
            const f32x4 v = value_of(u, j);
            *reinterpret_cast<bf16x4*>(image + row * 128 + pc * 16 + (lq & 1) * 8) = pack4<F16>(v[0], v[1], v[2], v[3]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r8 = 0; r8 < 4; ++r8) {
        const int row = r8 * 8 + lrow;
        const int m = mrow_first + row;
        const uint4 v = *reinterpret_cast<const uint4*>(image + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
        if (row < nblk * 16 && m < M && nvl) *reinterpret_cast<uint4*>(out + (size_t)m * ld + ncol0 + lc * 8) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// staged_store_2blocks for a tile that lies inside the matrix: two full row blocks, nothing predicated (one basic block)
template <bool F16, typename ValueOf>
__device__ __forceinline__ void staged_store_2blocks_interior(char* image, int lane, bf16_t* out_rows, int ld, ValueOf value_of) {
    const int lr = lane & 15, lq = lane >> 4, lc = lane & 7, lrow = lane >> 3;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = u * 16 + lr;
            const int pc = (j * 2 + (lq >> 1)) ^ ((row >> 1) & 7);
            const f32x4 v = value_of(u, j);
            *reinterpret_cast<bf16x4*>(image + row * 128 + pc * 16 + (lq & 1) * 8) = pack4<F16>(v[0], v[1], v[2], v[3]);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r8 = 0; r8 < 4; ++r8) {
        const int row = r8 * 8 + lrow;
        const uint4 v = *reinterpret_cast<const uint4*>(image + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
        *reinterpret_cast<uint4*>(out_rows + (size_t)row * ld + lc * 8) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A (16 MR) x 32 block of 16-bit values (the SwiGLU product: half as many output columns as accumulator columns):
// 64 B per row, the image geometry of staged_store_rows8 with 8 B per (row block, column block) -- lane (lr, lq) writes
// columns 16 jj + 4 lq ..+3 of row 16 i + lr; bank = 16 (row & 3) + 4 (chunk ^ (row >> 2) & 3) + 2 (lq & 1) per half wave.
template <int MR, bool F16, typename ValueOf>
__device__ __forceinline__ void staged_store_half_rows(char* region, int lane, int mrow0, int M, bf16_t* out, int ld, int ocol0, int Nout,
                                                       ValueOf value_of) {
    const int lr = lane & 15, lq = lane >> 4, lc = lane & 3, lrow = lane >> 2;
    const bool nvl = ocol0 + lc * 8 < Nout;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = pass * 4 + ii;
            if (i >= MR) continue;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int row = ii * 16 + lr;
                const int pc = (jj * 2 + (lq >> 1)) ^ ((row >> 2) & 3);
                const f32x4 v = value_of(i, jj);
                *reinterpret_cast<bf16x4*>(region + row * 64 + pc * 16 + (lq & 1) * 8) = pack4<F16>(v[0], v[1], v[2], v[3]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (HIPTS_STAGED_INTERIOR && mrow0 + pass * 64 + 64 <= M && pass * 64 + 64 <= MR * 16 && ocol0 + 32 <= Nout) {
            // the pass lies inside the matrix (wave-uniform): reads, then stores from one base address (see staged_store_rows)
            bf16_t* base = out + (size_t)(mrow0 + pass * 64 + lrow) * ld + ocol0 + lc * 8;
            uint4 v[4];
#pragma unroll
            for (int r16 = 0; r16 < 4; ++r16) {
                const int row = r16 * 16 + lrow;
                v[r16] = *reinterpret_cast<const uint4*>(region + row * 64 + ((lc ^ ((row >> 2) & 3)) * 16));
            }
#pragma unroll
            for (int r16 = 0; r16 < 4; ++r16) *reinterpret_cast<uint4*>(base + (size_t)r16 * 16 * ld) = v[r16];
        } else {
#pragma unroll
            for (int r16 = 0; r16 < 4; ++r16) {
                const int row = r16 * 16 + lrow;
                const int m = mrow0 + pass * 64 + row;
                const uint4 v = *reinterpret_cast<const uint4*>(region + row * 64 + ((lc ^ ((row >> 2) & 3)) * 16));
                if (pass * 64 + row >= MR * 16 || m >= M || !nvl) continue;
                *reinterpret_cast<uint4*>(out + (size_t)m * ld + ocol0 + lc * 8) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Epilogue shared by both main-loop variants.  acc[i][j]: 16 x 16 tile (i: 16-row block of the
// wave's 128 rows, j: 16-column block of its 64 columns).  Loads that feed the epilogue (bias,
// positional embedding, residual) are issued in batches of four before their first use so their
// latencies overlap instead of forming a chain of 32 dependent round trips.
template <int EPI, int MR = 8, bool F16 = false, bool INT = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[MR][4], int m0, int n0, int wave_m, int wave_n,
                                              int lane, const f32x4* bias_pre = nullptr, char* scratch = nullptr, const float2* st_lds = nullptr,
                                              const f32x4* colvec = nullptr, bool finish_stats = true) {
    // finish_stats = false (gemm4.hip: a wave holds two wave-parts and calls this twice): RESID_XG / RESID_XGI leave their row sums in
    // `red` and the caller adds the four parts of a row once both calls are through (resid_xg_finish_stats)
    // colvec (INT): gamma | bias | col_u of ALL the launch's columns (N <= 1024; 256 f32x4 each), staged into LDS once per workgroup: a
    // per-tile load issued here, behind the next tile's sixteen LDS-DMA requests, is only back when those have all landed.
    // INT (RESID_XG / RESID_XGI, MR 8): the launcher promises that every tile of the launch lies inside the matrix (M and N multiples of
    // 256), that there is no positional table and that the 16-bit copy and the row sums are both wanted
    // st_lds: (rstd, rstd * mean) of the tile's rows m0 .. m0 + 255 finished into LDS by the caller (RESID_ROWSTAT / RESID_XGI with the
    // statistics still as the producer's partials); null: read per lane (row_stat)
    const int lr = lane & 15, lq = lane >> 4;
    const int ld = a.ld_out ? a.ld_out : a.N;
    f32x4 bias_v[4];
    if (bias_pre) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias_v[j] = bias_pre[j];
    } else {
        load_bias<EPI>(a, n0, wave_n, lane, bias_v);
    }
    if constexpr (EPI == EPI_VT) {
        // natural order: lane = column n, registers = 4 consecutive rows (tokens)
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = bias_v[j][0];
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int m = m0 + wave_m * (MR * 16) + i * 16 + 4 * lq;
            if (m >= a.M) continue;
            const int b = m / a.tokens, t = m - b * a.tokens;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wave_n * 64 + j * 16 + lr;
                if (n >= a.N) continue;
                const f32x4 c = acc[i][j];
                const int head = n >> a.hd_log2, d = n & ((1 << a.hd_log2) - 1);
                const bf16x4 o = pack4<F16>(c[0] + bv[j], c[1] + bv[j], c[2] + bv[j], c[3] + bv[j]);
                *reinterpret_cast<bf16x4*>(a.out_bf16 + ((((size_t)(b * a.heads + head)) << a.hd_log2) + d) * a.tokens_pad + t) = o;
            }
        }
    } else if constexpr (EPI == EPI_HEAD) {
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int m = m0 + wave_m * (MR * 16) + i * 16 + lr;
            if (m >= a.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wave_n * 64 + j * 16 + 4 * lq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < a.N) {
                        const float v = acc[i][j][e] + a.bias[n + e];
                        if (a.out_f32) a.out_f32[(size_t)m * ld + n + e] = v;
                        if (a.out2_f32) a.out2_f32[(size_t)m * ld + n + e] = 1.0f / (1.0f + expf(-v));
                    }
                }
            }
        }
    } else {
        // swapped order: lane = row m, registers = 4 consecutive columns n
        int nc[4];
        bool nv[4];
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            nc[j] = n0 + wave_n * 64 + j * 16 + 4 * lq;
            nv[j] = nc[j] < a.N;
            bv[j] = bias_v[j];
        }
        if constexpr (EPI == EPI_RESID_LN) {
            // ---- pass 1: the residual read-modify-write; the new row values stay in acc
            const bool scaled = a.res_scale != nullptr;
            f32x4 rs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rs[j] = (scaled && nv[j]) ? *reinterpret_cast<const f32x4*>(a.res_scale + nc[j]) : f32x4{1.f, 1.f, 1.f, 1.f};
            constexpr int RB = 2;       // (4 spills registers here: mean / rstd / gamma live on top of acc)
#pragma unroll
            for (int i2 = 0; i2 < MR; i2 += RB) {
                f32x4 xv[RB][4];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (i2 + u >= MR) continue;
                    const int m = m0 + wave_m * (MR * 16) + (i2 + u) * 16 + lr;
                    const int mr = m < a.M ? m : a.M - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[u][j] = nv[j] ? *reinterpret_cast<const f32x4*>(a.out_f32 + x_off(a, mr, nc[j], ld)) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (i2 + u >= MR) continue;
                    const int m = m0 + wave_m * (MR * 16) + (i2 + u) * 16 + lr;
                    const int mr = m < a.M ? m : a.M - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i2 + u][j] = nv[j] ? xv[u][j] * rs[j] + (acc[i2 + u][j] + bv[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
                        if (nv[j] && m < a.M) *reinterpret_cast<f32x4*>(a.out_f32 + x_off(a, mr, nc[j], ld)) = acc[i2 + u][j];
                    }
                }
            }
            // ---- pass 2 / 3: row mean and variance.  A row's N <= 256 columns are spread over the four waves
            // of a wave group (64 each) and, inside a wave, over the four lane quarters: butterfly over the
            // quarters, then the wave partials meet in LDS (red[row][wave_n]).  Two passes like layernorm_kernel.
            float* red = reinterpret_cast<float*>(scratch);
            const float inv_n = 1.0f / (float)a.N;
            float mean[MR], rstd[MR];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    float sacc = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!nv[j]) continue;
                        if (pass == 0) {
                            sacc += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
                        } else {
                            const f32x4 d = acc[i][j] - mean[i];
                            sacc += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                        }
                    }
                    sacc += __shfl_xor(sacc, 16);
                    sacc += __shfl_xor(sacc, 32);
                    if (lq == 0) red[(wave_m * (MR * 16) + i * 16 + lr) * 4 + wave_n] = sacc;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const f32x4 p = *reinterpret_cast<const f32x4*>(red + (wave_m * (MR * 16) + i * 16 + lr) * 4);
                    const float t = ((p[0] + p[1]) + (p[2] + p[3])) * inv_n;
                    if (pass == 0) mean[i] = t;
                    else rstd[i] = 1.0f / sqrtf(t + a.ln_eps);
                }
                __builtin_amdgcn_s_barrier();       // red is rewritten by the next pass / the next tile
            }
            // ---- pass 4: normalise and store the bf16 operand of the next GEMM
            f32x4 gv[4], bt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gv[j] = nv[j] ? *reinterpret_cast<const f32x4*>(a.ln_gamma + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
                bt[j] = (a.ln_beta && nv[j]) ? *reinterpret_cast<const f32x4*>(a.ln_beta + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // through the wave's private 8 KB LDS image (red is dead: every wave is past the last barrier), so that
            // the stores are whole 128 B lines
            const int wave_id = wave_m * 4 + wave_n;
            if (a.out8)
                staged_store_rows8<MR>(scratch + wave_id * 8192, lane, m0 + wave_m * (MR * 16), a.M, reinterpret_cast<uint8_t*>(a.out_bf16), ld,
                                       n0 + wave_n * 64, a.N, [&](int i, int j) { return (acc[i][j] - mean[i]) * rstd[i] * gv[j] + bt[j]; });
            else
            staged_store_rows<MR, F16>(scratch + wave_id * 8192, lane, m0 + wave_m * (MR * 16), a.M, a.out_bf16, ld, n0 + wave_n * 64, a.N,
                                       [&](int i, int j) { return (acc[i][j] - mean[i]) * rstd[i] * gv[j] + bt[j]; });
            return;
        }
        if constexpr (EPI == EPI_RESID_XG || EPI == EPI_RESID_XGI) {
            constexpr bool fold = EPI == EPI_RESID_XGI;
            // (A software-pipelined form with FOUR row blocks per step -- loads of rows 64..127 in flight under the LDS-staged copy
            // of rows 0..63 -- needs 64 load registers, 128 accumulators, gamma and bias at once: 95 spilled registers, 242 us.)
            f32x4 uv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) uv[j] = (!INT && fold && nv[j]) ? *reinterpret_cast<const f32x4*>(a.col_u + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
            // ---- two 16-row blocks at a time: residual read-modify-write, then -- while the next blocks' loads are issued -- this
            // pair's share of the row sums and its gamma-scaled 16-bit copy through a 4 KB wave-private LDS image.  (With the copy
            // and the statistics after the whole read-modify-write the 16-bit stores formed a tail of their own: 215 us per
            // launch; two row blocks of loads in flight and four measure the same.)
            constexpr int RB = 2;
            const int ncol0 = n0 + wave_n * 64;
            const bool copy = a.out_bf16 != nullptr;
            f32x4 gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gv[j] = (!INT && copy && nv[j]) ? *reinterpret_cast<const f32x4*>(a.ln_gamma + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
            char* image = scratch + (wave_m * 4 + wave_n) * 4096;                     // 8 x 4 KB
            float2* red = reinterpret_cast<float2*>(scratch + 8 * 4096);              // [256 rows][4 waves] behind the images
            // ---- INT: launches whose tiles all lie inside the matrix (round 4; the ViT's 23 launches per forward) get an instantiation of
            // their own in which nothing is predicated per lane.  The general loop below compiles into one exec-masked basic block per
            // load and per store -- 142 branches, the block's loads drained with s_waitcnt vmcnt(0) together with the stores of the
            // block before, four drains per tile (profiles/r04_c_resid_epilogue.txt).  Here the loop is one basic block: the next
            // pair's eight loads are requested BEFORE this pair's read-modify-write, statistics and staged 16-bit copy (two register
            // sets; the waits are counted), and the stores never wait.  Same operations in the same order: the bits do not change.
            if constexpr (INT) {
                static_assert(MR == 8, "interior epilogue: full tiles");
                // the per-column vectors (gamma, bias, col_u) are read from LDS where they are used instead of living in 48 registers:
                // with them, 128 accumulators and two sets of 32 load registers the allocator spilled addresses, and a scratch reload is
                // a VMEM operation -- s_waitcnt vmcnt(0), i.e. a drain of every store in flight, in front of each use
                const f32x4* cvec = colvec + (ncol0 >> 2);
                // row-major: row block i adds 16 i ld, column block j 16 floats; blocked: block (i, j) is 256 floats, a row block ld / 16 blocks
                const int lane_off = a.x_blocked ? lr * 16 + 4 * lq : lr * ld + 4 * lq;
                const size_t rb_step = a.x_blocked ? (size_t)(ld >> 4) * 256 : (size_t)16 * ld;
                const int cb_step = a.x_blocked ? 256 : 16;
                float* base = a.out_f32 + x_off(a, m0 + wave_m * 128, ncol0, ld);            // uniform
                // HIPTS_INT_DEPTH (2 / 3): register sets of loads in flight.  With the stream in blocks the address path is no longer what a
                // step waits for, so a third set (two steps ahead, 32 more registers: the instantiation is built for 256) was tried
                constexpr int DEPTH = HIPTS_INT_DEPTH;
                f32x4 xv[DEPTH][RB][4];
                auto request = [&](int buf, int i2) {
#pragma unroll
                    for (int u = 0; u < RB; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            xv[buf][u][j] = *reinterpret_cast<const f32x4*>(base + (size_t)(i2 + u) * rb_step + lane_off + j * cb_step);
                };
                request(0, 0);
                if constexpr (DEPTH == 3) request(1, RB);
#pragma unroll
                for (int i2 = 0; i2 < 8; i2 += RB) {
                    epi_turn(a, wave_m, i2 / RB);
                    const int cur = (i2 / RB) % DEPTH;
                    if constexpr (DEPTH == 3) {
                        if (i2 + 2 * RB < 8) request((cur + 2) % 3, i2 + 2 * RB);
                    } else {
                        if (i2 + RB < 8) request(cur ^ 1, i2 + RB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        float2 st = make_float2(1.f, 0.f);
                        if constexpr (fold) st = st_lds ? st_lds[wave_m * 128 + (i2 + u) * 16 + lr] : row_stat(a, m0 + wave_m * 128 + (i2 + u) * 16 + lr);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 bj = cvec[256 + j * 4 + lq];
                            if constexpr (fold) acc[i2 + u][j] = xv[cur][u][j] + (acc[i2 + u][j] * st.x + (bj - cvec[512 + j * 4 + lq] * st.y));
                            else acc[i2 + u][j] = xv[cur][u][j] + (acc[i2 + u][j] + bj);
                            // (non-temporal loads / loads and stores of this stream, __builtin_nontemporal_*: 5277 / 5256-5294 images/s against
                            // 5294-5324 with the default policy on one box -- no gain, not kept)
                            *reinterpret_cast<f32x4*>(base + (size_t)(i2 + u) * rb_step + lane_off + j * cb_step) = acc[i2 + u][j];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int i = i2 + u;
                        f32x4 t = acc[i][0], q = acc[i][0] * acc[i][0];
#pragma unroll
                        for (int j = 1; j < 4; ++j) {
                            t = t + acc[i][j];
                            q = q + acc[i][j] * acc[i][j];
                        }
                        float s1 = (t[0] + t[1]) + (t[2] + t[3]), s2 = (q[0] + q[1]) + (q[2] + q[3]);
                        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                        if (lq == 0) red[(wave_m * 128 + i * 16 + lr) * 4 + wave_n] = make_float2(s1, s2);
                    }
                    staged_store_2blocks_interior<F16>(image, lane, a.out_bf16 + (size_t)(m0 + wave_m * 128 + i2 * 16) * ld + ncol0, ld,
                                                       [&](int u, int j) { return acc[i2 + u][j] * cvec[j * 4 + lq]; });
                    __builtin_amdgcn_sched_barrier(0);      // the pair after next is not requested early (registers)
                }
            } else {
#pragma unroll
            for (int i2 = 0; i2 < MR; i2 += RB) {
                epi_turn(a, wave_m, i2 / RB);
                f32x4 xv[RB][4];
                float2 st[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (i2 + u >= MR) continue;
                    const int m = m0 + wave_m * (MR * 16) + (i2 + u) * 16 + lr;
                    const int mc = m < a.M ? m : a.M - 1;
                    // a.pos: the patch embedding as the first "residual" GEMM -- x = acc * qscale + bias + pos[token] instead of x += ...
                    st[u] = fold ? (st_lds ? st_lds[mc - m0] : row_stat(a, mc)) : make_float2(1.f, 0.f);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xv[u][j] = !nv[j] ? f32x4{0.f, 0.f, 0.f, 0.f}
                                          : *reinterpret_cast<const f32x4*>(a.pos ? a.pos + (size_t)(mc % a.tokens) * a.N + nc[j] : a.out_f32 + x_off(a, mc, nc[j], ld));
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (i2 + u >= MR) continue;
                    const int m = m0 + wave_m * (MR * 16) + (i2 + u) * 16 + lr;
                    const int mcs = m < a.M ? m : a.M - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (fold)
                            acc[i2 + u][j] = nv[j] ? xv[u][j] + (acc[i2 + u][j] * st[u].x + (bv[j] - uv[j] * st[u].y)) : f32x4{0.f, 0.f, 0.f, 0.f};
                        else if (a.res_scale)      // (round 5) the CAFormer's scaled residual, EPI_RESCALE's expression: x = x * rs + (acc + bias)
                            acc[i2 + u][j] = nv[j] ? xv[u][j] * *reinterpret_cast<const f32x4*>(a.res_scale + nc[j]) + (acc[i2 + u][j] + bv[j])
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
                        else
                            acc[i2 + u][j] = nv[j] ? (a.pos ? acc[i2 + u][j] * a.qscale + bv[j] + xv[u][j] : xv[u][j] + (acc[i2 + u][j] + bv[j]))
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
                        if (nv[j] && m < a.M) *reinterpret_cast<f32x4*>(a.out_f32 + x_off(a, mcs, nc[j], ld)) = acc[i2 + u][j];
                    }
                }
                if (!copy) continue;
                if (a.stat_part) {
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int i = i2 + u;
                        if (i >= MR) continue;
                        f32x4 t = acc[i][0], q = acc[i][0] * acc[i][0];
#pragma unroll
                        for (int j = 1; j < 4; ++j) {
                            t = t + acc[i][j];
                            q = q + acc[i][j] * acc[i][j];
                        }
                        float s1 = (t[0] + t[1]) + (t[2] + t[3]), s2 = (q[0] + q[1]) + (q[2] + q[3]);
                        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                        if (lq == 0) red[(wave_m * (MR * 16) + i * 16 + lr) * 4 + wave_n] = make_float2(s1, s2);      // columns past N hold zeros
                    }
                }
                staged_store_2blocks<F16>(image, lane, m0 + wave_m * (MR * 16) + i2 * 16, (MR - i2) < RB ? (MR - i2) : RB, a.M, a.out_bf16, ld, ncol0, a.N,
                                          [&](int u, int j) { return acc[i2 + u][j] * gv[j]; });
            }
            }       // !INT
            if (copy && a.stat_part && finish_stats) {
                // the four waves that hold a row meet (uniform); red is written again only after the next tile's main loop
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int t = wave_n * 64 + lane;                     // 256 threads of a wave group own its MR * 16 rows
                if (t < MR * 16) {
                    const int r = wave_m * (MR * 16) + t;
                    const float2 p0 = red[r * 4], p1 = red[r * 4 + 1], p2 = red[r * 4 + 2], p3 = red[r * 4 + 3];
                    const int m = m0 + r;
                    if (m < a.M)
                        *reinterpret_cast<float2*>(a.stat_part + 2 * ((size_t)(n0 >> 8) * a.stat_stride + m)) =
                            make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
                }
            }
            return;
        }
        if constexpr (EPI == EPI_RESID || EPI == EPI_RESCALE || EPI == EPI_RESID_ROWSTAT) {
            f32x4 rs[4];
            if constexpr (EPI == EPI_RESCALE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rs[j] = nv[j] ? *reinterpret_cast<const f32x4*>(a.res_scale + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (EPI == EPI_RESID_ROWSTAT) {      // rs = col_u
#pragma unroll
                for (int j = 0; j < 4; ++j) rs[j] = nv[j] ? *reinterpret_cast<const f32x4*>(a.col_u + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // read-modify-write of the fp32 residual stream: RB x 4 loads of 16 B per lane in flight before the
            // first dependent add (RB 16-row blocks; the fragment registers of the main loop are free here),
            // so a CU keeps RB x 32 KB outstanding -- the epilogue is bound by HBM latency x bytes in flight.
            // A tile that lies inside the matrix (all but the last row / column of tiles) takes the branch-free copy of the
            // loop: with per-lane `if (valid) store` every store sits in its own exec-masked basic block, and the compiler
            // opens each block with s_waitcnt vmcnt(0) -- CDNA4 counts stores in vmcnt, so each of the 32 stores waited for
            // the previous one to be acknowledged.
            constexpr int RB = 4;
            auto rmw = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
                for (int i2 = 0; i2 < MR; i2 += RB) {
                    epi_turn(a, wave_m, i2 / RB);
                    f32x4 xv[RB][4];
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        if (i2 + u >= MR) continue;
                        const int m = m0 + wave_m * (MR * 16) + (i2 + u) * 16 + lr;
                        const int mr = (INTERIOR || m < a.M) ? m : a.M - 1;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            xv[u][j] = (INTERIOR || nv[j]) ? *reinterpret_cast<const f32x4*>(a.out_f32 + x_off(a, mr, nc[j], ld)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        if (i2 + u >= MR) continue;
                        const int m = m0 + wave_m * (MR * 16) + (i2 + u) * 16 + lr;
                        const int mc = (INTERIOR || m < a.M) ? m : a.M - 1;
                        // x_blocked with resid_rowmajor_out set (the last residual launch of a forward): the new rows leave row-major for
                        // the kernels behind the stream (final norm + pool)
                        float* row = (a.x_blocked && a.resid_rowmajor_out) ? a.resid_rowmajor_out + (size_t)mc * ld : nullptr;
                        float2 st = make_float2(1.f, 0.f);
                        if constexpr (EPI == EPI_RESID_ROWSTAT) st = st_lds ? st_lds[mc - m0] : row_stat(a, mc);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 v;
                            if constexpr (EPI == EPI_RESCALE)
                                v = xv[u][j] * rs[j] + (acc[i2 + u][j] + bv[j]);
                            else if constexpr (EPI == EPI_RESID_ROWSTAT)
                                v = xv[u][j] + ((acc[i2 + u][j] * st.x - rs[j] * st.y) + bv[j]);
                            else
                                v = xv[u][j] + (acc[i2 + u][j] + bv[j]);
                            if (INTERIOR || (nv[j] && m < a.M)) *reinterpret_cast<f32x4*>(row ? row + nc[j] : a.out_f32 + x_off(a, mc, nc[j], ld)) = v;
                        }
                    }
                }
            };
            if (m0 + 2 * MR * 16 <= a.M && n0 + BN <= a.N) rmw(std::true_type{});      // uniform over the workgroup
            else rmw(std::false_type{});
            return;
        }
        if constexpr (EPI == EPI_PATCH || EPI == EPI_BIAS) {
            // interior tiles: branch-free (see the residual epilogue above)
            if (m0 + 2 * MR * 16 <= a.M && n0 + BN <= a.N) {
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const int m = m0 + wave_m * (MR * 16) + i * 16 + lr;
                    f32x4 pv[4];
                    if constexpr (EPI == EPI_PATCH) {
                        const int t = m % a.tokens;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pv[j] = *reinterpret_cast<const f32x4*>(a.pos + (size_t)t * a.N + nc[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (EPI == EPI_PATCH) *reinterpret_cast<f32x4*>(a.out_f32 + (size_t)m * ld + nc[j]) = acc[i][j] * a.qscale + bv[j] + pv[j];
                        else *reinterpret_cast<f32x4*>(a.out_f32 + x_off(a, m, nc[j], ld)) = acc[i][j] + bv[j];
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int m = m0 + wave_m * (MR * 16) + i * 16 + lr;
            if (m >= a.M) continue;
            if constexpr (EPI == EPI_PATCH) {
                const int t = m % a.tokens;
                f32x4 pv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pv[j] = nv[j] ? *reinterpret_cast<const f32x4*>(a.pos + (size_t)t * a.N + nc[j]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (nv[j]) *reinterpret_cast<f32x4*>(a.out_f32 + (size_t)m * ld + nc[j]) = acc[i][j] * a.qscale + bv[j] + pv[j];
            } else if constexpr (EPI == EPI_BIAS) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (nv[j]) *reinterpret_cast<f32x4*>(a.out_f32 + x_off(a, m, nc[j], ld)) = acc[i][j] + bv[j];
            } else if constexpr (EPI == EPI_STAR) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!nv[j]) continue;
                    const f32x4 v = star_relu4(acc[i][j] + bv[j], a.star_scale, a.star_bias, a.star_kind);
                    *reinterpret_cast<bf16x4*>(a.out_bf16 + (size_t)m * ld + nc[j]) = pack4<F16>(v[0], v[1], v[2], v[3]);
                }
            } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!nv[j]) continue;
                    const f32x4 v = acc[i][j] + bv[j];
                    const f32x4 gv = gelu_f4(v, a.gelu_tanh);           // the staged epilogue's function, so a value does not depend on the path
                    const bf16x4 o = pack4<F16>(gv[0], gv[1], gv[2], gv[3]);
                    *reinterpret_cast<bf16x4*>(a.out_bf16 + (size_t)m * ld + nc[j]) = o;
                }
            } else if constexpr (EPI == EPI_QK) {
                const int b = m / a.tokens, t = m - b * a.tokens;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!nv[j]) continue;
                    const int which = nc[j] >= 2 * a.dim ? 2 : (nc[j] >= a.dim ? 1 : 0);       // q, k or (N = 3 dim) v
                    const int nn = nc[j] - which * a.dim;
                    const int head = nn >> a.hd_log2, d = nn & ((1 << a.hd_log2) - 1);
                    const float sc = which ? 1.0f : a.qscale;
                    const f32x4 v = acc[i][j] + bv[j];
                    const bf16x4 o = pack4<F16>(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
                    bf16_t* base = which == 2 ? a.out3_bf16 : (which ? a.out2_bf16 : a.out_bf16);
                    *reinterpret_cast<bf16x4*>(base + ((((size_t)(b * a.heads + head) * a.tokens_pad + t)) << a.hd_log2) + d) = o;
                }
            }
        }
    }
}

// Half-precision epilogues through LDS.  After the main loop one stage of LDS is free, so every wave
// packs its 16 MR x 64 output block, 64 rows at a time, into a private 8 KB LDS image and reads it back
// row-contiguous: each global store instruction then writes whole 128 B lines (8 rows x 128 B) instead of
// 16 scattered 32 B pieces.  Both LDS passes are bank-conflict free: the 16 B chunk of an image row is
// XOR-swizzled with (row >> 1) & 7 and rows alternate between the two 128 B halves of the 64 banks.
// Wave-private image: no workgroup barrier, LDS operations of one wave execute in order.
template <int EPI, int MR, bool F16>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& a, f32x4 (&acc)[MR][4], int m0, int n0, int wave_m, int wave_n,
                                                     int lane, char* region, const f32x4* bias_pre = nullptr, const float2* st_lds = nullptr,
                                                     float2* red = nullptr) {
    // red (SWIGLU with stat_part, persistent loop): [256 rows][4 waves] in LDS -- the four waves of a row leave their partial row sums
    // there and the caller adds them into ONE pair per (256-column tile, row)
    // st_lds: (rstd, rstd * mean) of the tile's rows m0 .. m0 + 255, finished from the producer's partials by the caller
    f32x4 bias_v[4];
    if (bias_pre) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bias_v[j] = bias_pre[j];
    } else {
        load_bias<EPI>(a, n0, wave_n, lane, bias_v);
    }
    static_assert(EPI == EPI_VT || EPI == EPI_GELU || EPI == EPI_STAR || EPI == EPI_QK || EPI == EPI_QK_ROPE || EPI == EPI_SWIGLU,
                  "staged epilogue: half-precision outputs only");
    constexpr bool IS_QK = EPI == EPI_QK || EPI == EPI_QK_ROPE;
    // LayerNorm folded into this GEMM (see EPI_RESID_XG): value = acc * rstd[row] + (bias[col] - col_u[col] * rstd_mean[row]);
    // without it the row pair is (1, 0) and col_u is zero, the same expression
    const bool fold = a.rowstat != nullptr || a.stat_in != nullptr;
    f32x4 cu[4];
    if (fold) load_cols<EPI>(a.col_u, a, n0, wave_n, lane, cu);
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) cu[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int lr = lane & 15, lq = lane >> 4;
    const int ncol0 = n0 + wave_n * 64;
    const int mrow0 = m0 + wave_m * (MR * 16);
    const int lc = lane & 7, lrow = lane >> 3;
    if constexpr (EPI == EPI_VT) {
        // acc[i][j][e] = (token mrow0 + 16 i + 4 lq + e, column ncol0 + 16 j + lr); image [64 columns][128 B = 64 tokens]
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = bias_v[j][0];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = pass * 4 + ii;
                if (i >= MR) continue;
                f32x4 sx{1.f, 1.f, 1.f, 1.f}, sy{0.f, 0.f, 0.f, 0.f};      // this lane's four rows (tokens)
                if (fold) {
                    const int mr = mrow0 + i * 16 + 4 * lq;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float2 st = st_lds ? st_lds[mr + e - m0] : row_stat(a, mr + e);
                        sx[e] = st.x;
                        sy[e] = st.y;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 c = acc[i][j] * sx + (bv[j] - sy * cu[j][0]);
                    const int d = j * 16 + lr;
                    const int pc = (ii * 2 + (lq >> 1)) ^ ((d >> 1) & 7);
                    *reinterpret_cast<bf16x4*>(region + d * 128 + pc * 16 + (lq & 1) * 8) = pack4<F16>(c[0], c[1], c[2], c[3]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int m = mrow0 + pass * 64 + lc * 8;
            const int b = m / a.tokens, t = m - b * a.tokens;
            const bool mv = pass * 64 + lc * 8 < MR * 16 && m < a.M;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int d = r8 * 8 + lrow;
                const int n = ncol0 + d;
                const uint4 v = *reinterpret_cast<const uint4*>(region + d * 128 + ((lc ^ ((d >> 1) & 7)) * 16));
                if (mv && n < a.N)
                    *reinterpret_cast<uint4*>(a.out_bf16 +
                                              ((((size_t)(b * a.heads + (n >> a.hd_log2))) << a.hd_log2) + (n & ((1 << a.hd_log2) - 1))) * a.tokens_pad + t) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else {
        // acc[i][j][e] = (row mrow0 + 16 i + lr, column ncol0 + 16 j + 4 lq + e); image [64 rows][128 B = 64 columns]
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = bias_v[j];
        if constexpr (EPI == EPI_STAR) {
            if (a.out8) {
                staged_store_rows8<MR>(region, lane, mrow0, a.M, reinterpret_cast<uint8_t*>(a.out_bf16), a.ld_out ? a.ld_out : a.N, ncol0, a.N,
                                       [&](int i, int j) { return star_relu4(acc[i][j] + bv[j], a.star_scale, a.star_bias, a.star_kind); });
                return;
            }
        }
        if constexpr (EPI == EPI_SWIGLU) {
            // acc[i][0..1]: gate columns, acc[i][2..3]: value columns of the same 32 hidden units
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                float2 st = make_float2(1.f, 0.f);
                if (fold) st = st_lds ? st_lds[mrow0 - m0 + i * 16 + lr] : row_stat(a, mrow0 + i * 16 + lr);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[i][jj] = star_relu4(acc[i][jj] * st.x + (bv[jj] - cu[jj] * st.y), 0.f, 0.f, 1) *
                                 (acc[i][jj + 2] * st.x + (bv[jj + 2] - cu[jj + 2] * st.y));
                if (a.stat_part) {
                    // this wave's share of the row statistics of the product (a LayerNorm over the hidden units follows):
                    // sum over the lane's 8 values, then over the four lane quarters
                    const f32x4 t = acc[i][0] + acc[i][1], q = acc[i][0] * acc[i][0] + acc[i][1] * acc[i][1];
                    float s1 = (t[0] + t[1]) + (t[2] + t[3]), s2 = (q[0] + q[1]) + (q[2] + q[3]);
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                    const int m = mrow0 + i * 16 + lr;
                    if (red) {
                        if (lq == 0) red[(mrow0 - m0 + i * 16 + lr) * 4 + wave_n] = ncol0 < a.N ? make_float2(s1, s2) : make_float2(0.f, 0.f);
                    } else if (lq == 0 && m < a.M && ncol0 < a.N)
                        *reinterpret_cast<float2*>(a.stat_part + 2 * ((size_t)(ncol0 >> 6) * a.stat_stride + m)) = make_float2(s1, s2);
                }
                // the gamma of the LayerNorm that follows, applied to the stored operand (the statistics above are of the raw
                // product): the next GEMM then keeps its own weights unchanged
                if (a.ln_gamma) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int u0 = (ncol0 >> 1) + jj * 16 + 4 * lq;
                        if (u0 < (a.N >> 1)) acc[i][jj] = acc[i][jj] * *reinterpret_cast<const f32x4*>(a.ln_gamma + u0);
                    }
                }
            }
            staged_store_half_rows<MR, F16>(region, lane, mrow0, a.M, a.out_bf16, a.ld_out ? a.ld_out : a.N / 2, ncol0 >> 1, a.N >> 1,
                                            [&](int i, int jj) { return acc[i][jj]; });
            return;
        }
        const int which = !IS_QK ? 0 : (ncol0 >= 2 * a.dim ? 2 : (ncol0 >= a.dim ? 1 : 0));       // q, k or (N = 3 dim) v: uniform over the wave's 64 columns
        const float sc = (IS_QK && !which) ? a.qscale : 1.0f;
        const bool nvl = ncol0 + lc * 8 < a.N;
        const int ld = a.ld_out ? a.ld_out : a.N;
        bf16_t* qk_base = nullptr;
        int head = 0, hd_off = 0;          // this lane's 8 columns: head and offset inside the head
        if constexpr (IS_QK) {
            qk_base = which == 2 ? a.out3_bf16 : (which ? a.out2_bf16 : a.out_bf16);
            const int col = ncol0 - which * a.dim + lc * 8;
            head = col >> a.hd_log2;
            hd_off = col & ((1 << a.hd_log2) - 1);
        }
        // (image, token) of this lane's first output row; later rows advance by 8 without dividing again
        int qb = 0, qt = 0;
        if constexpr (IS_QK) {
            qb = (mrow0 + lrow) / a.tokens;
            qt = (mrow0 + lrow) - qb * a.tokens;
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = pass * 4 + ii;
                if (i >= MR) continue;
                epi_turn(a, wave_m, i);
                float2 st = make_float2(1.f, 0.f);
                if (fold) st = st_lds ? st_lds[mrow0 - m0 + i * 16 + lr] : row_stat(a, mrow0 + i * 16 + lr);
                f32x4 rp[4];                // QK_ROPE: (sin, cos, sin, cos) of this lane's two column pairs, per column block
                bool rok = false;
                if constexpr (EPI == EPI_QK_ROPE) {
                    const int m = mrow0 + i * 16 + lr;
                    const int t = m - (m / a.tokens) * a.tokens;
                    rok = t >= 1 && t <= a.rope_tokens && which < 2;         // v (the third column range of a fused q | k | v launch) is not rotated
                    const f32x4* rr = reinterpret_cast<const f32x4*>(a.rope) + (size_t)(rok ? t - 1 : 0) * 16 + lq;
#pragma unroll
                    for (int j = 0; j < 4; ++j) rp[j] = rr[j * 4];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = ii * 16 + lr;
                    const int pc = (j * 2 + (lq >> 1)) ^ ((row >> 1) & 7);
                    f32x4 v = acc[i][j] * st.x + (bv[j] - cu[j] * st.y);
                    if constexpr (EPI == EPI_QK_ROPE) {
                        if (rok)
                            v = f32x4{v[0] * rp[j][1] - v[1] * rp[j][0], v[1] * rp[j][1] + v[0] * rp[j][0],
                                      v[2] * rp[j][3] - v[3] * rp[j][2], v[3] * rp[j][3] + v[2] * rp[j][2]};
                    }
                    bf16x4 o;
                    if constexpr (EPI == EPI_GELU) {
                        const f32x4 gv = gelu_f4(v, a.gelu_tanh);
                        o = pack4<F16>(gv[0], gv[1], gv[2], gv[3]);
                    } else if constexpr (EPI == EPI_STAR) {
                        const f32x4 gv = star_relu4(v, a.star_scale, a.star_bias, a.star_kind);
                        o = pack4<F16>(gv[0], gv[1], gv[2], gv[3]);
                    } else
                        o = pack4<F16>(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
                    *reinterpret_cast<bf16x4*>(region + row * 128 + pc * 16 + (lq & 1) * 8) = o;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (EPI == EPI_GELU || EPI == EPI_STAR) {
                // all 64 rows x 64 columns of this pass exist (wave-uniform: every tile but the last row / column of tiles): eight LDS reads,
                // then eight stores from one base address (see staged_store_rows)
                if (HIPTS_STAGED_INTERIOR && mrow0 + pass * 64 + 64 <= a.M && pass * 64 + 64 <= MR * 16 && ncol0 + 64 <= a.N) {
                    bf16_t* base = a.out_bf16 + (size_t)(mrow0 + pass * 64 + lrow) * ld + ncol0 + lc * 8;
                    uint4 v[8];
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int row = r8 * 8 + lrow;
                        v[r8] = *reinterpret_cast<const uint4*>(region + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) *reinterpret_cast<uint4*>(base + (size_t)r8 * 8 * ld) = v[r8];
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    continue;
                }
            }
            if constexpr (IS_QK) {
                // the same for the q | k | v layouts: the eight (image, token) addresses first, then reads, then stores
                if (HIPTS_STAGED_INTERIOR && mrow0 + pass * 64 + 64 <= a.M && pass * 64 + 64 <= MR * 16 && ncol0 + 64 <= a.N) {
                    uint4 v[8];
                    size_t off[8];
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int row = r8 * 8 + lrow;
                        v[r8] = *reinterpret_cast<const uint4*>(region + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
                        off[r8] = ((((size_t)(qb * a.heads + head) * a.tokens_pad + qt)) << a.hd_log2) + hd_off;
                        qt += 8;
                        while (qt >= a.tokens) {
                            qt -= a.tokens;
                            ++qb;
                        }
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) *reinterpret_cast<uint4*>(qk_base + off[r8]) = v[r8];
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    continue;
                }
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int row = r8 * 8 + lrow;
                const int m = mrow0 + pass * 64 + row;
                const uint4 v = *reinterpret_cast<const uint4*>(region + row * 128 + ((lc ^ ((row >> 1) & 7)) * 16));
                const int qb_now = qb, qt_now = qt;
                if constexpr (IS_QK) {
                    qt += 8;
                    while (qt >= a.tokens) {      // at most once unless an image has fewer than 8 tokens
                        qt -= a.tokens;
                        ++qb;
                    }
                }
                if (pass * 64 + row >= MR * 16 || m >= a.M || !nvl) continue;
                if constexpr (EPI == EPI_GELU || EPI == EPI_STAR) {
                    *reinterpret_cast<uint4*>(a.out_bf16 + (size_t)m * ld + ncol0 + lc * 8) = v;
                } else {
                    *reinterpret_cast<uint4*>(qk_base + ((((size_t)(qb_now * a.heads + head) * a.tokens_pad + qt_now)) << a.hd_log2) + hd_off) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
