// d2v.hip -- Doc2Vec PV-DBOW inference, one wavefront per document.
//
// Replaces gensim Doc2Vec.infer_vector as the reference calls it (genmodel.py:169 for every
// document, webui.py:106,185 per query tag / per reranked document; model built at
// genmodel.py:159 with dm=0, vector_size=300, negative=5, hs=0, sample=1e-3).
// Algorithm restated from gensim 4.3.3 (doc2vec.py::infer_vector, doc2vec_inner.pyx::
// train_document_dbow / fast_document_dbow_neg, word2vec_inner.pyx: 48-bit LCG, EXP_TABLE,
// bisect_left over cum_table).  gensim itself is not available: parity is against
// oracle/csrc/oracle.c, which restates the same algorithm with the same explicit inputs
// (start vector v0, per-document seed -> per-epoch LCG state via splitmix64).
//
// A document is a serial chain of ~epochs*words*(1+negative) dot/axpy steps on one 300-d vector,
// so the parallelism is across documents: the vector lives in 5 VGPRs per lane (element i on lane
// i%64), dot products are 5 fused multiply-adds + a 6-step xor butterfly, control flow is
// wave-uniform.  Bound by dependent-load latency (syn1neg row gather, cum_table search), not by
// HBM or MFMA.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

using namespace hipts;

struct hipts_d2v {
    int device = 0;
    int64_t V = 0;
    int dim = 0, negative = 5;
    double exp_scale = 83.0;
    bool has_sample = false;
    DevBuf syn1neg, syn_lane, cum_table, sample_int, exp_table, word_vectors;      // word_vectors: PV-DM inference only (hipts_d2v_set_word_vectors)
    // syn_lane: the lane-major copy the planned kernel gathers (below)
    DevBuf ws_ptr, ws_words, ws_v0, ws_seeds, ws_out;
};

namespace {

constexpr int EXP_TABLE_SIZE = 1000;
constexpr int MAX_EXP = 6;
constexpr uint64_t LCG_MOD = 281474976710655ULL;   // 2^48 - 1

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// The dot product's cross-lane sum: p[l] + p[l ^ 32], then ^16, ^8, ^4, ^2, ^1 -- the order oracle/csrc/oracle.c::dot_wave64
// defines.  Every stage adds each lane's value and its partner's (a + b on one side, b + a on the other: the same float), so any
// way of fetching the partner gives the same bits; __shfl_xor fetches it through the LDS crossbar (ds_bpermute, ~100 cycles a
// stage with its wait), these forms stay in the vector ALU: v_permlane32_swap / v_permlane16_swap for the two widest stages
// (swap with a copy: one register then holds the lower, the other the upper partner of every pair), DPP for the rest
// (row_ror:8; row_shl:4 | row_shr:4 under bank masks; quad_perm).  ~0.55 -> 0.3 us per step of the serial chain.
__device__ __forceinline__ float wave_sum_butterfly(float p) {
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const int pi = __float_as_int(p);
    p = p + __int_as_float(__builtin_amdgcn_update_dpp(pi, pi, 0x128, 0xf, 0xf, false));                 // row_ror:8  = lane ^ 8
    {
        const int qi = __float_as_int(p);
        int t = __builtin_amdgcn_update_dpp(qi, qi, 0x104, 0xf, 0x5, false);                               // row_shl:4: lanes 0-3, 8-11 of a row read lane + 4
        t = __builtin_amdgcn_update_dpp(t, qi, 0x114, 0xf, 0xa, false);                                    // row_shr:4: lanes 4-7, 12-15 read lane - 4
        p = p + __int_as_float(t);
    }
    {
        const int qi = __float_as_int(p);
        p = p + __int_as_float(__builtin_amdgcn_update_dpp(qi, qi, 0x4e, 0xf, 0xf, false));              // quad_perm [2,3,0,1] = lane ^ 2
    }
    {
        const int qi = __float_as_int(p);
        p = p + __int_as_float(__builtin_amdgcn_update_dpp(qi, qi, 0xb1, 0xf, 0xf, false));              // quad_perm [1,0,3,2] = lane ^ 1
    }
    return p;
}

// bisect_left(a, x, 0, n): smallest i with a[i] >= x.  64-ary search: each lane probes the last
// element of its chunk, the ballot tells which chunk holds the answer (3 rounds for n = 10k).
__device__ __forceinline__ uint32_t bisect_left_wave(const uint32_t* __restrict__ a, uint32_t x, uint32_t n, int lane) {
    uint32_t lo = 0;
    while (n > 0) {
        const uint32_t step = (n + 63) >> 6;
        uint32_t idx = lo + (uint32_t)(lane + 1) * step - 1;
        const uint32_t last = lo + n - 1;
        idx = idx < last ? idx : last;
        const bool below = a[idx] < x;
        const uint64_t mask = __ballot(below);
        const uint32_t c = (uint32_t)__popcll(mask);
        if (c == 64) return lo + n;
        const uint32_t nlo = lo + c * step;
        uint32_t chunk_last = nlo + step - 1;
        chunk_last = chunk_last < last ? chunk_last : last;
        lo = nlo;
        n = chunk_last - nlo;     // a[chunk_last] >= x is known
    }
    return lo;
}

template <int EPL>
__global__ __launch_bounds__(256) void d2v_infer_kernel(const float* __restrict__ syn1neg, const uint32_t* __restrict__ cum_table,
                                                        const uint32_t* __restrict__ sample_int, int64_t V, int dim,
                                                        const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ words,
                                                        int64_t ndocs, const float* __restrict__ v0,
                                                        const uint64_t* __restrict__ seeds, int epochs, float alpha0,
                                                        float min_alpha, int negative, double exp_scale,
                                                        const float* __restrict__ exp_table_g, float* __restrict__ out) {
    __shared__ float exp_table[EXP_TABLE_SIZE];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += 256) exp_table[i] = exp_table_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t doc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (doc >= ndocs) return;                       // whole wave exits together
    float v[EPL], work[EPL], rw[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) v[c] = (lane + 64 * c < dim) ? v0[doc * dim + lane + 64 * c] : 0.0f;
    const int64_t wb = doc_ptr[doc], we = doc_ptr[doc + 1];
    const uint64_t seed = seeds[doc];
    const uint32_t cum_last = cum_table[V - 1];
    double alpha = (double)alpha0;
    const double alpha_delta = ((double)alpha0 - (double)min_alpha) / (double)(epochs - 1 > 1 ? epochs - 1 : 1);
    for (int e = 0; e < epochs; ++e) {
        uint64_t next_random = uniform64(splitmix64(seed + (uint64_t)e) & LCG_MOD);
        const float a = (float)alpha;
        for (int64_t i = wb; i < we; ++i) {
            const int32_t w = __builtin_amdgcn_readfirstlane(words[i]);
            if (w < 0 || w >= V) continue;
            if (sample_int) {
                const uint64_t r = next_random >> 16;
                next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                if ((uint64_t)sample_int[w] < r) continue;
            }
#pragma unroll
            for (int c = 0; c < EPL; ++c) work[c] = 0.0f;
            for (int d = 0; d < negative + 1; ++d) {
                uint32_t target;
                float label;
                if (d == 0) {
                    target = (uint32_t)w;
                    label = 1.0f;
                } else {
                    const uint32_t x = (uint32_t)(next_random >> 16) % cum_last;
                    target = __builtin_amdgcn_readfirstlane(bisect_left_wave(cum_table, x, (uint32_t)V, lane));
                    next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                    if (target == (uint32_t)w) continue;
                    label = 0.0f;
                }
                const float* __restrict__ row = syn1neg + (int64_t)target * dim;
                float p = 0.0f;
#pragma unroll
                for (int c = 0; c < EPL; ++c) {
                    rw[c] = (lane + 64 * c < dim) ? row[lane + 64 * c] : 0.0f;
                    p = fmaf(v[c], rw[c], p);
                }
                p = wave_sum_butterfly(p);
                float f = p;
                if (f <= -(float)MAX_EXP || f >= (float)MAX_EXP) continue;
                f = exp_table[(int)((double)(f + (float)MAX_EXP) * exp_scale)];
                const float g = (label - f) * a;
#pragma unroll
                for (int c = 0; c < EPL; ++c) work[c] = fmaf(g, rw[c], work[c]);
            }
#pragma unroll
            for (int c = 0; c < EPL; ++c) v[c] = v[c] + work[c];
        }
        alpha -= alpha_delta;
    }
#pragma unroll
    for (int c = 0; c < EPL; ++c)
        if (lane + 64 * c < dim) out[doc * dim + lane + 64 * c] = v[c];
}

// ---------------------------------------------------------------------------------------------
// PV-DM inference (dm = 1, sum or mean; BASELINE.json's north_star names it, the reference runs dm = 0): gensim's
// train_document_dm with frozen word vectors and hidden layer -- oracle/csrc/oracle.c::orc_d2v_infer_dm is the restatement this
// follows operation for operation.  A wave per document as d2v_infer_kernel; per epoch the kept words (in vocabulary, surviving
// sub-sampling) and their reduced windows go into the wave's LDS lists (written by lane 0 from the scalar LCG walk), then every
// kept position builds l1 = (context word vectors, ascending position, + the document vector) [* 1 / count] and runs the
// (1 + negative) targets against it.  The same lane owns element lane + 64 c of every vector, so sums and the dot product have
// the oracle's order.
// ---------------------------------------------------------------------------------------------
constexpr int DM_CAP = 512;        // kept words of one document (host-checked against the raw length)

template <int EPL>
__global__ __launch_bounds__(256) void d2v_infer_dm_kernel(const float* __restrict__ syn1neg, const float* __restrict__ wv,
                                                           const uint32_t* __restrict__ cum_table, const uint32_t* __restrict__ sample_int,
                                                           int64_t V, int dim, const int64_t* __restrict__ doc_ptr,
                                                           const int32_t* __restrict__ words, int64_t ndocs, const float* __restrict__ v0,
                                                           const uint64_t* __restrict__ seeds, int epochs, float alpha0, float min_alpha,
                                                           int negative, double exp_scale, const float* __restrict__ exp_table_g, int window,
                                                           int dm_mean, float* __restrict__ out) {
    __shared__ float exp_table[EXP_TABLE_SIZE];
    __shared__ int32_t s_kept[4][DM_CAP];
    __shared__ int32_t s_red[4][DM_CAP];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += 256) exp_table[i] = exp_table_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t doc = (int64_t)blockIdx.x * 4 + wid;
    if (doc >= ndocs) return;                       // whole wave exits together
    int32_t* kept = s_kept[wid];
    int32_t* red = s_red[wid];
    float v[EPL], work[EPL], rw[EPL], l1[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) v[c] = (lane + 64 * c < dim) ? v0[doc * dim + lane + 64 * c] : 0.0f;
    const int64_t wb = doc_ptr[doc], we = doc_ptr[doc + 1];
    const uint64_t seed = seeds[doc];
    const uint32_t cum_last = cum_table[V - 1];
    double alpha = (double)alpha0;
    const double alpha_delta = ((double)alpha0 - (double)min_alpha) / (double)(epochs - 1 > 1 ? epochs - 1 : 1);
    for (int e = 0; e < epochs; ++e) {
        uint64_t next_random = uniform64(splitmix64(seed + (uint64_t)e) & LCG_MOD);
        const float a = (float)alpha;
        int n = 0;
        for (int64_t i = wb; i < we; ++i) {
            const int32_t w = __builtin_amdgcn_readfirstlane(words[i]);
            if (w < 0 || w >= V) continue;
            if (sample_int) {
                const uint64_t r = next_random >> 16;
                next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                if ((uint64_t)sample_int[w] < r) continue;
            }
            if (lane == 0) kept[n] = w;
            ++n;
        }
        for (int i = 0; i < n; ++i) {
            if (lane == 0) red[i] = (int32_t)((uint32_t)(next_random >> 16) % (uint32_t)window);
            next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
        }
        // lane 0's LDS writes are read by every lane of this wave below: LDS operations of one wave execute in order, the fence
        // only keeps the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = 0; i < n; ++i) {
            const int b = __builtin_amdgcn_readfirstlane(red[i]);
            int j = i - window + b, k = i + window + 1 - b;
            j = j < 0 ? 0 : j;
            k = k > n ? n : k;
#pragma unroll
            for (int c = 0; c < EPL; ++c) l1[c] = 0.0f;
            float count = 0.0f;
            for (int m = j; m < k; ++m) {
                if (m == i) continue;
                count += 1.0f;
                const int32_t wm = __builtin_amdgcn_readfirstlane(kept[m]);
                const float* __restrict__ row = wv + (int64_t)wm * dim;
#pragma unroll
                for (int c = 0; c < EPL; ++c) l1[c] = l1[c] + ((lane + 64 * c < dim) ? row[lane + 64 * c] : 0.0f);
            }
            count += 1.0f;                                  // the document tag
#pragma unroll
            for (int c = 0; c < EPL; ++c) l1[c] = l1[c] + v[c];
            const float inv_count = 1.0f / count;
            if (dm_mean) {
#pragma unroll
                for (int c = 0; c < EPL; ++c) l1[c] = l1[c] * inv_count;
            }
#pragma unroll
            for (int c = 0; c < EPL; ++c) work[c] = 0.0f;
            const int32_t w = __builtin_amdgcn_readfirstlane(kept[i]);
            for (int d = 0; d < negative + 1; ++d) {
                uint32_t target;
                float label;
                if (d == 0) {
                    target = (uint32_t)w;
                    label = 1.0f;
                } else {
                    const uint32_t x = (uint32_t)(next_random >> 16) % cum_last;
                    target = __builtin_amdgcn_readfirstlane(bisect_left_wave(cum_table, x, (uint32_t)V, lane));
                    next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                    if (target == (uint32_t)w) continue;
                    label = 0.0f;
                }
                const float* __restrict__ row = syn1neg + (int64_t)target * dim;
                float p = 0.0f;
#pragma unroll
                for (int c = 0; c < EPL; ++c) {
                    rw[c] = (lane + 64 * c < dim) ? row[lane + 64 * c] : 0.0f;
                    p = fmaf(l1[c], rw[c], p);
                }
                p = wave_sum_butterfly(p);
                float f = p;
                if (f <= -(float)MAX_EXP || f >= (float)MAX_EXP) continue;
                f = exp_table[(int)((double)(f + (float)MAX_EXP) * exp_scale)];
                const float g = (label - f) * a;
#pragma unroll
                for (int c = 0; c < EPL; ++c) work[c] = fmaf(g, rw[c], work[c]);
            }
            if (!dm_mean) {
#pragma unroll
                for (int c = 0; c < EPL; ++c) work[c] = work[c] * inv_count;
            }
#pragma unroll
            for (int c = 0; c < EPL; ++c) v[c] = v[c] + work[c];
        }
        // the next epoch's list writes must not overtake this epoch's list reads (same wave, in order; compiler fence only)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        alpha -= alpha_delta;
    }
#pragma unroll
    for (int c = 0; c < EPL; ++c)
        if (lane + 64 * c < dim) out[doc * dim + lane + 64 * c] = v[c];
}

// ---------------------------------------------------------------------------------------------
// Planned inference (default).  d2v_infer_kernel above pays, per dot/axpy step, a 64-ary search of cum_table (three dependent
// rounds of loads) and a dependent gather of the target row: ~2 us per step, ~30 ms for one document of 100 epochs -- the
// latency of find_similar_documents' rerank (webui.py:198-199: ten documents re-inferred per query).  But the random stream does
// not depend on the vector being trained: which words survive sub-sampling and which rows the negative samples hit follows
// from the LCG alone.  So per epoch and chunk of words a wave
//   1. walks the LCG (uniform, scalar): one entry per positive target, one raw draw per negative sample, into LDS;
//   2. resolves all draws of the chunk AT ONCE, a lane per draw: modulo, bisect_left over cum_table (its first levels from a
//      1024-entry coarse table in LDS), drops the samples that hit their own word, compacts;
//   3. runs the serial chain over the compacted plan a WORD at a time: the (1 + negative) dot products of a word are
//      independent (the vector changes only after the word), the next word's rows are already requested.
// The arithmetic of a step and its order are those of d2v_infer_kernel: results are bit-identical (same tests).
// ---------------------------------------------------------------------------------------------
constexpr int PLAN_WORDS = 32;            // words per chunk
constexpr int PLAN_MAX_NEG = 5;             // a word's group: 1 positive + up to 5 negative samples (gensim's default negative = 5)
constexpr int PLAN_GROUP = 1 + PLAN_MAX_NEG;
constexpr int PLAN_CAP = PLAN_WORDS * (1 + PLAN_MAX_NEG);     // 32 x 6 = 192 plan entries per wave
constexpr int PLAN_COARSE = 1024;
constexpr uint32_t PLAN_POS = 0x80000000u;

__device__ __forceinline__ uint32_t bisect_left_coarse(const uint32_t* __restrict__ a, const uint32_t* coarse, uint32_t x, uint32_t n, uint32_t stride) {
    // coarse[c] = a[min((c + 1) * stride, n) - 1]: the last element of block c.  Smallest block whose last element is >= x, then
    // the smallest element >= x inside it: exactly bisect_left (a is non-decreasing); x above every element returns n.
    const uint32_t nblk = (n + stride - 1) / stride;
    uint32_t lo = 0, hi = nblk;
    while (hi > lo) {
        const uint32_t mid = (lo + hi) >> 1;
        if (coarse[mid] >= x) hi = mid;
        else lo = mid + 1;
    }
    if (lo == nblk) return n;
    uint32_t b = lo * stride, e = b + stride < n ? b + stride : n;
    while (e > b) {
        const uint32_t mid = (b + e) >> 1;
        if (a[mid] >= x) e = mid;
        else b = mid + 1;
    }
    return b;
}

// Lane-major copy of syn1neg for the planned kernel.  A lane owns elements lane, lane + 64, ... of a row (that is what fixes the bits
// of the dot product), so in the row-major matrix its EPL values are 256 B apart: EPL global_load_dword per row and lane, 30 vector-
// memory instructions per word at negative = 5 -- and the wave's time went into ISSUING them (~50-75 cycles each).  Here a row is
// stored as [64 lanes][4] blocks for c = 0..3, 4..7, ... followed by one [64][tw] block for the remaining EPL % 4 values (tw = 1, 2,
// or 4 with padding for 3): the same values in the same registers from EPL / 4 dwordx4 loads plus at most one more (300-d: 2 loads
// per row instead of 5, 1 280 B per row).
__host__ __device__ inline int lane_tail_width(int epl) { return (epl & 3) == 3 ? 4 : (epl & 3); }
__host__ __device__ inline int lane_row_floats(int epl) { return 64 * (4 * (epl >> 2) + lane_tail_width(epl)); }

__global__ __launch_bounds__(256) void d2v_lane_major_kernel(const float* __restrict__ syn1neg, float* __restrict__ out, int64_t V, int dim, int epl) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= V) return;
    const int n4 = epl >> 2, tw = lane_tail_width(epl);
    float* __restrict__ row = out + w * lane_row_floats(epl);
    const float* __restrict__ src = syn1neg + w * dim;
    for (int q = 0; q < n4; ++q)
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * (4 * q + i);
            row[(q * 64 + lane) * 4 + i] = e < dim ? src[e] : 0.0f;
        }
    for (int i = 0; i < tw; ++i) {
        const int e = lane + 64 * (4 * n4 + i);
        row[n4 * 256 + lane * tw + i] = (4 * n4 + i < epl && e < dim) ? src[e] : 0.0f;
    }
}

template <int EPL>
__global__ __launch_bounds__(256) void d2v_infer_plan_kernel(const float* __restrict__ syn1neg, const uint32_t* __restrict__ cum_table,
                                                             const uint32_t* __restrict__ sample_int, int64_t V, int dim,
                                                             const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ words,
                                                             int64_t ndocs, const float* __restrict__ v0, const uint64_t* __restrict__ seeds,
                                                             int epochs, float alpha0, float min_alpha, int negative, double exp_scale,
                                                             const float* __restrict__ exp_table_g, float* __restrict__ out) {
    __shared__ float exp_table[EXP_TABLE_SIZE];
    __shared__ uint32_t coarse[PLAN_COARSE];
    __shared__ uint32_t plan_raw[4][PLAN_CAP], plan_own[4][PLAN_CAP], plan[4][PLAN_CAP];
    __shared__ int32_t cw[4][PLAN_WORDS];
    __shared__ uint32_t csi[4][PLAN_WORDS];
    const uint32_t stride = (uint32_t)((V + PLAN_COARSE - 1) / PLAN_COARSE);
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += 256) exp_table[i] = exp_table_g[i];
    for (uint32_t c = threadIdx.x; c * stride < (uint32_t)V; c += 256) {
        const uint32_t last = (c + 1) * stride < (uint32_t)V ? (c + 1) * stride : (uint32_t)V;
        coarse[c] = cum_table[last - 1];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t doc = (int64_t)blockIdx.x * 4 + wv;
    if (doc >= ndocs) return;                       // whole wave exits together (no barrier below)
    float v[EPL], work[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) v[c] = (lane + 64 * c < dim) ? v0[doc * dim + lane + 64 * c] : 0.0f;
    const int64_t wb = doc_ptr[doc], we = doc_ptr[doc + 1];
    const uint64_t seed = seeds[doc];
    const uint32_t cum_last = cum_table[V - 1];
    double alpha = (double)alpha0;
    const double alpha_delta = ((double)alpha0 - (double)min_alpha) / (double)(epochs - 1 > 1 ? epochs - 1 : 1);
    uint32_t* praw = plan_raw[wv];
    uint32_t* pown = plan_own[wv];
    uint32_t* pl = plan[wv];
    // (syn1neg here is the LANE-MAJOR copy, d2v_lane_major_kernel: dwordx4 blocks, then the tail block)
    constexpr int N4 = EPL >> 2, TW = (EPL & 3) == 3 ? 4 : (EPL & 3), ROWF = 64 * (4 * N4 + TW);
    auto load_row = [&](uint32_t e, float (&rw)[EPL]) {
        const float* __restrict__ row = syn1neg + (int64_t)(e & ~PLAN_POS) * ROWF;
#pragma unroll
        for (int q = 0; q < N4; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(row + (q * 64 + lane) * 4);
            rw[4 * q] = x.x; rw[4 * q + 1] = x.y; rw[4 * q + 2] = x.z; rw[4 * q + 3] = x.w;
        }
        if constexpr (TW == 1) {
            rw[4 * N4] = row[N4 * 256 + lane];
        } else if constexpr (TW == 2) {
            const float2 x = *reinterpret_cast<const float2*>(row + N4 * 256 + lane * 2);
            rw[4 * N4] = x.x; rw[4 * N4 + 1] = x.y;
        } else if constexpr (TW == 4) {
            const float4 x = *reinterpret_cast<const float4*>(row + N4 * 256 + lane * 4);
            rw[4 * N4] = x.x; rw[4 * N4 + 1] = x.y; rw[4 * N4 + 2] = x.z;
        }
    };
    for (int e = 0; e < epochs; ++e) {
        uint64_t next_random = uniform64(splitmix64(seed + (uint64_t)e) & LCG_MOD);
        const float a = (float)alpha;
        for (int64_t c0 = wb; c0 < we; c0 += PLAN_WORDS) {
            const int nc = (int)(we - c0 < PLAN_WORDS ? we - c0 : PLAN_WORDS);
            // the chunk's words and their sub-sampling thresholds: one coalesced load, then uniform reads from LDS
            if (lane < nc) {
                const int32_t w = words[c0 + lane];
                cw[wv][lane] = w;
                csi[wv][lane] = (sample_int && w >= 0 && w < V) ? sample_int[w] : 0xffffffffu;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-level LDS fence: the phases hand data from lane to lane
            // ---- 1. the random walk (uniform)
            int n = 0;
            for (int i = 0; i < nc; ++i) {
                const int32_t w = __builtin_amdgcn_readfirstlane(cw[wv][i]);
                if (w < 0 || w >= V) continue;
                if (sample_int) {
                    const uint64_t r = next_random >> 16;
                    next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                    if ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(csi[wv][i]) < r) continue;      // (the builtin returns int: no sign extension)
                }
                if (lane == 0) {
                    praw[n] = (uint32_t)w;
                    pown[n] = 0xffffffffu;          // marks a positive target (a raw draw may have any of its 32 bits set)
                }
                for (int d = 1; d <= negative; ++d) {
                    if (lane == 0) {
                        praw[n + d] = (uint32_t)(next_random >> 16);
                        pown[n + d] = (uint32_t)w;
                    }
                    next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                }
                n += 1 + negative;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-level LDS fence: the phases hand data from lane to lane
            // ---- 2. resolve every draw of the chunk, a lane per entry; drop samples that hit their own word; compact in order
            int m = 0;
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                uint32_t entry = 0;
                bool keep = false;
                if (j < n) {
                    const uint32_t r = praw[j], own = pown[j];
                    if (own == 0xffffffffu) {
                        entry = PLAN_POS | r;
                        keep = true;
                    } else {
                        const uint32_t t = bisect_left_coarse(cum_table, coarse, r % cum_last, (uint32_t)V, stride);
                        entry = t;
                        keep = t != own;
                    }
                }
                const uint64_t mask = __ballot(keep);
                if (keep) pl[m + __popcll(mask & ((1ull << lane) - 1))] = entry;
                m += __popcll(mask);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-level LDS fence: the phases hand data from lane to lane
            // ---- 3. the serial chain, a WORD at a time.  The (1 + negative) dot products of a word all use the document vector as it
            // stood before the word (v changes only after it), so they are independent: their rows are requested together, the dots,
            // the cross-lane sums and the sigmoid look-ups of the group overlap in the pipeline, and only the accumulation of `work`
            // keeps the reference's order.  The next word's rows are requested before this word is computed.
            if (m == 0) continue;
            constexpr int G = PLAN_GROUP;
            auto read_group = [&](int j, uint32_t (&ge)[G], int& gs) {
                const uint32_t x = (lane <= G && j + lane < m) ? pl[j + lane] : PLAN_POS;      // past the end counts as "next word"
                const uint64_t rest = __ballot((x & PLAN_POS) != 0) >> 1;                       // entry j itself is a positive target
                const int nxt = rest ? (int)__ffsll((unsigned long long)rest) : G;
                gs = nxt < G ? nxt : G;
#pragma unroll
                for (int g = 0; g < G; ++g) ge[g] = (uint32_t)__builtin_amdgcn_readlane((int)x, g);
            };
            auto load_rows = [&](const uint32_t (&ge)[G], int gs, float (&R)[G][EPL]) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (g < gs) load_row(ge[g], R[g]);
            };
            auto process = [&](int gs, const float (&R)[G][EPL]) {
                float p[G];
#pragma unroll
                for (int g = 0; g < G; ++g) p[g] = 0.0f;
#pragma unroll
                for (int c = 0; c < EPL; ++c)
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (g < gs) p[g] = fmaf(v[c], R[g][c], p[g]);
                float tv[G];
                bool ok[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float f = g < gs ? wave_sum_butterfly(p[g]) : 0.0f;
                    ok[g] = g < gs && !(f <= -(float)MAX_EXP || f >= (float)MAX_EXP);
                    const int idx = ok[g] ? (int)((double)(f + (float)MAX_EXP) * exp_scale) : 0;
                    tv[g] = exp_table[idx];
                }
                float wk[EPL];
#pragma unroll
                for (int c = 0; c < EPL; ++c) wk[c] = 0.0f;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (ok[g]) {
                        const float gg = ((g == 0 ? 1.0f : 0.0f) - tv[g]) * a;
#pragma unroll
                        for (int c = 0; c < EPL; ++c) wk[c] = fmaf(gg, R[g][c], wk[c]);
                    }
#pragma unroll
                for (int c = 0; c < EPL; ++c) v[c] = v[c] + wk[c];
            };
            uint32_t ea[G], eb[G];
            int ga = 0, gb = 0;
            float Ra[G][EPL], Rb[G][EPL];
            int j = 0;
            read_group(0, ea, ga);
            load_rows(ea, ga, Ra);
            for (;;) {
                int jn = j + ga;
                bool more = jn < m;
                if (more) {
                    read_group(jn, eb, gb);
                    load_rows(eb, gb, Rb);
                }
                process(ga, Ra);
                if (!more) break;
                j = jn;
                jn = j + gb;
                more = jn < m;
                if (more) {
                    read_group(jn, ea, ga);
                    load_rows(ea, ga, Ra);
                }
                process(gb, Rb);
                if (!more) break;
                j = jn;
            }
        }
        alpha -= alpha_delta;
    }
#pragma unroll
    for (int c = 0; c < EPL; ++c)
        if (lane + 64 * c < dim) out[doc * dim + lane + 64 * c] = v[c];
}

// ---------------------------------------------------------------------------------------------
// Training (genmodel.py:159-162: Doc2Vec(vector_size=300, window=50, min_count=1, workers=1, dm=0), build_vocab, train(epochs=100)).
// One document's pass = the inference step above PLUS the hidden-layer update syn1neg[target] += g * doc_vector
// (doc2vec_inner.pyx::fast_document_dbow_neg with learn_hidden = 1), the document vector updated in place in the table.
// A job's alpha (word2vec.py::_get_next_alpha): alpha0 - (alpha0 - min_alpha) * (epoch + first_doc_of_job / ndocs) / epochs.
//   SEQUENTIAL (mode 0)  one wavefront walks every document of every epoch in corpus order -- the reference's workers=1
//                        semantics; bit-identical to oracle/csrc/oracle.c::orc_d2v_train.  For small corpora and parity.
//   PARALLEL (mode 1)    a wavefront per document, the corpus in chunks of HIPTS_D2V_CHUNK (default 2048) documents per launch:
//                        documents of a chunk train concurrently against the hidden layer as the chunks before left it, their
//                        own updates of it are float atomic adds (gensim's worker threads race with plain stores, which works
//                        while collisions are rare; with thousands of concurrent documents on a 10 k-word vocabulary plain
//                        stores lose nearly every update -- measured: neighbour purity at chance).  Not reproducible run to
//                        run; its quality is judged downstream (tests/test_gpu_d2v_train.py).
// ---------------------------------------------------------------------------------------------
template <int EPL, bool ATOMIC>
__device__ __forceinline__ void train_document(float* __restrict__ syn1neg, const uint32_t* __restrict__ cum_table,
                                               const uint32_t* __restrict__ sample_int, int64_t V, int dim, const int32_t* __restrict__ words,
                                               int64_t wb, int64_t we, float* __restrict__ vrow, uint64_t lcg0, float a, int negative,
                                               double exp_scale, const float* exp_table, uint32_t cum_last, int lane) {
    float v[EPL], work[EPL], rw[EPL];
#pragma unroll
    for (int c = 0; c < EPL; ++c) v[c] = (lane + 64 * c < dim) ? vrow[lane + 64 * c] : 0.0f;
    uint64_t next_random = uniform64(lcg0);
    for (int64_t i = wb; i < we; ++i) {
        const int32_t w = __builtin_amdgcn_readfirstlane(words[i]);
        if (w < 0 || w >= V) continue;
        if (sample_int) {
            const uint64_t r = next_random >> 16;
            next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
            if ((uint64_t)sample_int[w] < r) continue;
        }
#pragma unroll
        for (int c = 0; c < EPL; ++c) work[c] = 0.0f;
        for (int d = 0; d < negative + 1; ++d) {
            uint32_t target;
            float label;
            if (d == 0) {
                target = (uint32_t)w;
                label = 1.0f;
            } else {
                const uint32_t x = (uint32_t)(next_random >> 16) % cum_last;
                target = __builtin_amdgcn_readfirstlane(bisect_left_wave(cum_table, x, (uint32_t)V, lane));
                next_random = (next_random * 25214903917ULL + 11) & LCG_MOD;
                if (target == (uint32_t)w) continue;
                label = 0.0f;
            }
            float* __restrict__ row = syn1neg + (int64_t)target * dim;
            float p = 0.0f;
#pragma unroll
            for (int c = 0; c < EPL; ++c) {
                rw[c] = (lane + 64 * c < dim) ? row[lane + 64 * c] : 0.0f;
                p = fmaf(v[c], rw[c], p);
            }
            p = wave_sum_butterfly(p);
            float f = p;
            if (f <= -(float)MAX_EXP || f >= (float)MAX_EXP) continue;
            f = exp_table[(int)((double)(f + (float)MAX_EXP) * exp_scale)];
            const float g = (label - f) * a;
#pragma unroll
            for (int c = 0; c < EPL; ++c) {
                work[c] = fmaf(g, rw[c], work[c]);                                   // work += g * syn1neg[target]
                if (lane + 64 * c < dim) {                                           // syn1neg[target] += g * doc   (learn_hidden)
                    if (ATOMIC) unsafeAtomicAdd(&row[lane + 64 * c], g * v[c]);      // concurrent documents: no update is lost
                    else row[lane + 64 * c] = fmaf(g, v[c], rw[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < EPL; ++c) v[c] = v[c] + work[c];
    }
#pragma unroll
    for (int c = 0; c < EPL; ++c)
        if (lane + 64 * c < dim) vrow[lane + 64 * c] = v[c];
}

__device__ __forceinline__ float job_alpha(int epoch, int64_t job_first, int64_t ndocs, int epochs, float alpha0, float min_alpha) {
    const double progress = ((double)epoch + (double)job_first / (double)ndocs) / (double)epochs;
    double al = (double)alpha0 - ((double)alpha0 - (double)min_alpha) * progress;
    if (al < (double)min_alpha) al = (double)min_alpha;
    return (float)al;
}

// mode 1: grid over documents [doc0, doc1) of ONE epoch.  mode 0: one wavefront (grid 1, 64 threads), epochs [epoch0, epoch1).
template <int EPL, bool SEQUENTIAL>
__global__ __launch_bounds__(256) void d2v_train_kernel(float* __restrict__ syn1neg, float* __restrict__ doc_vectors,
                                                        const uint32_t* __restrict__ cum_table, const uint32_t* __restrict__ sample_int,
                                                        int64_t V, int dim, const int64_t* __restrict__ doc_ptr, const int32_t* __restrict__ words,
                                                        const int64_t* __restrict__ job_first, int64_t ndocs, int epoch0, int epoch1, int epochs,
                                                        float alpha0, float min_alpha, int negative, double exp_scale, uint64_t seed,
                                                        const float* __restrict__ exp_table_g, int64_t doc0, int64_t doc1) {
    __shared__ float exp_table[EXP_TABLE_SIZE];
    for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) exp_table[i] = exp_table_g[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t cum_last = cum_table[V - 1];
    if (SEQUENTIAL) {
        if (blockIdx.x != 0 || threadIdx.x >= 64) return;
        for (int e = epoch0; e < epoch1; ++e)
            for (int64_t doc = 0; doc < ndocs; ++doc) {
                const float a = job_alpha(e, job_first[doc], ndocs, epochs, alpha0, min_alpha);
                const uint64_t lcg0 = splitmix64(seed + (uint64_t)e * (uint64_t)ndocs + (uint64_t)doc) & LCG_MOD;
                train_document<EPL, false>(syn1neg, cum_table, sample_int, V, dim, words, doc_ptr[doc], doc_ptr[doc + 1], doc_vectors + doc * dim, lcg0,
                                           a, negative, exp_scale, exp_table, cum_last, lane);
            }
    } else {
        const int64_t doc = doc0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        if (doc >= doc1) return;
        const float a = job_alpha(epoch0, job_first[doc], ndocs, epochs, alpha0, min_alpha);
        const uint64_t lcg0 = splitmix64(seed + (uint64_t)epoch0 * (uint64_t)ndocs + (uint64_t)doc) & LCG_MOD;
        train_document<EPL, true>(syn1neg, cum_table, sample_int, V, dim, words, doc_ptr[doc], doc_ptr[doc + 1], doc_vectors + doc * dim, lcg0, a,
                                  negative, exp_scale, exp_table, cum_last, lane);
    }
}

}  // namespace

extern "C" {

int hipts_d2v_create(const float* syn1neg, const uint32_t* cum_table, const uint32_t* sample_int, int64_t vocab, int dim,
                     int negative, double exp_scale, int device, hipts_d2v_t** out) {
    HIPTS_REQUIRE(syn1neg && cum_table && out && vocab >= 1 && vocab < (1ll << 31), "hipts_d2v_create: bad arguments");
    HIPTS_REQUIRE(dim >= 1 && dim <= 512, "hipts_d2v_create: dim must be in [1, 512]");
    HIPTS_REQUIRE(negative >= 0 && negative <= 64, "hipts_d2v_create: negative out of range");
    HIPTS_REQUIRE(cum_table[vocab - 1] > 0, "hipts_d2v_create: cum_table[-1] must be positive");
    HIPTS_TRY(use_device(device));
    auto* h = new hipts_d2v();
    h->device = device;
    h->V = vocab;
    h->dim = dim;
    h->negative = negative;
    h->exp_scale = exp_scale;
    h->has_sample = sample_int != nullptr;
    // word2vec_inner.pyx: EXP_TABLE[i] = exp((i / 1000 * 2 - 1) * 6); EXP_TABLE[i] /= (EXP_TABLE[i] + 1)   (REAL_t)
    float table[EXP_TABLE_SIZE];
    for (int i = 0; i < EXP_TABLE_SIZE; ++i) {
        const float e = (float)exp((i / (float)EXP_TABLE_SIZE * 2 - 1) * MAX_EXP);
        table[i] = (float)(e / (e + 1));
    }
    int st;
    if ((st = h->syn1neg.alloc((size_t)vocab * dim * 4)) || (st = h->cum_table.alloc((size_t)vocab * 4)) ||
        (st = h->exp_table.alloc(sizeof(table))) || (st = upload(h->syn1neg.p, syn1neg, (size_t)vocab * dim * 4)) ||
        (st = upload(h->cum_table.p, cum_table, (size_t)vocab * 4)) || (st = upload(h->exp_table.p, table, sizeof(table))) ||
        (sample_int && ((st = h->sample_int.alloc((size_t)vocab * 4)) || (st = upload(h->sample_int.p, sample_int, (size_t)vocab * 4))))) {
        delete h;
        return st;
    }
    {
        const int epl = (dim + 63) / 64;
        if ((st = h->syn_lane.alloc((size_t)vocab * lane_row_floats(epl) * 4))) {
            delete h;
            return st;
        }
        d2v_lane_major_kernel<<<ceil_div(vocab, 4), 256>>>(h->syn1neg.as<float>(), h->syn_lane.as<float>(), vocab, dim, epl);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            delete h;
            return set_error(HIPTS_ERR_HIP, "hipts_d2v_create: building the lane-major copy failed");
        }
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_d2v_destroy(hipts_d2v_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_d2v_infer(hipts_d2v_t* h, const int64_t* doc_ptr, const int32_t* words, int64_t ndocs, const float* v0,
                    const uint64_t* seeds, int epochs, float alpha, float min_alpha, float* out, int out_memspace,
                    void* stream) {
    HIPTS_REQUIRE(h && doc_ptr && v0 && seeds && out && ndocs >= 1 && epochs >= 1, "hipts_d2v_infer: bad arguments");
    HIPTS_TRY(use_device(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nw = doc_ptr[ndocs];
    HIPTS_REQUIRE(doc_ptr[0] == 0 && nw >= 0 && (words || nw == 0), "hipts_d2v_infer: bad CSR");
    HIPTS_TRY(h->ws_ptr.reserve((size_t)(ndocs + 1) * 8));
    HIPTS_TRY(h->ws_words.reserve((size_t)nw * 4));
    HIPTS_TRY(h->ws_v0.reserve((size_t)ndocs * h->dim * 4));
    HIPTS_TRY(h->ws_seeds.reserve((size_t)ndocs * 8));
    HIPTS_HIP(hipMemcpyAsync(h->ws_ptr.p, doc_ptr, (size_t)(ndocs + 1) * 8, hipMemcpyHostToDevice, s));
    if (nw) HIPTS_HIP(hipMemcpyAsync(h->ws_words.p, words, (size_t)nw * 4, hipMemcpyHostToDevice, s));
    HIPTS_HIP(hipMemcpyAsync(h->ws_v0.p, v0, (size_t)ndocs * h->dim * 4, hipMemcpyHostToDevice, s));
    HIPTS_HIP(hipMemcpyAsync(h->ws_seeds.p, seeds, (size_t)ndocs * 8, hipMemcpyHostToDevice, s));
    float* out_dev = out;
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(h->ws_out.reserve((size_t)ndocs * h->dim * 4));
        out_dev = h->ws_out.as<float>();
    }
    const int grid = ceil_div(ndocs, 4);
    const int epl = (h->dim + 63) / 64;
    static const bool planned_ok = !(getenv("HIPTS_D2V_PLAN") && strcmp(getenv("HIPTS_D2V_PLAN"), "0") == 0);      // A/B switch
    const bool planned = planned_ok && h->negative <= PLAN_MAX_NEG;
#define D2V_LAUNCH(E)                                                                                                   \
    if (planned)                                                                                                         \
        d2v_infer_plan_kernel<E><<<grid, 256, 0, s>>>(h->syn_lane.as<float>(), h->cum_table.as<uint32_t>(),                \
                                             h->has_sample ? h->sample_int.as<uint32_t>() : nullptr, h->V, h->dim,      \
                                             h->ws_ptr.as<int64_t>(), h->ws_words.as<int32_t>(), ndocs,                 \
                                             h->ws_v0.as<float>(), h->ws_seeds.as<uint64_t>(), epochs, alpha, min_alpha, \
                                             h->negative, h->exp_scale, h->exp_table.as<float>(), out_dev);              \
    else                                                                                                                 \
        d2v_infer_kernel<E><<<grid, 256, 0, s>>>(h->syn1neg.as<float>(), h->cum_table.as<uint32_t>(),                   \
                                             h->has_sample ? h->sample_int.as<uint32_t>() : nullptr, h->V, h->dim,      \
                                             h->ws_ptr.as<int64_t>(), h->ws_words.as<int32_t>(), ndocs,                 \
                                             h->ws_v0.as<float>(), h->ws_seeds.as<uint64_t>(), epochs, alpha, min_alpha, \
                                             h->negative, h->exp_scale, h->exp_table.as<float>(), out_dev)
    switch (epl) {
        case 1: D2V_LAUNCH(1); break;
        case 2: D2V_LAUNCH(2); break;
        case 3: D2V_LAUNCH(3); break;
        case 4: D2V_LAUNCH(4); break;
        case 5: D2V_LAUNCH(5); break;
        case 6: D2V_LAUNCH(6); break;
        case 7: D2V_LAUNCH(7); break;
        default: D2V_LAUNCH(8); break;
    }
#undef D2V_LAUNCH
    HIPTS_LAUNCH_CHECK();
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_HIP(hipMemcpyAsync(out, out_dev, (size_t)ndocs * h->dim * 4, hipMemcpyDeviceToHost, s));
    }
    HIPTS_HIP(hipStreamSynchronize(s));   // staging buffers are reused by the next call
    return HIPTS_OK;
}

int hipts_d2v_set_word_vectors(hipts_d2v_t* h, const float* word_vectors) {
    HIPTS_REQUIRE(h && word_vectors, "hipts_d2v_set_word_vectors: null argument");
    HIPTS_TRY(use_device(h->device));
    HIPTS_TRY(h->word_vectors.alloc((size_t)h->V * h->dim * 4));
    return upload(h->word_vectors.p, word_vectors, (size_t)h->V * h->dim * 4);
}

int hipts_d2v_infer_dm(hipts_d2v_t* h, const int64_t* doc_ptr, const int32_t* words, int64_t ndocs, const float* v0,
                       const uint64_t* seeds, int epochs, float alpha, float min_alpha, int window, int dm_mean, float* out,
                       int out_memspace, void* stream) {
    HIPTS_REQUIRE(h && doc_ptr && v0 && seeds && out && ndocs >= 1 && epochs >= 1, "hipts_d2v_infer_dm: bad arguments");
    HIPTS_REQUIRE(window >= 1, "hipts_d2v_infer_dm: window must be >= 1");
    if (!h->word_vectors.p)
        return set_error(HIPTS_ERR_STATE, "hipts_d2v_infer_dm: the model has no word vectors (hipts_d2v_set_word_vectors)");
    HIPTS_TRY(use_device(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t nw = doc_ptr[ndocs];
    HIPTS_REQUIRE(doc_ptr[0] == 0 && nw >= 0 && (words || nw == 0), "hipts_d2v_infer_dm: bad CSR");
    for (int64_t d = 0; d < ndocs; ++d)
        HIPTS_REQUIRE(doc_ptr[d + 1] >= doc_ptr[d] && doc_ptr[d + 1] - doc_ptr[d] <= DM_CAP,
                      "hipts_d2v_infer_dm: document %lld has %lld words, at most %d are supported", (long long)d,
                      (long long)(doc_ptr[d + 1] - doc_ptr[d]), DM_CAP);
    HIPTS_TRY(h->ws_ptr.reserve((size_t)(ndocs + 1) * 8));
    HIPTS_TRY(h->ws_words.reserve((size_t)(nw ? nw : 1) * 4));
    HIPTS_TRY(h->ws_v0.reserve((size_t)ndocs * h->dim * 4));
    HIPTS_TRY(h->ws_seeds.reserve((size_t)ndocs * 8));
    HIPTS_HIP(hipMemcpyAsync(h->ws_ptr.p, doc_ptr, (size_t)(ndocs + 1) * 8, hipMemcpyHostToDevice, s));
    if (nw) HIPTS_HIP(hipMemcpyAsync(h->ws_words.p, words, (size_t)nw * 4, hipMemcpyHostToDevice, s));
    HIPTS_HIP(hipMemcpyAsync(h->ws_v0.p, v0, (size_t)ndocs * h->dim * 4, hipMemcpyHostToDevice, s));
    HIPTS_HIP(hipMemcpyAsync(h->ws_seeds.p, seeds, (size_t)ndocs * 8, hipMemcpyHostToDevice, s));
    float* out_dev = out;
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(h->ws_out.reserve((size_t)ndocs * h->dim * 4));
        out_dev = h->ws_out.as<float>();
    }
    const int grid = ceil_div(ndocs, 4);
    const int epl = (h->dim + 63) / 64;
#define D2V_DM_LAUNCH(E)                                                                                                              \
    d2v_infer_dm_kernel<E><<<grid, 256, 0, s>>>(h->syn1neg.as<float>(), h->word_vectors.as<float>(), h->cum_table.as<uint32_t>(),         \
                                                h->has_sample ? h->sample_int.as<uint32_t>() : nullptr, h->V, h->dim,                 \
                                                h->ws_ptr.as<int64_t>(), h->ws_words.as<int32_t>(), ndocs, h->ws_v0.as<float>(),       \
                                                h->ws_seeds.as<uint64_t>(), epochs, alpha, min_alpha, h->negative, h->exp_scale,      \
                                                h->exp_table.as<float>(), window, dm_mean ? 1 : 0, out_dev)
    switch (epl) {
        case 1: D2V_DM_LAUNCH(1); break;
        case 2: D2V_DM_LAUNCH(2); break;
        case 3: D2V_DM_LAUNCH(3); break;
        case 4: D2V_DM_LAUNCH(4); break;
        case 5: D2V_DM_LAUNCH(5); break;
        case 6: D2V_DM_LAUNCH(6); break;
        case 7: D2V_DM_LAUNCH(7); break;
        default: D2V_DM_LAUNCH(8); break;
    }
#undef D2V_DM_LAUNCH
    HIPTS_LAUNCH_CHECK();
    if (out_memspace != HIPTS_DEVICE) HIPTS_HIP(hipMemcpyAsync(out, out_dev, (size_t)ndocs * h->dim * 4, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipStreamSynchronize(s));   // staging buffers are reused by the next call
    return HIPTS_OK;
}

int hipts_d2v_train(const uint32_t* cum_table, const uint32_t* sample_int, int64_t vocab, int dim, int negative, double exp_scale,
                    const int64_t* doc_ptr, const int32_t* words, int64_t ndocs, float* doc_vectors, float* syn1neg, int epochs,
                    float alpha, float min_alpha, uint64_t seed, int batch_words, int mode, int device, void* stream) {
    HIPTS_REQUIRE(cum_table && doc_ptr && doc_vectors && syn1neg && vocab >= 1 && vocab < (1ll << 31) && ndocs >= 1 && epochs >= 1,
                  "hipts_d2v_train: bad arguments");
    HIPTS_REQUIRE(dim >= 1 && dim <= 512 && negative >= 0 && negative <= 64 && batch_words >= 1 && (mode == 0 || mode == 1),
                  "hipts_d2v_train: dim in [1, 512], negative in [0, 64], batch_words >= 1, mode 0 (sequential) or 1 (parallel)");
    HIPTS_REQUIRE(cum_table[vocab - 1] > 0, "hipts_d2v_train: cum_table[-1] must be positive");
    const int64_t nw = doc_ptr[ndocs];
    HIPTS_REQUIRE(doc_ptr[0] == 0 && nw >= 0 && (words || nw == 0), "hipts_d2v_train: bad CSR");
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    // jobs (word2vec.py::_job_producer): consecutive documents while their raw word counts fit batch_words
    std::vector<int64_t> job_first((size_t)ndocs);
    {
        int64_t first = 0, jw = 0;
        for (int64_t d = 0; d < ndocs; ++d) {
            const int64_t n = doc_ptr[d + 1] - doc_ptr[d];
            if (d == 0 || jw + n > batch_words) {
                first = d;
                jw = 0;
            }
            jw += n;
            job_first[(size_t)d] = first;
        }
    }
    float table[EXP_TABLE_SIZE];
    for (int i = 0; i < EXP_TABLE_SIZE; ++i) {
        const float e = (float)exp((i / (float)EXP_TABLE_SIZE * 2 - 1) * MAX_EXP);
        table[i] = (float)(e / (e + 1));
    }
    DevBuf d_syn, d_dv, d_cum, d_si, d_ptr, d_words, d_job, d_exp;
    HIPTS_TRY(d_syn.alloc((size_t)vocab * dim * 4));
    HIPTS_TRY(d_dv.alloc((size_t)ndocs * dim * 4));
    HIPTS_TRY(d_cum.alloc((size_t)vocab * 4));
    HIPTS_TRY(d_ptr.alloc((size_t)(ndocs + 1) * 8));
    HIPTS_TRY(d_words.alloc((size_t)(nw > 0 ? nw : 1) * 4));
    HIPTS_TRY(d_job.alloc((size_t)ndocs * 8));
    HIPTS_TRY(d_exp.alloc(sizeof(table)));
    HIPTS_TRY(upload(d_syn.p, syn1neg, (size_t)vocab * dim * 4, s));
    HIPTS_TRY(upload(d_dv.p, doc_vectors, (size_t)ndocs * dim * 4, s));
    HIPTS_TRY(upload(d_cum.p, cum_table, (size_t)vocab * 4, s));
    HIPTS_TRY(upload(d_ptr.p, doc_ptr, (size_t)(ndocs + 1) * 8, s));
    if (nw) HIPTS_TRY(upload(d_words.p, words, (size_t)nw * 4, s));
    HIPTS_TRY(upload(d_job.p, job_first.data(), (size_t)ndocs * 8, s));
    HIPTS_TRY(upload(d_exp.p, table, sizeof(table), s));
    if (sample_int) {
        HIPTS_TRY(d_si.alloc((size_t)vocab * 4));
        HIPTS_TRY(upload(d_si.p, sample_int, (size_t)vocab * 4, s));
    }
    const int epl = (dim + 63) / 64;
#define D2V_TRAIN(E, SEQ, GRID, THREADS, E0, E1, D0, D1)                                                                                              \
    d2v_train_kernel<E, SEQ><<<GRID, THREADS, 0, s>>>(d_syn.as<float>(), d_dv.as<float>(), d_cum.as<uint32_t>(),                               \
                                                      sample_int ? d_si.as<uint32_t>() : nullptr, vocab, dim, d_ptr.as<int64_t>(),              \
                                                      d_words.as<int32_t>(), d_job.as<int64_t>(), ndocs, E0, E1, epochs, alpha, min_alpha,      \
                                                      negative, exp_scale, seed, d_exp.as<float>(), D0, D1)
#define D2V_TRAIN_EPL(SEQ, GRID, THREADS, E0, E1, D0, D1)                 \
    switch (epl) {                                                \
        case 1: D2V_TRAIN(1, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        case 2: D2V_TRAIN(2, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        case 3: D2V_TRAIN(3, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        case 4: D2V_TRAIN(4, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        case 5: D2V_TRAIN(5, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        case 6: D2V_TRAIN(6, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        case 7: D2V_TRAIN(7, SEQ, GRID, THREADS, E0, E1, D0, D1); break;  \
        default: D2V_TRAIN(8, SEQ, GRID, THREADS, E0, E1, D0, D1); break; \
    }
    if (mode == 0) {
        // one wavefront; a launch per epoch keeps each kernel bounded (a watchdog-safe few seconds even for mid-sized corpora)
        for (int e = 0; e < epochs; ++e) {
            D2V_TRAIN_EPL(true, 1, 64, e, e + 1, 0, ndocs);
            HIPTS_LAUNCH_CHECK();
        }
    } else {
        static const int64_t chunk = getenv("HIPTS_D2V_CHUNK") && atoll(getenv("HIPTS_D2V_CHUNK")) > 0 ? atoll(getenv("HIPTS_D2V_CHUNK")) : 2048;
        for (int e = 0; e < epochs; ++e)
            for (int64_t d0 = 0; d0 < ndocs; d0 += chunk) {
                const int64_t d1 = d0 + chunk < ndocs ? d0 + chunk : ndocs;
                D2V_TRAIN_EPL(false, ceil_div(d1 - d0, 4), 256, e, e + 1, d0, d1);
                HIPTS_LAUNCH_CHECK();
            }
    }
#undef D2V_TRAIN_EPL
#undef D2V_TRAIN
    HIPTS_HIP(hipMemcpyAsync(syn1neg, d_syn.p, (size_t)vocab * dim * 4, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipMemcpyAsync(doc_vectors, d_dv.p, (size_t)ndocs * dim * 4, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipStreamSynchronize(s));
    return HIPTS_OK;
}

}  // extern "C"
