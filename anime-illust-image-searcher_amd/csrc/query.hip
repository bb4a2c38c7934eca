// query.hip -- query-scoring path: BM25, dense similarity index, combine, top-k, fused search.
//
// Reference behaviour followed (file:line relative to the reference repository):
//   BM25 statistics      genmodel.py:51-99
//   BM25 scoring         webui.py:119-172
//   index[query]         webui.py:205,352   (gensim Similarity / MatrixSimilarity, dense float32)
//   normalise + combine  webui.py:377-383
//   ranking              webui.py:191-192   (stable sort by -score: ties keep ascending doc id)
//
// All of it is HBM/L2-bound streaming work: coalesced reads, wave reductions,
// no GEMM reshaping except the index product, which uses the exact-f32 MFMA because its fixed
// k-ordered fma chain makes the float32 result bit-reproducible (and equal to the CPU oracle).
// This file is compiled with -ffp-contract=off: every float64 expression below evaluates in
// numpy's operation order with one rounding per operation.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "../../include/hip_tagsearch_debug.h"

using namespace hipts;

// =============================================================================================
// BM25
// =============================================================================================
struct hipts_bm25 {
    int device = 0;
    int64_t D = 0, nnz = 0;
    int32_t V = 0;
    double avgdl = 0.0;
    std::vector<int64_t> h_ptr, h_dl, h_df;
    std::vector<int32_t> h_term, h_tf;
    std::vector<double> h_idf;
    DevBuf d_ptr, d_term, d_tf, d_dl, d_idf;   // int64[D+1], int32[nnz], int32[nnz], int32[D], double[V]
    DevBuf d_tptr, d_tdoc, d_ttf;              // term-major postings: int64[V+1], int32[nnz], int32[nnz]
    DevBuf d_tslice;                           // int64[V][BM25_SLICES + 1]: a term's postings cut at fixed document boundaries (bm25_postings_sliced_kernel)
    DevBuf ws_maxpart;                         // [queries][slices] partial row maxima of that kernel
    int64_t slice_docs = 0;                    // documents per slice (a multiple of 8); 0: the index is too small to be sliced
    DevBuf ws_q, ws_scores, ws_sims, ws_max, ws_final, ws_mark, ws_out, ws_parts;
    PinBuf pin_in, pin_out;                    // hipts_search: one H2D of the packed queries, one D2H of the packed results
    // hipts_search_submit / _collect (round 4): two batches in flight -- the host packs and launches batch i + 1 while the device runs batch i.
    // The device workspaces are shared (the batches run on one stream, in order); only the pinned buffers exist per slot.
    struct SearchSlot {
        PinBuf pin_in, pin_out;
        hipEvent_t done = nullptr;
        int nq = 0, k = 0, kk = 0;
        hipStream_t stream = nullptr;                   // the stream of the pending batch (both slots share the device workspaces: one stream)
        bool pending = false, host_result = false;      // host_result: the one-query path ran synchronously, its results wait in ids1 / vals1
        std::vector<int32_t> ids1;
        std::vector<double> vals1;
    } slot[2];
    DevBuf s1_state;                           // one-query path: Search1State + group maxima + per-workgroup candidate slots
    uint32_t s1_seq = 0;                       // sequence number of the last one-query call (completion flag in pinned memory)
    bool s1_dirty = true;                      // s1_state may hold leftovers (first use, or a call that failed half way)
    // batched hipts_search (round 4): BM25 postings on a side stream beside the index product -- the two do not depend on each other
    hipStream_t side = nullptr;
    hipEvent_t ev_in = nullptr, ev_bm25 = nullptr;
    // per-kernel HIP-event timing (hipts_query_profile_*): events on the stream each kernel is launched on
    bool prof = false;
    struct ProfRec { int cat; hipEvent_t a, b; double bytes; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[HIPTS_QUERY_PROF_CATEGORIES] = {0}, prof_bytes[HIPTS_QUERY_PROF_CATEGORIES] = {0};
    int64_t prof_n[HIPTS_QUERY_PROF_CATEGORIES] = {0};
};

namespace {

constexpr double BM25_K1 = 1.5;    // webui.py:126
constexpr double BM25_B = 0.75;    // webui.py:127
constexpr double REQUIRE_MAGIC = 1000.0;   // webui.py:60

// One thread per (document, query).  Document-major CSR: the ~20 (term, tf) pairs of a document
// are contiguous, consecutive threads own consecutive documents, so a wave sweeps one contiguous
// span of the postings arrays.
__global__ __launch_bounds__(256) void bm25_score_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ term,
                                                         const int32_t* __restrict__ tf, const int32_t* __restrict__ dl,
                                                         const double* __restrict__ idf, int32_t V, double avgdl, int64_t D,
                                                         const int32_t* __restrict__ q_terms, const double* __restrict__ q_weights,
                                                         const int32_t* __restrict__ q_ptr, double* __restrict__ out) {
    const int q = blockIdx.y;
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int qb = q_ptr[q], qe = q_ptr[q + 1];
    const int64_t b = ptr[d], e = ptr[d + 1];
    // webui.py:144  k1 * (1 - b + b * (dl / avgdl))   [(1-b) folds to 0.25 exactly]
    const double dlv = (double)dl[d];
    const double nrm = BM25_K1 * ((1.0 - BM25_B) + BM25_B * (dlv / avgdl));
    double s = 0.0;
    bool masked = false;
    for (int j = qb; j < qe; ++j) {
        const int32_t t = q_terms[j];
        const double w = q_weights[j];
        int32_t tfv = 0;
        for (int64_t i = b; i < e; ++i)
            if (term[i] == t) tfv = tf[i];
        const double idf_t = (t >= 0 && t < V) ? idf[t] : 0.0;   // bm25_idf.get(term_id, 0)   :140
        const double tfd = (double)tfv;
        const double denom = tfd + nrm;                          // :144
        const double numer = tfd * (BM25_K1 + 1.0);              // :145
        const double sc = idf_t * (numer / denom);               // :146
        if (w < 0.0) {                                           // :154-160 exclude
            if (tfv > 0) masked = true;
        } else if (w > REQUIRE_MAGIC) {                          // :161-168 required
            s += (w - REQUIRE_MAGIC) * sc;
            if (tfv == 0) masked = true;
        } else {                                                 // :169-170
            s += w * sc;
        }
    }
    out[(int64_t)q * D + d] = masked ? -INFINITY : s;
}

// Postings form of the same arithmetic (default): one workgroup per query walks the posting list
// of each query term IN THE QUERY'S TERM ORDER (a barrier between terms keeps the float64 additions
// of one document in the reference's order).  Documents that do not contain a term would add
// idf * (0 / denom) = +0.0, which leaves a non-negative score unchanged, so skipping them is exact.
// Reads sum(df) * 8 B + a few passes over D instead of the whole corpus per query.
// mark[d]: bit 7 = an excluded term is present, low bits = number of required terms present.
#ifdef HIPTS_X_TOPK_STAMPS
__device__ unsigned long long g_bm25_stamps[8];
#define BM25_STAMP(i) do { __syncthreads(); if (blockIdx.x == HIPTS_X_TOPK_STAMPS && threadIdx.x == 0) g_bm25_stamps[i] = wall_clock64(); } while (0)
#else
#define BM25_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void bm25_postings_kernel(const int64_t* __restrict__ tptr, const int32_t* __restrict__ tdoc,
                                                             const int32_t* __restrict__ ttf, const int32_t* __restrict__ dl,
                                                             const double* __restrict__ idf, int32_t V, double avgdl, int64_t D,
                                                             const int32_t* __restrict__ q_terms, const double* __restrict__ q_weights,
                                                             const int32_t* __restrict__ q_ptr, double* __restrict__ out,
                                                             uint8_t* __restrict__ mark_all, double* __restrict__ max_out) {
    const int q = blockIdx.x, tid = threadIdx.x;
    const int qb = q_ptr[q], qe = q_ptr[q + 1];
    double* __restrict__ scores = out + (int64_t)q * D;
    uint8_t* __restrict__ mark = mark_all + (int64_t)q * D;
    int n_required = 0;
    bool masking = false;
    for (int j = qb; j < qe; ++j) {
        const double w = q_weights[j];
        if (w > REQUIRE_MAGIC) ++n_required;
        if (w > REQUIRE_MAGIC || w < 0.0) masking = true;
    }
    BM25_STAMP(0);
    // the query's term metadata (posting range, idf) is requested before the row is cleared: per term the dependent chain is then
    // posting -> {document length, score} instead of term -> range -> posting -> ... (~2 us per term of a workgroup's 60-135 us)
    constexpr int TPRE = 8;
    int64_t t_b[TPRE], t_e[TPRE];
    double t_idf[TPRE];
#pragma unroll
    for (int u = 0; u < TPRE; ++u) {
        const int j = qb + u;
        const int32_t t = j < qe ? q_terms[j] : -1;
        const bool ok = t >= 0 && t < V;
        t_b[u] = ok ? tptr[t] : 0;
        t_e[u] = ok ? tptr[t + 1] : 0;
        t_idf[u] = ok ? idf[t] : 0.0;
    }
    // 8 documents per thread and step where the row allows it (64 B of scores, 8 B of marks per thread: the byte-wide mark accesses
    // made the masked rows' last pass 65-72 us against 15-29 us for unmasked ones -- tools/bm25_stamps.py)
    const bool vec8 = (D & 7) == 0;
    if (vec8) {
        // lane-contiguous 16-byte stores (1 KB per wave instruction; 64 B per lane at a 64 B stride wrote every line in four pieces: 59 us)
        const double2 z2 = make_double2(0.0, 0.0);
        for (int64_t g = tid; g < (D >> 1); g += 1024) reinterpret_cast<double2*>(scores)[g] = z2;
        if (masking)
            for (int64_t g = tid; g < (D >> 3); g += 1024) reinterpret_cast<uint64_t*>(mark)[g] = 0ull;
    } else {
#pragma unroll 8
        for (int64_t d = tid; d < D; d += 1024) {
            scores[d] = 0.0;
            if (masking) mark[d] = 0;
        }
    }
    __syncthreads();
    BM25_STAMP(1);
    for (int j = qb; j < qe; ++j) {
        const int32_t t = q_terms[j];
        const double w = q_weights[j];
        if (t >= 0 && t < V) {
            double idf_t;
            int64_t b, e;
            const int u = j - qb;
            if (u < TPRE) {                       // uniform
                idf_t = t_idf[0]; b = t_b[0]; e = t_e[0];
#pragma unroll
                for (int x = 1; x < TPRE; ++x)
                    if (u == x) { idf_t = t_idf[x]; b = t_b[x]; e = t_e[x]; }
            } else {
                idf_t = idf[t]; b = tptr[t]; e = tptr[t + 1];
            }
            // Eight postings per thread in flight: the documents of one term are distinct, but the compiler cannot know that a
            // load of scores[d'] may pass the store to scores[d], so an unrolled loop still ran its read-modify-writes one after
            // the other (~6 us per 1024 x 4 postings); here the gathers of a batch are all issued before the first store.
            constexpr int PU = 8;
            const bool req = w > REQUIRE_MAGIC;
            const double ww = req ? (w - REQUIRE_MAGIC) : w;
            for (int64_t i0 = b + tid; i0 < e; i0 += (int64_t)PU * 1024) {
                int32_t dd[PU];
                double tfd[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int64_t i = i0 + (int64_t)u * 1024;
                    dd[u] = i < e ? tdoc[i] : -1;
                    tfd[u] = i < e ? (double)ttf[i] : 0.0;
                }
                if (w < 0.0) {
#pragma unroll
                    for (int u = 0; u < PU; ++u)
                        if (dd[u] >= 0) mark[dd[u]] |= 0x80;
                } else {
                    double dlv[PU], sv[PU];
                    uint8_t mk[PU];
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        dlv[u] = dd[u] >= 0 ? (double)dl[dd[u]] : 0.0;
                        sv[u] = dd[u] >= 0 ? scores[dd[u]] : 0.0;
                        mk[u] = (req && dd[u] >= 0) ? mark[dd[u]] : (uint8_t)0;
                    }
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        if (dd[u] < 0) continue;
                        const double nrm = BM25_K1 * ((1.0 - BM25_B) + BM25_B * (dlv[u] / avgdl));
                        const double sc = idf_t * ((tfd[u] * (BM25_K1 + 1.0)) / (tfd[u] + nrm));
                        scores[dd[u]] = sv[u] + ww * sc;
                        if (req) mark[dd[u]] = (uint8_t)(mk[u] + 1);
                    }
                }
            }
        }
        __syncthreads();
    }
    BM25_STAMP(2);
    double mx = -INFINITY;
    if (masking && vec8) {
        // two documents per lane and access (16 B of scores, 2 B of marks), four accesses in flight; every wave instruction covers a contiguous KB
        const int64_t pairs = D >> 1;
        for (int64_t g0 = tid; g0 < pairs; g0 += 4 * 1024) {
            double2 v[4];
            uint32_t m2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * 1024;
                v[u] = g < pairs ? reinterpret_cast<const double2*>(scores)[g] : make_double2(-INFINITY, -INFINITY);
                m2[u] = g < pairs ? reinterpret_cast<const uint16_t*>(mark)[g] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * 1024;
                if (g >= pairs) continue;
                const uint32_t ma = m2[u] & 0xffu, mb_ = m2[u] >> 8;
                const bool ka = (ma & 0x80u) || (int)(ma & 0x7fu) != n_required, kb = (mb_ & 0x80u) || (int)(mb_ & 0x7fu) != n_required;
                if (ka) v[u].x = -INFINITY;
                if (kb) v[u].y = -INFINITY;
                if (ka || kb) reinterpret_cast<double2*>(scores)[g] = v[u];
                mx = fmax(mx, fmax(v[u].x, v[u].y));
            }
        }
    } else if (masking) {
#pragma unroll 8
        for (int64_t d = tid; d < D; d += 1024) {
            const uint8_t m = mark[d];
            double v = scores[d];
            if ((m & 0x80) || (m & 0x7f) != n_required) scores[d] = v = -INFINITY;
            mx = fmax(mx, v);
        }
    } else if (max_out && vec8) {
        const int64_t pairs = D >> 1;
#pragma unroll 4
        for (int64_t g = tid; g < pairs; g += 1024) {
            const double2 a0 = reinterpret_cast<const double2*>(scores)[g];
            mx = fmax(mx, fmax(a0.x, a0.y));
        }
    } else if (max_out) {
#pragma unroll 8
        for (int64_t d = tid; d < D; d += 1024) mx = fmax(mx, scores[d]);
    }
    BM25_STAMP(3);
    if (max_out) {      // row maximum for the normalisation of webui.py:379-380, fused here
        __shared__ double part[16];
        for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) part[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            double m = part[0];
            for (int w = 1; w < 16; ++w) m = fmax(m, part[w]);
            max_out[q] = m;
        }
    }
}

// The same walk with the document range cut into slices (round 4).  One workgroup per query is one workgroup per CU for a batch of 256:
// 16 waves that clear, scatter into and re-read an 800 KB row at ~15 GB/s -- the kernel took 166 us for 620 MB, most of it latency.  A
// document's float64 additions only have to keep the QUERY'S TERM ORDER, and documents are independent, so a workgroup may take any
// document range: grid (query, part), each workgroup the postings of its range -- found through a table of every term's posting offsets
// at BM25_SLICES fixed document boundaries, built with the index -- and a partial row maximum; four 512-thread workgroups share a CU.
// Same instructions per document as bm25_postings_kernel: the same bits.
constexpr int BM25_SLICES = 8;
__global__ __launch_bounds__(512) void bm25_postings_sliced_kernel(const int64_t* __restrict__ tslice, const int32_t* __restrict__ tdoc,
                                                                   const int32_t* __restrict__ ttf, const int32_t* __restrict__ dl,
                                                                   const double* __restrict__ idf, int32_t V, double avgdl, int64_t D, int64_t slice_docs,
                                                                   int per_part, const int32_t* __restrict__ q_terms,
                                                                   const double* __restrict__ q_weights, const int32_t* __restrict__ q_ptr,
                                                                   double* __restrict__ out, uint8_t* __restrict__ mark_all, double* __restrict__ max_part) {
    constexpr int NT = 512;
    const int q = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int s0 = part * per_part, s1 = s0 + per_part;                       // fixed slices [s0, s1)
    const int64_t lo = (int64_t)s0 * slice_docs, hi = s1 == BM25_SLICES ? D : (int64_t)s1 * slice_docs;      // lo, hi multiples of 8 (D % 8 == 0)
    const int qb = q_ptr[q], qe = q_ptr[q + 1];
    double* __restrict__ scores = out + (int64_t)q * D;
    uint8_t* __restrict__ mark = mark_all + (int64_t)q * D;
    int n_required = 0;
    bool masking = false;
    for (int j = qb; j < qe; ++j) {
        const double w = q_weights[j];
        if (w > REQUIRE_MAGIC) ++n_required;
        if (w > REQUIRE_MAGIC || w < 0.0) masking = true;
    }
    constexpr int TPRE = 8;
    int64_t t_b[TPRE], t_e[TPRE];
    double t_idf[TPRE];
#pragma unroll
    for (int u = 0; u < TPRE; ++u) {
        const int j = qb + u;
        const int32_t t = j < qe ? q_terms[j] : -1;
        const bool ok = t >= 0 && t < V;
        t_b[u] = ok ? tslice[(int64_t)t * (BM25_SLICES + 1) + s0] : 0;
        t_e[u] = ok ? tslice[(int64_t)t * (BM25_SLICES + 1) + s1] : 0;
        t_idf[u] = ok ? idf[t] : 0.0;
    }
    {
        const double2 z2 = make_double2(0.0, 0.0);
        for (int64_t g = (lo >> 1) + tid; g < (hi >> 1); g += NT) reinterpret_cast<double2*>(scores)[g] = z2;
        if (masking)
            for (int64_t g = (lo >> 3) + tid; g < (hi >> 3); g += NT) reinterpret_cast<uint64_t*>(mark)[g] = 0ull;
    }
    __syncthreads();
    for (int j = qb; j < qe; ++j) {
        const int32_t t = q_terms[j];
        const double w = q_weights[j];
        if (t >= 0 && t < V) {
            double idf_t;
            int64_t b, e;
            const int u = j - qb;
            if (u < TPRE) {                       // uniform
                idf_t = t_idf[0]; b = t_b[0]; e = t_e[0];
#pragma unroll
                for (int x = 1; x < TPRE; ++x)
                    if (u == x) { idf_t = t_idf[x]; b = t_b[x]; e = t_e[x]; }
            } else {
                idf_t = idf[t]; b = tslice[(int64_t)t * (BM25_SLICES + 1) + s0]; e = tslice[(int64_t)t * (BM25_SLICES + 1) + s1];
            }
            constexpr int PU = 8;
            const bool req = w > REQUIRE_MAGIC;
            const double ww = req ? (w - REQUIRE_MAGIC) : w;
            for (int64_t i0 = b + tid; i0 < e; i0 += (int64_t)PU * NT) {
                int32_t dd[PU];
                double tfd[PU];
#pragma unroll
                for (int u2 = 0; u2 < PU; ++u2) {
                    const int64_t i = i0 + (int64_t)u2 * NT;
                    dd[u2] = i < e ? tdoc[i] : -1;
                    tfd[u2] = i < e ? (double)ttf[i] : 0.0;
                }
                if (w < 0.0) {
#pragma unroll
                    for (int u2 = 0; u2 < PU; ++u2)
                        if (dd[u2] >= 0) mark[dd[u2]] |= 0x80;
                } else {
                    double dlv[PU], sv[PU];
                    uint8_t mk[PU];
#pragma unroll
                    for (int u2 = 0; u2 < PU; ++u2) {
                        dlv[u2] = dd[u2] >= 0 ? (double)dl[dd[u2]] : 0.0;
                        sv[u2] = dd[u2] >= 0 ? scores[dd[u2]] : 0.0;
                        mk[u2] = (req && dd[u2] >= 0) ? mark[dd[u2]] : (uint8_t)0;
                    }
#pragma unroll
                    for (int u2 = 0; u2 < PU; ++u2) {
                        if (dd[u2] < 0) continue;
                        const double nrm = BM25_K1 * ((1.0 - BM25_B) + BM25_B * (dlv[u2] / avgdl));
                        const double sc = idf_t * ((tfd[u2] * (BM25_K1 + 1.0)) / (tfd[u2] + nrm));
                        scores[dd[u2]] = sv[u2] + ww * sc;
                        if (req) mark[dd[u2]] = (uint8_t)(mk[u2] + 1);
                    }
                }
            }
        }
        __syncthreads();
    }
    double mx = -INFINITY;
    const int64_t p0 = lo >> 1, p1 = hi >> 1;
    if (masking) {
        for (int64_t g0 = p0 + tid; g0 < p1; g0 += 4 * NT) {
            double2 v[4];
            uint32_t m2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * NT;
                v[u] = g < p1 ? reinterpret_cast<const double2*>(scores)[g] : make_double2(-INFINITY, -INFINITY);
                m2[u] = g < p1 ? reinterpret_cast<const uint16_t*>(mark)[g] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t g = g0 + u * NT;
                if (g >= p1) continue;
                const uint32_t ma = m2[u] & 0xffu, mb_ = m2[u] >> 8;
                const bool ka = (ma & 0x80u) || (int)(ma & 0x7fu) != n_required, kb = (mb_ & 0x80u) || (int)(mb_ & 0x7fu) != n_required;
                if (ka) v[u].x = -INFINITY;
                if (kb) v[u].y = -INFINITY;
                if (ka || kb) reinterpret_cast<double2*>(scores)[g] = v[u];
                mx = fmax(mx, fmax(v[u].x, v[u].y));
            }
        }
    } else if (max_part) {
#pragma unroll 4
        for (int64_t g = p0 + tid; g < p1; g += NT) {
            const double2 a0 = reinterpret_cast<const double2*>(scores)[g];
            mx = fmax(mx, fmax(a0.x, a0.y));
        }
    }
    if (max_part) {
        __shared__ double part_s[NT / 64];
        for (int o = 32; o >= 1; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) part_s[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            double m = part_s[0];
            for (int w = 1; w < NT / 64; ++w) m = fmax(m, part_s[w]);
            max_part[(int64_t)q * gridDim.y + part] = m;
        }
    }
}
__global__ void bm25_max_reduce_kernel(const double* __restrict__ parts, int nparts, double* __restrict__ max_out, int nq) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    double m = parts[(int64_t)q * nparts];
    for (int p = 1; p < nparts; ++p) m = fmax(m, parts[(int64_t)q * nparts + p]);
    max_out[q] = m;
}

// ---------------------------------------------------------------------------------------------
// row maxima (numpy .max(): NaN-free inputs assumed)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void rowmax_kernel(const T* __restrict__ v, int64_t n, T* __restrict__ out) {
    const T* row = v + (int64_t)blockIdx.x * n;
    T m = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmax(m, row[i]);
    for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    __shared__ T part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) {
        T x = threadIdx.x < 16 ? part[threadIdx.x] : (T)-INFINITY;
        for (int o = 8; o >= 1; o >>= 1) x = fmax(x, __shfl_xor(x, o));
        if (threadIdx.x == 0) out[blockIdx.x] = x;
    }
}

// order-preserving u32 image of a float (for atomicMax); 0 is below every real value
__device__ __forceinline__ uint32_t float_order_key(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_from_key(uint32_t k) {
    return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
}

// webui.py:377-383 (and :208 with norm flags off):
//   out = wa * (a / max_a) + (double)((float)wb * (b / max_b))
__global__ __launch_bounds__(256) void combine_kernel(const double* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                      double wa, float wb, const double* __restrict__ max_a,
                                                      const float* __restrict__ max_b, const uint32_t* __restrict__ max_b_keys,
                                                      double* __restrict__ out) {
    const int q = blockIdx.y;
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n) return;
    double A = a[(int64_t)q * n + d];
    float B = b[(int64_t)q * n + d];
    if (max_a) {
        const double m = max_a[q];
        if (m > 0.0) A = A / m;
    }
    if (max_b || max_b_keys) {
        const float m = max_b ? max_b[q] : float_from_key(max_b_keys[q]);
        if (m > 0.0f) B = B / m;
    }
    const float wB = wb * B;                      // python float * float32 array stays float32
    out[(int64_t)q * n + d] = wa * A + (double)wB;
}

// =============================================================================================
// Dense similarity: scores[q][d] = chain_k fma(row[d][k], query[q][k], acc)
// v_mfma_f32_32x32x2_f32: A = 32 queries x 2 k, B = 2 k x 32 documents, D[query][document].
// Per wave: a 32-document tile, the whole K; up to 32 queries per pass for the price of one
// (the kernel is HBM-bound: 32 x 1200 B per 150 MFMAs per wave).
// =============================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));


// TILED: `index` is the tile-major copy of the rows, float4 tl[((T * K/4 + kq) * 32 + c)] = row[32 T + c][4 kq .. 4 kq + 3]:
// the 32 lanes of a half-wave read 512 contiguous bytes per load.  With the plain row-major matrix every lane
// walks its own 1200 B row -- 32 different cache lines per load instruction, each revisited four times -- and the
// kernel ran at 2 TB/s of algorithmic traffic (61 us per pass over 100k x 300); K % 4 != 0 keeps that path.
constexpr int SIM_THREADS = 512;     // 8 waves share one LDS query tile: 3 workgroups = 6 dependent MFMA chains per SIMD
template <bool TILED>
__global__ __launch_bounds__(SIM_THREADS) void sim_mfma_kernel(const float* __restrict__ index, int64_t D, int K, int64_t ld,
                                                       const float* __restrict__ q, int nq, float* __restrict__ out,
                                                       int64_t out_ld) {
    extern __shared__ __attribute__((aligned(16))) float qT[];   // [K][32]: qT[k*32 + j] = q[j][k]
    const int tid = threadIdx.x;
    for (int i = tid; i < K * 32; i += SIM_THREADS) {
        const int k = i >> 5, j = i & 31;
        qT[i] = j < nq ? q[(int64_t)j * K + k] : 0.0f;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (D + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * (SIM_THREADS / 64) + wave; tile < ntiles; tile += (int64_t)gridDim.x * (SIM_THREADS / 64)) {
        int64_t doc = tile * 32 + r;
        const int64_t docc = doc < D ? doc : D - 1;
        const float* __restrict__ row = index + docc * ld;
        const float4* __restrict__ trow = reinterpret_cast<const float4*>(index) + tile * (K >> 2) * 32 + r;    // TILED
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        int k = 0;
#pragma unroll 2
        for (; k + 8 <= K; k += 8) {
            const float4 v0 = TILED ? trow[(k >> 2) * 32] : *reinterpret_cast<const float4*>(row + k);
            const float4 v1 = TILED ? trow[((k >> 2) + 1) * 32] : *reinterpret_cast<const float4*>(row + k + 4);
            const float b0 = h ? v0.y : v0.x, b1 = h ? v0.w : v0.z, b2 = h ? v1.y : v1.x, b3 = h ? v1.w : v1.z;
            const float* qk = qT + (k + h) * 32 + r;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qk[0], b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qk[64], b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qk[128], b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qk[192], b3, acc, 0, 0, 0);
        }
        for (; k + 2 <= K; k += 2) {   // K % 8 tail (K is even, checked on the host)
            float b0;
            if (TILED) {
                const float4 t4 = trow[(k >> 2) * 32];
                const int e = (k & 3) + h;                      // k is even here: e in {0,1} or {2,3}
                b0 = e == 0 ? t4.x : (e == 1 ? t4.y : (e == 2 ? t4.z : t4.w));
            } else {
                b0 = row[k + h];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qT[(k + h) * 32 + r], b0, acc, 0, 0, 0);
        }
        // D layout: column = lane & 31 = document, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = query
        if (doc < D) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int qi = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (qi < nq) out[(int64_t)qi * out_ld + doc] = acc[reg];
            }
        }
    }
}

// Up to 256 queries per pass over the index (round 3).  The kernel above streams the whole index once per 32 queries -- 8 passes of
// 120 MB for the bench's 256-query batch, HBM-bound at 41 us each -- although a 32-document tile already in registers can be multiplied
// against every query block for the price of the (exact-f32) MFMAs alone.  Here a workgroup of 8 waves takes 8 document tiles and walks K
// in chunks of 8: the chunk of the query matrix ([8 k][256 queries], 9 KB) is staged in LDS by all threads (double-buffered, one barrier
// per chunk), each wave multiplies its documents' 8 k-values against NQB query blocks (4 NQB MFMAs) and keeps NQB accumulator tiles.
// Per (query, document) the MFMA sequence -- operands and order -- is the one of the kernel above: the same bits.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {      // v of the lane the DPP control names (all rows and banks enabled)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int SIMW_QS = 288;       // floats per k-row of the LDS chunk: 256 queries + 32, so that the two half-waves (k, k + 1) use different banks
template <int NQB>
__global__ __launch_bounds__(512) void sim_mfma_wide_kernel(const float4* __restrict__ tiled, int64_t D, int K, const float* __restrict__ q, int nq,
                                                            float* __restrict__ out, int64_t out_ld, float* __restrict__ part_max) {
    // part_max (round 4, optional): [gridDim.x][256] -- this workgroup's maximum of every query's products, taken from the accumulators
    // (the row maximum webui.py:377-380 divides by; a second pass over the 100 MB of stored products took 24 us per 256 queries).  The
    // consumer takes the maximum over the workgroups: max is exact, so the value is rowmax_kernel's bit for bit.
    __shared__ float qs[2][8][SIMW_QS];
    __shared__ uint32_t wmax[256];               // float_order_key images; 0 = nothing seen
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) wmax[tid] = 0u;               // (published by the first barrier of the first walk)
    const int r = lane & 31, h = lane >> 5;
    const int64_t ntiles = (D + 31) / 32;
    const int KQ = K >> 2;                       // float4 per document row
    const int nchunks = (KQ + 1) >> 1;           // 8 k per chunk; the last one holds 4 when K % 8 == 4
    const int sj = tid >> 1, sh = tid & 1;       // staging role: query sj, k-half sh of the chunk
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_q = [&](int c) {
        const int kq = 2 * c + sh;
        return (sj < nq && kq < KQ) ? *reinterpret_cast<const float4*>(q + (int64_t)sj * K + 4 * kq) : zero4;
    };
    auto store_q = [&](int buf, float4 v) {
        qs[buf][4 * sh + 0][sj] = v.x;
        qs[buf][4 * sh + 1][sj] = v.y;
        qs[buf][4 * sh + 2][sj] = v.z;
        qs[buf][4 * sh + 3][sj] = v.w;
    };
    // One walk over K for this wave: document tile `tile` (tv: the wave has one) against NB query blocks starting at block qb0.
    auto walk = [&](auto nb_c, int64_t tile, bool tv, int qb0) __attribute__((always_inline)) {
        constexpr int NB = decltype(nb_c)::value;
        const float4* __restrict__ trow = tiled + (tv ? tile : ntiles - 1) * KQ * 32 + r;
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.0f;
        float4 d0 = trow[0], d1 = 1 < KQ ? trow[32] : zero4;
        store_q(0, load_q(0));                   // buffer 0 was last read before the final barrier of the previous walk
        __syncthreads();
        const int nfull = KQ >> 1;               // chunks of 8 k; K % 8 == 4 leaves a chunk of 4 behind them
        for (int c = 0; c < nfull; ++c) {
            const bool more = c + 1 < nchunks;
            float4 qn = zero4, e0 = zero4, e1 = zero4;
            if (more) {                          // the next chunk's loads are in flight under this chunk's MFMAs
                qn = load_q(c + 1);
                e0 = trow[(2 * c + 2) * 32];
                e1 = 2 * c + 3 < KQ ? trow[(2 * c + 3) * 32] : zero4;
            }
            const float b0 = h ? d0.y : d0.x, b1 = h ? d0.w : d0.z, b2 = h ? d1.y : d1.x, b3 = h ? d1.w : d1.z;
            const float* qrow = &qs[c & 1][h][qb0 * 32 + r];
            // k-pair by k-pair across the blocks: consecutive MFMAs are independent (per block the k order is unchanged)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[b * 32], b0, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[b * 32 + 2 * SIMW_QS], b1, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[b * 32 + 4 * SIMW_QS], b2, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[b * 32 + 6 * SIMW_QS], b3, acc[b], 0, 0, 0);
            if (more) {
                store_q((c + 1) & 1, qn);
                d0 = e0;
                d1 = e1;
            }
            __syncthreads();
        }
        if (KQ & 1) {                            // the chunk of 4 (staged by the last iteration above, or by the prologue when K == 4)
            const float b0 = h ? d0.y : d0.x, b1 = h ? d0.w : d0.z;
            const float* qrow = &qs[nfull & 1][h][qb0 * 32 + r];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float* qk = qrow + b * 32;
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(qk[0], b0, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(qk[2 * SIMW_QS], b1, acc[b], 0, 0, 0);
            }
            __syncthreads();                     // the next walk's prologue writes buffer 0 again
        }
        const int64_t doc = tile * 32 + r;
        if (tv && doc < D) {
            // D layout: column = lane & 31 = document, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = query of the block.  One per-lane
            // pointer, the row offsets as scalar products of a stride the compiler may not hoist (left to itself it keeps 16 NQB 64-bit
            // offsets and as many lane masks alive across the whole tile loop: 256 VGPRs and 190 spilled for NQB = 8)
            int64_t old = out_ld;
            int nqv = nq;
            asm volatile("" : "+s"(old), "+s"(nqv));
            float* __restrict__ base = out + (int64_t)(qb0 * 32 + 4 * h) * old + doc;
            const int nleft = nqv - qb0 * 32;    // queries from this wave's first block on
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b * 32 + 32 <= nleft) {                     // uniform: a whole block of queries, no per-lane test
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) base[(int64_t)(b * 32 + (reg & 3) + 8 * (reg >> 2)) * old] = acc[b][reg];
                } else if (b * 32 < nleft) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int cqi = b * 32 + (reg & 3) + 8 * (reg >> 2);
                        if (cqi + 4 * h < nleft) base[(int64_t)cqi * old] = acc[b][reg];
                    }
                }
            }
        }
        if (part_max) {                          // uniform
            // Row maxima out of the accumulators (stored above, so the registers are free): the 32 documents of a half-wave share the
            // query.  Per value five DPP steps on the ORDER KEY -- unsigned max fuses with the DPP move into one v_max_u32_dpp: within
            // the quad, the half row, the row of 16, then row_bcast15 hands row 0's result to row 1 (rows 1 and 3 enabled) -- and no LDS
            // instruction at all; lanes 16 and 48 then hold the half-waves' maxima and add them to the workgroup's table with LDS
            // atomics issued back to back under ONE branch.  (Measured against the 24.6 us of the second pass this replaces: five
            // __shfl_xor per value +53 us on the kernel, float DPP steps + a swizzle +31 us, key DPP steps + a swizzle +27 us -- every
            // swizzle / shuffle is an LDS round trip the next value's waits for.)
            const bool live = tv && doc < D;     // padding documents of the last tile and idle waves do not take part
#pragma unroll
            for (int b = 0; b < NB; ++b) {       // sixteen values at a time: 128 keys beside the accumulators spilled
                uint32_t kv[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    uint32_t k = live ? float_order_key(acc[b][reg]) : 0u;
                    k = max(k, dpp_u32<0xB1>(k));       // quad_perm [1, 0, 3, 2]
                    k = max(k, dpp_u32<0x4E>(k));       // quad_perm [2, 3, 0, 1]
                    k = max(k, dpp_u32<0x141>(k));      // row_half_mirror
                    k = max(k, dpp_u32<0x140>(k));      // row_mirror
                    k = max(k, (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x142, 0xA, 0xF, false));      // row_bcast15 into rows 1 and 3
                    kv[reg] = k;
                }
                if (r == 16) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) atomicMax(&wmax[(qb0 + b) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h], kv[reg]);
                }
            }
        }
    };
    // Full rounds: every wave of every workgroup one tile against all NQB blocks.  What is left over (fewer tiles than wave slots) may be
    // split f ways across the query blocks (f waves per tile, NQB / f blocks each) when that shortens the tail.  A walk costs about
    // a + b NB with a = 2.5 b (measured at 100 k x 300, 256 queries: 114 us for NB = 8, 38 us for NB = 1 -- the per-chunk load -> LDS ->
    // barrier chain does not shrink with the MFMAs), so the split pays for small indices that would leave most workgroups without a tile
    // (32 tiles: one walk of 1 block on 32 workgroups instead of 8 blocks on 4), not for 3125 tiles on 2048 slots (measured: 302 us with
    // f = 8 against 228 us unsplit).
    const int64_t G = gridDim.x, slots = 8 * G;
    const int64_t full_rounds = ntiles / slots;
    for (int64_t rd = 0; rd < full_rounds; ++rd) walk(std::integral_constant<int, NQB>{}, (rd * G + blockIdx.x) * 8 + wave, true, 0);
    const int64_t base_tile = full_rounds * slots, rem = ntiles - base_tile;
    if (rem > 0) {
        int f = 1;
        int64_t best = (rem + slots - 1) / slots * (5 + 2 * NQB);        // walks x (a + b NB), a : b = 5 : 2
        for (int c = 2; c <= NQB; c *= 2) {
            const int64_t len = (rem * c + slots - 1) / slots * (5 + 2 * (NQB / c));
            if (len < best) {
                best = len;
                f = c;
            }
        }
        const int64_t jobs = rem * f, walks = (jobs + slots - 1) / slots;
        for (int64_t w = 0; w < walks; ++w) {
            const int64_t J = (w * G + blockIdx.x) * 8 + wave;
            const bool tv = J < jobs;
            const int64_t tile = base_tile + (tv ? J / f : 0);
            const int grp = (int)(J % f);
            if (f == 1) walk(std::integral_constant<int, NQB>{}, tile, tv, 0);
            if constexpr (NQB >= 2) { if (f == 2) walk(std::integral_constant<int, NQB / 2>{}, tile, tv, grp * (NQB / 2)); }
            if constexpr (NQB >= 4) { if (f == 4) walk(std::integral_constant<int, NQB / 4>{}, tile, tv, grp * (NQB / 4)); }
            if constexpr (NQB >= 8) { if (f == 8) walk(std::integral_constant<int, NQB / 8>{}, tile, tv, grp * (NQB / 8)); }
        }
    }
    if (part_max) {
        __syncthreads();
        if (tid < 256) part_max[(int64_t)blockIdx.x * 256 + tid] = wmax[tid] ? float_from_key(wmax[tid]) : -INFINITY;
    }
}

// (re)build tiles [t0, t1) of the tile-major copy from the row-major rows; rows >= len read as zero
__global__ __launch_bounds__(256) void retile_kernel(const float* __restrict__ rows, float4* __restrict__ tl, int64_t len, int K,
                                                     int64_t t0, int64_t t1) {
    const int KQ = K >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (t1 - t0) * KQ * 32;
    if (idx >= total) return;
    const int c = (int)(idx & 31);
    const int kq = (int)((idx >> 5) % KQ);
    const int64_t T = t0 + (idx >> 5) / KQ;
    const int64_t doc = T * 32 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (doc < len) v = *reinterpret_cast<const float4*>(rows + doc * K + 4 * kq);
    tl[(T * KQ + kq) * 32 + c] = v;
}

// =============================================================================================
// top-k by (value descending, index ascending): one workgroup per query.
// MSB-first radix select over an order-preserving u64 image of the float64 value (12-bit digits,
// LDS histogram), early exit once the survivors fit the LDS candidate buffer, then a bitonic
// sort of the candidates.  Exact for ties (ordered gather of the lowest indices).
// =============================================================================================
constexpr int TOPK_CAP = 2048;
constexpr int TOPK_MAX_K = 1024;

__device__ __forceinline__ uint64_t order_key(double x) {
    if (x == 0.0) x = 0.0;   // -0.0 and +0.0 compare equal in the reference's sort
    uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(uint64_t k) {
    uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// block-wide exclusive scan of one int per thread (1024 threads); returns exclusive prefix, total in *total
__device__ __forceinline__ int block_excl_scan(int x, int* scratch /*[17]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) {
            int t = scratch[w];
            scratch[w] = run;
            run += t;
        }
        scratch[16] = run;
    }
    __syncthreads();
    const int res = scratch[wave] + incl - x;
    *total = scratch[16];
    __syncthreads();
    return res;
}

// Monotone (non-decreasing) 12-bit digit of a score, uniform in VALUE over [-2, 2): the fast path of the
// top-k histograms on it.  Everything below -2 (and NaN) is digit 0, everything from 2 up is 4095.
__device__ __forceinline__ uint32_t value_digit(double x) {
    const double t = (x + 2.0) * 1024.0;
    return t >= 4095.0 ? 4095u : (t > 0.0 ? (uint32_t)t : 0u);
}

// Histogram increment that survives concentration: when many lanes of a wave hold the SAME digit (ties: every
// document a required term excludes scores -inf; exponent digits of like-sized scores) plain LDS atomics
// serialise on one address.  The lanes that share the first active lane's digit are counted with one ballot
// and added once; the remaining lanes add individually.
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t d, bool active) {
    const uint64_t act = __ballot(active);
    if (act == 0) return;
    const int leader = __ffsll((unsigned long long)act) - 1;
    const uint32_t dl = __shfl(d, leader);
    const uint64_t same = __ballot(active && d == dl);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
    else if (active && d != dl) atomicAdd(&hist[d], 1u);
}

// A workgroup streams its query's scores several times; each pass is bound by load latency x loads in flight,
// so every thread keeps TOPK_U independent 8-byte loads outstanding (128 KB per workgroup).
constexpr int TOPK_U = 16;
constexpr int TOPK_SORT_TARGET = 256;

struct Search1State;
struct Search1Witness;
__device__ void search1_state_clear(Search1State* st);
__device__ void search1_state_debug(Search1State* st, uint32_t cnt, uint32_t fast);
constexpr int S1_BLOCK_CAP = 64;     // candidates one search1_collect_kernel workgroup (1024 documents) may hand on: at least this many (search_one sizes it)
constexpr int S1_GATHER_BLOCKS = 1024;  // up to this many workgroups' slots are gathered through an offset table in LDS

#ifdef HIPTS_X_TOPK_STAMPS          // measurement-only build (tools/gpurun/r3_topk_stamps.sh): wall-clock stamps of workgroup 0, 10 ns units
__device__ unsigned long long g_topk_stamps[16];
#define TOPK_STAMP(i) do { __syncthreads(); if (blockIdx.x == HIPTS_X_TOPK_STAMPS && threadIdx.x == 0) g_topk_stamps[i] = wall_clock64(); } while (0)
#else
#define TOPK_STAMP(i) do { } while (0)
#endif

// The score rows as the two addends of webui.py:377-383 instead of their sum: with `a` set the kernel computes
// wa * (a / max_a) + (double)(wb * (b / max_b)) -- combine_kernel's expression, operation for operation -- wherever it reads a score,
// and the batched search neither writes nor re-reads the combined rows (round 3: 20 B per score of traffic less, one launch less).
struct TopkFused {
    const double* a = nullptr;       // [nq][n] BM25 scores
    const float* b = nullptr;        // [nq][n] index products
    double wa = 0.0;
    float wb = 0.f;
    const double* max_a = nullptr;   // [nq] row maxima (normalise when > 0)
    const float* max_b = nullptr;
    const float* max_b_parts = nullptr;      // instead of max_b: [query / 256][SIMW_MAX_GRID][256] per-workgroup maxima of sim_mfma_wide_kernel,
    int parts = 0;                           //   `parts` workgroups each
};

__global__ __launch_bounds__(1024) void topk_kernel(const double* __restrict__ vals, int64_t n, int k,
                                                    int32_t* __restrict__ ids_out, double* __restrict__ vals_out,
                                                    Search1State* __restrict__ pre = nullptr, const uint32_t* __restrict__ pre_cnt = nullptr,
                                                    const uint32_t* __restrict__ pre_flag = nullptr, const unsigned long long* __restrict__ pre_key = nullptr,
                                                    const uint32_t* __restrict__ pre_id = nullptr, int pre_blocks = 0,
                                                    uint32_t* __restrict__ done_flag = nullptr, uint32_t done_seq = 0, int pre_cap = S1_BLOCK_CAP,
                                                    const TopkFused fz = TopkFused()) {
    __shared__ uint32_t hist[4096];
    __shared__ uint64_t ckey[TOPK_CAP];
    __shared__ uint32_t cid[TOPK_CAP];
    __shared__ int scratch[17];
    __shared__ int sh_digit, sh_need, sh_bin, sh_cnt;
    __shared__ unsigned long long sh_z0;       // largest order key among the scores of digit 0 (fast path that has to dip into that bin)
    __shared__ int soff[S1_GATHER_BLOCKS];
    const int tid = threadIdx.x;
    const bool fused = fz.a != nullptr;
    const double* __restrict__ v = fused ? nullptr : vals + (int64_t)blockIdx.x * n;
    const double* __restrict__ fa = fused ? fz.a + (int64_t)blockIdx.x * n : nullptr;
    const float* __restrict__ fb = fused ? fz.b + (int64_t)blockIdx.x * n : nullptr;
    const double f_ma = fused ? fz.max_a[blockIdx.x] : 0.0;
    float f_mb = 0.f;
    if (fused) {
        if (fz.max_b_parts) {        // the maximum over the product kernel's workgroups (exact: the same value rowmax_kernel finds in the stored row)
            const float* pp = fz.max_b_parts + (int64_t)(blockIdx.x >> 8) * (256 * 256) + (blockIdx.x & 255);
            float v = tid < fz.parts ? pp[(int64_t)tid * 256] : -INFINITY;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
            __shared__ float pmax[16];
            if ((tid & 63) == 0) pmax[tid >> 6] = v;
            __syncthreads();
            v = pmax[0];
#pragma unroll
            for (int w = 1; w < 16; ++w) v = fmaxf(v, pmax[w]);
            f_mb = v;
        } else {
            f_mb = fz.max_b[blockIdx.x];
        }
    }
    auto comb = [&](double A, float B) -> double {       // combine_kernel, operation for operation (the file is built with -ffp-contract=off)
        if (f_ma > 0.0) A = A / f_ma;
        if (f_mb > 0.0f) B = B / f_mb;
        const float wB = fz.wb * B;
        return fz.wa * A + (double)wB;
    };
    auto val = [&](int64_t i) -> double { return fused ? comb(fa[i], fb[i]) : v[i]; };
    auto val2 = [&](int64_t i) -> double2 {              // scores i, i + 1 (i even, rows 16-byte aligned: `wide`)
        if (fused) {
            const double2 A = *reinterpret_cast<const double2*>(fa + i);
            const float2 B = *reinterpret_cast<const float2*>(fb + i);
            return make_double2(comb(A.x, B.x), comb(A.y, B.y));
        }
        return *reinterpret_cast<const double2*>(v + i);
    };
    if ((int64_t)k > n) k = (int)n;
    // ---- fast path (measured: the exact radix select below spends ~200 us of a 233 us single-query call in the
    // LDS atomics of its first histogram, 100 k of them).  Estimate the threshold from a 1/8 sample instead:
    // histogram the sample over a VALUE-uniform 12-bit digit (scores are normalised sums in about [-1, 1], so
    // the bins spread and the atomics do not pile up on a few exponents), pick the digit below which the sample
    // holds ~k/8 + 3 sigma entries, and collect every score with digit >= that one.  The digit is monotone in
    // the score, so the collected set is exactly "all scores >= a pivot": if it has at least k and at most CAP
    // members it contains the top k and the sort below finishes the job; otherwise the exact path runs.
    bool done_fast = false;
    int fill_need = 0;        // results still missing after the candidates: the lowest-index -inf scores (candidate path only)
    TOPK_STAMP(12);
    if (pre) {
        // Candidates handed on by search1_collect_kernel (per-workgroup slots): every score whose digit is at or above a threshold
        // digit, i.e. all scores >= a pivot.  With at least k and at most CAP of them the top k are among them.  With fewer than
        // k, and nothing but -inf below the pivot, they are ALL results and the rest are -inf ties in index order.
        int mine = 0, bad = 0, other = 0;
        for (int b = tid; b < pre_blocks; b += 1024) {
            const int cb = (int)pre_cnt[b];
            bad |= cb > pre_cap;
            mine += cb > pre_cap ? pre_cap : cb;
            other |= (int)pre_flag[b];
        }
        int c;
        const int excl = block_excl_scan(mine, scratch, &c);
        bad = __syncthreads_or(bad);
        const bool only_inf_below = __syncthreads_or(other) == 0;
        done_fast = !bad && c <= TOPK_CAP && (c >= k || only_inf_below);
        if (done_fast && pre_blocks <= S1_GATHER_BLOCKS) {
            // one thread per CANDIDATE: its workgroup by bisection of the offsets (a workgroup may hand on hundreds when k is a large
            // part of a small index; a thread per workgroup copying them one by one took a round trip each)
            if (tid < pre_blocks) soff[tid] = excl;
            __syncthreads();
            for (int j = tid; j < c; j += 1024) {
                int lo = 0, hi = pre_blocks - 1;                   // last block whose offset is <= j
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (soff[mid] <= j) lo = mid; else hi = mid - 1;
                }
                const int64_t src = (int64_t)lo * pre_cap + (j - soff[lo]);
                ckey[j] = pre_key[src];
                cid[j] = pre_id[src];
            }
            if (tid == 0) sh_cnt = c;
            if (c < k) fill_need = k - c;
        } else if (done_fast) {
            int off = excl;
            for (int b = tid; b < pre_blocks; b += 1024) {
                const int cb = (int)pre_cnt[b];
                for (int i = 0; i < cb; ++i) {
                    ckey[off + i] = pre_key[(int64_t)b * pre_cap + i];
                    cid[off + i] = pre_id[(int64_t)b * pre_cap + i];
                }
                off += cb;
            }
            if (tid == 0) sh_cnt = c;
            if (c < k) fill_need = k - c;
        }
        __syncthreads();
        TOPK_STAMP(13);
        search1_state_clear(pre);               // the maxima slots are zero again for the next query (search1_combine_kernel has read them)
        search1_state_debug(pre, (uint32_t)c, done_fast ? 1u : 0u);
    }
    TOPK_STAMP(0);
    if (!done_fast && n >= 8192) {
        for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        // the first 1024 of every 8192 scores; TOPK_U loads per thread in flight (one load per step made the sample a chain of ~13
        // dependent round trips for 100 k scores: ~25 us of the kernel's ~80 per workgroup)
        // 16-byte loads where the row allows it (even length, 16-byte aligned): 8-byte loads reach about 0.6 of the 16-byte rate on this
        // part (MI355X_MICROARCH.md, "8-B accesses 0.54-0.70x the 16-B rate"), and both passes of the fast path are pure streams
        const bool wide = (n & 1) == 0 && (fused ? ((reinterpret_cast<uintptr_t>(fa) & 15) == 0 && (reinterpret_cast<uintptr_t>(fb) & 7) == 0)
                                                 : (reinterpret_cast<uintptr_t>(v) & 15) == 0);
        constexpr int UW = TOPK_U / 2;
        uint32_t smax = 0u;
        bool shave = false;
        if (wide) {
            // the first 2048 of every 16384 scores (the same 1/8 sample)
            for (int64_t base0 = 0; base0 < n; base0 += (int64_t)UW * 16384) {
                double2 sx[UW];
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int64_t i = base0 + (int64_t)u * 16384 + 2 * tid;
                    sx[u] = i < n ? val2(i) : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int64_t i = base0 + (int64_t)u * 16384 + 2 * tid;
                    if (i < n) {
                        const uint32_t d0 = value_digit(sx[u].x), d1 = value_digit(sx[u].y);
                        smax = max(smax, max(d0, d1));
                        shave = true;
                    }
                }
            }
        } else
        for (int64_t base0 = 0; base0 < n; base0 += (int64_t)TOPK_U * 8192) {
            double sx[TOPK_U];
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = base0 + (int64_t)u * 8192 + tid;
                sx[u] = i < n ? val(i) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = base0 + (int64_t)u * 8192 + tid;
                if (i < n) {
                    smax = max(smax, value_digit(sx[u]));
                    shave = true;
                }
            }
        }
        // ONE histogram entry per thread: the largest digit among its ~12 samples.  The number of samples at or above a digit is at least
        // the number of thread maxima there, so the digit chosen below still leaves >= `want` samples above it (a few more when two of a
        // thread's samples qualify: 0.6 expected at k = 100, +9 % at k = 1024) -- and the histogram takes 1 k LDS atomics instead of 12.5 k
        // spread over 4096 bins (12 us of the kernel's 77 at k = 100: tools/topk_k.py, tools/topk_cases.py).
        TOPK_STAMP(1);
        hist_add(hist, smax, shave);
        __syncthreads();
        TOPK_STAMP(2);
        const int want = k / 8 + 3 * (int)ceilf(sqrtf((float)k / 8.0f)) + 4;
        int own[4], ssum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            own[j] = (int)hist[4095 - (4 * tid + j)];
            ssum += own[j];
        }
        int total;
        const int excl = block_excl_scan(ssum, scratch, &total);
        if (tid == 0) sh_digit = 0;                              // fewer sampled entries than `want`: take everything
        __syncthreads();
        if (excl < want && want <= excl + ssum) {
            int run = excl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (run < want && want <= run + own[j]) sh_digit = 4095 - (4 * tid + j);
                run += own[j];
            }
        }
        __syncthreads();
        // The sample holds fewer than `want` entries above digit 0 when a required term has left most of the row -inf (digit 0): the top k
        // then reach into that bin.  Round 2 took "everything", overflowed the candidate buffer and fell to the exact radix select -- six
        // passes, 130-200 us for a row with fewer than ~300 finite scores, and the slowest row sets the launch time (measured:
        // tools/topk_cases.py).  Now: collect everything ABOVE digit 0 and look at what the bin holds; if nothing but -inf (the common case),
        // the candidates are all results and the rest are -inf ties in index order (the ordered fill at the end of the kernel).
        TOPK_STAMP(3);
        const bool dip = sh_digit == 0;
        const uint32_t dmin = dip ? 1u : (uint32_t)sh_digit;
        if (tid == 0) sh_z0 = 0ull;
        __syncthreads();
        unsigned long long z0 = 0ull;
        // the next step's TOPK_U loads are requested before this step's values are examined (two register sets): the pass was a chain of
        // load round trips, one per 16 k scores
        // Candidates go into the wave's OWN 128 slots of ckey / cid, placed by ballot and a wave-uniform count -- no atomics: ~250 adds on
        // one LDS counter were 14 us of the kernel at k = 100.  A wave whose slots are full (skewed rows, k = 1024) appends to an overflow
        // list that lives in the histogram's memory (free now) through a shared counter; the lists are packed below.
        constexpr int WSLOTS = TOPK_CAP / 16, OVCAP = 1024;
        uint64_t* ovk = reinterpret_cast<uint64_t*>(hist);                  // 8 KB
        uint32_t* ovi = hist + 2 * OVCAP;                                   // 4 KB behind it
        const int wv = tid >> 6, ln = tid & 63;
        int wcnt = 0;
        __syncthreads();                                                    // every thread has read its bins of the histogram
        auto examine = [&](double xv, int64_t i) {
            const bool c = i < n && value_digit(xv) >= dmin;
            const unsigned long long m = __ballot(c);
            if (m) {
                if (c) {
                    const int pos = wcnt + __popcll(m & ((1ull << ln) - 1ull));
                    if (pos < WSLOTS) {
                        ckey[wv * WSLOTS + pos] = order_key(xv);
                        cid[wv * WSLOTS + pos] = (uint32_t)i;
                    } else {
                        const int slot = atomicAdd(&sh_cnt, 1);
                        if (slot < OVCAP) {
                            ovk[slot] = order_key(xv);
                            ovi[slot] = (uint32_t)i;
                        }
                    }
                }
                wcnt += __popcll(m);
            }
            if (!c && dip && i < n) {
                const unsigned long long kx = order_key(xv);
                z0 = kx > z0 ? kx : z0;
            }
        };
        if (wide) {
            double2 x[UW], xn[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int64_t i = (int64_t)u * 2048 + 2 * tid;
                x[u] = i < n ? val2(i) : make_double2(-INFINITY, -INFINITY);
            }
            for (int64_t i0 = 0; i0 < n; i0 += UW * 2048) {
                const int64_t i1 = i0 + UW * 2048;
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int64_t i = i1 + u * 2048 + 2 * tid;
                    xn[u] = i < n ? val2(i) : make_double2(-INFINITY, -INFINITY);
                }
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int64_t i = i0 + u * 2048 + 2 * tid;
                    examine(x[u].x, i);
                    examine(x[u].y, i + 1);
                }
#pragma unroll
                for (int u = 0; u < UW; ++u) x[u] = xn[u];
            }
        } else {
        double x[TOPK_U], xn[TOPK_U];
#pragma unroll
        for (int u = 0; u < TOPK_U; ++u) {
            const int64_t i = (int64_t)u * 1024 + tid;
            x[u] = i < n ? val(i) : -INFINITY;
        }
        for (int64_t i0 = 0; i0 < n; i0 += TOPK_U * 1024) {
            const int64_t i1 = i0 + TOPK_U * 1024;
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = i1 + u * 1024 + tid;
                xn[u] = i < n ? val(i) : -INFINITY;
            }
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) examine(x[u], i0 + u * 1024 + tid);
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) x[u] = xn[u];
        }
        }
        if (dip) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long other = __shfl_xor(z0, o);
                z0 = other > z0 ? other : z0;
            }
            if ((tid & 63) == 0) atomicMax(&sh_z0, z0);
        }
        TOPK_STAMP(4);
        // pack the sixteen wave lists and the overflow list into ckey / cid [0, total): through registers (source and destination overlap)
        if (ln == 0) scratch[wv] = wcnt < WSLOTS ? wcnt : WSLOTS;
        __syncthreads();
        {
            const int nov = sh_cnt;                                         // entries the waves tried to append to the overflow list
            int off[17];
            off[0] = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) off[w + 1] = off[w] + scratch[w];
            const int total = off[16] + (nov < OVCAP ? nov : OVCAP);
            uint64_t mk[3];
            uint32_t mi[3];
            int md[3];
#pragma unroll
            for (int e = 0; e < 2; ++e) {                                    // wave-list slots tid and tid + 1024
                const int sl = tid + e * 1024, w = sl / WSLOTS, j = sl - w * WSLOTS;
                md[e] = j < scratch[w] ? off[w] + j : -1;
                mk[e] = ckey[sl];
                mi[e] = cid[sl];
            }
            md[2] = tid < nov && tid < OVCAP ? off[16] + tid : -1;           // overflow slot tid
            mk[2] = ovk[tid];
            mi[2] = ovi[tid];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (md[e] >= 0 && md[e] < TOPK_CAP) {
                    ckey[md[e]] = mk[e];
                    cid[md[e]] = mi[e];
                }
            __syncthreads();
            if (tid == 0) sh_cnt = nov > OVCAP ? TOPK_CAP + 1 : total;      // an overflowing overflow list: not a fast-path row
        }
        __syncthreads();
        if (dip) {
            // digit 0 empty (z0 == 0: every score was collected) or nothing but -inf in it
            done_fast = sh_cnt <= TOPK_CAP && (sh_z0 == 0ull || sh_z0 == order_key(-INFINITY));
            if (done_fast && sh_cnt < k) fill_need = k - sh_cnt;
        } else {
            done_fast = sh_cnt >= k && sh_cnt <= TOPK_CAP;
        }
        __syncthreads();
    }
    TOPK_STAMP(5);
    uint64_t prefix = 0;
    int pbits = 0;
    int need = k;            // how many of the keys matching `prefix` are still wanted
    bool fits = done_fast;
    const int shifts[6] = {52, 40, 28, 16, 4, 0};
    for (int pass = 0; pass < 6 && !fits; ++pass) {
        const int shift = shifts[pass];
        const int dbits = pass == 5 ? 4 : 12;
        for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int64_t i0 = 0; i0 < n; i0 += TOPK_U * 1024) {
            uint64_t key[TOPK_U];
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = i0 + u * 1024 + tid;
                key[u] = i < n ? order_key(val(i)) : 0;
            }
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = i0 + u * 1024 + tid;
                hist_add(hist, (uint32_t)(key[u] >> shift) & ((1u << dbits) - 1), i < n && (pbits == 0 || (key[u] >> (64 - pbits)) == prefix));
            }
        }
        __syncthreads();
        // walk bins from the top: thread t owns reversed bins 4t..4t+3
        int own[4], s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            own[j] = (int)hist[4095 - (4 * tid + j)];
            s += own[j];
        }
        int total;
        int excl = block_excl_scan(s, scratch, &total);
        if (excl < need && need <= excl + s) {
            int run = excl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (run < need && need <= run + own[j]) {
                    sh_digit = 4095 - (4 * tid + j);
                    sh_need = need - run;
                    sh_bin = own[j];
                }
                run += own[j];
            }
        }
        __syncthreads();
        prefix = (prefix << dbits) | (uint64_t)(sh_digit & ((1 << dbits) - 1));
        pbits += dbits;
        const int above = k - sh_need;          // keys strictly above the chosen bin (all wanted)
        need = sh_need;
        // stop refining once the candidate set is small enough to SORT cheaply: the final bitonic sort costs
        // log^2 barriers (2048 candidates = 66 stages ~ 100 us, 256 = 36), a further pass over the scores ~15 us;
        // past the last digit (64 bits) whatever fits the LDS buffers is taken
        fits = above + sh_bin <= (pbits < 64 ? max(TOPK_SORT_TARGET, k + 64) : TOPK_CAP);
        __syncthreads();
        if (pass == 0 && !fits && sh_bin > TOPK_CAP) {
            // A first bin with more members than the buffers hold is usually one value repeated (every document
            // a required term rules out scores -inf).  One pass decides: if the smallest and the largest key
            // in the bin agree, the remaining five digit passes are known in advance.
            uint64_t mn = ~0ull, mx = 0ull;
            for (int64_t i = tid; i < n; i += 1024) {
                const uint64_t key = order_key(val(i));
                if ((key >> (64 - pbits)) == prefix) {
                    mn = key < mn ? key : mn;
                    mx = key > mx ? key : mx;
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const uint64_t a = __shfl_xor(mn, o), b = __shfl_xor(mx, o);
                mn = a < mn ? a : mn;
                mx = b > mx ? b : mx;
            }
            if ((tid & 63) == 0) {
                ckey[tid >> 6] = mn;
                ckey[16 + (tid >> 6)] = mx;
            }
            __syncthreads();
            mn = ckey[0];
            mx = ckey[16];
            for (int w = 1; w < 16; ++w) {
                mn = ckey[w] < mn ? ckey[w] : mn;
                mx = ckey[16 + w] > mx ? ckey[16 + w] : mx;
            }
            __syncthreads();
            if (mn == mx) {
                prefix = mn;
                pbits = 64;
                break;              // fits stays false: the ordered tie compaction below takes the `need` lowest indices
            }
        }
    }
    const uint64_t low = pbits == 64 ? prefix : (prefix << (64 - pbits));
    if (!done_fast) {
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    if (fits) {
        for (int64_t i0 = 0; i0 < n; i0 += TOPK_U * 1024) {
            uint64_t key[TOPK_U];
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = i0 + u * 1024 + tid;
                key[u] = i < n ? order_key(val(i)) : 0;
            }
#pragma unroll
            for (int u = 0; u < TOPK_U; ++u) {
                const int64_t i = i0 + u * 1024 + tid;
                if (i < n && key[u] >= low) {
                    const int slot = atomicAdd(&sh_cnt, 1);
                    ckey[slot] = key[u];
                    cid[slot] = (uint32_t)i;
                }
            }
        }
        __syncthreads();
    } else {
        // pbits == 64 and more than CAP exact ties at the threshold: everything above it, then the
        // `need` lowest indices among the ties (ordered compaction).
        for (int64_t i = tid; i < n; i += 1024) {
            const uint64_t key = order_key(val(i));
            if (key > low) {
                const int slot = atomicAdd(&sh_cnt, 1);
                ckey[slot] = key;
                cid[slot] = (uint32_t)i;
            }
        }
        __syncthreads();
        int base = sh_cnt, taken = 0;
        for (int64_t i0 = 0; i0 < n && taken < need; i0 += 1024) {
            const int64_t i = i0 + tid;
            const int flag = (i < n && order_key(val(i)) == low) ? 1 : 0;
            int total;
            const int excl = block_excl_scan(flag, scratch, &total);
            if (flag && taken + excl < need) {
                ckey[base + taken + excl] = low;
                cid[base + taken + excl] = (uint32_t)i;
            }
            taken += total;
        }
        __syncthreads();
        if (tid == 0) sh_cnt = base + (taken < need ? taken : need);
        __syncthreads();
    }
    }
    const int cnt = sh_cnt;
    const int kout = cnt < k ? cnt : k;
    TOPK_STAMP(6);
    if (cnt <= 640) {
        // few candidates: rank by counting -- rank(i) = #{j : j before i in (key desc, id asc)} -- no barriers, LDS broadcasts
        // P adjacent lanes (as many as 1024 threads allow) share a candidate, each counting over every P-th entry (one thread per candidate walked the whole list as a
        // chain of LDS round trips: 26 us of the workgroup's 67 at k = 100 -- tools/topk_stamps.py); partial ranks meet by lane swaps
        __syncthreads();
        const int P = cnt <= 64 ? 16 : cnt <= 128 ? 8 : cnt <= 256 ? 4 : cnt <= 512 ? 2 : 1;
        const int c = tid / P, part = tid - c * P;
        const bool live = c < cnt;
        const uint64_t ki = live ? ckey[c] : 0ull;
        const uint32_t ii = live ? cid[c] : 0u;
        int rank = 0;
        if (live) {
#pragma unroll 4
            for (int j = part; j < cnt; j += P) {
                const uint64_t kj = ckey[j];
                const uint32_t ij = cid[j];
                rank += (kj > ki || (kj == ki && ij < ii)) ? 1 : 0;
            }
        }
        if (P >= 2) rank += __shfl_xor(rank, 1);
        if (P >= 4) rank += __shfl_xor(rank, 2);
        if (P >= 8) rank += __shfl_xor(rank, 4);
        if (P >= 16) rank += __shfl_xor(rank, 8);
        if (live && part == 0 && rank < kout) {
            ids_out[(int64_t)blockIdx.x * k + rank] = (int32_t)ii;
            vals_out[(int64_t)blockIdx.x * k + rank] = key_value(ki);
        }
    } else {
        int np2 = 64;
        while (np2 < cnt) np2 <<= 1;
        for (int i = cnt + tid; i < np2; i += 1024) {
            ckey[i] = 0;
            cid[i] = 0xffffffffu;
        }
        __syncthreads();
        // bitonic sort, "greater first": (key desc, id asc)
        for (int size = 2; size <= np2; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (np2 >> 1); t += 1024) {
                    const int lo = ((t / stride) * stride * 2) + (t % stride);
                    const int hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const uint64_t ka = ckey[lo], kb = ckey[hi];
                    const uint32_t ia = cid[lo], ib = cid[hi];
                    const bool a_first = (ka > kb) || (ka == kb && ia < ib);
                    if (a_first != desc) {
                        ckey[lo] = kb; ckey[hi] = ka;
                        cid[lo] = ib; cid[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < kout; i += 1024) {
            ids_out[(int64_t)blockIdx.x * k + i] = (int32_t)cid[i];
            vals_out[(int64_t)blockIdx.x * k + i] = key_value(ckey[i]);
        }
    }
    TOPK_STAMP(7);
    if (fill_need > 0) {
        // the remaining results are -inf scores in ascending index order (ordered compaction; stops as soon as enough are found)
        const uint64_t ninf = order_key(-INFINITY);
        int taken = 0;
        for (int64_t i0 = 0; i0 < n && taken < fill_need; i0 += 1024) {
            const int64_t i = i0 + tid;
            const int flag = (i < n && order_key(val(i)) == ninf) ? 1 : 0;
            int total;
            const int excl = block_excl_scan(flag, scratch, &total);
            if (flag && taken + excl < fill_need) {
                ids_out[(int64_t)blockIdx.x * k + cnt + taken + excl] = (int32_t)i;
                vals_out[(int64_t)blockIdx.x * k + cnt + taken + excl] = -INFINITY;
            }
            taken += total;
        }
    }
    TOPK_STAMP(14);
    if (done_flag) {
        // The results above went to pinned host memory.  Publish them to the HOST without waiting for the runtime's completion
        // signal (a hipStreamSynchronize wake-up costs ~10 us): every storing thread fences at system scope, the workgroup meets,
        // one lane releases the sequence number the host is spinning on.
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    TOPK_STAMP(15);
}

// =============================================================================================
// One query at a time -- the reference's real call (webui.py:586: find_similar_documents(query, topn=800)).
// The batched kernels above give one workgroup to a query for BM25, the row maxima and the top-k: at nq = 1 that is
// one CU of 256 busy for ~120 of the ~170 us a query took.  Here every step runs over the whole chip, thread per
// document, and the query itself travels in the kernel arguments (no staging copy):
//   search1_score    BM25 in the reference's own shape (webui.py:139-170: for each query term a pass over the documents,
//                    here the document's own (term, tf) list, document-major CSR) + the index product as the k-ordered
//                    fmaf chain (bit-equal to the exact-f32 MFMA chain of sim_mfma_kernel and to the oracle) from the
//                    tile-major copy; per-workgroup maxima into 256 slots
//   search1_combine  webui.py:377-383 with the global maxima; the maximum of every group of 64 * gw documents
//   search1_collect  threshold digit = digit of the k-th largest group maximum; candidates (score digit >= threshold) into
//                    the workgroup's own slots
//   topk_kernel      gathers the candidates and ranks them (by counting when few, bitonic sort otherwise); fills up with -inf
//                    ties when a required / excluded term left fewer than k finite scores; if the candidates overflow their
//                    buffers the exact radix select runs instead -- same kernel, same results
// No global atomics on shared addresses after the score kernel: same-address atomics serialise at ~50-60 ns each on this part
// (measured: 1563 histogram adds over ~100 bins cost 5 us, ~100 adds on one counter 6 us).
// Algorithmic bytes per query: D * dim * 4 (index) + nnz * 8 + D * 12 (CSR) + D * (8 + 4) * 2 + D * 8 * 2 (scores, final).
// =============================================================================================
constexpr int S1_MAX_TERMS = 16;
constexpr int S1_MAX_DIM = 768;
constexpr int S1_THREADS = 128;
constexpr int S1_SLOTS = 256;
constexpr int S1_MIN_DOCS = 8192;
constexpr int S1_LDS_TERMS = 6144;   // term ids of one workgroup's 128 documents staged in LDS (48 per document)
constexpr int S1_GROUPS = 4096;      // at most this many document groups (a group = 64 * gw consecutive documents, one wave of search1_combine_kernel)

struct Search1Query {            // passed by value: 3.3 KB of the 4 KB kernel-argument segment
    int32_t nt, dim;
    int32_t terms[S1_MAX_TERMS];
    double weights[S1_MAX_TERMS];
    float q[S1_MAX_DIM];
};

struct Search1State {            // device resident; the maxima slots are all zero between queries (the last kernel clears them)
    unsigned long long max_a[S1_SLOTS];      // order_key images of the per-workgroup BM25 maxima
    uint32_t max_b[S1_SLOTS];                // float_order_key images of the index-product maxima
    uint32_t dbg[4];                         // NOT cleared: {candidates, took the candidate path, 0, 0} of the last query (hiptsdbg_search1_last)
};

struct Search1Witness {          // 32 B per wave of search1_score_kernel (64 documents)
    double bm_a;                 // the wave's largest BM25 score ...
    float sim_a, pad0;           // ... and that document's index product
    double bm_b;                 // the BM25 score of the document with ...
    float sim_b, pad1;           // ... the wave's largest index product
};

__device__ void search1_state_clear(Search1State* st) {
    uint32_t* w = reinterpret_cast<uint32_t*>(st);
    for (int i = threadIdx.x; i < (int)(offsetof(Search1State, dbg) / 4); i += blockDim.x) w[i] = 0;
}
__device__ void search1_state_debug(Search1State* st, uint32_t cnt, uint32_t fast) {
    if (threadIdx.x == 0) {
        st->dbg[0] = cnt;
        st->dbg[1] = fast;
    }
}

// PIPE (round 5, the default; HIPTS_S1_PIPE=0 runs the sequential form): the BM25 part is a chain of dependent round trips -- the span's
// bounds, its term ids (up to 12 passes of four loads -> LDS, each a round trip of its own), the matches' tf, the idf values -- about 12 of
// the kernel's 32 us during which only ONE round of the index stream was in flight.  Here the rounds of the stream are dealt between the
// chain's steps (two rounds in flight throughout, every consume waits only for loads older than the chain's), the term ids are requested 16
// per lane at once, and the idf values go through LDS, requested with the first round.  Same operations per document in the same order.
template <bool PIPE>
__global__ __launch_bounds__(S1_THREADS) void search1_score_kernel(const Search1Query Q, const float4* __restrict__ tiled, int64_t D,
                                                                   const int64_t* __restrict__ ptr, const int32_t* __restrict__ term,
                                                                   const int32_t* __restrict__ tf, const int32_t* __restrict__ dl,
                                                                   const double* __restrict__ idf, int32_t V, double avgdl,
                                                                   double* __restrict__ bm_out, float* __restrict__ sim_out,
                                                                   Search1State* __restrict__ st, Search1Witness* __restrict__ wit) {
    __shared__ int32_t sterm[S1_LDS_TERMS];
    __shared__ int sptr[S1_THREADS + 1];
    __shared__ int32_t stf[S1_MAX_TERMS][S1_THREADS];          // tf of query term j in this thread's document (0 = absent)
    const int tid = threadIdx.x;
    const int64_t d = (int64_t)blockIdx.x * S1_THREADS + tid;
    const bool valid = d < D;
    const int64_t dd = valid ? d : D - 1;
    // ---- index product: acc = fmaf(row[k], q[k], acc), k ascending (the order of sim_mfma_kernel's chain and of the oracle).
    // The kernel is a stream of D * dim * 4 bytes: two register buffers of U float4 per lane, the loads of round r + 1 are issued
    // before round r is consumed, and the first round is requested before the BM25 part below so that it flies meanwhile.
    // Rounds past the last k re-read the last float4 (a cache hit) against Q.q's zero padding: fmaf(x, +0, acc) = acc exactly.
    // (Measured: U = 8 or 16 and 128- or 256-thread workgroups make no difference; 64-thread workgroups are 40 % slower.)
    const int KQ = Q.dim >> 2;
    const float4* __restrict__ p = tiled + ((dd >> 5) * KQ) * 32 + (dd & 31);
    constexpr int U = 8;
    static_assert(S1_MAX_DIM % (4 * U) == 0, "Q.q must cover whole rounds");
    float4 va[U], vb[U], vc[U];      // (vc: the PIPE form keeps THREE rounds in flight -- the chain's LDS phases are longer than a round)
    auto request = [&](float4(&v)[U], int kq0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kq = kq0 + u < KQ ? kq0 + u : KQ - 1;
            v[u] = p[(int64_t)kq * 32];
        }
    };
    float acc = 0.0f;
    auto consume = [&](const float4(&v)[U], int kq0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* qq = Q.q + 4 * (kq0 + u);
            acc = fmaf(v[u].x, qq[0], acc);
            acc = fmaf(v[u].y, qq[1], acc);
            acc = fmaf(v[u].z, qq[2], acc);
            acc = fmaf(v[u].w, qq[3], acc);
        }
    };
    const int rounds = (KQ + U - 1) / U;
    // one step of the stream: round r is consumed, round r + 2 requested into its buffer (past the end: a clamped re-read nobody consumes)
    constexpr int AHEAD = PIPE ? 3 : 2;
    auto step_a = [&](int r) {
        if (r < rounds) {
            consume(va, r * U);
            request(va, (r + AHEAD) * U);
        }
    };
    auto step_b = [&](int r) {
        if (r < rounds) {
            consume(vb, r * U);
            request(vb, (r + AHEAD) * U);
        }
    };
    auto step_c = [&](int r) {
        if (r < rounds) {
            consume(vc, r * U);
            request(vc, (r + AHEAD) * U);
        }
    };
    request(va, 0);
    // ---- BM25 (webui.py:139-170), the arithmetic of bm25_score_kernel.  The (term, tf) lists of a workgroup's documents are one
    // contiguous span of the document-major CSR: its term ids are staged in LDS with coalesced loads, then walked ONCE, coalesced:
    // a lane compares its entry with the (scalar) query terms; on a match -- at most nt per document -- it finds the entry's
    // document by bisection of the 128 start offsets and records the entry's tf in the slot of (query term, document).
    // (Also measured without gain: the BM25 part on two waves of its own beside two index-stream waves (32.7 us); the tf staged
    // with the term id so that a match needs no global load (35.5 us).  The index stream alone runs at 6 TB/s -- sim1_kernel back to
    // back, 20 us for these 120 MB -- so ~12 us of this kernel are the BM25 part's dependent chain, not bandwidth.)
    // (Measured alternatives, all within 2 us of each other at ~33 us for the kernel: every thread walking its own list from memory
    // or from LDS; a staged entry -> document map with 16-bit LDS stores was slower, 50 us.)
    const int64_t d_first = (int64_t)blockIdx.x * S1_THREADS;
    const int64_t d_last = d_first + S1_THREADS < D ? d_first + S1_THREADS : D;
    const int64_t b0 = ptr[d_first], e0 = ptr[d_last];
    const bool staged = e0 - b0 <= S1_LDS_TERMS;
    const int nspan = staged ? (int)(e0 - b0) : 0;
    __shared__ double sidf[S1_MAX_TERMS];
    const int64_t b = ptr[dd], e = ptr[dd + 1];
    const double dlv = (double)dl[dd];
    double s = 0.0;
    bool masked = false;
    if constexpr (PIPE) {
        double idf_mine = 0.0;
        if (tid < Q.nt) {
            const int32_t t = Q.terms[tid];
            idf_mine = (t >= 0 && t < V) ? idf[t] : 0.0;
        }
        request(vb, U);
        request(vc, 2 * U);
        __builtin_amdgcn_sched_barrier(0);
        step_a(0);
        // the span's term ids, 16 per lane in flight (2048 ids per pass: most spans in one)
        constexpr int TU = 16;
        int32_t t16[TU];
        auto req_terms = [&](int i0) {
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int i = i0 + u * S1_THREADS + tid;
                t16[u] = i < nspan ? term[b0 + i] : 0;
            }
        };
        auto put_terms = [&](int i0) {
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int i = i0 + u * S1_THREADS + tid;
                if (i < nspan) sterm[i] = t16[u];
            }
        };
        if (nspan > 0) req_terms(0);
        __builtin_amdgcn_sched_barrier(0);
        step_b(1);
        __builtin_amdgcn_sched_barrier(0);
        if (nspan > 0) put_terms(0);
        for (int i0 = TU * S1_THREADS; i0 < nspan; i0 += TU * S1_THREADS) {
            req_terms(i0);
            put_terms(i0);
        }
        for (int j = 0; j < Q.nt; ++j) stf[j][tid] = 0;
        sptr[tid] = valid ? (int)(b - b0) : (int)(e0 - b0);
        if (tid == 0) sptr[S1_THREADS] = (int)(e0 - b0);
        if (tid < Q.nt) sidf[tid] = idf_mine;
        __syncthreads();
        step_c(2);
        __builtin_amdgcn_sched_barrier(0);
        if (staged) {
            for (int i = tid; i < nspan; i += S1_THREADS) {
                const int32_t ti = sterm[i];
                for (int j = 0; j < Q.nt; ++j)
                    if (ti == Q.terms[j]) {
                        int lo = 0, hi = S1_THREADS;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (sptr[mid] <= i) lo = mid;
                            else hi = mid;
                        }
                        stf[j][lo] = tf[b0 + i];
                    }
            }
            __syncthreads();
        } else {
            for (int64_t i = b; i < e; ++i) {
                const int32_t ti = term[i];
                for (int j = 0; j < Q.nt; ++j)
                    if (ti == Q.terms[j]) stf[j][tid] = tf[i];
            }
        }
        step_a(3);
        __builtin_amdgcn_sched_barrier(0);
        const double nrm = BM25_K1 * ((1.0 - BM25_B) + BM25_B * (dlv / avgdl));
        for (int j = 0; j < Q.nt; ++j) {
            const double w = Q.weights[j];
            const int32_t tfv = stf[j][tid];
            const double idf_t = sidf[j];
            const double tfd = (double)tfv;
            const double sc = idf_t * ((tfd * (BM25_K1 + 1.0)) / (tfd + nrm));
            if (w < 0.0) {
                if (tfv > 0) masked = true;
            } else if (w > REQUIRE_MAGIC) {
                s += (w - REQUIRE_MAGIC) * sc;
                if (tfv == 0) masked = true;
            } else {
                s += w * sc;
            }
        }
        if (masked) s = -INFINITY;
        // ---- the rest of the index stream
        step_b(4);
        step_c(5);
        for (int r = 6; r < rounds; r += 3) {
            step_a(r);
            step_b(r + 1);
            step_c(r + 2);
        }
    } else {
    for (int i0 = 0; i0 < nspan; i0 += 4 * S1_THREADS) {
        int32_t t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * S1_THREADS + tid;
            t4[u] = i < nspan ? term[b0 + i] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * S1_THREADS + tid;
            if (i < nspan) sterm[i] = t4[u];
        }
    }
    for (int j = 0; j < Q.nt; ++j) stf[j][tid] = 0;
    sptr[tid] = valid ? (int)(b - b0) : (int)(e0 - b0);         // where this thread's document starts inside the staged span (threads past D: nowhere)
    if (tid == 0) sptr[S1_THREADS] = (int)(e0 - b0);
    __syncthreads();
    if (staged) {
        for (int i = tid; i < nspan; i += S1_THREADS) {
            const int32_t ti = sterm[i];
            for (int j = 0; j < Q.nt; ++j)
                if (ti == Q.terms[j]) {
                    int lo = 0, hi = S1_THREADS;            // largest lo with sptr[lo] <= i (an empty document shares its start with the next)
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (sptr[mid] <= i) lo = mid;
                        else hi = mid;
                    }
                    stf[j][lo] = tf[b0 + i];
                }
        }
        __syncthreads();
    } else {        // a span too long for LDS (documents with hundreds of tags): every thread walks its own list in memory
        for (int64_t i = b; i < e; ++i) {
            const int32_t ti = term[i];
            for (int j = 0; j < Q.nt; ++j)
                if (ti == Q.terms[j]) stf[j][tid] = tf[i];
        }
    }
    const double nrm = BM25_K1 * ((1.0 - BM25_B) + BM25_B * (dlv / avgdl));
    for (int j = 0; j < Q.nt; ++j) {
        const int32_t t = Q.terms[j];
        const double w = Q.weights[j];
        const int32_t tfv = stf[j][tid];
        const double idf_t = (t >= 0 && t < V) ? idf[t] : 0.0;
        const double tfd = (double)tfv;
        const double sc = idf_t * ((tfd * (BM25_K1 + 1.0)) / (tfd + nrm));
        if (w < 0.0) {
            if (tfv > 0) masked = true;
        } else if (w > REQUIRE_MAGIC) {
            s += (w - REQUIRE_MAGIC) * sc;
            if (tfv == 0) masked = true;
        } else {
            s += w * sc;
        }
    }
    if (masked) s = -INFINITY;
    // ---- the index stream
    int r = 0;
    for (; r + 2 <= rounds; r += 2) {
        request(vb, (r + 1) * U);
        consume(va, r * U);
        request(va, (r + 2) * U);               // past the end: a clamped re-read that nobody consumes
        consume(vb, (r + 1) * U);
    }
    if (r < rounds) consume(va, r * U);
    }
    if (valid) {
        bm_out[d] = s;
        sim_out[d] = acc;
    }
    // ---- per-workgroup maxima -> one of 256 slots (atomicMax on order-preserving images; 0 is below every value): ~3 adds per slot
    double ma = valid ? s : -INFINITY;
    float mb = valid ? acc : -INFINITY;
    const double s_own = ma;
    const float acc_own = mb;
    for (int o = 32; o >= 1; o >>= 1) {
        ma = fmax(ma, __shfl_xor(ma, o));
        mb = fmaxf(mb, __shfl_xor(mb, o));
    }
    if (wit) {
        // Two WITNESS documents per wave for search1_finish_kernel's threshold: the one with the wave's largest BM25 score and the one with
        // its largest index product, each with BOTH of its scores.  Whatever the global maxima turn out to be, the combined score of a real
        // document is a lower bound of its wave's best combined score -- and the wave's best is, in practice, one of these two.
        const unsigned long long ba = __ballot(s_own == ma), bb = __ballot(acc_own == mb);
        const int la = ba ? __ffsll(ba) - 1 : 0, lb = bb ? __ffsll(bb) - 1 : 0;      // (an all-NaN wave: any lane; its witness then scores NaN = digit 0)
        const float sim_at_a = __shfl(acc_own, la);
        const double bm_at_b = __shfl(s_own, lb);
        if ((tid & 63) == 0) {
            Search1Witness w;
            w.bm_a = ma; w.sim_a = sim_at_a; w.bm_b = bm_at_b; w.sim_b = mb; w.pad0 = 0.f; w.pad1 = 0.f;
            wit[d >> 6] = w;                        // d of lane 0 = the wave's first document (D rounded up to whole waves is allocated)
        }
    }
    __shared__ double pa[S1_THREADS / 64];
    __shared__ float pb[S1_THREADS / 64];
    if ((tid & 63) == 0) {
        pa[tid >> 6] = ma;
        pb[tid >> 6] = mb;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < S1_THREADS / 64; ++w) {
            ma = fmax(ma, pa[w]);
            mb = fmaxf(mb, pb[w]);
        }
        atomicMax(&st->max_a[blockIdx.x % S1_SLOTS], (unsigned long long)order_key(ma));
        atomicMax(&st->max_b[blockIdx.x % S1_SLOTS], float_order_key(mb));
    }
}

__device__ __forceinline__ void search1_maxima(const Search1State* __restrict__ st, double* ma, float* mb) {
    // every wave folds the slots itself (3 KB from L2)
    const int lane = threadIdx.x & 63;
    unsigned long long ka = 0;
    uint32_t kb = 0;
#pragma unroll
    for (int i = 0; i < S1_SLOTS / 64; ++i) {
        const unsigned long long a = st->max_a[i * 64 + lane];
        const uint32_t b = st->max_b[i * 64 + lane];
        ka = a > ka ? a : ka;
        kb = b > kb ? b : kb;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long oa = __shfl_xor(ka, o);
        const uint32_t ob = __shfl_xor(kb, o);
        ka = oa > ka ? oa : ka;
        kb = ob > kb ? ob : kb;
    }
    *ma = ka ? key_value(ka) : -INFINITY;
    *mb = kb ? float_from_key(kb) : -INFINITY;
}


// index[query] for ONE query (webui.py:205: the rerank product; webui.py:306-309 restated: the character-feature differences):
// the index-stream half of search1_score_kernel alone -- thread per document, k-ordered fmaf chain from the tile-major copy,
// the query in the kernel arguments.  Same bits as sim_mfma_kernel's chain; one pass costs the same bytes but no LDS query
// tile, no 32-wide MFMA for one useful column: 100k x 768 in ~60 us instead of ~270 us per call.
struct Sim1Query {
    int32_t dim, pad[3];
    float q[S1_MAX_DIM];
};

__global__ __launch_bounds__(128) void sim1_kernel(const Sim1Query Q, const float4* __restrict__ tiled, int64_t D, float* __restrict__ out) {
    const int64_t d = (int64_t)blockIdx.x * 128 + threadIdx.x;
    const int64_t dd = d < D ? d : D - 1;
    const int KQ = Q.dim >> 2;
    const float4* __restrict__ p = tiled + ((dd >> 5) * KQ) * 32 + (dd & 31);
    constexpr int U = 8;
    float4 va[U], vb[U];
    auto request = [&](float4(&v)[U], int kq0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kq = kq0 + u < KQ ? kq0 + u : KQ - 1;
            v[u] = p[(int64_t)kq * 32];
        }
    };
    float acc = 0.0f;
    auto consume = [&](const float4(&v)[U], int kq0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* qq = Q.q + 4 * (kq0 + u);
            acc = fmaf(v[u].x, qq[0], acc);
            acc = fmaf(v[u].y, qq[1], acc);
            acc = fmaf(v[u].z, qq[2], acc);
            acc = fmaf(v[u].w, qq[3], acc);
        }
    };
    request(va, 0);
    const int rounds = (KQ + U - 1) / U;
    int r = 0;
    for (; r + 2 <= rounds; r += 2) {
        request(vb, (r + 1) * U);
        consume(va, r * U);
        request(va, (r + 2) * U);
        consume(vb, (r + 1) * U);
    }
    if (r < rounds) consume(va, r * U);
    if (d < D) out[d] = acc;
}

// Threshold without sampling: the documents are cut into G <= 4096 groups of 64 * gw consecutive documents (one wave of this
// kernel each); the k-th largest of the G group maxima is a LOWER bound of the k-th largest score (k groups hold a score at
// least that large), so "every score whose digit is >= the digit of that bound" always contains the top k -- about
// G * -ln(1 - k/G) of them, a few percent more than k (measured, 100k documents: 100-110 candidates for k = 100, 1630-1710 for
// k = 1024).  Each wave stores its maximum in its own slot: no atomics.
__global__ __launch_bounds__(256) void search1_combine_kernel(const double* __restrict__ bm, const float* __restrict__ sim, int64_t D, double wa,
                                                              float wb, double* __restrict__ final_out, const Search1State* __restrict__ st,
                                                              unsigned long long* __restrict__ wmax, int gw, int gl) {
    double ma;
    float mb;
    search1_maxima(st, &ma, &mb);
    const int lane = threadIdx.x & 63;
    const int64_t group = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    double m = -INFINITY;
    for (int c = 0; c < gw; ++c) {
        const int64_t d = (group * gw + c) * 64 + lane;
        if (d < D) {
            double A = bm[d];
            float B = sim[d];
            if (ma > 0.0) A = A / ma;                    // webui.py:379-380
            if (mb > 0.0f) B = B / mb;                   // webui.py:377-378
            const float wB = wb * B;
            const double f = wa * A + (double)wB;        // webui.py:383
            final_out[d] = f;
            m = fmax(m, f);
        }
    }
    // gl < 64 (only with gw == 1): the wave's 64 documents form 64 / gl groups of gl consecutive ones -- when k is a large part of a
    // small index there must be several times k groups for the k-th largest group maximum to be a useful bound
    for (int o = (gl >> 1); o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    if ((lane & (gl - 1)) == 0) wmax[group * (64 / gl) + lane / gl] = (unsigned long long)order_key(m);      // a group past the last document stores the image of -inf
}

constexpr int S1_COLLECT_THREADS = 1024;
__global__ __launch_bounds__(S1_COLLECT_THREADS) void search1_collect_kernel(const double* __restrict__ final_in, int64_t D, int k,
                                                                             const unsigned long long* __restrict__ wmax, int groups,
                                                                             uint32_t* __restrict__ bcnt, uint32_t* __restrict__ bflag,
                                                                             unsigned long long* __restrict__ bkey, uint32_t* __restrict__ bid, int bcap) {
    __shared__ uint32_t hist[4096];
    __shared__ int scan[17];
    __shared__ int sh_dmin;
    __shared__ uint32_t sh_n, sh_other;
    const int tid = threadIdx.x, lane = tid & 63;
    // threshold digit: the digit of the k-th largest group maximum (every workgroup finds it itself: 8 B x G from L2)
    for (int i = tid; i < 4096; i += S1_COLLECT_THREADS) hist[i] = 0;
    if (tid == 0) {
        sh_n = 0;
        sh_other = 0;
    }
    __syncthreads();
    for (int g0 = 0; g0 < groups; g0 += S1_COLLECT_THREADS) {
        const int g = g0 + tid;
        const uint32_t dg = g < groups ? value_digit(key_value(wmax[g])) : 0u;
        hist_add(hist, dg, dg != 0u);
    }
    __syncthreads();
    // Digit 0 is everything below -2: combined scores are >= -1 unless a required / excluded term made them -inf, so with fewer
    // than k groups above it the threshold is digit 1 -- every finite score is a candidate and the ranking fills up with -inf ties.
    int own[4], ssum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int bin = 4095 - (4 * tid + j);
        own[j] = bin >= 1 ? (int)hist[bin] : 0;
        ssum += own[j];
    }
    int total;
    const int excl = block_excl_scan(ssum, scan, &total);
    if (tid == 0) sh_dmin = 1;
    __syncthreads();
    if (excl < k && k <= excl + ssum) {
        int run = excl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (run < k && k <= run + own[j]) sh_dmin = 4095 - (4 * tid + j);
            run += own[j];
        }
    }
    __syncthreads();
    const uint32_t dmin = (uint32_t)sh_dmin;
    const int64_t d = (int64_t)blockIdx.x * S1_COLLECT_THREADS + tid;
    const double f = d < D ? final_in[d] : -INFINITY;
    const bool take = d < D && value_digit(f) >= dmin;
    if (d < D && !take && f != -INFINITY) sh_other = 1u;
    const uint64_t m = __ballot(take);
    if (m != 0) {                                   // candidates go to this workgroup's own slots (LDS counter, one add per wave)
        uint32_t base = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) base = atomicAdd(&sh_n, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (take) {
            const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (slot < (uint32_t)bcap) {
                bkey[(int64_t)blockIdx.x * bcap + slot] = order_key(f);
                bid[(int64_t)blockIdx.x * bcap + slot] = (uint32_t)d;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        bcnt[blockIdx.x] = sh_n;                    // may exceed the cap: the ranking kernel then takes the exact path
        bflag[blockIdx.x] = sh_other;
    }
}

// search1_combine_kernel + search1_collect_kernel in ONE launch (round 4): the threshold no longer comes from the group maxima of the
// combined scores (which only exist after a pass over all documents, hence the launch boundary between the two kernels) but from the
// WITNESS documents the score kernel left behind, two per wave: a workgroup combines the 2 G witnesses itself (G x 32 B from L2),
// takes per wave the larger of its two, and uses the digit of the k-th largest of those G values -- still a lower bound of the k-th
// largest combined score, because every value belongs to a real document and no two waves share one.  Then it combines its own 1024
// documents (webui.py:377-383, combine_kernel's expression, operation for operation), stores them for the fallback paths and the
// caller, and hands on the candidates as search1_collect_kernel does.  Used when a group is a whole wave (gl == 64, gw == 1: 160 k <=
// D <= 262 k documents at k = 100); other sizes keep the two kernels.
__global__ __launch_bounds__(S1_COLLECT_THREADS) void search1_finish_kernel(const double* __restrict__ bm, const float* __restrict__ sim, int64_t D, int k,
                                                                            double wa, float wb, double* __restrict__ final_out,
                                                                            const Search1State* __restrict__ st, const Search1Witness* __restrict__ wit,
                                                                            int groups, uint32_t* __restrict__ bcnt, uint32_t* __restrict__ bflag,
                                                                            unsigned long long* __restrict__ bkey, uint32_t* __restrict__ bid, int bcap) {
    __shared__ uint32_t hist[4096];
    __shared__ int scan[17];
    __shared__ int sh_dmin;
    __shared__ uint32_t sh_n, sh_other;
    const int tid = threadIdx.x, lane = tid & 63;
    double ma;
    float mb;
    search1_maxima(st, &ma, &mb);
    auto comb = [&](double A, float B) -> double {       // search1_combine_kernel / combine_kernel, operation for operation
        if (ma > 0.0) A = A / ma;
        if (mb > 0.0f) B = B / mb;
        const float wB = wb * B;
        return wa * A + (double)wB;
    };
    for (int i = tid; i < 4096; i += S1_COLLECT_THREADS) hist[i] = 0;
    if (tid == 0) {
        sh_n = 0;
        sh_other = 0;
    }
    __syncthreads();
    for (int g0 = 0; g0 < groups; g0 += S1_COLLECT_THREADS) {
        const int g = g0 + tid;
        uint32_t dg = 0u;
        if (g < groups) {
            const Search1Witness w = wit[g];
            dg = value_digit(fmax(comb(w.bm_a, w.sim_a), comb(w.bm_b, w.sim_b)));
        }
        hist_add(hist, dg, dg != 0u);
    }
    __syncthreads();
    int own[4], ssum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int bin = 4095 - (4 * tid + j);
        own[j] = bin >= 1 ? (int)hist[bin] : 0;
        ssum += own[j];
    }
    int total;
    const int excl = block_excl_scan(ssum, scan, &total);
    if (tid == 0) sh_dmin = 1;
    __syncthreads();
    if (excl < k && k <= excl + ssum) {
        int run = excl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (run < k && k <= run + own[j]) sh_dmin = 4095 - (4 * tid + j);
            run += own[j];
        }
    }
    __syncthreads();
    const uint32_t dmin = (uint32_t)sh_dmin;
    const int64_t d = (int64_t)blockIdx.x * S1_COLLECT_THREADS + tid;
    double f = -INFINITY;
    if (d < D) {
        f = comb(bm[d], sim[d]);
        final_out[d] = f;
    }
    const bool take = d < D && value_digit(f) >= dmin;
    if (d < D && !take && f != -INFINITY) sh_other = 1u;
    const uint64_t m = __ballot(take);
    if (m != 0) {
        uint32_t base = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane == leader) base = atomicAdd(&sh_n, (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (take) {
            const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (slot < (uint32_t)bcap) {
                bkey[(int64_t)blockIdx.x * bcap + slot] = order_key(f);
                bid[(int64_t)blockIdx.x * bcap + slot] = (uint32_t)d;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        bcnt[blockIdx.x] = sh_n;
        bflag[blockIdx.x] = sh_other;
    }
}

// Handle-less entry points (hipts_combine / hipts_topk) keep their small scratch here: one slot
// per (device, purpose), intentionally never freed (freeing at static-destruction time would
// race the HIP runtime's own teardown).  Callers are single-threaded per device by contract.
DevBuf& scratch_buf(int device, int which) {
    static DevBuf* bufs = new DevBuf[64 * 4];
    return bufs[(device & 63) * 4 + (which & 3)];
}

int copy_out(void* dst, const void* src_dev, size_t bytes, int memspace, hipStream_t s) {
    if (!dst || bytes == 0) return HIPTS_OK;
    if (memspace == HIPTS_DEVICE) {
        if (dst != src_dev) HIPTS_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToDevice, s));
    } else {
        HIPTS_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, s));
        HIPTS_HIP(hipStreamSynchronize(s));
    }
    return HIPTS_OK;
}

enum QueryProfCat { QP_BM25 = 0, QP_SIM, QP_ROWMAX, QP_COMBINE, QP_TOPK, QP_S1_SCORE, QP_S1_COMBINE, QP_S1_COLLECT, QP_S1_TOPK, QP_COUNT };
static_assert(QP_COUNT == HIPTS_QUERY_PROF_CATEGORIES, "category count");
const char* const kQueryProfNames[QP_COUNT] = {"bm25_postings_kernel", "sim_mfma_kernel", "rowmax_kernel<float>", "combine_kernel", "topk_kernel",
                                               "search1_score_kernel", "search1_combine_kernel", "search1_collect_kernel | search1_finish_kernel", "topk_kernel<candidates>"};

struct QueryProfScope {
    hipts_bm25* h;
    hipStream_t s;
    hipts_bm25::ProfRec r{};
    bool on;
    QueryProfScope(hipts_bm25* h_, hipStream_t s_, int cat, double bytes) : h(h_), s(s_), on(h_ && h_->prof) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e = nullptr;
            if (!h->prof_pool.empty()) {
                e = h->prof_pool.back();
                h->prof_pool.pop_back();
            } else if (hipEventCreate(&e) != hipSuccess) {
                e = nullptr;
            }
            return e;
        };
        r.cat = cat;
        r.bytes = bytes;
        r.a = get();
        r.b = get();
        if (r.a) (void)hipEventRecord(r.a, s);
    }
    ~QueryProfScope() {
        if (!on) return;
        if (r.b) (void)hipEventRecord(r.b, s);
        h->prof_recs.push_back(r);
    }
};

}  // namespace

// =============================================================================================
// Dense index handle
// =============================================================================================
struct hipts_index {
    int device = 0;
    int dim = 0;
    int64_t len = 0, cap = 0;
    DevBuf rows;      // float [cap][dim]
    DevBuf tiled;     // tile-major copy for the query kernel (dim % 4 == 0): float4 [ceil(cap/32)][dim/4][32]
    DevBuf ws_q, ws_out;
};

namespace {

// HIPTS_SIM_PARTS=0: the batched search takes the index products' row maxima with rowmax_kernel again (A/B; default: from the accumulators)
bool sim_parts_enabled() {
    static const bool on = !(getenv("HIPTS_SIM_PARTS") && atoi(getenv("HIPTS_SIM_PARTS")) == 0);
    return on;
}
bool sim_wide_enabled() {
    static const bool on = !(getenv("HIPTS_SIM_WIDE") && atoi(getenv("HIPTS_SIM_WIDE")) == 0) && !(getenv("HIPTS_SIM") && strcmp(getenv("HIPTS_SIM"), "rows") == 0);
    return on;
}
// passes over the index that launch_sim makes for nq queries (the profile's algorithmic bytes)
int sim_index_passes(bool tiled, int K, int nq) {
    if (!(tiled && sim_wide_enabled() && K % 4 == 0)) return (nq + 31) / 32;
    int passes = 0, done = 0;
    while (nq - done > 32) {
        done += std::min(256, nq - done);
        ++passes;
    }
    return passes + (done < nq ? 1 : 0);
}

// parts / parts_grid (optional): when every query goes through the wide kernel, its per-workgroup row maxima are left in parts
// ([query group of 256][SIMW_MAX_GRID][256] floats) and *parts_grid says how many workgroups wrote; *parts_grid = 0 otherwise (the caller
// then runs rowmax_kernel over the stored products as before).
constexpr int SIMW_MAX_GRID = 256;
int launch_sim(const float* index, const float* tiled, int64_t D, int K, const float* q_dev, int nq, float* out_dev, int64_t out_ld,
               hipStream_t s, float* parts = nullptr, int* parts_grid = nullptr) {
    if (parts_grid) *parts_grid = 0;
    const size_t lds = (size_t)K * 32 * sizeof(float);
    HIPTS_REQUIRE(lds <= 160 * 1024, "index dim %d too large for the query tile in LDS", K);
    static PerDevice attr_set;
    {
        int dev = 0;
        HIPTS_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(attr_set.mu);
        if (!attr_set.done(dev)) {
            HIPTS_HIP(hipFuncSetAttribute((const void*)sim_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIPTS_HIP(hipFuncSetAttribute((const void*)sim_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set.mark(dev);
        }
    }
    static const bool use_tiled = !(getenv("HIPTS_SIM") && strcmp(getenv("HIPTS_SIM"), "rows") == 0);
    const int64_t ntiles = (D + 31) / 32;
    constexpr int WPB = SIM_THREADS / 64;
    int grid = (int)std::min<int64_t>((ntiles + WPB - 1) / WPB, 256 * 4);
    if (grid < 1) grid = 1;
    // more than 32 queries over the tile-major copy: one pass per 256 queries (HIPTS_SIM_WIDE=0: the 32-query passes, for A/B)
    const bool use_wide = sim_wide_enabled();
    int q_done = 0;
    if (tiled && use_tiled && use_wide && K % 4 == 0) {
        // up to one workgroup per CU; a small index still fills them: the kernel splits the tiles it has across the query blocks
        auto grid_for = [&](int nqb) { return (int)std::max<int64_t>(1, std::min<int64_t>((ntiles * nqb + 7) / 8, SIMW_MAX_GRID)); };
        // the row maxima come out of the accumulators when no query is left for the 32-query kernel below and the groups share a grid
        bool want_parts = parts && parts_grid && nq > 32;
        if (want_parts) {
            int rem = nq, g0 = -1;
            while (rem > 32) {
                const int n = std::min(256, rem), g = grid_for(n <= 64 ? 2 : n <= 128 ? 4 : 8);
                if (g0 >= 0 && g != g0) want_parts = false;
                g0 = g;
                rem -= n;
            }
            if (rem > 0) want_parts = false;
        }
        while (nq - q_done > 32) {
            const int n = std::min(256, nq - q_done);
            const float4* t4 = reinterpret_cast<const float4*>(tiled);
            const float* qp = q_dev + (int64_t)q_done * K;
            float* op = out_dev + (int64_t)q_done * out_ld;
            float* pm = want_parts ? parts + (int64_t)(q_done / 256) * SIMW_MAX_GRID * 256 : nullptr;
            const int nqb = n <= 64 ? 2 : n <= 128 ? 4 : 8;
            if (nqb == 2) sim_mfma_wide_kernel<2><<<grid_for(2), 512, 0, s>>>(t4, D, K, qp, n, op, out_ld, pm);
            else if (nqb == 4) sim_mfma_wide_kernel<4><<<grid_for(4), 512, 0, s>>>(t4, D, K, qp, n, op, out_ld, pm);
            else sim_mfma_wide_kernel<8><<<grid_for(8), 512, 0, s>>>(t4, D, K, qp, n, op, out_ld, pm);
            HIPTS_LAUNCH_CHECK();
            if (want_parts) *parts_grid = grid_for(nqb);
            q_done += n;
        }
    }
    for (int q0 = q_done; q0 < nq; q0 += 32) {
        const int n = std::min(32, nq - q0);
        if (tiled && use_tiled) sim_mfma_kernel<true><<<grid, SIM_THREADS, lds, s>>>(tiled, D, K, K, q_dev + (int64_t)q0 * K, n, out_dev + (int64_t)q0 * out_ld, out_ld);
        else sim_mfma_kernel<false><<<grid, SIM_THREADS, lds, s>>>(index, D, K, K, q_dev + (int64_t)q0 * K, n, out_dev + (int64_t)q0 * out_ld, out_ld);
        HIPTS_LAUNCH_CHECK();
    }
    return HIPTS_OK;
}

int launch_bm25(hipts_bm25* h, const int32_t* qt_dev, const double* qw_dev, const int32_t* qp_dev, int nq, double* out_dev,
                hipStream_t s, double* max_out = nullptr) {
    static const bool scan = getenv("HIPTS_BM25") && strcmp(getenv("HIPTS_BM25"), "scan") == 0;   // A/B: document-major scan
    // HIPTS_BM25_PARTS: workgroups per query of the sliced kernel (1 = the one-workgroup-per-query kernel; default: enough to put ~1024 on the chip)
    static const int parts_env = getenv("HIPTS_BM25_PARTS") ? atoi(getenv("HIPTS_BM25_PARTS")) : 0;
    if (!scan && h->slice_docs > 0 && parts_env != 1) {
        int parts = parts_env;
        if (parts != 2 && parts != 4 && parts != 8) parts = nq >= 256 ? 4 : 8;
        HIPTS_TRY(h->ws_mark.reserve((size_t)nq * h->D));
        if (max_out) HIPTS_TRY(h->ws_maxpart.reserve((size_t)nq * parts * 8));
        bm25_postings_sliced_kernel<<<dim3(nq, parts), 512, 0, s>>>(h->d_tslice.as<int64_t>(), h->d_tdoc.as<int32_t>(), h->d_ttf.as<int32_t>(),
                                                                     h->d_dl.as<int32_t>(), h->d_idf.as<double>(), h->V, h->avgdl, h->D, h->slice_docs,
                                                                     BM25_SLICES / parts, qt_dev, qw_dev, qp_dev, out_dev, h->ws_mark.as<uint8_t>(),
                                                                     max_out ? h->ws_maxpart.as<double>() : nullptr);
        HIPTS_LAUNCH_CHECK();
        if (max_out) {
            bm25_max_reduce_kernel<<<ceil_div(nq, 256), 256, 0, s>>>(h->ws_maxpart.as<double>(), parts, max_out, nq);
            HIPTS_LAUNCH_CHECK();
        }
        return HIPTS_OK;
    }
    if (!scan) {
        HIPTS_TRY(h->ws_mark.reserve((size_t)nq * h->D));
        bm25_postings_kernel<<<nq, 1024, 0, s>>>(h->d_tptr.as<int64_t>(), h->d_tdoc.as<int32_t>(), h->d_ttf.as<int32_t>(),
                                                 h->d_dl.as<int32_t>(), h->d_idf.as<double>(), h->V, h->avgdl, h->D, qt_dev, qw_dev,
                                                 qp_dev, out_dev, h->ws_mark.as<uint8_t>(), max_out);
        HIPTS_LAUNCH_CHECK();
        return HIPTS_OK;
    }
    dim3 grid(ceil_div(h->D, 256), nq);
    bm25_score_kernel<<<grid, 256, 0, s>>>(h->d_ptr.as<int64_t>(), h->d_term.as<int32_t>(), h->d_tf.as<int32_t>(),
                                           h->d_dl.as<int32_t>(), h->d_idf.as<double>(), h->V, h->avgdl, h->D, qt_dev, qw_dev,
                                           qp_dev, out_dev);
    HIPTS_LAUNCH_CHECK();
    if (max_out) {
        rowmax_kernel<double><<<nq, 1024, 0, s>>>(out_dev, h->D, max_out);
        HIPTS_LAUNCH_CHECK();
    }
    return HIPTS_OK;
}

// stage the CSR query description in the handle's workspace; returns device pointers
int stage_queries(hipts_bm25* h, const int32_t* q_terms, const double* q_weights, const int32_t* q_ptr, int nq,
                  const int32_t** qt, const double** qw, const int32_t** qp, hipStream_t s) {
    const int nt = q_ptr[nq];
    HIPTS_REQUIRE(q_ptr[0] == 0 && nt >= 0, "q_ptr must start at 0 and be non-decreasing");
    const size_t off_w = ((size_t)nt * 4 + 15) / 16 * 16;
    const size_t off_p = off_w + (size_t)nt * 8;
    const size_t total = off_p + (size_t)(nq + 1) * 4;
    HIPTS_TRY(h->ws_q.reserve(total));
    std::vector<char> host(total, 0);
    memcpy(host.data(), q_terms, (size_t)nt * 4);
    memcpy(host.data() + off_w, q_weights, (size_t)nt * 8);
    memcpy(host.data() + off_p, q_ptr, (size_t)(nq + 1) * 4);
    HIPTS_HIP(hipMemcpyAsync(h->ws_q.p, host.data(), total, hipMemcpyHostToDevice, s));
    HIPTS_HIP(hipStreamSynchronize(s));   // `host` goes out of scope
    *qt = h->ws_q.as<int32_t>();
    *qw = reinterpret_cast<const double*>(h->ws_q.as<char>() + off_w);
    *qp = reinterpret_cast<const int32_t*>(h->ws_q.as<char>() + off_p);
    return HIPTS_OK;
}

// hipts_search for ONE query (webui.py:345-383 + the ranking of :191-192): four launches, no staging copy, results written by the
// last kernel straight into pinned host memory and published by a sequence number the host spins on.
int search_one(hipts_bm25* bm25, hipts_index* index, const int32_t* q_terms, const double* q_weights, int nt, const float* q_vector,
               double w_bm25, double w_sim, int k, int32_t* ids_out, double* vals_out, double* final_out_device, hipStream_t s) {
    const int64_t D = bm25->D;
    Search1Query Q;
    memset(&Q, 0, sizeof(Q));
    Q.nt = nt;
    Q.dim = index->dim;
    for (int j = 0; j < nt; ++j) {
        Q.terms[j] = q_terms[j];
        Q.weights[j] = q_weights[j];
    }
    memcpy(Q.q, q_vector, (size_t)index->dim * 4);
    HIPTS_TRY(bm25->ws_scores.reserve((size_t)D * 8));
    HIPTS_TRY(bm25->ws_sims.reserve((size_t)D * 4));
    double* final_dev = final_out_device;
    if (!final_dev) {
        HIPTS_TRY(bm25->ws_final.reserve((size_t)D * 8));
        final_dev = bm25->ws_final.as<double>();
    }
    const int64_t waves = (D + 63) / 64;
    const int gw = (int)((waves + S1_GROUPS - 1) / S1_GROUPS);                   // 64 * gw documents per group
    const int blocks2 = (int)((waves + (int64_t)gw * 4 - 1) / ((int64_t)gw * 4));
    const int blocks3 = ceil_div(D, S1_COLLECT_THREADS);
    const int kk = (int)std::min<int64_t>(k, D);
    // lanes per group: 64 unless that leaves fewer than 2.5 k groups (then the k-th largest group maximum bounds little: with fewer
    // groups than k every finite score is a candidate); down to 4.  G groups give about G * -ln(1 - k / G) candidates.
    int gl = 64;
    while (gw == 1 && gl > 4 && D / gl < (int64_t)kk * 5 / 2) gl >>= 1;
    const int groups = blocks2 * 4 * (64 / gl);
    // candidate slots per collect workgroup (1024 documents): four times the expected share of ~1.7 k candidates, 64 .. 1024
    int bcap = S1_BLOCK_CAP;
    while (bcap < 1024 && (int64_t)bcap * D < (int64_t)4 * 1024 * 17 * kk / 10) bcap <<= 1;
    // workspace: state | group maxima u64[groups] | per-workgroup counts u32[blocks3] | flags u32[blocks3] | keys u64[blocks3][CAP] | ids u32[blocks3][CAP]
    const size_t off_wmax = (sizeof(Search1State) + 15) / 16 * 16;
    const size_t off_bcnt = off_wmax + (size_t)groups * 8;
    const size_t off_bflag = off_bcnt + (size_t)blocks3 * 4;
    const size_t off_bkey = (off_bflag + (size_t)blocks3 * 4 + 15) / 16 * 16;
    const size_t off_bid = off_bkey + (size_t)blocks3 * bcap * 8;
    // combine + collect in one launch (search1_finish_kernel) when a group is exactly one wave of the score kernel: its threshold comes from
    // the score kernel's witness records (HIPTS_SEARCH1_FINISH=0: the two kernels, for A/B)
    static const bool allow_finish = !(getenv("HIPTS_SEARCH1_FINISH") && atoi(getenv("HIPTS_SEARCH1_FINISH")) == 0);
    const bool finish = allow_finish && gl == 64 && gw == 1;
    const size_t off_wit = (off_bid + (size_t)blocks3 * bcap * 4 + 31) / 32 * 32;
    const size_t ws_bytes = off_wit + (finish ? (size_t)ceil_div(D, S1_THREADS) * (S1_THREADS / 64) * sizeof(Search1Witness) : 0);
    if (bm25->s1_state.bytes < ws_bytes) {
        HIPTS_TRY(bm25->s1_state.alloc(ws_bytes));
        bm25->s1_dirty = true;
    }
    char* ws = bm25->s1_state.as<char>();
    Search1State* st = reinterpret_cast<Search1State*>(ws);
    if (bm25->s1_dirty) HIPTS_HIP(hipMemsetAsync(st, 0, sizeof(Search1State), s));
    bm25->s1_dirty = true;                    // until the last kernel (which clears the maxima slots) has been enqueued
    unsigned long long* wmax = reinterpret_cast<unsigned long long*>(ws + off_wmax);
    uint32_t* bcnt = reinterpret_cast<uint32_t*>(ws + off_bcnt);
    uint32_t* bflag = reinterpret_cast<uint32_t*>(ws + off_bflag);
    unsigned long long* bkey = reinterpret_cast<unsigned long long*>(ws + off_bkey);
    uint32_t* bid = reinterpret_cast<uint32_t*>(ws + off_bid);
    Search1Witness* wit = finish ? reinterpret_cast<Search1Witness*>(ws + off_wit) : nullptr;
    const size_t out_bytes = (size_t)kk * 12;
    HIPTS_TRY(bm25->pin_out.reserve(out_bytes + 192));
    double* hv = bm25->pin_out.as<double>();                  // pinned + mapped: the last kernel stores the results here
    int32_t* hi = reinterpret_cast<int32_t*>(hv + kk);
    static const bool via_copy = getenv("HIPTS_SEARCH1_D2H") && strcmp(getenv("HIPTS_SEARCH1_D2H"), "1") == 0;     // A/B: device buffer + copy
    static const bool allow_flag = !(getenv("HIPTS_SEARCH1_FLAG") && strcmp(getenv("HIPTS_SEARCH1_FLAG"), "0") == 0);   // A/B: completion by stream synchronise
    const bool use_flag = allow_flag && !via_copy && !bm25->prof;
    uint32_t* flag = reinterpret_cast<uint32_t*>(bm25->pin_out.as<char>() + ((out_bytes + 63) / 64 * 64));       // its own cache line behind the results
    if (++bm25->s1_seq == 0) bm25->s1_seq = 1;
    const uint32_t seq = bm25->s1_seq;
    if (use_flag) __atomic_store_n(flag, 0u, __ATOMIC_RELAXED);          // (fresh pinned memory is not zeroed)
    double* ov = hv;
    int32_t* oi = hi;
    if (via_copy) {
        HIPTS_TRY(bm25->ws_out.reserve(out_bytes + 64));
        ov = bm25->ws_out.as<double>();
        oi = reinterpret_cast<int32_t*>(ov + kk);
    }
    {
        QueryProfScope ps(bm25, s, QP_S1_SCORE, (double)D * index->dim * 4.0 + (double)bm25->nnz * 8.0 + (double)D * (8 + 4 + 8 + 4));
        static const bool pipe = !(getenv("HIPTS_S1_PIPE") && atoi(getenv("HIPTS_S1_PIPE")) == 0);      // A/B: the sequential form
        if (pipe)
            search1_score_kernel<true><<<ceil_div(D, S1_THREADS), S1_THREADS, 0, s>>>(Q, index->tiled.as<float4>(), D, bm25->d_ptr.as<int64_t>(),
                                                                                  bm25->d_term.as<int32_t>(), bm25->d_tf.as<int32_t>(),
                                                                                  bm25->d_dl.as<int32_t>(), bm25->d_idf.as<double>(), bm25->V, bm25->avgdl,
                                                                                  bm25->ws_scores.as<double>(), bm25->ws_sims.as<float>(), st, wit);
        else
            search1_score_kernel<false><<<ceil_div(D, S1_THREADS), S1_THREADS, 0, s>>>(Q, index->tiled.as<float4>(), D, bm25->d_ptr.as<int64_t>(),
                                                                                   bm25->d_term.as<int32_t>(), bm25->d_tf.as<int32_t>(),
                                                                                   bm25->d_dl.as<int32_t>(), bm25->d_idf.as<double>(), bm25->V, bm25->avgdl,
                                                                                   bm25->ws_scores.as<double>(), bm25->ws_sims.as<float>(), st, wit);
        HIPTS_LAUNCH_CHECK();
    }
    if (finish) {
        QueryProfScope ps(bm25, s, QP_S1_COLLECT, (double)D * 20.0 + (double)blocks3 * (double)waves * 32.0);
        search1_finish_kernel<<<blocks3, S1_COLLECT_THREADS, 0, s>>>(bm25->ws_scores.as<double>(), bm25->ws_sims.as<float>(), D, kk, w_bm25, (float)w_sim,
                                                                     final_dev, st, wit, (int)waves, bcnt, bflag, bkey, bid, bcap);
        HIPTS_LAUNCH_CHECK();
    } else {
        {
            QueryProfScope ps(bm25, s, QP_S1_COMBINE, (double)D * 20.0);
            search1_combine_kernel<<<blocks2, 256, 0, s>>>(bm25->ws_scores.as<double>(), bm25->ws_sims.as<float>(), D, w_bm25, (float)w_sim, final_dev,
                                                           st, wmax, gw, gl);
            HIPTS_LAUNCH_CHECK();
        }
        {
            QueryProfScope ps(bm25, s, QP_S1_COLLECT, (double)D * 8.0);
            search1_collect_kernel<<<blocks3, S1_COLLECT_THREADS, 0, s>>>(final_dev, D, kk, wmax, groups, bcnt, bflag, bkey, bid, bcap);
            HIPTS_LAUNCH_CHECK();
        }
    }
    {
        QueryProfScope ps(bm25, s, QP_S1_TOPK, (double)kk * 24.0);
        topk_kernel<<<1, 1024, 0, s>>>(final_dev, D, kk, oi, ov, st, bcnt, bflag, bkey, bid, blocks3, use_flag ? flag : nullptr, seq, bcap);
        HIPTS_LAUNCH_CHECK();
    }
    bm25->s1_dirty = false;
    if (via_copy) HIPTS_HIP(hipMemcpyAsync(hv, ov, out_bytes, hipMemcpyDeviceToHost, s));
    bool seen = false;
    if (use_flag) {
        // spin on the sequence number the last kernel releases after its result stores (bounded: then the ordinary synchronise)
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) {
                seen = true;
                break;
            }
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    if (!seen) HIPTS_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < k; ++i) {
        ids_out[i] = i < kk ? hi[i] : -1;
        vals_out[i] = i < kk ? hv[i] : -INFINITY;
    }
    return HIPTS_OK;
}

}  // namespace

extern "C" {

int hipts_query_profile_enable(hipts_bm25_t* h, int enable) {
    HIPTS_REQUIRE(h, "null handle");
    h->prof = enable != 0;
    if (enable) {
        for (int c = 0; c < QP_COUNT; ++c) {
            h->prof_ms[c] = 0.0;
            h->prof_bytes[c] = 0.0;
            h->prof_n[c] = 0;
        }
    }
    return HIPTS_OK;
}

int hipts_query_profile_read(hipts_bm25_t* h, int category, double* total_ms, int64_t* launches, double* total_bytes) {
    HIPTS_REQUIRE(h && category >= 0 && category < QP_COUNT, "hipts_query_profile_read: bad arguments");
    HIPTS_TRY(use_device(h->device));
    for (auto& r : h->prof_recs) {
        if (r.a && r.b) {
            HIPTS_HIP(hipEventSynchronize(r.b));
            float ms = 0.f;
            HIPTS_HIP(hipEventElapsedTime(&ms, r.a, r.b));
            h->prof_ms[r.cat] += ms;
            h->prof_n[r.cat] += 1;
            h->prof_bytes[r.cat] += r.bytes;
        }
        if (r.a) h->prof_pool.push_back(r.a);
        if (r.b) h->prof_pool.push_back(r.b);
    }
    h->prof_recs.clear();
    if (total_ms) *total_ms = h->prof_ms[category];
    if (launches) *launches = h->prof_n[category];
    if (total_bytes) *total_bytes = h->prof_bytes[category];
    return HIPTS_OK;
}

int hiptsdbg_topk_stamps(unsigned long long* host16) {
#ifdef HIPTS_X_TOPK_STAMPS
    HIPTS_HIP(hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_topk_stamps), 16 * 8));
    if (getenv("HIPTS_DBG_BM25_STAMPS"))      // slots 8..11 <- the BM25 kernel's clear / postings / mask+max (tools/bm25_stamps.py)
        HIPTS_HIP(hipMemcpyFromSymbol(host16 + 8, HIP_SYMBOL(g_bm25_stamps), 4 * 8));
    return HIPTS_OK;
#else
    (void)host16;
    return set_error(HIPTS_ERR_STATE, "built without HIPTS_X_TOPK_STAMPS");
#endif
}

int hiptsdbg_search1_last(hipts_bm25_t* h, uint32_t* candidates, uint32_t* took_candidate_path) {
    HIPTS_REQUIRE(h && h->s1_state.p && candidates && took_candidate_path, "hiptsdbg_search1_last: no one-query search has run on this handle");
    HIPTS_TRY(use_device(h->device));
    uint32_t dbg[4];
    HIPTS_HIP(hipMemcpy(dbg, reinterpret_cast<const char*>(h->s1_state.p) + offsetof(Search1State, dbg), sizeof(dbg), hipMemcpyDeviceToHost));
    *candidates = dbg[0];
    *took_candidate_path = dbg[1];
    return HIPTS_OK;
}

int hipts_query_profile_name(int category, char* buf, size_t n) {
    HIPTS_REQUIRE(buf && n > 0 && category >= 0 && category < QP_COUNT, "hipts_query_profile_name: bad arguments");
    snprintf(buf, n, "%s", kQueryProfNames[category]);
    return HIPTS_OK;
}

// ---------------------------------------------------------------------------------------------
int hipts_bm25_build(const int64_t* doc_ptr, const int32_t* term_ids, int64_t num_docs, int32_t vocab, int device,
                     hipts_bm25_t** out) {
    HIPTS_REQUIRE(doc_ptr && out && num_docs >= 0 && vocab >= 0, "hipts_bm25_build: bad arguments");
    HIPTS_REQUIRE(term_ids || doc_ptr[num_docs] == 0, "hipts_bm25_build: term_ids is NULL");
    HIPTS_TRY(use_device(device));
    auto* h = new hipts_bm25();
    h->device = device;
    h->D = num_docs;
    h->V = vocab;
    h->h_ptr.assign(1, 0);
    h->h_df.assign((size_t)vocab, 0);
    h->h_dl.reserve((size_t)num_docs);
    std::unordered_map<int32_t, int32_t> slot;   // term -> position in this document's list (dict insertion order)
    int64_t sum_dl = 0;
    for (int64_t d = 0; d < num_docs; ++d) {     // genmodel.py:57-73
        slot.clear();
        const size_t base = h->h_term.size();
        int64_t dl = 0;
        for (int64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
            const int32_t t = term_ids[i];
            if (t < 0 || t >= vocab) continue;   // tag not in dictionary (:59)
            ++dl;
            auto it = slot.find(t);
            if (it == slot.end()) {
                slot.emplace(t, (int32_t)(h->h_term.size() - base));
                h->h_term.push_back(t);
                h->h_tf.push_back(1);
            } else {
                h->h_tf[base + it->second] += 1;
            }
        }
        for (size_t i = base; i < h->h_term.size(); ++i) h->h_df[h->h_term[i]] += 1;
        h->h_dl.push_back(dl);
        sum_dl += dl;
        h->h_ptr.push_back((int64_t)h->h_term.size());
    }
    h->nnz = (int64_t)h->h_term.size();
    // np.mean of an int64 array: exact integer sum (< 2^53) then one division          genmodel.py:76
    h->avgdl = num_docs > 0 ? (double)sum_dl / (double)num_docs : std::numeric_limits<double>::quiet_NaN();
    h->h_idf.assign((size_t)vocab, 0.0);
    for (int32_t t = 0; t < vocab; ++t) {        // genmodel.py:80-82
        const int64_t df = h->h_df[t];
        if (df > 0) h->h_idf[t] = std::log(1 + ((double)(num_docs - df) + 0.5) / ((double)df + 0.5));
    }
    std::vector<int32_t> dl32((size_t)num_docs);
    for (int64_t d = 0; d < num_docs; ++d) dl32[d] = (int32_t)h->h_dl[d];
    // term-major postings (documents ascending inside a term) for the postings kernel
    std::vector<int64_t> tptr((size_t)vocab + 1, 0);
    for (int32_t t = 0; t < vocab; ++t) tptr[t + 1] = tptr[t] + h->h_df[t];
    std::vector<int32_t> tdoc((size_t)h->nnz), ttf((size_t)h->nnz);
    {
        std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
        for (int64_t d = 0; d < num_docs; ++d)
            for (int64_t i = h->h_ptr[d]; i < h->h_ptr[d + 1]; ++i) {
                const int64_t pos = fill[h->h_term[i]]++;
                tdoc[pos] = (int32_t)d;
                ttf[pos] = h->h_tf[i];
            }
    }
    // every term's posting offsets at BM25_SLICES fixed document boundaries (bm25_postings_sliced_kernel); the boundaries are multiples
    // of 8 documents so that a slice of a score row starts on a 64-byte line
    std::vector<int64_t> tslice;
    h->slice_docs = (num_docs % 8 == 0 && num_docs >= 8192) ? (num_docs / 8 / BM25_SLICES) * 8 : 0;
    if (h->slice_docs > 0) {
        tslice.resize((size_t)vocab * (BM25_SLICES + 1));
        for (int32_t t = 0; t < vocab; ++t) {
            const int32_t* first = tdoc.data() + tptr[t];
            const int32_t* last = tdoc.data() + tptr[t + 1];
            for (int sl = 0; sl <= BM25_SLICES; ++sl) {
                const int64_t bound = sl == BM25_SLICES ? num_docs : (int64_t)sl * h->slice_docs;
                tslice[(size_t)t * (BM25_SLICES + 1) + sl] = tptr[t] + (std::lower_bound(first, last, (int32_t)std::min<int64_t>(bound, INT32_MAX)) - first);
            }
        }
    }
    int st = HIPTS_OK;
    if (h->slice_docs > 0 && ((st = h->d_tslice.alloc(tslice.size() * 8)) || (st = upload(h->d_tslice.p, tslice.data(), tslice.size() * 8)))) {
        delete h;
        return st;
    }
    if ((st = h->d_ptr.alloc((size_t)(num_docs + 1) * 8)) || (st = h->d_term.alloc((size_t)h->nnz * 4)) ||
        (st = h->d_tf.alloc((size_t)h->nnz * 4)) || (st = h->d_dl.alloc((size_t)num_docs * 4)) ||
        (st = h->d_idf.alloc((size_t)vocab * 8)) || (st = upload(h->d_ptr.p, h->h_ptr.data(), (size_t)(num_docs + 1) * 8)) ||
        (st = upload(h->d_term.p, h->h_term.data(), (size_t)h->nnz * 4)) ||
        (st = upload(h->d_tf.p, h->h_tf.data(), (size_t)h->nnz * 4)) || (st = upload(h->d_dl.p, dl32.data(), (size_t)num_docs * 4)) ||
        (st = upload(h->d_idf.p, h->h_idf.data(), (size_t)vocab * 8)) || (st = h->d_tptr.alloc((size_t)(vocab + 1) * 8)) ||
        (st = h->d_tdoc.alloc((size_t)h->nnz * 4)) || (st = h->d_ttf.alloc((size_t)h->nnz * 4)) ||
        (st = upload(h->d_tptr.p, tptr.data(), (size_t)(vocab + 1) * 8)) || (st = upload(h->d_tdoc.p, tdoc.data(), (size_t)h->nnz * 4)) ||
        (st = upload(h->d_ttf.p, ttf.data(), (size_t)h->nnz * 4))) {
        delete h;
        return st;
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_bm25_destroy(hipts_bm25_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        if (h->side) {
            (void)hipStreamSynchronize(h->side);
            (void)hipStreamDestroy(h->side);
        }
        for (auto& sl : h->slot)
            if (sl.done) (void)hipEventDestroy(sl.done);
        if (h->ev_in) (void)hipEventDestroy(h->ev_in);
        if (h->ev_bm25) (void)hipEventDestroy(h->ev_bm25);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_bm25_info(const hipts_bm25_t* h, int64_t* num_docs, int64_t* nnz, int32_t* vocab, double* avgdl) {
    HIPTS_REQUIRE(h, "null handle");
    if (num_docs) *num_docs = h->D;
    if (nnz) *nnz = h->nnz;
    if (vocab) *vocab = h->V;
    if (avgdl) *avgdl = h->avgdl;
    return HIPTS_OK;
}

int hipts_bm25_export(const hipts_bm25_t* h, int64_t* csr_ptr, int32_t* csr_term, int32_t* csr_tf, int64_t* doc_len,
                      int64_t* df, double* idf) {
    HIPTS_REQUIRE(h, "null handle");
    if (csr_ptr) memcpy(csr_ptr, h->h_ptr.data(), h->h_ptr.size() * 8);
    if (csr_term) memcpy(csr_term, h->h_term.data(), h->h_term.size() * 4);
    if (csr_tf) memcpy(csr_tf, h->h_tf.data(), h->h_tf.size() * 4);
    if (doc_len) memcpy(doc_len, h->h_dl.data(), h->h_dl.size() * 8);
    if (df) memcpy(df, h->h_df.data(), h->h_df.size() * 8);
    if (idf) memcpy(idf, h->h_idf.data(), h->h_idf.size() * 8);
    return HIPTS_OK;
}

int hipts_bm25_set_idf(hipts_bm25_t* h, const double* idf) {
    HIPTS_REQUIRE(h && idf, "null argument");
    HIPTS_TRY(use_device(h->device));
    h->h_idf.assign(idf, idf + h->V);
    return upload(h->d_idf.p, h->h_idf.data(), (size_t)h->V * 8);
}

int hipts_bm25_set_avgdl(hipts_bm25_t* h, double avgdl) {
    HIPTS_REQUIRE(h, "null argument");
    HIPTS_REQUIRE(avgdl > 0.0, "hipts_bm25_set_avgdl: avgdl must be positive");
    h->avgdl = avgdl;
    return HIPTS_OK;
}

int hipts_bm25_score(hipts_bm25_t* h, const int32_t* q_terms, const double* q_weights, const int32_t* q_ptr, int nq,
                     double* scores_out, int out_memspace, void* stream) {
    HIPTS_REQUIRE(h && q_ptr && scores_out && nq >= 1, "hipts_bm25_score: bad arguments");
    HIPTS_TRY(use_device(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int32_t* qt;
    const double* qw;
    const int32_t* qp;
    HIPTS_TRY(stage_queries(h, q_terms, q_weights, q_ptr, nq, &qt, &qw, &qp, s));
    double* out_dev = scores_out;
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(h->ws_scores.reserve((size_t)nq * h->D * 8));
        out_dev = h->ws_scores.as<double>();
    }
    HIPTS_TRY(launch_bm25(h, qt, qw, qp, nq, out_dev, s));
    return copy_out(scores_out, out_dev, (size_t)nq * h->D * 8, out_memspace, s);
}

// ---------------------------------------------------------------------------------------------
int hipts_index_create(int dim, int64_t capacity, int device, hipts_index_t** out) {
    HIPTS_REQUIRE(out && dim >= 4 && dim % 4 == 0 && dim <= 1280, "hipts_index_create: dim must be a multiple of 4 and <= 1280");
    HIPTS_TRY(use_device(device));
    auto* h = new hipts_index();
    h->device = device;
    h->dim = dim;
    h->cap = capacity > 0 ? capacity : 1024;
    int st = h->rows.alloc((size_t)h->cap * dim * 4);
    if (st) {
        delete h;
        return st;
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_index_destroy(hipts_index_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_index_add(hipts_index_t* h, const float* rows, int64_t nrows, int rows_memspace) {
    HIPTS_REQUIRE(h && (rows || nrows == 0) && nrows >= 0, "hipts_index_add: bad arguments");
    HIPTS_TRY(use_device(h->device));
    if (h->len + nrows > h->cap) {
        int64_t ncap = std::max<int64_t>(h->cap * 2, h->len + nrows);
        DevBuf nb;
        HIPTS_TRY(nb.alloc((size_t)ncap * h->dim * 4));
        HIPTS_HIP(hipMemcpy(nb.p, h->rows.p, (size_t)h->len * h->dim * 4, hipMemcpyDeviceToDevice));
        std::swap(nb.p, h->rows.p);
        std::swap(nb.bytes, h->rows.bytes);
        h->cap = ncap;
    }
    HIPTS_HIP(hipMemcpy(h->rows.as<float>() + h->len * h->dim, rows, (size_t)nrows * h->dim * 4,
                        rows_memspace == HIPTS_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    const int64_t old_len = h->len;
    h->len += nrows;
    if (h->dim % 4 == 0 && nrows > 0) {
        // the tile-major copy the product kernel reads: size it like `rows`, then rewrite the tiles the new rows touch
        const size_t need = (size_t)((h->cap + 31) / 32) * 32 * h->dim * 4;
        if (h->tiled.bytes < need) {
            DevBuf nb;
            HIPTS_TRY(nb.alloc(need));
            if (h->tiled.p && old_len > 0)
                HIPTS_HIP(hipMemcpy(nb.p, h->tiled.p, (size_t)((old_len + 31) / 32) * 32 * h->dim * 4, hipMemcpyDeviceToDevice));
            std::swap(nb.p, h->tiled.p);
            std::swap(nb.bytes, h->tiled.bytes);
        }
        const int64_t t0 = old_len / 32, t1 = (h->len + 31) / 32;
        const int64_t total = (t1 - t0) * (h->dim / 4) * 32;
        retile_kernel<<<ceil_div(total, 256), 256>>>(h->rows.as<float>(), h->tiled.as<float4>(), h->len, h->dim, t0, t1);
        HIPTS_LAUNCH_CHECK();
        HIPTS_HIP(hipDeviceSynchronize());
    }
    return HIPTS_OK;
}

int hipts_index_len(const hipts_index_t* h, int64_t* nrows) {
    HIPTS_REQUIRE(h && nrows, "null argument");
    *nrows = h->len;
    return HIPTS_OK;
}

int hipts_index_vector_by_id(const hipts_index_t* h, int64_t id, float* out_host) {
    HIPTS_REQUIRE(h && out_host && id >= 0 && id < h->len, "hipts_index_vector_by_id: id out of range");
    HIPTS_TRY(use_device(h->device));
    HIPTS_HIP(hipMemcpy(out_host, h->rows.as<float>() + id * h->dim, (size_t)h->dim * 4, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}

int hipts_index_export(const hipts_index_t* h, int64_t first, int64_t nrows, float* out_host) {
    HIPTS_REQUIRE(h && first >= 0 && nrows >= 0 && first + nrows <= h->len, "hipts_index_export: rows [%lld, %lld) out of range (len %lld)",
                  (long long)first, (long long)(first + nrows), (long long)h->len);
    if (nrows == 0) return HIPTS_OK;
    HIPTS_REQUIRE(out_host, "hipts_index_export: null output");
    HIPTS_TRY(use_device(h->device));
    HIPTS_HIP(hipMemcpy(out_host, h->rows.as<float>() + first * h->dim, (size_t)nrows * h->dim * 4, hipMemcpyDeviceToHost));
    return HIPTS_OK;
}

int hipts_index_data(const hipts_index_t* h, void** device_ptr) {
    HIPTS_REQUIRE(h && device_ptr, "null argument");
    *device_ptr = h->rows.p;
    return HIPTS_OK;
}

int hipts_index_query(hipts_index_t* h, const float* queries, int queries_memspace, int nq, float* scores_out,
                      int out_memspace, void* stream) {
    HIPTS_REQUIRE(h && queries && scores_out && nq >= 1, "hipts_index_query: bad arguments");
    HIPTS_REQUIRE(h->len > 0, "hipts_index_query: empty index");
    HIPTS_TRY(use_device(h->device));
    hipStream_t s = (hipStream_t)stream;
    float* out_dev = scores_out;
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(h->ws_out.reserve((size_t)nq * h->len * 4));
        out_dev = h->ws_out.as<float>();
    }
    static const bool allow_one = !(getenv("HIPTS_SEARCH1") && strcmp(getenv("HIPTS_SEARCH1"), "0") == 0);
    if (nq == 1 && allow_one && queries_memspace != HIPTS_DEVICE && h->dim <= S1_MAX_DIM && h->dim % 4 == 0 && h->tiled.p && h->len >= S1_MIN_DOCS) {
        Sim1Query Q;                      // a host query goes straight into the kernel arguments: no staging copy
        memset(&Q, 0, sizeof(Q));
        Q.dim = h->dim;
        memcpy(Q.q, queries, (size_t)h->dim * 4);
        sim1_kernel<<<ceil_div(h->len, 128), 128, 0, s>>>(Q, h->tiled.as<float4>(), h->len, out_dev);
        HIPTS_LAUNCH_CHECK();
        return copy_out(scores_out, out_dev, (size_t)h->len * 4, out_memspace, s);
    }
    const float* q_dev = queries;
    if (queries_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(h->ws_q.reserve((size_t)nq * h->dim * 4));
        HIPTS_HIP(hipMemcpyAsync(h->ws_q.p, queries, (size_t)nq * h->dim * 4, hipMemcpyHostToDevice, s));
        q_dev = h->ws_q.as<float>();
    }
    HIPTS_TRY(launch_sim(h->rows.as<float>(), h->tiled.as<float>(), h->len, h->dim, q_dev, nq, out_dev, h->len, s));
    return copy_out(scores_out, out_dev, (size_t)nq * h->len * 4, out_memspace, s);
}

// ---------------------------------------------------------------------------------------------
int hipts_combine(const double* a, const float* b, int nq, int64_t n, double wa, double wb, int norm_a, int norm_b,
                  double* out, int device, void* stream) {
    HIPTS_REQUIRE(a && b && out && nq >= 1 && n >= 1, "hipts_combine: bad arguments");
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf& maxbuf = scratch_buf(device, 0);
    HIPTS_TRY(maxbuf.reserve((size_t)nq * 16));
    double* ma = maxbuf.as<double>();
    float* mb = reinterpret_cast<float*>(ma + nq);
    if (norm_a) {
        rowmax_kernel<double><<<nq, 1024, 0, s>>>(a, n, ma);
        HIPTS_LAUNCH_CHECK();
    }
    if (norm_b) {
        rowmax_kernel<float><<<nq, 1024, 0, s>>>(b, n, mb);
        HIPTS_LAUNCH_CHECK();
    }
    dim3 grid(ceil_div(n, 256), nq);
    combine_kernel<<<grid, 256, 0, s>>>(a, b, n, wa, (float)wb, norm_a ? ma : nullptr, norm_b ? mb : nullptr, nullptr, out);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

int hipts_rowmax(const double* a, const float* b, int nq, int64_t n, double* max_a_out, float* max_b_out, int device,
                 void* stream) {
    HIPTS_REQUIRE(nq >= 1 && n >= 1 && (!a || max_a_out) && (!b || max_b_out), "hipts_rowmax: bad arguments");
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    if (a) {
        rowmax_kernel<double><<<nq, 1024, 0, s>>>(a, n, max_a_out);
        HIPTS_LAUNCH_CHECK();
    }
    if (b) {
        rowmax_kernel<float><<<nq, 1024, 0, s>>>(b, n, max_b_out);
        HIPTS_LAUNCH_CHECK();
    }
    return HIPTS_OK;
}

int hipts_combine_with_max(const double* a, const float* b, int nq, int64_t n, double wa, double wb, const double* max_a,
                           const float* max_b, double* out, int device, void* stream) {
    HIPTS_REQUIRE(a && b && out && nq >= 1 && n >= 1, "hipts_combine_with_max: bad arguments");
    HIPTS_TRY(use_device(device));
    dim3 grid(ceil_div(n, 256), nq);
    combine_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a, b, n, wa, (float)wb, max_a, max_b, nullptr, out);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

int hipts_topk(const double* vals, int nq, int64_t n, int k, int32_t* ids_out, double* vals_out, int out_memspace,
               int device, void* stream) {
    HIPTS_REQUIRE(vals && ids_out && vals_out && nq >= 1 && n >= 1, "hipts_topk: bad arguments");
    HIPTS_REQUIRE(k >= 1 && k <= TOPK_MAX_K, "hipts_topk: k must be in [1, %d]", TOPK_MAX_K);
    HIPTS_REQUIRE(n < (1ll << 32), "hipts_topk: n too large");
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    const int kk = (int)std::min<int64_t>(k, n);
    if (out_memspace == HIPTS_DEVICE) {
        HIPTS_REQUIRE(kk == k, "hipts_topk: k > n needs host outputs");
        topk_kernel<<<nq, 1024, 0, s>>>(vals, n, k, ids_out, vals_out);
        HIPTS_LAUNCH_CHECK();
        return HIPTS_OK;
    }
    DevBuf& obuf = scratch_buf(device, 1);
    HIPTS_TRY(obuf.reserve((size_t)nq * kk * 12 + 64));
    double* ov = obuf.as<double>();
    int32_t* oi = reinterpret_cast<int32_t*>(ov + (size_t)nq * kk);
    topk_kernel<<<nq, 1024, 0, s>>>(vals, n, kk, oi, ov);
    HIPTS_LAUNCH_CHECK();
    std::vector<int32_t> hi((size_t)nq * kk);
    std::vector<double> hv((size_t)nq * kk);
    HIPTS_HIP(hipMemcpyAsync(hi.data(), oi, hi.size() * 4, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipMemcpyAsync(hv.data(), ov, hv.size() * 8, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipStreamSynchronize(s));
    for (int q = 0; q < nq; ++q)
        for (int i = 0; i < k; ++i) {
            ids_out[(size_t)q * k + i] = i < kk ? hi[(size_t)q * kk + i] : -1;
            vals_out[(size_t)q * k + i] = i < kk ? hv[(size_t)q * kk + i] : -INFINITY;
        }
    return HIPTS_OK;
}

// scores ranked at or before (after_val, after_id) in the order (value descending, index ascending) become -inf: the top k of the
// result are the NEXT k of that order
__global__ __launch_bounds__(256) void mask_ranked_before_kernel(const double* __restrict__ vals, int64_t n, double after_val, int64_t after_id,
                                                                 double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = vals[i];
    const bool later = v < after_val || (v == after_val && i > after_id);
    out[i] = later ? v : -INFINITY;
}

int hipts_topk_after(const double* vals, int64_t n, int k, double after_val, int64_t after_id, int32_t* ids_out, double* vals_out,
                     int device, void* stream) {
    HIPTS_REQUIRE(vals && ids_out && vals_out && n >= 1 && k >= 1 && k <= TOPK_MAX_K, "hipts_topk_after: bad arguments");
    HIPTS_REQUIRE(after_val == after_val, "hipts_topk_after: after_val is NaN");
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf& mbuf = scratch_buf(device, 2);
    HIPTS_TRY(mbuf.reserve((size_t)n * 8));
    mask_ranked_before_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(vals, n, after_val, after_id, mbuf.as<double>());
    HIPTS_LAUNCH_CHECK();
    return hipts_topk(mbuf.as<double>(), 1, n, k, ids_out, vals_out, HIPTS_HOST, device, stream);
}

// One batch of queries: pack, copy, launch everything and the copy back into slot `sl`'s pinned buffer, record its event.  Does not wait.
static int search_submit_impl(hipts_bm25_t* bm25, hipts_index_t* index, const int32_t* q_terms, const double* q_weights, const int32_t* q_ptr,
                              const float* q_vectors, int nq, double w_bm25, double w_sim, int k, double* final_out_device, int sl, hipStream_t s) {
    HIPTS_REQUIRE(bm25 && index && q_ptr && q_vectors && nq >= 1, "hipts_search: bad arguments");
    HIPTS_REQUIRE(sl == 0 || sl == 1, "hipts_search_submit: slot must be 0 or 1");
    HIPTS_REQUIRE(bm25->device == index->device, "hipts_search: handles live on different devices");
    HIPTS_REQUIRE(bm25->D == index->len, "hipts_search: BM25 corpus has %lld documents, index has %lld rows",
                  (long long)bm25->D, (long long)index->len);
    HIPTS_TRY(use_device(bm25->device));
    hipts_bm25::SearchSlot& S = bm25->slot[sl];
    HIPTS_REQUIRE(!S.pending, "hipts_search_submit: slot %d still holds an uncollected batch", sl);
    // ws_q / ws_scores / ws_sims / ws_out are shared by the two slots: stream order is what keeps batch i + 1 off batch i's rows
    HIPTS_REQUIRE(!bm25->slot[sl ^ 1].pending || bm25->slot[sl ^ 1].stream == s,
                  "hipts_search_submit: slot %d holds a batch submitted on another stream (all batches of a handle use ONE stream)", sl ^ 1);
    S.stream = s;
    const int64_t D = bm25->D;
    HIPTS_REQUIRE(k >= 1 && k <= TOPK_MAX_K, "hipts_search: k must be in [1, %d]", TOPK_MAX_K);
    const int nt = q_ptr[nq];
    HIPTS_REQUIRE(q_ptr[0] == 0 && nt >= 0, "q_ptr must start at 0 and be non-decreasing");
    S.nq = nq;
    S.k = k;
    static const bool allow_one = !(getenv("HIPTS_SEARCH1") && strcmp(getenv("HIPTS_SEARCH1"), "0") == 0);       // A/B switch
    if (nq == 1 && allow_one && nt <= S1_MAX_TERMS && index->dim <= S1_MAX_DIM && index->dim % 4 == 0 && index->tiled.p && D >= S1_MIN_DOCS) {
        // the one-query path completes inside the call (its last kernel stores into pinned memory and the host spins on a flag)
        S.ids1.resize((size_t)k);
        S.vals1.resize((size_t)k);
        HIPTS_TRY(search_one(bm25, index, q_terms, q_weights, nt, q_vectors, w_bm25, w_sim, k, S.ids1.data(), S.vals1.data(), final_out_device, s));
        S.host_result = true;
        S.pending = true;
        return HIPTS_OK;
    }
    S.host_result = false;
    if (!S.done) HIPTS_HIP(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
    // queries, weights, offsets and the query vectors travel as ONE packed copy from pinned memory (no bounce buffer, no
    // synchronisation before the kernels); the results come back the same way
    const size_t off_w = ((size_t)nt * 4 + 15) / 16 * 16;
    const size_t off_p = off_w + (size_t)nt * 8;
    const size_t off_v = (off_p + (size_t)(nq + 1) * 4 + 15) / 16 * 16;
    const size_t in_bytes = off_v + (size_t)nq * index->dim * 4;
    HIPTS_TRY(S.pin_in.reserve(in_bytes));
    HIPTS_TRY(bm25->ws_q.reserve(in_bytes));
    char* hin = S.pin_in.as<char>();
    if (nt) {
        memcpy(hin, q_terms, (size_t)nt * 4);
        memcpy(hin + off_w, q_weights, (size_t)nt * 8);
    }
    memcpy(hin + off_p, q_ptr, (size_t)(nq + 1) * 4);
    memcpy(hin + off_v, q_vectors, (size_t)nq * index->dim * 4);
    HIPTS_HIP(hipMemcpyAsync(bm25->ws_q.p, hin, in_bytes, hipMemcpyHostToDevice, s));
    const int32_t* qt = bm25->ws_q.as<int32_t>();
    const double* qw = reinterpret_cast<const double*>(bm25->ws_q.as<char>() + off_w);
    const int32_t* qp = reinterpret_cast<const int32_t*>(bm25->ws_q.as<char>() + off_p);
    const float* qvec = reinterpret_cast<const float*>(bm25->ws_q.as<char>() + off_v);
    HIPTS_TRY(bm25->ws_scores.reserve((size_t)nq * D * 8));
    HIPTS_TRY(bm25->ws_sims.reserve((size_t)nq * D * 4));
    // Without a caller who wants the combined rows they are never written: the top-k kernel combines on the fly (TopkFused;
    // HIPTS_SEARCH_FUSED_TOPK=0 restores combine_kernel + top-k over the stored rows, for A/B).
    static const bool fuse_topk = !(getenv("HIPTS_SEARCH_FUSED_TOPK") && atoi(getenv("HIPTS_SEARCH_FUSED_TOPK")) == 0);
    const bool fused = fuse_topk && final_out_device == nullptr;
    double* final_dev = final_out_device;
    if (!final_dev && !fused) {
        HIPTS_TRY(bm25->ws_final.reserve((size_t)nq * D * 8));
        final_dev = bm25->ws_final.as<double>();
    }
    // row maxima (webui.py:377-380): the BM25 workgroup reduces its own row while it still owns it; the
    // index product's maximum is a separate pass (folding it into the product with one atomicMax per
    // tile and query was measured 3.5x SLOWER: 200k atomics on 32 addresses).
    HIPTS_TRY(bm25->ws_max.reserve((size_t)nq * 16));
    double* ma = bm25->ws_max.as<double>();
    float* mb = reinterpret_cast<float*>(ma + nq);
    // BM25 and the index product share no data.  Round 4 measured them on two streams (HIPTS_SEARCH_OVERLAP=1): 396-398 k queries/s either
    // way -- side by side BM25 takes 234-336 us instead of 156-167 and the product 262-363 instead of 233: both fill the wave slots of
    // every CU, so they take turns rather than share.  One stream stays the default.
    static const bool overlap = getenv("HIPTS_SEARCH_OVERLAP") && atoi(getenv("HIPTS_SEARCH_OVERLAP")) != 0;
    int parts_grid = 0;          // > 0: the index product left per-workgroup row maxima in ws_parts
    hipStream_t sb = s;
    if (overlap) {
        if (!bm25->side) {
            HIPTS_HIP(hipStreamCreateWithFlags(&bm25->side, hipStreamNonBlocking));
            HIPTS_HIP(hipEventCreateWithFlags(&bm25->ev_in, hipEventDisableTiming));
            HIPTS_HIP(hipEventCreateWithFlags(&bm25->ev_bm25, hipEventDisableTiming));
        }
        sb = bm25->side;
        HIPTS_HIP(hipEventRecord(bm25->ev_in, s));              // the packed queries are on the device
        HIPTS_HIP(hipStreamWaitEvent(sb, bm25->ev_in, 0));
    }
    {
        // algorithmic bytes: the posting lists of the queries' terms (8 B per entry + the document length gathered with it) and the
        // three passes over each query's score row (clear, mask / maximum) -- DESIGN.md section 4
        double pb = 0.0;
        for (int j = 0; j < nt; ++j)
            if (q_terms[j] >= 0 && q_terms[j] < bm25->V) pb += (double)bm25->h_df[q_terms[j]] * 12.0;
        QueryProfScope ps(bm25, sb, QP_BM25, pb + (double)nq * D * 8.0 * 3);
        HIPTS_TRY(launch_bm25(bm25, qt, qw, qp, nq, bm25->ws_scores.as<double>(), sb, ma));
    }
    if (overlap) HIPTS_HIP(hipEventRecord(bm25->ev_bm25, sb));
    {
        QueryProfScope ps(bm25, s, QP_SIM, (double)sim_index_passes(index->tiled.p != nullptr, index->dim, nq) * D * index->dim * 4.0 + (double)nq * D * 4.0);
        if (fused && sim_parts_enabled()) HIPTS_TRY(bm25->ws_parts.reserve((size_t)((nq + 255) / 256) * SIMW_MAX_GRID * 256 * sizeof(float)));
        HIPTS_TRY(launch_sim(index->rows.as<float>(), index->tiled.as<float>(), D, index->dim, qvec, nq, bm25->ws_sims.as<float>(), D, s,
                             fused && sim_parts_enabled() ? bm25->ws_parts.as<float>() : nullptr, &parts_grid));
    }
    if (parts_grid == 0) {       // (otherwise the maxima came out of the product kernel's accumulators: the top-k kernel reduces its workgroups' values)
        QueryProfScope ps(bm25, s, QP_ROWMAX, (double)nq * D * 4.0);
        rowmax_kernel<float><<<nq, 1024, 0, s>>>(bm25->ws_sims.as<float>(), D, mb);
        HIPTS_LAUNCH_CHECK();
    }
    if (overlap) HIPTS_HIP(hipStreamWaitEvent(s, bm25->ev_bm25, 0));      // BM25 rows and their maxima are complete
    if (!fused) {
        QueryProfScope ps(bm25, s, QP_COMBINE, (double)nq * D * 20.0);
        dim3 grid(ceil_div(D, 256), nq);
        combine_kernel<<<grid, 256, 0, s>>>(bm25->ws_scores.as<double>(), bm25->ws_sims.as<float>(), D, w_bm25, (float)w_sim, ma,
                                            mb, nullptr, final_dev);
        HIPTS_LAUNCH_CHECK();
    }
    const int kk = (int)std::min<int64_t>(k, D);
    S.kk = kk;
    const size_t out_bytes = (size_t)nq * kk * 12;
    HIPTS_TRY(bm25->ws_out.reserve(out_bytes + 64));
    HIPTS_TRY(S.pin_out.reserve(out_bytes + 64));
    double* ov = bm25->ws_out.as<double>();
    int32_t* oi = reinterpret_cast<int32_t*>(ov + (size_t)nq * kk);
    {
        QueryProfScope ps(bm25, s, QP_TOPK, (double)nq * D * (fused ? 12.0 : 8.0));
        if (fused) {
            TopkFused fz;
            fz.a = bm25->ws_scores.as<double>();
            fz.b = bm25->ws_sims.as<float>();
            fz.wa = w_bm25;
            fz.wb = (float)w_sim;
            fz.max_a = ma;
            fz.max_b = mb;
            if (parts_grid > 0) {
                fz.max_b = nullptr;
                fz.max_b_parts = bm25->ws_parts.as<float>();
                fz.parts = parts_grid;
            }
            topk_kernel<<<nq, 1024, 0, s>>>(nullptr, D, kk, oi, ov, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0u, S1_BLOCK_CAP, fz);
        } else {
            topk_kernel<<<nq, 1024, 0, s>>>(final_dev, D, kk, oi, ov);
        }
        HIPTS_LAUNCH_CHECK();
    }
    HIPTS_HIP(hipMemcpyAsync(S.pin_out.p, ov, out_bytes, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipEventRecord(S.done, s));
    S.pending = true;
    return HIPTS_OK;
}

static int search_collect_impl(hipts_bm25_t* bm25, int sl, int32_t* ids_out, double* vals_out) {
    HIPTS_REQUIRE(bm25 && ids_out && vals_out && (sl == 0 || sl == 1), "hipts_search_collect: bad arguments");
    hipts_bm25::SearchSlot& S = bm25->slot[sl];
    HIPTS_REQUIRE(S.pending, "hipts_search_collect: slot %d holds no submitted batch", sl);
    S.pending = false;
    const int k = S.k;
    if (S.host_result) {
        memcpy(ids_out, S.ids1.data(), (size_t)k * 4);
        memcpy(vals_out, S.vals1.data(), (size_t)k * 8);
        return HIPTS_OK;
    }
    HIPTS_TRY(use_device(bm25->device));
    HIPTS_HIP(hipEventSynchronize(S.done));
    const int nq = S.nq, kk = S.kk;
    const double* hv = S.pin_out.as<double>();
    const int32_t* hi = reinterpret_cast<const int32_t*>(hv + (size_t)nq * kk);
    if (kk == k) {
        memcpy(ids_out, hi, (size_t)nq * k * 4);
        memcpy(vals_out, hv, (size_t)nq * k * 8);
    } else {
        for (int q = 0; q < nq; ++q)
            for (int i = 0; i < k; ++i) {
                ids_out[(size_t)q * k + i] = i < kk ? hi[(size_t)q * kk + i] : -1;
                vals_out[(size_t)q * k + i] = i < kk ? hv[(size_t)q * kk + i] : -INFINITY;
            }
    }
    return HIPTS_OK;
}

int hipts_search(hipts_bm25_t* bm25, hipts_index_t* index, const int32_t* q_terms, const double* q_weights,
                 const int32_t* q_ptr, const float* q_vectors, int nq, double w_bm25, double w_sim, int k, int32_t* ids_out,
                 double* vals_out, double* final_out_device, void* stream) {
    HIPTS_REQUIRE(bm25 && ids_out && vals_out, "hipts_search: bad arguments");
    HIPTS_REQUIRE(!bm25->slot[0].pending, "hipts_search: a submitted batch waits in slot 0 (hipts_search_collect it first)");
    HIPTS_TRY(search_submit_impl(bm25, index, q_terms, q_weights, q_ptr, q_vectors, nq, w_bm25, w_sim, k, final_out_device, 0, (hipStream_t)stream));
    return search_collect_impl(bm25, 0, ids_out, vals_out);
}

int hipts_search_submit(hipts_bm25_t* bm25, hipts_index_t* index, const int32_t* q_terms, const double* q_weights, const int32_t* q_ptr,
                        const float* q_vectors, int nq, double w_bm25, double w_sim, int k, int slot, void* stream) {
    return search_submit_impl(bm25, index, q_terms, q_weights, q_ptr, q_vectors, nq, w_bm25, w_sim, k, nullptr, slot, (hipStream_t)stream);
}

int hipts_search_collect(hipts_bm25_t* bm25, int slot, int32_t* ids_out, double* vals_out) {
    return search_collect_impl(bm25, slot, ids_out, vals_out);
}

}  // extern "C"
