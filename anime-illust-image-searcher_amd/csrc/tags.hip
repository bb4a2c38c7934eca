// tags.hip -- per-image tag selection on the device.
//
// Replaces the numpy/Python post-processing of Predictor.predict (tagging.py:185-227) and
// mcut_threshold (tagging.py:61-66): probabilities are widened to float64 (:186), each category
// (general = 0, character = 4; :137-139) gets its Maximum-Cut threshold = midpoint of the first
// largest gap between consecutive sorted probabilities, characters additionally max(0.15, t)
// (:201), labels with p > t (strict) are kept and emitted by descending probability, ties in
// label order (Python's sorted(reverse=True) is stable).
//
// One workgroup per image.  The category's (probability, position) pairs are sorted once in LDS
// (bitonic, 64-bit composite keys: probability descending, position ascending); the kept labels
// are then exactly a prefix of that order, so threshold, count and output order all come from the
// one sort.  float32 -> float64 widening is exact and order preserving, so sorting the float32
// bit patterns equals sorting the float64 values; differences, the argmax and the comparisons
// against the threshold are done in float64 like the reference.  LDS-bound, ~43 KB read per image.
#include <vector>

#include "common.h"

using namespace hipts;

struct hipts_tagsel {
    int device = 0;
    int C = 0, max_batch = 0;
    int ng = 0, nc = 0, npad = 0;
    DevBuf gidx, cidx;                       // int32 label ids of each category, ascending
    DevBuf ws_probs, ws_counts, ws_ids, ws_thresh;
    int ws_row_cap = 0;
};

namespace {

__device__ __forceinline__ uint32_t order_key32(float x) {
    if (x == 0.0f) x = 0.0f;
    const uint32_t u = __float_as_uint(x);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key32_value(uint32_t k) {
    const uint32_t u = (k >> 31) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// sorts keys[0..npad) descending
__device__ void bitonic_desc(uint64_t* keys, int npad) {
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) != desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// One category of one image.  Returns (through shared) count and threshold; writes label ids.
__device__ void select_category(const float* __restrict__ probs, const int32_t* __restrict__ idx, int n, int npad,
                                bool mcut, double thresh_in, bool floor015, uint64_t* keys, double* sh_d, int* sh_i,
                                int32_t* __restrict__ ids_out, int cap, int* count_out, double* thresh_out) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < npad; i += nthr) {
        uint64_t k = 0;
        if (i < n) k = ((uint64_t)order_key32(probs[idx[i]]) << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
        keys[i] = k;
    }
    __syncthreads();
    bitonic_desc(keys, npad);
    double thresh = thresh_in;
    if (mcut && n >= 2) {
        // difs[i] = s[i] - s[i+1]; t = first argmax                                   tagging.py:63-64
        double best = -INFINITY;
        int besti = 0x7fffffff;
        for (int i = tid; i < n - 1; i += nthr) {
            const double a = (double)key32_value((uint32_t)(keys[i] >> 32));
            const double b = (double)key32_value((uint32_t)(keys[i + 1] >> 32));
            const double dif = a - b;
            if (dif > best) {
                best = dif;
                besti = i;
            }
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(besti, o);
            if (ob > best || (ob == best && oi < besti)) {
                best = ob;
                besti = oi;
            }
        }
        if ((tid & 63) == 0) {
            sh_d[tid >> 6] = best;
            sh_i[tid >> 6] = besti;
        }
        __syncthreads();
        if (tid == 0) {
            double b = sh_d[0];
            int bi = sh_i[0];
            for (int w = 1; w < (nthr >> 6); ++w)
                if (sh_d[w] > b || (sh_d[w] == b && sh_i[w] < bi)) {
                    b = sh_d[w];
                    bi = sh_i[w];
                }
            const double a0 = (double)key32_value((uint32_t)(keys[bi] >> 32));
            const double a1 = (double)key32_value((uint32_t)(keys[bi + 1] >> 32));
            double t = (a0 + a1) / 2;                                                // :65
            if (floor015 && !(t > 0.15)) t = 0.15;                                   // :201 max(0.15, t)
            sh_d[16] = t;
        }
        __syncthreads();
        thresh = sh_d[16];
        __syncthreads();
    } else if (mcut && floor015 && !(thresh > 0.15)) {
        thresh = 0.15;
    }
    // kept labels are a prefix of the sorted order: count entries with p > thresh
    int cnt = 0;
    for (int i = tid; i < n; i += nthr)
        if ((double)key32_value((uint32_t)(keys[i] >> 32)) > thresh) ++cnt;
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) sh_i[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int w = 0; w < (nthr >> 6); ++w) c += sh_i[w];
        sh_i[16] = c;
    }
    __syncthreads();
    const int total = sh_i[16];
    for (int i = tid; i < total && i < cap; i += nthr)
        ids_out[i] = idx[0xffffffffu - (uint32_t)(keys[i] & 0xffffffffu)];
    if (tid == 0) {
        *count_out = total;
        *thresh_out = thresh;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void tagsel_kernel(const float* __restrict__ probs, int C, const int32_t* __restrict__ gidx,
                                                      int ng, const int32_t* __restrict__ cidx, int nc, int npad,
                                                      double g_thresh, int g_mcut, double c_thresh, int c_mcut,
                                                      int32_t* __restrict__ counts, int counts_stride,
                                                      int32_t* __restrict__ ids, int ids_stride, int row_cap,
                                                      double* __restrict__ thresh) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    __shared__ double sh_d[17];
    __shared__ int sh_i[17];
    __shared__ int sh_cnt[2];
    __shared__ double sh_thr[2];
    const int b = blockIdx.x;
    const float* __restrict__ row = probs + (int64_t)b * C;
    int32_t* __restrict__ out = ids + (int64_t)b * ids_stride;
    int npg = 64;
    while (npg < ng) npg <<= 1;
    int npc = 64;
    while (npc < nc) npc <<= 1;
    select_category(row, gidx, ng, npg, g_mcut != 0, g_thresh, false, keys, sh_d, sh_i, out, row_cap, &sh_cnt[0], &sh_thr[0]);
    const int n_g = sh_cnt[0];
    const int used = n_g < row_cap ? n_g : row_cap;
    select_category(row, cidx, nc, npc, c_mcut != 0, c_thresh, true, keys, sh_d, sh_i, out + used, row_cap - used, &sh_cnt[1],
                    &sh_thr[1]);
    if (threadIdx.x == 0) {
        counts[(int64_t)b * counts_stride] = sh_cnt[0];
        counts[(int64_t)b * counts_stride + 1] = sh_cnt[1];
        if (thresh) {
            thresh[2 * b] = sh_thr[0];
            thresh[2 * b + 1] = sh_thr[1];
        }
    }
}

}  // namespace

extern "C" {

int hipts_tagsel_create(const int32_t* category, int num_classes, int device, int max_batch, hipts_tagsel_t** out) {
    HIPTS_REQUIRE(category && out && num_classes >= 1 && max_batch >= 1, "hipts_tagsel_create: bad arguments");
    HIPTS_TRY(use_device(device));
    std::vector<int32_t> g, c;
    for (int i = 0; i < num_classes; ++i) {
        if (category[i] == 0) g.push_back(i);        // tagging.py:138
        else if (category[i] == 4) c.push_back(i);   // tagging.py:139
    }
    int npad = 64;
    while (npad < (int)g.size() || npad < (int)c.size()) npad <<= 1;
    HIPTS_REQUIRE((size_t)npad * 8 <= 144 * 1024, "hipts_tagsel_create: a category with %zu labels does not fit the LDS sort",
                  g.size() > c.size() ? g.size() : c.size());
    auto* h = new hipts_tagsel();
    h->device = device;
    h->C = num_classes;
    h->max_batch = max_batch;
    h->ng = (int)g.size();
    h->nc = (int)c.size();
    h->npad = npad;
    int st;
    if ((st = h->gidx.alloc(g.size() * 4)) || (st = h->cidx.alloc(c.size() * 4)) || (st = upload(h->gidx.p, g.data(), g.size() * 4)) ||
        (st = upload(h->cidx.p, c.data(), c.size() * 4))) {
        delete h;
        return st;
    }
    hipError_t e = hipFuncSetAttribute((const void*)tagsel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) {
        delete h;
        return set_error(HIPTS_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return HIPTS_OK;
}

int hipts_tagsel_destroy(hipts_tagsel_t* h) {
    if (h) {
        (void)hipSetDevice(h->device);
        delete h;
    }
    return HIPTS_OK;
}

int hipts_tagsel_run(hipts_tagsel_t* h, const float* probs, int probs_memspace, int batch, double general_thresh,
                     int general_mcut, double character_thresh, int character_mcut, int32_t* counts_out, int32_t* ids_out,
                     int row_cap, double* thresh_out, int out_memspace, void* stream) {
    HIPTS_REQUIRE(h && probs && counts_out && ids_out && batch >= 1 && row_cap >= 1, "hipts_tagsel_run: bad arguments");
    HIPTS_TRY(use_device(h->device));
    hipStream_t s = (hipStream_t)stream;
    const float* p_dev = probs;
    if (probs_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(h->ws_probs.reserve((size_t)batch * h->C * 4));
        HIPTS_HIP(hipMemcpyAsync(h->ws_probs.p, probs, (size_t)batch * h->C * 4, hipMemcpyHostToDevice, s));
        p_dev = h->ws_probs.as<float>();
    }
    int32_t* cnt_dev = counts_out;
    int32_t* ids_dev = ids_out;
    double* thr_dev = thresh_out;
    const bool host_out = out_memspace != HIPTS_DEVICE;
    if (host_out || !thresh_out) {
        HIPTS_TRY(h->ws_thresh.reserve((size_t)batch * 16));
        thr_dev = h->ws_thresh.as<double>();
    }
    if (host_out) {
        HIPTS_TRY(h->ws_counts.reserve((size_t)batch * 8));
        HIPTS_TRY(h->ws_ids.reserve((size_t)batch * row_cap * 4));
        cnt_dev = h->ws_counts.as<int32_t>();
        ids_dev = h->ws_ids.as<int32_t>();
    }
    tagsel_kernel<<<batch, 1024, (size_t)h->npad * 8, s>>>(p_dev, h->C, h->gidx.as<int32_t>(), h->ng, h->cidx.as<int32_t>(), h->nc,
                                                           h->npad, general_thresh, general_mcut, character_thresh,
                                                           character_mcut, cnt_dev, 2, ids_dev, row_cap, row_cap, thr_dev);
    HIPTS_LAUNCH_CHECK();
    if (host_out) {
        HIPTS_HIP(hipMemcpyAsync(counts_out, cnt_dev, (size_t)batch * 8, hipMemcpyDeviceToHost, s));
        HIPTS_HIP(hipMemcpyAsync(ids_out, ids_dev, (size_t)batch * row_cap * 4, hipMemcpyDeviceToHost, s));
        if (thresh_out) HIPTS_HIP(hipMemcpyAsync(thresh_out, thr_dev, (size_t)batch * 16, hipMemcpyDeviceToHost, s));
        HIPTS_HIP(hipStreamSynchronize(s));
    }
    return HIPTS_OK;
}

int hipts_tagsel_run_rows(hipts_tagsel_t* h, const float* probs_device, int batch, double general_thresh, int general_mcut,
                          double character_thresh, int character_mcut, int32_t* rows_device, int row_width, void* stream) {
    HIPTS_REQUIRE(h && probs_device && rows_device && batch >= 1 && row_width >= 3, "hipts_tagsel_run_rows: bad arguments");
    HIPTS_TRY(use_device(h->device));
    tagsel_kernel<<<batch, 1024, (size_t)h->npad * 8, (hipStream_t)stream>>>(
        probs_device, h->C, h->gidx.as<int32_t>(), h->ng, h->cidx.as<int32_t>(), h->nc, h->npad, general_thresh, general_mcut,
        character_thresh, character_mcut, rows_device, row_width, rows_device + 2, row_width, row_width - 2, nullptr);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // extern "C"
