// metric.hip -- pairwise CCIP differences of a batch of features.  Stands where gen_cfeatures.py:257-274 (ccip_batch_differences)
// runs its metric model: `output, = metric_model.run(['output'], {'input': features[N, 768]})` -> float32 [N, N].
// The reference's metric graph is an opaque ONNX file fetched from the hub (not on disk here); BASELINE.json configs[4] restates the
// difference as the cosine form, which is kind 0 below: rows unit-normalised in float32 (x / sqrt(sum x^2), the expression of
// hiptagsearch.cfeatures.CharacterFeatureIndex.add_features), Gram matrix as the k-ordered fmaf chain (the index product's definition,
// bit-equal to oracle/csrc/oracle.c::orc_sim_chain), difference = 1 - cosine.  A real metric head takes its place behind the same
// signature when a checkpoint is present (`kind` is the switch).
#include "common.h"

#include "../../include/hip_tagsearch.h"

namespace hipts {
namespace {

// one wave per row: the float32 sum of squares in index order (numpy's pairwise sum differs: the caller normalises with THIS routine's
// definition, see the oracle restatement), then x / norm
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* __restrict__ x, float* __restrict__ u, int n, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* p = x + (size_t)row * dim;
    float s = 0.f;
    if (lane == 0)
        for (int k = 0; k < dim; ++k) s = fmaf(p[k], p[k], s);        // sequential chain: one defined order
    s = __shfl(s, 0);
    const float nrm = sqrtf(s);
    for (int k = lane; k < dim; k += 64) u[(size_t)row * dim + k] = nrm > 0.f ? p[k] / nrm : p[k];
}

__global__ __launch_bounds__(256) void gram_diff_kernel(const float* __restrict__ u, float* __restrict__ out, int n, int dim) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx - (int64_t)i * n);
    const float* a = u + (size_t)i * dim;
    const float* b = u + (size_t)j * dim;
    float acc = 0.f;
    for (int k = 0; k < dim; ++k) acc = fmaf(a[k], b[k], acc);
    out[idx] = 1.0f - acc;
}

}  // namespace
}  // namespace hipts

using namespace hipts;

extern "C" int hipts_ccip_metric(const float* features, int features_memspace, int n, int dim, int kind, float* diff_out, int out_memspace,
                                 int device, void* stream) {
    HIPTS_REQUIRE(features && diff_out && n >= 1 && dim >= 1, "hipts_ccip_metric: bad arguments");
    HIPTS_REQUIRE(kind == 0, "hipts_ccip_metric: kind %d is not available (0 = 1 - cosine; the reference's metric head needs its checkpoint)", kind);
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf in, unit, out;
    const float* fp = features;
    if (features_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(in.alloc((size_t)n * dim * 4));
        HIPTS_HIP(hipMemcpyAsync(in.p, features, (size_t)n * dim * 4, hipMemcpyHostToDevice, s));
        fp = in.as<float>();
    }
    HIPTS_TRY(unit.alloc((size_t)n * dim * 4));
    float* op = diff_out;
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(out.alloc((size_t)n * n * 4));
        op = out.as<float>();
    }
    unit_rows_kernel<<<(n + 3) / 4, 256, 0, s>>>(fp, unit.as<float>(), n, dim);
    HIPTS_LAUNCH_CHECK();
    gram_diff_kernel<<<(unsigned)(((int64_t)n * n + 255) / 256), 256, 0, s>>>(unit.as<float>(), op, n, dim);
    HIPTS_LAUNCH_CHECK();
    if (out_memspace != HIPTS_DEVICE) HIPTS_HIP(hipMemcpyAsync(diff_out, op, (size_t)n * n * 4, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipStreamSynchronize(s));      // the temporaries die with this call
    return HIPTS_OK;
}
