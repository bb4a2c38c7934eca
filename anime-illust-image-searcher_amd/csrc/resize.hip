// resize.hip -- PIL's two-pass resample on the device.  Replaces the host-side
//     image.resize((448, 448), BICUBIC)        timm eval transform inside tagging.py:241 (Resize(bicubic) on the padded square of :100-120)
//     image.resize((384, 384), BILINEAR)       gen_cfeatures.py:101
// bit for bit: Pillow's ImagingResample (libImaging/Resample.c) for 8-bit images is an integer algorithm -- per output column / row a window
// [xmin, xmin + n) of the source and n fixed-point coefficients (22 fractional bits) computed in double precision from the filter
// (support scaled by the shrink factor: the antialiasing), a horizontal pass into a uint8 temporary restricted to the rows the vertical pass
// needs, then the vertical pass; each output byte = clip8((2^21 + sum pixel * coeff) >> 22).  The coefficient tables are computed on the
// host with Pillow's own expressions in Pillow's order (this file is built with -ffp-contract=off, and Pillow's wheels carry no fused
// multiply-adds), cached per (source size, output size, filter) and handed to two small kernels: the decode workers then only decode.
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

#include "../../include/hip_tagsearch.h"

namespace hipts {
namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

inline double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Coeffs {
    int ksize = 0;
    std::vector<int> bounds;        // [out][2]: first source index, tap count
    std::vector<int> kk;            // [out][ksize] fixed-point coefficients
};

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box (in0 = 0, in1 = inSize)
Coeffs precompute(int inSize, int outSize, int filter) {
    const double fsupport = filter == 3 ? 2.0 : 1.0;
    double filterscale, scale;
    filterscale = scale = (double)((float)inSize - 0.0f) / outSize;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = fsupport * filterscale;
    Coeffs c;
    c.ksize = (int)ceil(support) * 2 + 1;
    c.bounds.resize((size_t)outSize * 2);
    c.kk.resize((size_t)outSize * c.ksize);
    std::vector<double> k(c.ksize);
    for (int xx = 0; xx < outSize; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > inSize) xmax = inSize;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            const double arg = (x + xmin - center + 0.5) * ss;
            const double w = filter == 3 ? bicubic_filter(arg) : bilinear_filter(arg);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < c.ksize; ++x) k[x] = 0;
        c.bounds[(size_t)xx * 2] = xmin;
        c.bounds[(size_t)xx * 2 + 1] = xmax;
        for (x = 0; x < c.ksize; ++x) {
            if (k[x] < 0) c.kk[(size_t)xx * c.ksize + x] = (int)(-0.5 + k[x] * (1 << PRECISION_BITS));
            else c.kk[(size_t)xx * c.ksize + x] = (int)(0.5 + k[x] * (1 << PRECISION_BITS));
        }
    }
    return c;
}

struct DevCoeffs {
    int ksize = 0;
    int first = 0, last = 0;        // source rows / columns the pass touches: [first, last)
    DevBuf bounds, kk;
};

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;            // arithmetic shift: floor, as Pillow's table index
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[y - y0][xx][c] = clip8(2^21 + sum_x src[y][xmin + x][c] * k[xx][x]),  y in [y0, y1)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, int src_w, uint8_t* __restrict__ tmp, int out_w, int y0, int rows,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * out_w) return;
    const int y = i / out_w, xx = i - y * out_w;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* p = src + ((size_t)(y0 + y) * src_w + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        s0 += p[3 * x] * w;
        s1 += p[3 * x + 1] * w;
        s2 += p[3 * x + 2] * w;
    }
    uint8_t* o = tmp + (size_t)i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// The same pass over an image that sits at (top, left) inside a white canvas (Predictor.prepare_image's centred padding, tagging.py:100-120):
// canvas row y0 + y, canvas columns xmin ..; pixels outside the image are 255.
__global__ __launch_bounds__(256) void resample_h_pad_kernel(const uint8_t* __restrict__ src, int img_h, int img_w, int top, int left, uint8_t* __restrict__ tmp,
                                                             int out_w, int y0, int rows, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * out_w) return;
    const int y = i / out_w, xx = i - y * out_w;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const int sy = y0 + y - top;
    const bool row_in = sy >= 0 && sy < img_h;
    const uint8_t* p = src + (size_t)(row_in ? sy : 0) * img_w * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        const int sx = xmin + x - left;
        const bool in = row_in && sx >= 0 && sx < img_w;
        s0 += (in ? p[3 * sx] : 255) * w;
        s1 += (in ? p[3 * sx + 1] : 255) * w;
        s2 += (in ? p[3 * sx + 2] : 255) * w;
    }
    uint8_t* o = tmp + (size_t)i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// dst[yy][x][c] = clip8(2^21 + sum_y tmp[ymin - y0 + y][x][c] * k[yy][y])
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, int w, uint8_t* __restrict__ dst, int out_h, int y0,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= out_h * w) return;
    const int yy = i / w, x = i - yy * w;
    const int ymin = bounds[2 * yy] - y0, n = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    const uint8_t* p = tmp + ((size_t)ymin * w + x) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int wgt = k[y];
        s0 += p[(size_t)y * w * 3] * wgt;
        s1 += p[(size_t)y * w * 3 + 1] * wgt;
        s2 += p[(size_t)y * w * 3 + 2] * wgt;
    }
    uint8_t* o = dst + (size_t)i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// Batched forms (round 4): one launch for up to RESIZE_CHUNK images of different sizes, blockIdx.y = image, the per-image parameters in
// the kernel arguments.  A batch of 64 decoded images was 128 launches of 5-13 us each; beside the forward's kernels they ran one after
// the other and cost the pipeline their summed duration (tools/jpeg_overlap_bench.py).
constexpr int RESIZE_CHUNK = 32;
struct ResizeJob {
    const uint8_t* src;       // image, [img_h][img_w][3]
    uint8_t* tmp;             // horizontal pass output, [rows][size][3]
    uint8_t* dst;             // [size][size][3]
    const int* hb;            // horizontal coefficient table (bounds, weights) and its row length
    const int* hk;
    const int* vb;
    const int* vk;
    int img_h, img_w, top, left, y0, rows, hks, vks;
};
struct ResizeJobs {
    ResizeJob j[RESIZE_CHUNK];
};
__global__ __launch_bounds__(256) void resample_h_pad_batch_kernel(const ResizeJobs jobs, int out_w) {
    const ResizeJob& J = jobs.j[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= J.rows * out_w) return;
    const int y = i / out_w, xx = i - y * out_w;
    const int xmin = J.hb[2 * xx], n = J.hb[2 * xx + 1];
    const int* k = J.hk + (size_t)xx * J.hks;
    const int sy = J.y0 + y - J.top;
    const bool row_in = sy >= 0 && sy < J.img_h;
    const uint8_t* p = J.src + (size_t)(row_in ? sy : 0) * J.img_w * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        const int sx = xmin + x - J.left;
        const bool in = row_in && sx >= 0 && sx < J.img_w;
        s0 += (in ? p[3 * sx] : 255) * w;
        s1 += (in ? p[3 * sx + 1] : 255) * w;
        s2 += (in ? p[3 * sx + 2] : 255) * w;
    }
    uint8_t* o = J.tmp + (size_t)i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}
__global__ __launch_bounds__(256) void resample_v_batch_kernel(const ResizeJobs jobs, int size) {
    const ResizeJob& J = jobs.j[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= size * size) return;
    const int yy = i / size, x = i - yy * size;
    const int ymin = J.vb[2 * yy] - J.y0, n = J.vb[2 * yy + 1];
    const int* k = J.vk + (size_t)yy * J.vks;
    const uint8_t* p = J.tmp + ((size_t)ymin * size + x) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const int wgt = k[y];
        s0 += p[(size_t)y * size * 3] * wgt;
        s1 += p[(size_t)y * size * 3 + 1] * wgt;
        s2 += p[(size_t)y * size * 3 + 2] * wgt;
    }
    uint8_t* o = J.dst + (size_t)i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

struct ResizeState {
    std::mutex mu;
    std::map<std::tuple<int, int, int, int>, DevCoeffs*> cache;      // (device, in, out, filter)
    DevBuf stage[64], tmp[64];                                         // per device: host-source staging, horizontal-pass temporary
    DevBuf bstage[64];                                                 // per device: the batch entry's staging of host images
    hipEvent_t last[64] = {};                                          // per device: end of the last call that used tmp (callers may be on different streams)
};
ResizeState& state() {
    static ResizeState* s = new ResizeState;       // never destroyed: see scratch_buf() in query.hip
    return *s;
}

int get_coeffs(int device, int in, int out, int filter, DevCoeffs** res) {
    ResizeState& st = state();
    const auto key = std::make_tuple(device, in, out, filter);
    auto it = st.cache.find(key);
    if (it == st.cache.end()) {
        const Coeffs c = precompute(in, out, filter);
        std::unique_ptr<DevCoeffs> d(new DevCoeffs);
        d->ksize = c.ksize;
        d->first = c.bounds[0];
        d->last = c.bounds[(size_t)out * 2 - 2] + c.bounds[(size_t)out * 2 - 1];
        HIPTS_TRY(d->bounds.alloc(c.bounds.size() * 4));
        HIPTS_TRY(d->kk.alloc(c.kk.size() * 4));
        HIPTS_TRY(upload(d->bounds.p, c.bounds.data(), c.bounds.size() * 4));
        HIPTS_TRY(upload(d->kk.p, c.kk.data(), c.kk.size() * 4));
        it = st.cache.emplace(key, d.release()).first;
    }
    *res = it->second;
    return HIPTS_OK;
}

}  // namespace
}  // namespace hipts

using namespace hipts;

extern "C" int hipts_resize_u8(const uint8_t* src, int src_memspace, int src_h, int src_w, uint8_t* dst_device, int dst_h, int dst_w, int filter,
                               int device, void* stream) {
    HIPTS_REQUIRE(src && dst_device && src_h >= 1 && src_w >= 1 && dst_h >= 1 && dst_w >= 1, "hipts_resize_u8: bad arguments");
    HIPTS_REQUIRE(filter == 2 || filter == 3, "hipts_resize_u8: filter must be 2 (PIL BILINEAR) or 3 (PIL BICUBIC)");
    HIPTS_REQUIRE(device >= 0 && device < 64, "hipts_resize_u8: device index");
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    ResizeState& st = state();
    std::lock_guard<std::mutex> lock(st.mu);
    const uint8_t* sp = src;
    const size_t src_bytes = (size_t)src_h * src_w * 3;
    if (src_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(st.stage[device].reserve(src_bytes));
        HIPTS_HIP(hipMemcpyAsync(st.stage[device].p, src, src_bytes, hipMemcpyHostToDevice, s));
        sp = st.stage[device].as<uint8_t>();
    }
    const bool need_h = dst_w != src_w, need_v = dst_h != src_h;
    if (!need_h && !need_v) {
        HIPTS_HIP(hipMemcpyAsync(dst_device, sp, src_bytes, hipMemcpyDeviceToDevice, s));
        if (src_memspace != HIPTS_DEVICE) HIPTS_HIP(hipStreamSynchronize(s));
        return HIPTS_OK;
    }
    DevCoeffs *ch = nullptr, *cv = nullptr;
    HIPTS_TRY(get_coeffs(device, src_w, dst_w, filter, &ch));
    HIPTS_TRY(get_coeffs(device, src_h, dst_h, filter, &cv));
    // Resample.c: the horizontal pass covers only the source rows the vertical pass reads
    const int y0 = need_v ? cv->first : 0, y1 = need_v ? cv->last : src_h;
    const uint8_t* vin = sp;
    int vin_w = src_w, v_y0 = 0;
    bool used_tmp = false;
    if (need_h) {
        uint8_t* out_h = dst_device;
        if (need_v) {
            // the temporary is shared by every caller of this device: order this call behind the last one that used it
            if (st.last[device]) HIPTS_HIP(hipStreamWaitEvent(s, st.last[device], 0));
            else HIPTS_HIP(hipEventCreateWithFlags(&st.last[device], hipEventDisableTiming));
            HIPTS_TRY(st.tmp[device].reserve((size_t)(y1 - y0) * dst_w * 3));
            out_h = st.tmp[device].as<uint8_t>();
            used_tmp = true;
        }
        const int total = (y1 - y0) * dst_w;
        resample_h_kernel<<<(total + 255) / 256, 256, 0, s>>>(sp, src_w, out_h, dst_w, y0, y1 - y0, ch->bounds.as<int>(), ch->kk.as<int>(), ch->ksize);
        HIPTS_LAUNCH_CHECK();
        vin = out_h;
        vin_w = dst_w;
        v_y0 = y0;
    }
    if (need_v) {
        const int total = dst_h * vin_w;
        resample_v_kernel<<<(total + 255) / 256, 256, 0, s>>>(vin, vin_w, dst_device, dst_h, v_y0, cv->bounds.as<int>(), cv->kk.as<int>(), cv->ksize);
        HIPTS_LAUNCH_CHECK();
    }
    if (used_tmp) HIPTS_HIP(hipEventRecord(st.last[device], s));
    if (src_memspace != HIPTS_DEVICE) HIPTS_HIP(hipStreamSynchronize(s));       // the staging buffer is reused by the next call
    return HIPTS_OK;
}


// A batch of decoded images of different sizes -> uint8 [n][size][size][3]: image i is h x w (hw[2 i], hw[2 i + 1]) at src_base + i * slot_stride
// (a ring slot of hiptagsearch/pipeline.py's decode-only workers).  pad_square != 0: centred on a white max(h, w) square first (the tagger's
// prepare_image), then Resize(bicubic / bilinear) as hipts_resize_u8.  Host sources are copied with hipMemcpyAsync (asynchronous when the
// caller has registered the ring as pinned memory); everything is ordered on `stream`, nothing is synchronised.
extern "C" int hipts_resize_batch_u8(const uint8_t* src_base, int src_memspace, int64_t slot_stride, const int32_t* hw, int n, int pad_square,
                                     uint8_t* dst_device, int size, int filter, int device, void* stream) {
    HIPTS_REQUIRE(src_base && hw && dst_device && n >= 1 && size >= 1 && slot_stride >= 1, "hipts_resize_batch_u8: bad arguments");
    HIPTS_REQUIRE(filter == 2 || filter == 3, "hipts_resize_batch_u8: filter must be 2 (PIL BILINEAR) or 3 (PIL BICUBIC)");
    HIPTS_REQUIRE(device >= 0 && device < 64, "hipts_resize_batch_u8: device index");
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        HIPTS_REQUIRE(hw[2 * i] >= 1 && hw[2 * i + 1] >= 1 && (int64_t)hw[2 * i] * hw[2 * i + 1] * 3 <= slot_stride,
                      "hipts_resize_batch_u8: image %d is %d x %d, slot stride %lld", i, hw[2 * i], hw[2 * i + 1], (long long)slot_stride);
        total += ((size_t)hw[2 * i] * hw[2 * i + 1] * 3 + 255) / 256 * 256;
    }
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    ResizeState& st = state();
    std::lock_guard<std::mutex> lock(st.mu);
    // staging and temporary are shared by every caller of this device: order this call behind the last one that used them
    if (st.last[device]) HIPTS_HIP(hipStreamWaitEvent(s, st.last[device], 0));
    else HIPTS_HIP(hipEventCreateWithFlags(&st.last[device], hipEventDisableTiming));
    const bool host = src_memspace != HIPTS_DEVICE;
    if (host) HIPTS_TRY(st.bstage[device].reserve(total));
    // the horizontal pass of every image keeps its own temporary: the launches below cover many images at once
    std::vector<size_t> tmp_off((size_t)n);
    size_t tmp_total = 0;
    std::vector<DevCoeffs*> chs((size_t)n), cvs((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int h = hw[2 * i], w = hw[2 * i + 1];
        const int m = pad_square ? (h > w ? h : w) : 0;
        const int ch = pad_square ? m : h, cw = pad_square ? m : w;                 // canvas
        HIPTS_TRY(get_coeffs(device, cw, size, filter, &chs[i]));
        HIPTS_TRY(get_coeffs(device, ch, size, filter, &cvs[i]));
        tmp_off[i] = tmp_total;
        tmp_total += ((size_t)(cvs[i]->last - cvs[i]->first) * size * 3 + 255) / 256 * 256;
    }
    HIPTS_TRY(st.tmp[device].reserve(tmp_total));      // (grow-only, before anything of this call is launched)
    size_t off = 0;
    std::vector<const uint8_t*> srcs((size_t)n);
    for (int i = 0; i < n; ++i) {
        const size_t bytes = (size_t)hw[2 * i] * hw[2 * i + 1] * 3;
        srcs[i] = src_base + (size_t)i * slot_stride;
        if (host) {
            uint8_t* d = st.bstage[device].as<uint8_t>() + off;
            HIPTS_HIP(hipMemcpyAsync(d, srcs[i], bytes, hipMemcpyHostToDevice, s));
            srcs[i] = d;
            off += (bytes + 255) / 256 * 256;
        }
    }
    for (int c0 = 0; c0 < n; c0 += RESIZE_CHUNK) {
        const int nc = n - c0 < RESIZE_CHUNK ? n - c0 : RESIZE_CHUNK;
        ResizeJobs jobs{};
        int max_h_work = 0;
        for (int u = 0; u < nc; ++u) {
            const int i = c0 + u;
            const int h = hw[2 * i], w = hw[2 * i + 1];
            const int m = pad_square ? (h > w ? h : w) : 0;
            ResizeJob& J = jobs.j[u];
            J.src = srcs[i];
            J.tmp = st.tmp[device].as<uint8_t>() + tmp_off[i];
            J.dst = dst_device + (size_t)i * size * size * 3;
            J.hb = chs[i]->bounds.as<int>(); J.hk = chs[i]->kk.as<int>(); J.hks = chs[i]->ksize;
            J.vb = cvs[i]->bounds.as<int>(); J.vk = cvs[i]->kk.as<int>(); J.vks = cvs[i]->ksize;
            J.img_h = h; J.img_w = w;
            J.top = pad_square ? (m - h) / 2 : 0; J.left = pad_square ? (m - w) / 2 : 0;
            J.y0 = cvs[i]->first; J.rows = cvs[i]->last - cvs[i]->first;
            max_h_work = std::max(max_h_work, J.rows * size);
        }
        resample_h_pad_batch_kernel<<<dim3((max_h_work + 255) / 256, nc), 256, 0, s>>>(jobs, size);
        HIPTS_LAUNCH_CHECK();
        resample_v_batch_kernel<<<dim3((size * size + 255) / 256, nc), 256, 0, s>>>(jobs, size);
        HIPTS_LAUNCH_CHECK();
    }
    HIPTS_HIP(hipEventRecord(st.last[device], s));
    return HIPTS_OK;
}
