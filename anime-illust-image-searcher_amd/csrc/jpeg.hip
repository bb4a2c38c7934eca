// jpeg.hip -- device half of the hybrid JPEG decode (round 4; SURVEY.md section 8 f4 "or GPU decode"; jpeg_slot.h describes the split).
// The reference decodes with PIL's Image.open (tagging.py:234-252, gen_cfeatures.py:285-295), i.e. libjpeg-turbo with its defaults;
// end to end the tagger was bound by that decode on the host cores (profiles/r03_pipeline_e2e.txt: 2.8 k images/s against 5.2 k of the
// forward).  The serial part -- markers and Huffman decoding -- stays on the host (jpeg_host.c, in the decode worker processes); what is
// per block and per pixel runs here, byte for byte libjpeg's arithmetic:
//   jpeg_idct_kernel    dequantisation + jidctint.c's accurate integer inverse DCT (JDCT_ISLOW: CONST_BITS 13, PASS1_BITS 2; the
//                       zero-coefficient short cuts of the C code are value-identical to the full passes) + range limit -> sample planes
//   jpeg_rgb_kernel     jdsample.c's "fancy" (triangle) chroma upsampling for 4:2:0 / 4:2:2 with libjpeg's alternating rounding and its
//                       edge rules, jdcolor.c's 16-bit fixed-point YCbCr -> RGB -> uint8 [h][w][3]
// then resize.hip's pad + Pillow-exact resize.  Both kernels are HBM-bound byte work (3 B of coefficients in, 1.5 B of planes out and in,
// 3 B of RGB out per pixel for 4:2:0).
#include <algorithm>
#include <cstddef>
#include <mutex>
#include <vector>

#include "common.h"
#include "jpeg_slot.h"

#include "../../include/hip_tagsearch.h"

namespace hipts {
namespace {

struct JpegImage {
    const int16_t* coef;           // device: all components' blocks, [block][64] natural order
    const uint16_t* quant;         // device: [3][64]
    uint8_t* plane[3];             // device: [blocks_h * 8][blocks_w * 8]
    int blocks_w[3], blocks_h[3];
    int first_block[4];            // prefix sums of the components' block counts
    int dw[3], dh[3];              // real samples per component
    int ncomp, hmax, vmax, width, height;
};

// one pass of jidctint.c over d[0..7]; results descaled by SHIFT bits
template <int SHIFT>
__device__ __forceinline__ void idct_1d(int (&d)[8]) {
    constexpr int F0_298631336 = 2446, F0_390180644 = 3196, F0_541196100 = 4433, F0_765366865 = 6270, F0_899976223 = 7373, F1_175875602 = 9633,
                  F1_501321110 = 12299, F1_847759065 = 15137, F1_961570560 = 16069, F2_053119869 = 16819, F2_562915447 = 20995,
                  F3_072711026 = 25172;
    int z2 = d[2], z3 = d[6];
    int z1 = (z2 + z3) * F0_541196100;
    int tmp2 = z1 + z3 * (-F1_847759065);
    int tmp3 = z1 + z2 * F0_765366865;
    z2 = d[0];
    z3 = d[4];
    int tmp0 = (z2 + z3) * 8192;
    int tmp1 = (z2 - z3) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = d[7];
    tmp1 = d[5];
    tmp2 = d[3];
    tmp3 = d[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F1_175875602;
    tmp0 *= F0_298631336;
    tmp1 *= F2_053119869;
    tmp2 *= F3_072711026;
    tmp3 *= F1_501321110;
    z1 *= -F0_899976223;
    z2 *= -F2_562915447;
    z3 = z3 * (-F1_961570560) + z5;
    z4 = z4 * (-F0_390180644) + z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    d[0] = (tmp10 + tmp3 + R) >> SHIFT;
    d[7] = (tmp10 - tmp3 + R) >> SHIFT;
    d[1] = (tmp11 + tmp2 + R) >> SHIFT;
    d[6] = (tmp11 - tmp2 + R) >> SHIFT;
    d[2] = (tmp12 + tmp1 + R) >> SHIFT;
    d[5] = (tmp12 - tmp1 + R) >> SHIFT;
    d[3] = (tmp13 + tmp0 + R) >> SHIFT;
    d[4] = (tmp13 - tmp0 + R) >> SHIFT;
}

// the sample of a descaled pass-2 value: + CENTERJSAMPLE, limited to 0..255.  (jidctint.c indexes sample_range_limit with x & RANGE_MASK,
// which wraps beyond +-512; libjpeg-turbo's SIMD code -- what Pillow runs -- saturates.  The host half only lets blocks through whose values
// stay far inside both, jpeg_host.c COLSUM_LIMIT; saturation is the form taken here.)
__device__ __forceinline__ unsigned range_limit(int x) {
    x += 128;
    return (unsigned)(x < 0 ? 0 : x > 255 ? 255 : x);
}

// 32 blocks per workgroup, eight lanes per block: lane c runs column c of pass 1, then row c of pass 2 (the workspace goes through LDS,
// rows padded to nine words).  A wave reads 8 x 128 B of coefficients as whole lines and stores 8 B per lane.
__device__ __forceinline__ void jpeg_idct_body(const JpegImage& im, int block_x) {
    __shared__ int ws[32][72];
    const int t = threadIdx.x, lb = t >> 3, l8 = t & 7;
    const int blk = block_x * 32 + lb;
    const bool valid = blk < im.first_block[im.ncomp];
    const int c = !valid ? 0 : (blk >= im.first_block[1] && im.ncomp > 1) + (blk >= im.first_block[2] && im.ncomp > 2);
    int d[8];
    if (valid) {
        const int16_t* cp = im.coef + (size_t)blk * 64;
        const uint16_t* q = im.quant + c * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (int)cp[k * 8 + l8] * (int)q[k * 8 + l8];
        idct_1d<11>(d);
#pragma unroll
        for (int k = 0; k < 8; ++k) ws[lb][k * 9 + l8] = d[k];
    }
    __syncthreads();
    if (!valid) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = ws[lb][l8 * 9 + k];
    idct_1d<18>(d);
    const unsigned lo = range_limit(d[0]) | (range_limit(d[1]) << 8) | (range_limit(d[2]) << 16) | (range_limit(d[3]) << 24);
    const unsigned hi = range_limit(d[4]) | (range_limit(d[5]) << 8) | (range_limit(d[6]) << 16) | (range_limit(d[7]) << 24);
    const int local = blk - im.first_block[c];
    const int by = local / im.blocks_w[c], bx = local - by * im.blocks_w[c];
    *reinterpret_cast<uint2*>(im.plane[c] + (size_t)(by * 8 + l8) * (im.blocks_w[c] * 8) + bx * 8) = make_uint2(lo, hi);
}
__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegImage im) { jpeg_idct_body(im, blockIdx.x); }

// Up to JPEG_CHUNK images per launch (blockIdx.y = image, the descriptors in the kernel arguments): 64 images were 128 launches of ~5 us
// that ran one after the other beside the forward's kernels.
constexpr int JPEG_CHUNK = 16;
struct JpegBatch {
    JpegImage im[JPEG_CHUNK];
    uint8_t* rgb[JPEG_CHUNK];
};
__global__ __launch_bounds__(256) void jpeg_idct_batch_kernel(const JpegBatch b) {
    const JpegImage& im = b.im[blockIdx.y];
    if ((int)blockIdx.x * 32 >= im.first_block[im.ncomp]) return;      // (uniform per workgroup: the barrier inside is not reached by anybody)
    jpeg_idct_body(im, blockIdx.x);
}

// chroma sample of output pixel (x, y): jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample / fullsize
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int pw, int dw, int dh, int hmax, int vmax, int x, int y) {
    if (hmax == 1) return p[(size_t)y * pw + x];
    const int c = x >> 1;
    if (vmax == 1) {
        const uint8_t* row = p + (size_t)y * pw;
        const int v = row[c];
        if (x & 1) return c == dw - 1 ? v : (3 * v + row[c + 1] + 2) >> 2;
        return c == 0 ? v : (3 * v + row[c - 1] + 1) >> 2;
    }
    const int r = y >> 1;
    const int rn = (y & 1) ? (r + 1 < dh ? r + 1 : dh - 1) : (r > 0 ? r - 1 : 0);      // jdmainct.c: the context row beyond an edge is the edge row
    const uint8_t* r0 = p + (size_t)r * pw;
    const uint8_t* r1 = p + (size_t)rn * pw;
    const int cs = 3 * r0[c] + r1[c];
    if (x & 1) return c == dw - 1 ? (cs * 4 + 7) >> 4 : (cs * 3 + (3 * r0[c + 1] + r1[c + 1]) + 7) >> 4;
    return c == 0 ? (cs * 4 + 8) >> 4 : (cs * 3 + (3 * r0[c - 1] + r1[c - 1]) + 8) >> 4;
}

__device__ __forceinline__ int clamp8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// a thread per pixel pair (x even): 6 bytes of RGB
__device__ __forceinline__ void jpeg_rgb_body(const JpegImage& im, uint8_t* __restrict__ rgb, int block_x) {
    const int pairs = (im.width + 1) >> 1;
    const int idx = block_x * 256 + threadIdx.x;
    if (idx >= pairs * im.height) return;
    const int y = idx / pairs, x0 = (idx - y * pairs) * 2;
    const int pw0 = im.blocks_w[0] * 8;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int x = x0 + u;
        if (x >= im.width) break;
        const int Y = im.plane[0][(size_t)y * pw0 + x];
        uint8_t* o = rgb + ((size_t)y * im.width + x) * 3;
        if (im.ncomp == 1) {
            o[0] = o[1] = o[2] = (uint8_t)Y;
            continue;
        }
        const int cb = chroma_at(im.plane[1], im.blocks_w[1] * 8, im.dw[1], im.dh[1], im.hmax, im.vmax, x, y) - 128;
        const int cr = chroma_at(im.plane[2], im.blocks_w[2] * 8, im.dw[2], im.dh[2], im.hmax, im.vmax, x, y) - 128;
        // jdcolor.c build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
        o[0] = (uint8_t)clamp8(Y + ((91881 * cr + 32768) >> 16));
        o[1] = (uint8_t)clamp8(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
        o[2] = (uint8_t)clamp8(Y + ((116130 * cb + 32768) >> 16));
    }
}

__global__ __launch_bounds__(256) void jpeg_rgb_kernel(const JpegImage im, uint8_t* __restrict__ rgb) { jpeg_rgb_body(im, rgb, blockIdx.x); }
__global__ __launch_bounds__(256) void jpeg_rgb_batch_kernel(const JpegBatch b) { jpeg_rgb_body(b.im[blockIdx.y], b.rgb[blockIdx.y], blockIdx.x); }

struct JpegState {
    std::mutex mu;
    DevBuf stage[64], planes[64], rgb[64];      // per device: coefficient slots, sample planes, decoded images
    hipEvent_t last[64] = {};                   // per device: end of the last call (callers may be on different streams)
};
JpegState& jstate() {
    static JpegState* s = new JpegState;      // never destroyed: see scratch_buf() in query.hip
    return *s;
}

inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }

// header checks shared by both entry points; fills the image descriptor but for its device pointers
int describe(const hipts_jpeg_header* hd, int64_t slot_bytes, JpegImage* im, size_t* plane_bytes) {
    HIPTS_REQUIRE(hd->magic == HIPTS_JPEG_MAGIC && hd->kind == 1, "jpeg: the slot does not hold coefficient blocks");
    HIPTS_REQUIRE((hd->ncomp == 1 || hd->ncomp == 3) && hd->width >= 1 && hd->height >= 1 && hd->hmax >= 1 && hd->hmax <= 2 && hd->vmax >= 1 &&
                      hd->vmax <= 2 && !(hd->hmax == 1 && hd->vmax == 2),
                  "jpeg: unsupported geometry in the slot header");
    HIPTS_REQUIRE(hd->total_bytes >= HIPTS_JPEG_HEADER_BYTES && hd->total_bytes <= slot_bytes, "jpeg: slot of %lld bytes, header says %lld",
                  (long long)slot_bytes, (long long)hd->total_bytes);
    int64_t blocks = 0;
    size_t pb = 0;
    im->ncomp = hd->ncomp;
    im->hmax = hd->hmax;
    im->vmax = hd->vmax;
    im->width = hd->width;
    im->height = hd->height;
    for (int c = 0; c < 3; ++c) {
        im->plane[c] = nullptr;
        im->blocks_w[c] = im->blocks_h[c] = im->dw[c] = im->dh[c] = 0;
    }
    for (int c = 0; c < hd->ncomp; ++c) {
        const hipts_jpeg_component& k = hd->comp[c];
        HIPTS_REQUIRE(k.blocks_w >= 1 && k.blocks_h >= 1 && k.offset == blocks * 64 && k.dw >= 1 && k.dh >= 1 && k.dw <= k.blocks_w * 8 &&
                          k.dh <= k.blocks_h * 8,
                      "jpeg: inconsistent component %d in the slot header", c);
        // every sample the colour kernel reads exists: luma covers the image, chroma its (up to) half-size grid
        const int hs = c == 0 ? 1 : hd->hmax, vs = c == 0 ? 1 : hd->vmax;
        HIPTS_REQUIRE((int64_t)k.dw * hs >= hd->width && (int64_t)k.dh * vs >= hd->height && (hs == 1 || k.dw >= 3),
                      "jpeg: component %d does not cover the image", c);
        im->first_block[c] = (int)blocks;
        im->blocks_w[c] = k.blocks_w;
        im->blocks_h[c] = k.blocks_h;
        im->dw[c] = k.dw;
        im->dh[c] = k.dh;
        blocks += (int64_t)k.blocks_w * k.blocks_h;
        pb += align256((size_t)k.blocks_w * k.blocks_h * 64);
    }
    for (int c = hd->ncomp; c < 4; ++c) im->first_block[c] = (int)blocks;
    HIPTS_REQUIRE(HIPTS_JPEG_HEADER_BYTES + blocks * 128 == hd->total_bytes, "jpeg: %lld blocks do not fill %lld bytes", (long long)blocks,
                  (long long)hd->total_bytes);
    *plane_bytes = pb;
    return HIPTS_OK;
}

// the descriptor's device pointers: copy of the slot at `dev_slot`, planes at `planes`
void bind(JpegImage& im, const uint8_t* dev_slot, uint8_t* planes) {
    im.coef = reinterpret_cast<const int16_t*>(dev_slot + HIPTS_JPEG_HEADER_BYTES);
    im.quant = reinterpret_cast<const uint16_t*>(dev_slot + offsetof(hipts_jpeg_header, quant));
    size_t off = 0;
    for (int c = 0; c < im.ncomp; ++c) {
        im.plane[c] = planes + off;
        off += align256((size_t)im.blocks_w[c] * im.blocks_h[c] * 64);
    }
}

// one image: the two kernels
int launch_decode(JpegImage im, const uint8_t* dev_slot, uint8_t* planes, uint8_t* rgb, hipStream_t s) {
    bind(im, dev_slot, planes);
    const int blocks = im.first_block[im.ncomp];
    jpeg_idct_kernel<<<(blocks + 31) / 32, 256, 0, s>>>(im);
    HIPTS_LAUNCH_CHECK();
    const int work = ((im.width + 1) / 2) * im.height;
    jpeg_rgb_kernel<<<(work + 255) / 256, 256, 0, s>>>(im, rgb);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}

}  // namespace
}  // namespace hipts

using namespace hipts;

extern "C" int hipts_jpeg_decode_rgb(const void* slot, int64_t slot_bytes, uint8_t* rgb_out, int out_memspace, int64_t out_capacity, int device,
                                     void* stream) {
    HIPTS_REQUIRE(slot && rgb_out && slot_bytes >= HIPTS_JPEG_HEADER_BYTES, "hipts_jpeg_decode_rgb: bad arguments");
    HIPTS_REQUIRE(device >= 0 && device < 64, "hipts_jpeg_decode_rgb: device index");
    const hipts_jpeg_header* hd = static_cast<const hipts_jpeg_header*>(slot);
    JpegImage im{};
    size_t plane_bytes = 0;
    HIPTS_TRY(describe(hd, slot_bytes, &im, &plane_bytes));
    const size_t rgb_bytes = (size_t)hd->width * hd->height * 3;
    HIPTS_REQUIRE((int64_t)rgb_bytes <= out_capacity, "hipts_jpeg_decode_rgb: %d x %d needs %zu bytes, the output holds %lld", hd->height, hd->width,
                  rgb_bytes, (long long)out_capacity);
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    JpegState& st = jstate();
    std::lock_guard<std::mutex> lock(st.mu);
    if (st.last[device]) HIPTS_HIP(hipStreamWaitEvent(s, st.last[device], 0));
    else HIPTS_HIP(hipEventCreateWithFlags(&st.last[device], hipEventDisableTiming));
    HIPTS_TRY(st.stage[device].reserve((size_t)hd->total_bytes));
    HIPTS_TRY(st.planes[device].reserve(plane_bytes));
    uint8_t* out = rgb_out;
    if (out_memspace != HIPTS_DEVICE) {
        HIPTS_TRY(st.rgb[device].reserve(rgb_bytes));
        out = st.rgb[device].as<uint8_t>();
    }
    HIPTS_HIP(hipMemcpyAsync(st.stage[device].p, slot, (size_t)hd->total_bytes, hipMemcpyHostToDevice, s));
    HIPTS_TRY(launch_decode(im, st.stage[device].as<uint8_t>(), st.planes[device].as<uint8_t>(), out, s));
    if (out_memspace != HIPTS_DEVICE) HIPTS_HIP(hipMemcpyAsync(rgb_out, out, rgb_bytes, hipMemcpyDeviceToHost, s));
    HIPTS_HIP(hipEventRecord(st.last[device], s));
    HIPTS_HIP(hipStreamSynchronize(s));       // the caller's slot and (host) output are its own again
    return HIPTS_OK;
}

extern "C" int hipts_jpeg_batch_u8(const uint8_t* slots, int64_t slot_stride, const int32_t* kinds, const int32_t* hw, int n, int pad_square,
                                   uint8_t* dst_device, int size, int filter, int device, void* stream) {
    HIPTS_REQUIRE(slots && kinds && hw && dst_device && n >= 1 && size >= 1 && slot_stride >= HIPTS_JPEG_HEADER_BYTES, "hipts_jpeg_batch_u8: bad arguments");
    HIPTS_REQUIRE(device >= 0 && device < 64, "hipts_jpeg_batch_u8: device index");
    std::vector<JpegImage> ims((size_t)n);
    std::vector<size_t> stage_off((size_t)n), plane_off((size_t)n);
    size_t stage_total = 0, plane_total = 0, rgb_stride = 0;
    for (int i = 0; i < n; ++i) {
        const int h = hw[2 * i], w = hw[2 * i + 1];
        HIPTS_REQUIRE(h >= 1 && w >= 1, "hipts_jpeg_batch_u8: image %d is %d x %d", i, h, w);
        rgb_stride = std::max(rgb_stride, align256((size_t)h * w * 3));
        if (kinds[i] == 1) {
            const hipts_jpeg_header* hd = reinterpret_cast<const hipts_jpeg_header*>(slots + (size_t)i * slot_stride);
            size_t pb = 0;
            HIPTS_TRY(describe(hd, slot_stride, &ims[i], &pb));
            HIPTS_REQUIRE(hd->height == h && hd->width == w, "hipts_jpeg_batch_u8: image %d: header %d x %d, caller %d x %d", i, hd->height, hd->width, h, w);
            stage_off[i] = stage_total;
            plane_off[i] = plane_total;
            stage_total += align256((size_t)hd->total_bytes);
            plane_total += pb;
        } else {
            HIPTS_REQUIRE(kinds[i] == 0 && (int64_t)h * w * 3 <= slot_stride, "hipts_jpeg_batch_u8: image %d: kind %d, %d x %d in a slot of %lld bytes", i,
                          kinds[i], h, w, (long long)slot_stride);
        }
    }
    HIPTS_TRY(use_device(device));
    hipStream_t s = (hipStream_t)stream;
    JpegState& st = jstate();
    {
        std::lock_guard<std::mutex> lock(st.mu);
        // the buffers are shared by every caller of this device: order this call behind the last one that used them
        if (st.last[device]) HIPTS_HIP(hipStreamWaitEvent(s, st.last[device], 0));
        else HIPTS_HIP(hipEventCreateWithFlags(&st.last[device], hipEventDisableTiming));
        HIPTS_TRY(st.stage[device].reserve(stage_total));
        HIPTS_TRY(st.planes[device].reserve(plane_total));
        HIPTS_TRY(st.rgb[device].reserve(rgb_stride * (size_t)n));
        uint8_t* rgb = st.rgb[device].as<uint8_t>();
        for (int i = 0; i < n; ++i) {
            const uint8_t* sp = slots + (size_t)i * slot_stride;
            if (kinds[i] == 1) {
                const hipts_jpeg_header* hd = reinterpret_cast<const hipts_jpeg_header*>(sp);
                uint8_t* d = st.stage[device].as<uint8_t>() + stage_off[i];
                HIPTS_HIP(hipMemcpyAsync(d, sp, (size_t)hd->total_bytes, hipMemcpyHostToDevice, s));
                bind(ims[i], d, st.planes[device].as<uint8_t>() + plane_off[i]);
            } else {
                HIPTS_HIP(hipMemcpyAsync(rgb + (size_t)i * rgb_stride, sp, (size_t)hw[2 * i] * hw[2 * i + 1] * 3, hipMemcpyHostToDevice, s));
            }
        }
        // the coefficient slots, JPEG_CHUNK images per launch
        {
            JpegBatch b{};
            int nb = 0, max_blocks = 0, max_work = 0;
            auto flush = [&]() -> int {
                if (nb == 0) return HIPTS_OK;
                jpeg_idct_batch_kernel<<<dim3((max_blocks + 31) / 32, nb), 256, 0, s>>>(b);
                HIPTS_LAUNCH_CHECK();
                jpeg_rgb_batch_kernel<<<dim3((max_work + 255) / 256, nb), 256, 0, s>>>(b);
                HIPTS_LAUNCH_CHECK();
                nb = max_blocks = max_work = 0;
                return HIPTS_OK;
            };
            for (int i = 0; i < n; ++i) {
                if (kinds[i] != 1) continue;
                b.im[nb] = ims[i];
                b.rgb[nb] = rgb + (size_t)i * rgb_stride;
                max_blocks = std::max(max_blocks, ims[i].first_block[ims[i].ncomp]);
                max_work = std::max(max_work, ((ims[i].width + 1) / 2) * ims[i].height);
                if (++nb == JPEG_CHUNK) HIPTS_TRY(flush());
            }
            HIPTS_TRY(flush());
        }
        // pad + resize of the decoded images (resize.hip; same stream, so ordered behind the kernels above)
        const int rs = hipts_resize_batch_u8(rgb, HIPTS_DEVICE, (int64_t)rgb_stride, hw, n, pad_square, dst_device, size, filter, device, stream);
        HIPTS_HIP(hipEventRecord(st.last[device], s));
        if (rs != HIPTS_OK) return rs;
    }
    return HIPTS_OK;
}
