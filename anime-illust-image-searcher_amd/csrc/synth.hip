// synth.hip -- the synthetic image corpus of SURVEY.md section 8(d), generated ON THE DEVICE: config[3] ("1 M synthetic images sharded over
// 8 GPUs") has no host I/O, every rank produces its own block of the corpus from a counter-based generator keyed by the GLOBAL image
// index, so any rank can regenerate any image (bench.py's order check of the gathered rows does) and the corpus does not depend on
// how many ranks share it.
#include "common.h"

#include "../../include/hip_tagsearch.h"

namespace hipts {
namespace {

// splitmix64 finaliser as a counter hash: 8 pixel bytes per call, a pure function of (seed, image, 8-byte word index)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_images_kernel(uint8_t* __restrict__ out, int64_t first_index, int64_t count, int64_t bytes_per_image,
                                                           unsigned long long seed) {
    const int64_t words = bytes_per_image / 16;                        // 16 B per thread and step
    const int64_t total = count * words;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t img = i / words, w = i - img * words;
        const unsigned long long key = mix64(seed ^ mix64((unsigned long long)(first_index + img)));
        const unsigned long long a = mix64(key + 2ull * (unsigned long long)w), b = mix64(key + 2ull * (unsigned long long)w + 1ull);
        *reinterpret_cast<ulonglong2*>(out + img * bytes_per_image + w * 16) = make_ulonglong2(a, b);
    }
}

}  // namespace
}  // namespace hipts

using namespace hipts;

extern "C" int hipts_synth_images_u8(uint8_t* images_device, int64_t first_index, int64_t count, int image_size, uint64_t seed, int device,
                                     void* stream) {
    HIPTS_REQUIRE(images_device && count >= 1 && image_size >= 4 && first_index >= 0, "hipts_synth_images_u8: bad arguments");
    const int64_t bytes = (int64_t)image_size * image_size * 3;
    HIPTS_REQUIRE(bytes % 16 == 0, "hipts_synth_images_u8: image_size^2 * 3 must be a multiple of 16");
    HIPTS_TRY(use_device(device));
    const int64_t total = count * (bytes / 16);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
    synth_images_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(images_device, first_index, count, bytes, (unsigned long long)seed);
    HIPTS_LAUNCH_CHECK();
    return HIPTS_OK;
}
