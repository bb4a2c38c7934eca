// Development probe (gpurun): lane maps and scale operands of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3
// operands, and the rounding / saturation of v_cvt_pk_fp8_f32, checked with exact small-integer data.
//   hipcc --offload-arch=gfx950 -O2 -o fp8_mfma fp8_mfma.hip && ./fp8_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: [16][128] e4m3 bytes (row-major), B^T: [16][128] (row n, k contiguous); D[m][n] = sum_k A[m][k] B^T[n][k] * 2^(sa-127) * 2^(sb-127)
__global__ void probe(const uint8_t* A, const uint8_t* Bt, float* D, int sa, int sb) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    i32x8 a, b;
    const int* ap = reinterpret_cast<const int*>(A + r * 128 + q * 32);
    const int* bp = reinterpret_cast<const int*>(Bt + r * 128 + q * 32);
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // operand "a" carries rows of A -> D rows = 4 q + reg ... (C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa * 0x01010101, 0, sb * 0x01010101);
    for (int e = 0; e < 4; ++e) D[(4 * q + e) * 16 + r] = c[e];
}

__global__ void cvt_probe(const float* x, uint8_t* y, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        const int p = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.0f, 0, false);
        y[i] = (uint8_t)(p & 0xff);
    }
}

static float e4m3_to_f32(uint8_t v) {          // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) f = NAN;
    else if (e == 0) f = ldexpf((float)m, -9);
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    // small exact values: e4m3 codes for 0, +-0.5, +-1, +-1.5, +-2, +-3
    const uint8_t codes[] = {0x00, 0x30, 0xb0, 0x38, 0xb8, 0x3c, 0xbc, 0x40, 0xc0, 0x44, 0xc4};
    for (auto& v : A) v = codes[rnd() % 11];
    for (auto& v : B) v = codes[rnd() % 11];
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int t = 0; t < 3; ++t) {
        const int sa = t == 1 ? 125 : 127, sb = t == 2 ? 130 : 127;
        probe<<<1, 64>>>(dA, dB, dD, sa, sb);
        std::vector<float> D(256);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)e4m3_to_f32(A[m * 128 + k]) * e4m3_to_f32(B[n * 128 + k]);
                ref = ldexp(ref, (sa - 127) + (sb - 127));
                if ((float)ref != D[m * 16 + n]) { if (bad < 4) printf("  D[%d][%d] = %g, expected %g\n", m, n, D[m * 16 + n], ref); ++bad; }
            }
        printf("scale_a %d scale_b %d: %d mismatches of 256\n", sa, sb, bad);
        bad_total += bad;
    }
    const float xs[] = {0.0f, 1.0f, -1.0f, 0.0625f, 0.001f, 0.002f, 1.0625f, 1.1875f, 447.0f, 448.0f, 460.0f, 480.0f, 1000.0f, -1000.0f, 1e30f, 0.0146f, 17.0f, 19.0f};
    const int n = sizeof(xs) / 4;
    float* dx; uint8_t* dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n);
    hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(dx, dy, n);
    std::vector<uint8_t> y(n);
    hipMemcpy(y.data(), dy, n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt %g -> 0x%02x = %g\n", xs[i], y[i], e4m3_to_f32(y[i]));
    return bad_total ? 1 : 0;
}
