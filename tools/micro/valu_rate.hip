// Development aid: issue rate of v_exp_f32 / v_rcp_f32 / v_pk_fma_f32 / v_fma_f32 / v_dot2c_f32_bf16 / v_perm_b32 on gfx950 (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    float a[24];
    f32x2 p[24];
    for (int i = 0; i < 24; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 2) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            if (OP == 3) p[i] = __builtin_elementwise_fma(p[i], f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
            if (OP == 4) a[i] = fmaxf(fmaxf(a[i], a[i]), 0.25f);
            if (OP == 5) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p[i][0]), __builtin_bit_cast(bf16x2, p[(i + 1) % 24][1]), a[i], false);
            if (OP == 7) a[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, p[i][0]), __builtin_bit_cast(f16x2, p[(i + 1) % 24][1]), a[i], false);
            if (OP == 8) a[i] = __builtin_fmaf((float)__builtin_bit_cast(f16x2, p[i][0])[0], p[(i + 1) % 24][1], a[i]);      // v_fma_mix_f32
            if (OP == 9) p[i][0] = __builtin_bit_cast(float, __builtin_elementwise_fma(__builtin_bit_cast(f16x2, p[i][0]), __builtin_bit_cast(f16x2, p[(i + 1) % 24][1]), __builtin_bit_cast(f16x2, p[i][1])));   // v_pk_fma_f16
            if (OP == 6) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, a[i]), __builtin_bit_cast(unsigned, a[(i + 1) % 24]), 0x05040100u));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 24; ++i) s += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 128);
    const int iters = 4096;
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_pk_fma_f32", "v_max3_f32", "v_dot2c_f32_bf16", "v_perm_b32", "v_dot2_f32_f16", "v_fma_mix_f32", "v_pk_fma_f16"};
    for (int rep = 0; rep < 2; ++rep) {
        k<0><<<256, 256>>>(out, iters, cyc); k<1><<<256, 256>>>(out, iters, cyc); k<2><<<256, 256>>>(out, iters, cyc);
        k<3><<<256, 256>>>(out, iters, cyc); k<4><<<256, 256>>>(out, iters, cyc); k<5><<<256, 256>>>(out, iters, cyc); k<6><<<256, 256>>>(out, iters, cyc);
        k<7><<<256, 256>>>(out, iters, cyc); k<8><<<256, 256>>>(out, iters, cyc); k<9><<<256, 256>>>(out, iters, cyc);
        hipDeviceSynchronize();
    }
    unsigned long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; ++i) printf("%-14s %.2f cycles per wave instruction (1 wave/SIMD)\n", names[i], (double)h[i] / (iters * 24.0));
    return 0;
}
