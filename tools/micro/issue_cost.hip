// Development aid: issue cost (cycles per instruction, one wave per SIMD) of the instructions of the attention softmax stream on gfx950,
// alone and as fillers behind v_mfma_f32_32x32x16_f16.  hipcc -O3 --offload-arch=gfx950 issue_cost.hip -o issue_cost && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define REP8(X) X X X X X X X X
#define T0() unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory")
#define T1(slot, n) do { unsigned long long t1_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory"); if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[2 * (slot)] = t1_ - t0_; cyc[2 * (slot) + 1] = (n); } } while (0)

template <int OP>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    float a[16], b[16];
    unsigned w[8];
    f32x16 acc[4], cc;
    f16x8 fa, fb;
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + i; cc[i] = -1.0f; }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * i); fb[i] = (_Float16)0.5f; w[i] = i; }
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float*>(lds)[i] = i;
    __syncthreads();
    const unsigned la = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 16;
    constexpr int IT = 64;
    T0();
    for (int it = 0; it < IT; ++it) {
        if constexpr (OP == 0) { REP8(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(a[0]), "+v"(a[1]));) }
        if constexpr (OP == 1) { REP8(asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(a[0]), "+v"(a[1]) : "v"(b[0]));) }
        if constexpr (OP == 2) { REP8(asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %3, %2" : "=v"(w[0]), "=v"(w[1]) : "v"(a[0]), "v"(a[1]));) }
        if constexpr (OP == 3) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %3, %2" : "=v"(w[0]), "=v"(w[1]) : "v"(a[0]), "v"(a[1]));) }
        if constexpr (OP == 4) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %2, %2, %1" : "+v"(*(double*)&a[0]) : "v"(*(double*)&b[0]), "v"(*(double*)&a[2]));) }
        if constexpr (OP == 5) { REP8(asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=v"(*(float4*)&a[0]), "=v"(*(float4*)&a[4]) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if constexpr (OP == 6) { REP8(asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:4096" : "=v"(*(double*)&a[0]), "=v"(*(double*)&a[2]) : "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if constexpr (OP == 7) { REP8(asm volatile("s_nop 0\n\ts_nop 0");) }
        if constexpr (OP == 8) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1" : "+v"(acc[0]), "+v"(acc[1]) : "v"(fa), "v"(fb));) }
        if constexpr (OP == 9) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %4\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %4" : "=v"(acc[0]), "=v"(acc[1]) : "v"(fa), "v"(fb), "v"(cc));) }
        // the attention slot: MFMA + 2 exp + 2 add + 1 cvt (half)
        if constexpr (OP == 10) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %2\n\tv_cvt_pk_f16_f32 %7, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(fa), "v"(fb), "v"(w[0]));) }
        if constexpr (OP == 11) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %2\n\tv_cvt_pk_bf16_f32 %7, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(fa), "v"(fb), "v"(w[0]));) }
        // ... without the adds
        if constexpr (OP == 12) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_cvt_pk_f16_f32 %5, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]) : "v"(fa), "v"(fb), "v"(w[0]));) }
        // ... with one v_pk_add_f32 for the two adds
        if constexpr (OP == 13) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_pk_add_f32 %6, %6, %7\n\tv_cvt_pk_f16_f32 %5, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]) : "v"(fa), "v"(fb), "v"(w[0]), "v"(*(double*)&a[4]), "v"(*(double*)&b[0]));) }
        // ... plus a sub per value (the no-accumulator-image form)
        if constexpr (OP == 14) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n\tv_sub_f32 %1, %1, %8\n\tv_sub_f32 %2, %2, %8\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %2\n\tv_cvt_pk_f16_f32 %7, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(fa), "v"(fb), "v"(w[0]), "v"(b[1]));) }
        // ... slot + one LDS read
        if constexpr (OP == 15) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %5, %6, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tds_read_b64_tr_b16 %8, %9\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %2\n\tv_cvt_pk_f16_f32 %7, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(fa), "v"(fb), "v"(w[0]), "v"(*(double*)&a[6]), "v"(la));) asm volatile("s_waitcnt lgkmcnt(0)"); }
        // ... exp with v_exp_f16 on packed halves?  v_exp_f16 x2
        if constexpr (OP == 16) { REP8(asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1" : "+v"(a[0]), "+v"(a[1]));) }
        if constexpr (OP == 17) { REP8(asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %2" : "=v"(a[0]), "=v"(a[1]) : "a"(b[0]));) }
        if constexpr (OP == 18) { REP8(asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(a[0]), "+v"(a[1]) : "v"(b[0]));) }
        if constexpr (OP == 19) { REP8(asm volatile("v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1" : "+v"(a[0]), "+v"(a[1]) : "v"(b[0]));) }

        if constexpr (OP == 20) { asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\tv_exp_f32 %4, %0\n\tv_exp_f32 %5, %1\n\tv_exp_f32 %6, %2\n\tv_exp_f32 %7, %3\n\tv_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\tv_exp_f32 %4, %0\n\tv_exp_f32 %5, %1\n\tv_exp_f32 %6, %2\n\tv_exp_f32 %7, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])); }
        if constexpr (OP == 21) { asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %7\n\tv_add_f32 %4, %4, %0\n\tv_add_f32 %5, %5, %1\n\tv_add_f32 %6, %6, %2\n\tv_add_f32 %7, %7, %3\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %5\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %7\n\tv_add_f32 %4, %4, %0\n\tv_add_f32 %5, %5, %1\n\tv_add_f32 %6, %6, %2\n\tv_add_f32 %7, %7, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])); }
        if constexpr (OP == 22) { asm volatile("v_cvt_pk_f16_f32 %0, %4, %4\n\tv_cvt_pk_f16_f32 %1, %5, %5\n\tv_cvt_pk_f16_f32 %2, %6, %6\n\tv_cvt_pk_f16_f32 %3, %7, %7\n\tv_cvt_pk_f16_f32 %4, %0, %0\n\tv_cvt_pk_f16_f32 %5, %1, %1\n\tv_cvt_pk_f16_f32 %6, %2, %2\n\tv_cvt_pk_f16_f32 %7, %3, %3\n\tv_cvt_pk_f16_f32 %0, %4, %4\n\tv_cvt_pk_f16_f32 %1, %5, %5\n\tv_cvt_pk_f16_f32 %2, %6, %6\n\tv_cvt_pk_f16_f32 %3, %7, %7\n\tv_cvt_pk_f16_f32 %4, %0, %0\n\tv_cvt_pk_f16_f32 %5, %1, %1\n\tv_cvt_pk_f16_f32 %6, %2, %2\n\tv_cvt_pk_f16_f32 %7, %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])); }
        if constexpr (OP == 23) { asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7\n\tv_pk_add_f32 %4, %4, %0\n\tv_pk_add_f32 %5, %5, %1\n\tv_pk_add_f32 %6, %6, %2\n\tv_pk_add_f32 %7, %7, %3\n\tv_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %5\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %7\n\tv_pk_add_f32 %4, %4, %0\n\tv_pk_add_f32 %5, %5, %1\n\tv_pk_add_f32 %6, %6, %2\n\tv_pk_add_f32 %7, %7, %3" : "+v"(*(double*)&a[0]), "+v"(*(double*)&a[2]), "+v"(*(double*)&a[4]), "+v"(*(double*)&a[6]), "+v"(*(double*)&a[8]), "+v"(*(double*)&a[10]), "+v"(*(double*)&a[12]), "+v"(*(double*)&a[14])); }
        if constexpr (OP == 24) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %1, %11\n\tv_exp_f32 %2, %12\n\tv_add_f32 %5, %5, %3\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_f16_f32 %7, %3, %4\n\tv_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %3, %13\n\tv_exp_f32 %4, %14\n\tv_add_f32 %5, %5, %1\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_f16_f32 %8, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(w[0]), "+v"(w[1]) : "v"(fa), "v"(fb), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));) }
        if constexpr (OP == 25) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %1, %11\n\tv_exp_f32 %2, %12\n\tv_cvt_pk_f16_f32 %7, %3, %4\n\tv_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %3, %13\n\tv_exp_f32 %4, %14\n\tv_cvt_pk_f16_f32 %8, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(w[0]), "+v"(w[1]) : "v"(fa), "v"(fb), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));) }
        if constexpr (OP == 26) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %1, %11\n\tv_exp_f32 %2, %12\n\tv_add_f32 %5, %5, %3\n\tv_add_f32 %6, %6, %4\n\tv_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %3, %13\n\tv_exp_f32 %4, %14\n\tv_add_f32 %5, %5, %1\n\tv_add_f32 %6, %6, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(w[0]), "+v"(w[1]) : "v"(fa), "v"(fb), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));) }
        if constexpr (OP == 27) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %1, %11\n\tv_add_f32 %5, %5, %3\n\tv_add_f32 %6, %6, %4\n\tv_cvt_pk_f16_f32 %7, %3, %4\n\tv_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %3, %13\n\tv_add_f32 %5, %5, %1\n\tv_add_f32 %6, %6, %2\n\tv_cvt_pk_f16_f32 %8, %1, %2" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(w[0]), "+v"(w[1]) : "v"(fa), "v"(fb), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));) }

        if constexpr (OP == 28) { asm volatile("v_dot2c_f32_f16 %0, %4, %8\n\tv_dot2c_f32_f16 %1, %5, %8\n\tv_dot2c_f32_f16 %2, %6, %8\n\tv_dot2c_f32_f16 %3, %7, %8\n\tv_dot2c_f32_f16 %4, %0, %8\n\tv_dot2c_f32_f16 %5, %1, %8\n\tv_dot2c_f32_f16 %6, %2, %8\n\tv_dot2c_f32_f16 %7, %3, %8\n\tv_dot2c_f32_f16 %0, %4, %8\n\tv_dot2c_f32_f16 %1, %5, %8\n\tv_dot2c_f32_f16 %2, %6, %8\n\tv_dot2c_f32_f16 %3, %7, %8\n\tv_dot2c_f32_f16 %4, %0, %8\n\tv_dot2c_f32_f16 %5, %1, %8\n\tv_dot2c_f32_f16 %6, %2, %8\n\tv_dot2c_f32_f16 %7, %3, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w[0])); }
        if constexpr (OP == 29) { asm volatile("v_dot2_f32_f16 %0, %4, %8, %0\n\tv_dot2_f32_f16 %1, %5, %8, %1\n\tv_dot2_f32_f16 %2, %6, %8, %2\n\tv_dot2_f32_f16 %3, %7, %8, %3\n\tv_dot2_f32_f16 %4, %0, %8, %4\n\tv_dot2_f32_f16 %5, %1, %8, %5\n\tv_dot2_f32_f16 %6, %2, %8, %6\n\tv_dot2_f32_f16 %7, %3, %8, %7\n\tv_dot2_f32_f16 %0, %4, %8, %0\n\tv_dot2_f32_f16 %1, %5, %8, %1\n\tv_dot2_f32_f16 %2, %6, %8, %2\n\tv_dot2_f32_f16 %3, %7, %8, %3\n\tv_dot2_f32_f16 %4, %0, %8, %4\n\tv_dot2_f32_f16 %5, %1, %8, %5\n\tv_dot2_f32_f16 %6, %2, %8, %6\n\tv_dot2_f32_f16 %7, %3, %8, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w[0])); }
        if constexpr (OP == 30) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %1, %11\n\tv_exp_f32 %2, %12\n\tv_cvt_pk_f16_f32 %7, %3, %4\n\tv_dot2c_f32_f16 %5, %8, %15\n\tv_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %3, %13\n\tv_exp_f32 %4, %14\n\tv_cvt_pk_f16_f32 %8, %1, %2\n\tv_dot2c_f32_f16 %6, %7, %15" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(w[0]), "+v"(w[1]) : "v"(fa), "v"(fb), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(w[2]));) }
        if constexpr (OP == 31) { REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %1, %11\n\tv_exp_f32 %2, %12\n\tv_cvt_pk_f16_f32 %7, %3, %4\n\tv_dot2_f32_f16 %5, %8, %15, %5\n\tv_mfma_f32_32x32x16_f16 %0, %9, %10, %0\n\tv_exp_f32 %3, %13\n\tv_exp_f32 %4, %14\n\tv_cvt_pk_f16_f32 %8, %1, %2\n\tv_dot2_f32_f16 %6, %7, %15, %6" : "+v"(acc[0]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(w[0]), "+v"(w[1]) : "v"(fa), "v"(fb), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(w[2]));) }
        // MFMA slot with the mixed form: v_fma_mix? -- left out
    }
    T1(OP, IT * 16);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + acc[i & 3][i];
    for (int i = 0; i < 8; ++i) s += w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 64 * 16);
    hipMemset(cyc, 0, 64 * 16);
    const char* names[] = {"v_exp_f32", "v_add_f32", "v_cvt_pk_f16_f32", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "ds_read_b128 (issue)", "ds_read_b64_tr_b16 (issue)", "s_nop 0",
                           "mfma 32x32x16 f16 in place", "mfma, C from other registers", "slot: mfma 2exp 2add cvt_f16", "slot: mfma 2exp 2add cvt_bf16", "slot: mfma 2exp cvt_f16",
                           "slot: mfma 2exp pk_add cvt_f16", "slot: mfma 2sub 2exp 2add cvt_f16", "slot + ds_read_tr", "v_exp_f16", "v_accvgpr_read", "v_mul_f32", "v_fma_f32", "v_exp_f32 independent", "v_add_f32 independent", "v_cvt_pk_f16_f32 independent", "v_pk_add_f32 independent", "skewed slot: mfma 2exp 2add cvt", "skewed slot without the adds", "skewed slot without the cvt", "skewed slot with ONE exp", "v_dot2c_f32_f16 independent", "v_dot2_f32_f16 independent", "skewed slot: mfma 2exp cvt dot2c", "skewed slot: mfma 2exp cvt dot2"};
    for (int rep = 0; rep < 2; ++rep) {
#define L(N) k<N><<<256, 256>>>(out, cyc);
        L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31)
        hipDeviceSynchronize();
    }
    unsigned long long h[128];
    hipMemcpy(h, cyc, 64 * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) {
        const double per = (double)h[2 * i] / (double)h[2 * i + 1];
        const bool slot = (i >= 10 && i <= 15);
        printf("%-36s %7.2f cycles per %s\n", names[i], slot ? per * 2 : per, slot ? "slot" : "instruction");     // slots: 8 per iteration, counted as 16
    }
    return 0;
}
