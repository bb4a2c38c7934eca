/* hip_tagsearch_debug.h -- diagnostic entry points of libhip_tagsearch.so.
 *
 * NOT part of the drop-in boundary (include/hip_tagsearch.h): nothing in the reference binds to these.  They exist for
 * the kernel tests (tests/test_gpu_gemm.py), the standalone GEMM timers (tools/gemm_bench.py) and bench.py's clock probe.
 * They may change or disappear between versions; HIPTS_ABI_VERSION does not cover them.
 */
#ifndef HIP_TAGSEARCH_DEBUG_H
#define HIP_TAGSEARCH_DEBUG_H

#include <stddef.h>
#include <stdint.h>

#include "hip_tagsearch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Average device time of `iters` back-to-back launches of the bf16 GEMM [M,K] x [N,K]^T with epilogue `epi`
 * (the GemmEpilogue enumerator of csrc/vit_internal.h) on synthetic operands. */
int hiptsdbg_gemm_time(int M, int N, int K, int epi, int iters, float* ms_out);
/* The same, and the shader clock (GHz) held inside the main loop of the stamped workgroup (s_memtime against the
 * 100 MHz s_memrealtime); 0 when the library was built without HIPTS_GEMM_STAMPS support for that epilogue. */
int hiptsdbg_gemm_clock(int M, int N, int K, int epi, int iters, float* ms_out, float* loop_ghz);
/* One GEMM launch through the 8-wave loop (csrc/gemm.hip) and through the 4-wave loop (csrc/gemm4.hip) on the same random operands: number
 * of output bytes that differ (must be 0) and the average launch time each way (ms_out[0]: 8-wave, ms_out[1]: 4-wave).  epi: 4 (GELU),
 * 1 (QK; N % 3 == 0: fused q | k | v), 13 (RESID_XG); M % 256 == 0, M % 784 == 0, N % 256 == 0, K % 128 == 0. */
int hiptsdbg_gemm_q4_compare(int M, int N, int K, int epi, int f16, int iters, long long* mismatch_out, float* ms_out);
/* y_host[i] = the GELU of the fc1 epilogue (csrc/gemm.hip gelu_f4) of x_host[i]; n % 4 == 0; tanh_form as hipts_vit_config::gelu_tanh. */
int hiptsdbg_gelu(const float* x_host, int n, int tanh_form, float* y_host);
/* The depthwise 7 x 7 (padding 3) of the CCIP encoder's SepConv blocks on its own: in / out IEEE-half bit patterns [batch][H][H][C] (host),
 * w float32 [C][49]; mode 0: the float32-FMA kernel, 1: the matrix-core kernel with its tile chosen by H, 2 / 3: its 32- / 48-column tile;
 * ms_out: average device time of `iters` launches after one untimed launch. */
int hiptsdbg_dwconv7(const uint16_t* in_f16, const float* w, uint16_t* out_f16, int batch, int H, int C, int mode, int iters, float* ms_out);
/* Phase time stamps (100 MHz) of one workgroup of the matrix-core kernel's last launch; zeros unless csrc/ccip.hip was built with
 * -DHIPTS_DW_STAMPS=<workgroup> (tools/dwconv_stamps.py). */
int hiptsdbg_dwconv7_stamps(unsigned long long* host, int n);
/* The fused MLP of the CCIP encoder's stages 0-1 on its own (csrc/mlp.hip): x[m] = rs * x[m] + StarReLU(xn[m] W1^T) W2^T, xn_out[m] = LayerNorm(x[m]) * gamma.
 * Host arrays: xn / xn_out IEEE-half bits [M][C], w1 [4C][C], w2 [C][4C], x [M][C] in / out, res_scale / gamma [C] or null; C = 128 or 256;
 * ms_out: average device time of iters - 1 launches (iters >= 2); waves: 4 / 8 waves per workgroup, 0 = chosen by the size of the launch. */
int hiptsdbg_mlp_fused(const uint16_t* xn, const float* w1, const float* w2, float* x, const float* res_scale, const float* gamma, uint16_t* xn_out,
                       int M, int C, float star_s, float star_b, float eps, int iters, float* ms_out, int waves);
/* Phase time stamps (100 MHz) of one wave of the fused MLP kernel's last launch; zeros unless csrc/mlp.hip was built with
 * -DHIPTS_MLP_STAMPS=<workgroup> (tools/mlp_stamps.py). */
int hiptsdbg_mlp_stamps(unsigned long long* host, int n);
/* Host only, no GPU call: the weight layouts the two round-4 CCIP kernels read (tests/test_host_layouts.py builds them again in numpy).
 * hiptsdbg_mlp_weight_image: per 32 hidden units one block of IEEE-half bits [32 rows of w1, pitch C + 16 halves][C rows of w2, pitch 40 halves,
 * position 8 q + e of a row = hidden unit 16 (e >> 2) + 4 q + (e & 3) of the chunk]; out_halves = (4C / 32) * (32 * (C + 16) + C * 40).
 * hiptsdbg_dw_toeplitz: out u32 [channels][7][64] -- per (channel, kernel row) the 7 weights as halves Z[16 + kx] inside 48 zero halves;
 * lane L < 24 holds (Z[2L], Z[2L+1]), lane 32 + L holds (Z[2L+1], Z[2L+2]), the other lanes zero. */
int hiptsdbg_mlp_weight_image(const float* w1, const float* w2, int C, uint16_t* out, long long out_halves);
int hiptsdbg_dw_toeplitz(const float* w, int channels, uint32_t* out);
/* out_host float32 [M][N] = A W^T for bf16 bit patterns a_bf16 [M][K], w_bf16 [N][K] (plain epilogue). */
int hiptsdbg_gemm_run(int M, int N, int K, const uint16_t* a_bf16, const uint16_t* w_bf16, float* out_host);
/* The e4m3 operand path: a_f32 / w_f32 are quantised by the library (per-tensor power-of-two weight scale returned in
 * w_exp); kind 0: float32 output, 1: e4m3 output through the StarReLU epilogue's store path. */
int hiptsdbg_gemm8_run(int M, int N, int K, const float* a_f32, const float* w_f32, int kind, void* out_host, int* w_exp);
/* Copies a named workspace tensor of the last forward ("x", "xn", "q", "k", "v", "att", "hmid") to the host. */
int hiptsdbg_vit_dump(hipts_vit_t* h, const char* name, void* out_host, size_t max_bytes, size_t* bytes);

/* The attention kernel alone: q, k 16-bit patterns [batch * heads][tokens_pad][head_dim] (q pre-scaled by head_dim^-0.5 * log2 e, rows past
 * `tokens` zero), vT [batch * heads][head_dim][tokens_pad]; out 16-bit patterns [batch][tokens][heads * head_dim].  f16: IEEE half operands. */
int hiptsdbg_attention_run(const uint16_t* q, const uint16_t* k, const uint16_t* vT, uint16_t* out_host, int batch, int heads, int tokens,
                           int tokens_pad, int head_dim, int f16);
/* The same launch timed: average device microseconds over `iters` launches (HIP events) with the chip to itself. */
int hiptsdbg_attention_time(const uint16_t* q, const uint16_t* k, const uint16_t* vT, int batch, int heads, int tokens, int tokens_pad,
                            int head_dim, int f16, int iters, double* avg_us);

/* The head_dim-64 attention kernel of round 3 (csrc/attn2.hip) alone: as above but v in its natural layout [batch * heads][tokens_pad][64].
 * iters == 0: one launch, result in out_host; iters > 0: average device microseconds of that many launches in *avg_us (out_host unused).
 * variant 0: the default geometry; 1: 4 waves x 64 query rows, 2: 8 waves x 32, 3: 4 waves x 32. */
int hiptsdbg_attention2(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out_host, int batch, int heads, int tokens, int tokens_pad,
                        int f16, int variant, int iters, double* avg_us);

/* Measurement-only builds of csrc/attn2.hip (-DHIPTS_X_STAMPS=<workgroup>): the cycle stamps of one wave (tools/attn2_check.py stamps);
 * HIPTS_ERR_STATE in an ordinary build. */
int hiptsdbg_attention2_stamps(unsigned long long* host, int n);

/* One-query search path (hipts_search with nq == 1): how many candidates its threshold step collected for the last query and
 * whether the ranking used them (1) or fell through to the exact radix select (0). */
int hiptsdbg_search1_last(hipts_bm25_t* h, uint32_t* candidates, uint32_t* took_candidate_path);
/* measurement-only builds (-DHIPTS_X_TOPK_STAMPS=<workgroup>): the 16 wall-clock stamps (10 ns units) of the last batched hipts_topk launch */
int hiptsdbg_topk_stamps(unsigned long long* host16);

#ifdef __cplusplus
}
#endif
#endif /* HIP_TAGSEARCH_DEBUG_H */
