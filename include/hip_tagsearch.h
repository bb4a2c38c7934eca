/* hip_tagsearch.h -- C ABI of libhip_tagsearch.so (MI355X / gfx950 only).
 *
 * The reference (ryogrid/anime-illust-image-searcher) has no FFI: its hot path sits behind
 * plain Python call sites into timm/torch, gensim and numpy.  Each entry point below is what
 * a ctypes binding for one of those call sites binds to; the call site it replaces is cited
 * as  file:line  relative to the reference repository.  INTEGRATION.md shows the
 * reference-side stubs.
 *
 * Conventions
 *   - every function returns HIPTS_OK (0) or a negative hipts_status; the message of the last
 *     failure on the calling thread is available from hipts_last_error().
 *   - handles are opaque, created/destroyed explicitly, bound to one device, single-caller.
 *   - the caller owns every input and output buffer.  `memspace` arguments say whether a
 *     pointer is host memory (HIPTS_HOST) or memory of the handle's device (HIPTS_DEVICE).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  With device outputs
 *     the call returns after enqueueing; with host outputs it returns after the copy-back.
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point
 *     fails with HIPTS_ERR_NO_DEVICE.
 */
#ifndef HIP_TAGSEARCH_H
#define HIP_TAGSEARCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPTS_ABI_VERSION 1

typedef enum hipts_status {
    HIPTS_OK = 0,
    HIPTS_ERR_INVALID = -1,     /* bad argument / shape the kernels do not support      */
    HIPTS_ERR_NO_DEVICE = -2,   /* no gfx950 device / device index out of range          */
    HIPTS_ERR_HIP = -3,         /* a HIP runtime call or kernel launch failed            */
    HIPTS_ERR_OOM = -4,
    HIPTS_ERR_STATE = -5        /* e.g. forward() before all weights were set            */
} hipts_status;

typedef enum hipts_memspace { HIPTS_HOST = 0, HIPTS_DEVICE = 1 } hipts_memspace;

int hipts_abi_version(void);
/* copies the calling thread's last error message (NUL terminated, truncated to n) */
int hipts_last_error(char* buf, size_t n);
int hipts_device_count(int* count);
/* sizeof() of a configuration structure as THIS LIBRARY was compiled: kind 0 = hipts_vit_config_t, 1 = hipts_eva_config_t,
 * 2 = hipts_ccip_config_t.  A binding in another language compares its own structure's size with it before the first
 * hipts_*_create (tests/test_abi.py does, for the ctypes structures and for the stubs printed in INTEGRATION.md). */
int hipts_sizeof_config(int kind, size_t* bytes);

/* ------------------------------------------------------------------------------------------
 * ViT tagger forward.   Replaces timm `model.forward(x)` + `F.sigmoid`      tagging.py:174,176
 * (model construction / weight load: tagging.py:146-148).
 * Graph: patch-embed conv k=s=patch -> +pos_embed -> depth x {LN, QKV, softmax(QK^T/sqrt(hd))V,
 * proj, +res, LN, fc1, GELU, fc2, +res} -> final LN -> token mean -> head  (no class token).
 * Matrix weights are held as bf16 (MFMA operands), everything else and all accumulation,
 * the residual stream, LayerNorm and softmax statistics in float32.
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_vit hipts_vit_t;

typedef struct hipts_vit_config {
    int32_t image_size;      /* 448                                   */
    int32_t patch;           /* 16                                    */
    int32_t dim;             /* 768                                   */
    int32_t depth;           /* 12                                    */
    int32_t heads;           /* 12  (head dim must be 64)             */
    int32_t mlp_dim;         /* 3072                                  */
    int32_t num_classes;     /* 10861                                 */
    float   ln_eps;          /* 1e-6                                  */
    int32_t gelu_tanh;       /* 1: tanh approximation, 0: erf         */
    int32_t pool_then_norm;  /* 0: LN(tokens) then mean (fc_norm=False); 1: mean then LN */
    int32_t max_batch;       /* workspace is sized for this many images per forward call */
    int32_t operand_f16;     /* bit 0 -- 0: bf16 MFMA operands (BASELINE.json configs[1]); 1: IEEE half operands --
                                same MFMA rate, 8x smaller activation rounding: keeps |dlogit| <= 1e-3 also on flat
                                images, where bf16 rounding is identical on every token and does not average out.
                                bit 4 (HIPTS_OPERAND_SPLIT_ATT) -- the attention output is handed to the output
                                projection as a hi | lo pair of 16-bit halves (22 significant bits with half operands):
                                removes the error class "one colour everywhere" (every token's attention output carries
                                the SAME rounding) for 2 dim^2 more flops per token and layer */
} hipts_vit_config_t;
#define HIPTS_OPERAND_F16 1
#define HIPTS_OPERAND_SPLIT_ATT 16

int hipts_vit_create(const hipts_vit_config_t* cfg, int device, hipts_vit_t** out);
int hipts_vit_destroy(hipts_vit_t* h);
/* One tensor of the checkpoint by its timm state_dict key ("patch_embed.proj.weight",
 * "pos_embed", "blocks.3.attn.qkv.weight", "norm.bias", "head.weight", ...); `data` is host
 * float32 in the timm layout, `numel` its element count (checked).     tagging.py:147-148 */
int hipts_vit_set_tensor(hipts_vit_t* h, const char* key, const float* data, int64_t numel);
/* images: uint8 [batch][H][W][3] RGB (what PIL hands to the timm transform for an image that
 * is already image_size^2); the kernel applies ToTensor + Normalize(.5,.5) + the RGB->BGR flip
 * of tagging.py:241-243 while forming the patch matrix.  logits_out / probs_out: float32
 * [batch][num_classes], either may be NULL. */
int hipts_vit_forward_u8(hipts_vit_t* h, const uint8_t* images, int images_memspace, int batch,
                         float* logits_out, float* probs_out, int out_memspace, void* stream);
/* x: float32 [batch][3][H][W], already normalised and BGR -- exactly the tensor
 * tagging.py:164 stacks and :174 passes to model.forward. */
int hipts_vit_forward_f32(hipts_vit_t* h, const float* x, int x_memspace, int batch,
                          float* logits_out, float* probs_out, int out_memspace, void* stream);
/* Per-kernel timing for roofline accounting (bench.py): while enabled (enable = n > 0), every kernel
 * launch of every n-th forward() call is bracketed by HIP events on the stream it is launched on
 * (n = 1: every call; sampling keeps the ~220 event records per call out of most steps).  read() resolves them
 * (synchronises) and returns, for one kernel category, the summed device time, the number of
 * launches and the ALGORITHMIC flops / bytes those launches stand for.
 * Categories are numbered 0 .. HIPTS_VIT_PROF_CATEGORIES-1; name() gives the kernel's name. */
#define HIPTS_VIT_PROF_CATEGORIES 11
int hipts_vit_profile_enable(hipts_vit_t* h, int enable);
/* Which categories are recorded while profiling is enabled (bit c = category c; default: all).  bench.py records only the dominant
 * kernel's launches inside its timed region -- the events of all ~130 launches of a step cost 0.7 % of the step -- and the full
 * breakdown in extra steps after it. */
int hipts_vit_profile_select(hipts_vit_t* h, uint32_t category_mask);
int hipts_vit_profile_read(hipts_vit_t* h, int category, double* total_ms, int64_t* launches,
                           double* total_flops, double* total_bytes);
int hipts_vit_profile_name(int category, char* buf, size_t n);
/* How many sub-batches a forward() call is split into, each on its own internal HIP stream (forked from
 * and joined to the caller's stream with events): 0 = default (2, or HIPTS_VIT_STREAMS), 1 = the whole
 * batch as one sequence of launches on the caller's stream, up to 4.  A kernel of one sub-batch then
 * fills the partial last round and the epilogue bubbles of the other's.  Results do not depend on it
 * (every image is computed by the same instruction sequence). */
int hipts_vit_set_sub_batches(hipts_vit_t* h, int n);
/* Deferred join (device outputs only): with it on, forward() returns once the sub-batch streams are fed and
 * does NOT make the caller's stream wait for them; hipts_vit_join(h, stream) makes `stream` wait for the
 * outputs of the most recent forward.  A tagging loop joins on the stream that consumes the probabilities
 * and keeps submitting forwards on the other: the sub-batch streams then run from one batch straight into
 * the next, and the low-occupancy first and last kernels of a forward (patch matrix, pooling, head) overlap
 * the other half's bulk.  Each sub-batch stream stays in order and workspaces are carved by image, so consecutive
 * forwards of the SAME batch size from device-resident input reuse them safely; a forward whose batch size, sub-batch
 * count or input memspace differs from the unjoined one before it first waits (on `stream`) for all of that forward.
 * The caller double-buffers what it hands in as outputs. */
int hipts_vit_set_deferred_join(hipts_vit_t* h, int on);
int hipts_vit_join(hipts_vit_t* h, void* stream);
/* algorithmic FLOPs of one image's forward (2*M*N*K of every contraction), for roofline use */
int hipts_vit_flops_per_image(const hipts_vit_t* h, double* flops);

/* ------------------------------------------------------------------------------------------
 * EVA02 tagger forward -- the model the reference actually loads (tagging.py:45: wd-eva02-large-tagger-v3 =
 * timm eva02_large_patch14_448).  Same call sites as the ViT above (tagging.py:174,176).
 * Graph: patch-embed conv k=s=patch -> [cls | patches] + pos_embed -> depth x {LN, q/k/v (separate projections,
 * k without bias), 2-D axial RoPE on the patch tokens, softmax(QK^T/8)V, proj, +res, LN, SwiGLU
 * (silu(fc1_g) * fc1_x -> LayerNorm -> fc2), +res} -> mean over the patch tokens -> fc_norm -> head.
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_eva hipts_eva_t;

typedef struct hipts_eva_config {
    int32_t image_size;      /* 448                                   */
    int32_t patch;           /* 14                                    */
    int32_t dim;             /* 1024 (multiple of 64, at most 1024)   */
    int32_t depth;           /* 24                                    */
    int32_t heads;           /* 16  (head dim must be 64)             */
    int32_t mlp_hidden;      /* 2730 = int(dim * 8 / 3)               */
    int32_t num_classes;     /* 10861                                 */
    float   ln_eps;          /* 1e-6                                  */
    int32_t rope_ref_grid;   /* 16: RoPE positions are rescaled to this reference grid (ref_feat_shape) */
    int32_t max_batch;
    int32_t operand_f16;     /* as hipts_vit_config_t.operand_f16     */
} hipts_eva_config_t;

int hipts_eva_create(const hipts_eva_config_t* cfg, int device, hipts_eva_t** out);
int hipts_eva_destroy(hipts_eva_t* h);
/* timm `Eva` state_dict keys: "patch_embed.proj.weight", "cls_token", "pos_embed", "blocks.0.attn.q_proj.weight",
 * "blocks.0.attn.k_proj.weight", "blocks.0.mlp.fc1_g.weight", "blocks.0.mlp.norm.weight", "fc_norm.weight", "head.weight", ... */
int hipts_eva_set_tensor(hipts_eva_t* h, const char* key, const float* data, int64_t numel);
/* same contracts as hipts_vit_forward_u8 / _f32 */
int hipts_eva_forward_u8(hipts_eva_t* h, const uint8_t* images, int images_memspace, int batch, float* logits_out, float* probs_out,
                         int out_memspace, void* stream);
int hipts_eva_forward_f32(hipts_eva_t* h, const float* x, int x_memspace, int batch, float* logits_out, float* probs_out,
                          int out_memspace, void* stream);
int hipts_eva_flops_per_image(const hipts_eva_t* h, double* flops);

/* ------------------------------------------------------------------------------------------
 * CCIP feature encoder.   Replaces the onnxruntime session of gen_cfeatures.py:112-118 and its
 * `session.run(['output'], {'input': x})` call (gen_cfeatures.py:158; batching :133-159).
 * Graph: CAFormer (timm MetaFormer) -- stem conv 7x7 s4 + bias-free LN; four stages (3x3 s2 conv
 * downsampling with pre-norm between them) of blocks x = rs1*x + mixer(LN(x)); x = rs2*x + MLP(LN(x));
 * mixer = SepConv (1x1 -> StarReLU -> depthwise 7x7 -> 1x1) in the first stages, self-attention with
 * head_dim 32 (no biases) from `attn_from_stage` on; MLP = fc1 -> StarReLU -> fc2 (x4, no biases);
 * rs = per-channel res_scale (attention stages only); output = LN(global average pool), dims[3] wide.
 * Matrix weights are bf16 MFMA operands; accumulation, residual stream, LayerNorm, softmax in float32.
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_ccip hipts_ccip_t;

typedef struct hipts_ccip_config {
    int32_t image_size;       /* 384 (a multiple of 32)                                   */
    int32_t dims[4];          /* 128, 256, 512, 768 (multiples of 64)                      */
    int32_t depths[4];        /* 3, 12, 18, 3                                              */
    int32_t head_dim;         /* 32                                                        */
    int32_t attn_from_stage;  /* 2: stages 0,1 SepConv, stages 2,3 attention               */
    float   ln_eps;           /* 1e-6                                                      */
    int32_t max_batch;        /* workspace is sized for this many images per forward call  */
    int32_t operand_f16;      /* 0: bf16, 1: IEEE half MFMA operands (hipts_vit_config_t.operand_f16 bit 0).  The value 2 of
                               * rounds 1-3 (OCP e4m3 operands for pwconv2 / fc1 / fc2: BASELINE.json configs[4]'s "fp8 MFMA") is
                               * refused since round 4 (HIPTS_ERR_INVALID): cosine 0.968 against the float32 forward, not
                               * faster than half operands, and per-32-element block scales do not change that (DESIGN.md
                               * section 6, tools/ccip_fp8_emulation.py).                                             */
} hipts_ccip_config_t;

int hipts_ccip_create(const hipts_ccip_config_t* cfg, int device, hipts_ccip_t** out);
int hipts_ccip_destroy(hipts_ccip_t* h);
/* One tensor by its timm MetaFormer state_dict key ("stem.conv.weight", "stages.1.downsample.conv.weight",
 * "stages.0.blocks.2.token_mixer.dwconv.weight", "stages.2.blocks.0.res_scale1.scale", "head.norm.bias", ...);
 * 1x1 convolutions may arrive as [out,in,1,1] or [out,in] (same element count). */
int hipts_ccip_set_tensor(hipts_ccip_t* h, const char* key, const float* data, int64_t numel);
/* images: uint8 [batch][S][S][3] RGB, already S x S (the resize of gen_cfeatures.py:101 done by the
 * caller); the kernel applies /255 and the CLIP mean / std of gen_cfeatures.py:103-110 while forming
 * the stem's patch matrix.  features_out: float32 [batch][dims[3]]. */
int hipts_ccip_forward_u8(hipts_ccip_t* h, const uint8_t* images, int images_memspace, int batch, float* features_out,
                          int out_memspace, void* stream);
/* x: float32 [batch][3][S][S], already normalised -- exactly the `input` array of gen_cfeatures.py:158. */
int hipts_ccip_forward_f32(hipts_ccip_t* h, const float* x, int x_memspace, int batch, float* features_out,
                           int out_memspace, void* stream);
int hipts_ccip_flops_per_image(const hipts_ccip_t* h, double* flops);

/* PIL's resample on the device, bit for bit (Pillow libImaging/Resample.c, 8-bit RGB): replaces the host-side resize of the input
 * pipeline -- the timm eval transform's Resize(bicubic) applied to the padded square of tagging.py:100-120 (tagging.py:241), and
 * image.resize((384, 384), BILINEAR) of gen_cfeatures.py:101 -- so that decode workers only decode.
 * src: uint8 [src_h][src_w][3] (host or device memory); dst_device: uint8 [dst_h][dst_w][3]; filter: PIL's enumerator, 2 = BILINEAR,
 * 3 = BICUBIC (both with PIL's antialiasing: the filter support grows with the shrink factor). */
int hipts_resize_u8(const uint8_t* src, int src_memspace, int src_h, int src_w, uint8_t* dst_device, int dst_h, int dst_w, int filter,
                    int device, void* stream);

/* A batch of decoded images of different sizes -> uint8 [n][size][size][3] on the device: image i is hw[2 i] x hw[2 i + 1] x 3 at
 * src_base + i * slot_stride (host or device memory: the ring slots of the decode-only worker processes, hiptagsearch/pipeline.py).
 * pad_square != 0: centred on a white max(h, w) square first -- Predictor.prepare_image, reference tagging.py:100-120 -- then the
 * Resize of hipts_resize_u8 (filter 3: tagging.py:241's transform; filter 2, pad_square 0: gen_cfeatures.py:101).  Host images are
 * copied with hipMemcpyAsync (register the ring as pinned memory for asynchronous DMA); all work is ordered on `stream`, the call does
 * not synchronise: keep the source bytes unchanged until the stream has passed this point. */
int hipts_resize_batch_u8(const uint8_t* src_base, int src_memspace, int64_t slot_stride, const int32_t* hw, int n, int pad_square,
                          uint8_t* dst_device, int size, int filter, int device, void* stream);

/* ---- Hybrid JPEG decode (round 4).  What stands behind it in the reference: PIL's Image.open(path) inside Predictor.prepare_image /
 * gen_image_tensor (tagging.py:100-120, 234-252) and _preprocess_image (gen_cfeatures.py:285-295), i.e. libjpeg-turbo with its defaults.
 * The serial half (markers, Huffman decoding) runs on the host, in the decode worker processes; the per-block / per-pixel half (accurate
 * integer inverse DCT, "fancy" chroma upsampling, YCbCr -> RGB) on the device, byte for byte libjpeg's arithmetic.
 *
 * hipts_jpeg_entropy_decode (HOST ONLY: also exported by libhipts_jpeg_host.so, which has no GPU runtime behind it -- the library the
 * worker processes load): file bytes -> a "slot" = hipts_jpeg_header (csrc/jpeg_slot.h) + int16 coefficient blocks.
 * Baseline, extended-sequential and progressive Huffman files.  Returns 0; 1: not a file the fast path takes (arithmetic coding / a
 * progression that leaves coefficients out or below full precision / 12-bit / CMYK / RGB-coded / sampling other than 4:4:4, 4:2:2,
 * 4:2:0 / a sequential file in several scans / smaller than 16 x 16 / not a JPEG); 2: the slot is too small; 3: the stream is irregular (truncated, bad codes,
 * restart markers out of order, coefficients beyond what an 8-bit image can produce).  On 1..3 the caller decodes the file with Pillow as before.  hipts_jpeg_slot_bytes: an upper bound of the
 * slot a width x height image needs. */
int hipts_jpeg_entropy_decode(const uint8_t* file_bytes, int64_t n, void* slot, int64_t slot_bytes);
int64_t hipts_jpeg_slot_bytes(int width, int height);

/* One slot (host memory) -> uint8 [height][width][3] RGB = the bytes of PIL.Image.open(file).convert("RGB"); rgb_out in host or device
 * memory with room for out_capacity bytes.  Synchronises `stream`. */
int hipts_jpeg_decode_rgb(const void* slot, int64_t slot_bytes, uint8_t* rgb_out, int out_memspace, int64_t out_capacity, int device, void* stream);

/* The batch entry of the decode pipeline (hiptagsearch/pipeline.py): n ring slots (host memory, slot i at slots + i * slot_stride), each
 * either a coefficient slot (kinds[i] = 1) or a decoded uint8 [h][w][3] image (kinds[i] = 0: the files Pillow had to decode), image i of
 * hw[2 i] x hw[2 i + 1] pixels -> uint8 [n][size][size][3] on the device: decode, then exactly hipts_resize_batch_u8 (pad_square, filter).
 * Everything is ordered on `stream`, nothing is synchronised: keep the slots unchanged until the stream has passed this point. */
int hipts_jpeg_batch_u8(const uint8_t* slots, int64_t slot_stride, const int32_t* kinds, const int32_t* hw, int n, int pad_square,
                        uint8_t* dst_device, int size, int filter, int device, void* stream);

/* The synthetic image corpus of the benchmark configurations (BASELINE.json configs[3]: "1M synthetic images sharded
 * 8xMI355X"), generated on the device with no host I/O: images first_index .. first_index + count - 1 of the corpus `seed`, uint8
 * [count][image_size][image_size][3], every byte a counter-hash of (seed, GLOBAL image index, byte offset) -- so a rank produces its own
 * contiguous block (tagging.py:276-359 cut by rank) and any rank can reproduce any image.  Replaces the decode of tagging.py:100-120 for
 * benchmark input only. */
int hipts_synth_images_u8(uint8_t* images_device, int64_t first_index, int64_t count, int image_size, uint64_t seed, int device,
                          void* stream);

/* Pairwise CCIP differences.   Replaces `metric_model.run(['output'], {'input': features})` of ccip_batch_differences /
 * ccip_difference                                                                  gen_cfeatures.py:212-274 (used by webui.py:303-335).
 * features: float32 [n][dim] (dim = 768); diff_out: float32 [n][n].  kind 0: 1 - cosine of the rows (unit-normalised in float32, Gram
 * matrix as the k-ordered fmaf chain of the dense index) -- how BASELINE.json configs[4] restates the reference's opaque metric graph;
 * other kinds are reserved for a real metric head (HIPTS_ERR_INVALID until one is loaded).  The cut applied to these differences is a
 * calibrated parameter of the caller (hiptagsearch.cfeatures.calibrate_threshold), not the reference's metric-model constant. */
int hipts_ccip_metric(const float* features, int features_memspace, int n, int dim, int kind, float* diff_out, int out_memspace,
                      int device, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tag selection.   Replaces the per-image numpy/Python post-processing   tagging.py:61-66,185-227
 * (float64 MCut threshold per category, strict '>' filter, stable descending order).
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_tagsel hipts_tagsel_t;
/* category[c]: 0 general, 4 character, anything else ignored (9 = rating)  tagging.py:137-139 */
int hipts_tagsel_create(const int32_t* category, int num_classes, int device, int max_batch,
                        hipts_tagsel_t** out);
int hipts_tagsel_destroy(hipts_tagsel_t* h);
/* probs: float32 [batch][num_classes].  Per image writes counts_out[2] = {#general, #character},
 * ids_out[row_cap] = label ids, general tags first then character tags, each group in output
 * order; thresh_out[2] = the float64 thresholds used.  A row that selects more than row_cap
 * labels is truncated and reports its full counts (caller can detect n_g+n_c > row_cap). */
int hipts_tagsel_run(hipts_tagsel_t* h, const float* probs, int probs_memspace, int batch,
                     double general_thresh, int general_mcut, double character_thresh, int character_mcut,
                     int32_t* counts_out, int32_t* ids_out, int row_cap, double* thresh_out,
                     int out_memspace, void* stream);

/* Device-resident variant producing the fixed-width tag rows that ranks all-gather: row b of
 * rows_device (int32 [batch][row_width]) = {#general, #character, ids[row_width-2]}. */
int hipts_tagsel_run_rows(hipts_tagsel_t* h, const float* probs_device, int batch,
                          double general_thresh, int general_mcut, double character_thresh, int character_mcut,
                          int32_t* rows_device, int row_width, void* stream);

/* ------------------------------------------------------------------------------------------
 * The one collective of the indexing path: ranks tag contiguous blocks of the file list -- the loop of
 * tagging.py:276-359 cut by rank, one process per GPU -- and ONE all-gather of their fixed-width rows (hipts_tagsel_run_rows; float32
 * feature rows reinterpreted as int32 for gen_cfeatures.py:337-459) restores file order: out_device [world][rows_per_rank][row_width],
 * rank order == file order.  RCCL over xGMI; resolved at run time (a PyTorch process uses the librccl it already maps).  The Python CLIs
 * use torch.distributed for the same exchange; these entry points are for hosts in other languages.  Rank 0 creates the 128-byte id
 * and hands it to the other ranks by its own means (file, socket, MPI ...).
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_comm hipts_comm_t;
#define HIPTS_COMM_ID_BYTES 128
int hipts_comm_unique_id(uint8_t* id_out, size_t bytes);
int hipts_comm_create(const uint8_t* id, size_t bytes, int rank, int world, int device, hipts_comm_t** out);
int hipts_comm_destroy(hipts_comm_t* comm);
int hipts_allgather_rows(hipts_comm_t* comm, const int32_t* rows_device, int64_t rows_per_rank, int row_width, int32_t* out_device,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * BM25.   build replaces gen_and_save_bm25_index                            genmodel.py:51-99
 *         score replaces compute_bm25_scores(query_weights=...)             webui.py:119-172
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_bm25 hipts_bm25_t;
/* Documents in CSR form: token ids of document d are term_ids[doc_ptr[d] .. doc_ptr[d+1]) in
 * document order, duplicates allowed (they count into tf), id < 0 = tag not in the dictionary
 * (dropped, as genmodel.py:59).  vocab = number of dictionary ids. */
int hipts_bm25_build(const int64_t* doc_ptr, const int32_t* term_ids, int64_t num_docs, int32_t vocab,
                     int device, hipts_bm25_t** out);
int hipts_bm25_destroy(hipts_bm25_t* h);
int hipts_bm25_info(const hipts_bm25_t* h, int64_t* num_docs, int64_t* nnz, int32_t* vocab, double* avgdl);
/* The five objects genmodel.py:84-97 pickles, as arrays (any pointer may be NULL):
 * csr_ptr[D+1], csr_term[nnz], csr_tf[nnz] (per-document term->tf in first-occurrence order),
 * doc_len[D], df[vocab], idf[vocab] (0 where df == 0). */
int hipts_bm25_export(const hipts_bm25_t* h, int64_t* csr_ptr, int32_t* csr_term, int32_t* csr_tf,
                      int64_t* doc_len, int64_t* df, double* idf);
/* Replace the idf table (float64 [vocab]).  build computes idf = log(1+(D-df+.5)/(df+.5)) with
 * the host libm, which is within 1 ulp of -- but not always bit-identical to -- numpy's log
 * (genmodel.py:81); a caller that holds the reference's own `bm25_idf` pickle, or wants
 * bit-identical pickles, sets those values here. */
int hipts_bm25_set_idf(hipts_bm25_t* h, const double* idf);
/* Replace avgdl (genmodel.py:76).  A handle built over one shard of the documents scores with the
 * statistics of the WHOLE corpus: the caller installs the global idf table and the global avgdl. */
int hipts_bm25_set_avgdl(hipts_bm25_t* h, double avgdl);
/* nq queries; query i has terms q_terms[q_ptr[i]..q_ptr[i+1]) with weights q_weights[...] in the
 * iteration order of the reference's dict (webui.py:139).  weight < 0: exclude, weight > 1000:
 * required with weight-1000 (webui.py:154-170).  scores_out: float64 [nq][num_docs]. */
int hipts_bm25_score(hipts_bm25_t* h, const int32_t* q_terms, const double* q_weights, const int32_t* q_ptr,
                     int nq, double* scores_out, int out_memspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense similarity index.   Replaces gensim Similarity / MatrixSimilarity as the reference
 * uses it: Similarity(prefix,[vec],num_features) / add_documents   genmodel.py:170-173,
 * gen_cfeatures.py:311-314;  index[query]                         webui.py:205,352;
 * vector_by_id / len                                               webui.py:306-309.
 * scores[q][d] = sum_k rows[d][k]*query[q][k], float32, accumulated as the k-ordered fused
 * chain acc = fma(row[k], query[k], acc) (bit-reproducible; v_mfma_f32_32x32x2_f32).
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_index hipts_index_t;
int hipts_index_create(int dim, int64_t capacity, int device, hipts_index_t** out);
int hipts_index_destroy(hipts_index_t* h);
int hipts_index_add(hipts_index_t* h, const float* rows, int64_t nrows, int rows_memspace);
int hipts_index_len(const hipts_index_t* h, int64_t* nrows);
int hipts_index_vector_by_id(const hipts_index_t* h, int64_t id, float* out_host);
/* device address of the row-major [len][dim] float32 matrix (for zero-copy producers) */
int hipts_index_data(const hipts_index_t* h, void** device_ptr);
/* rows [first, first + nrows) as one block into host memory (bulk form of vector_by_id: Similarity.save,
 * genmodel.py:175, gen_cfeatures.py:459, and the revision copy of gen_cfeatures.py:360-368) */
int hipts_index_export(const hipts_index_t* h, int64_t first, int64_t nrows, float* out_host);
int hipts_index_query(hipts_index_t* h, const float* queries, int queries_memspace, int nq,
                      float* scores_out, int out_memspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Score combination and ranking.   Replaces   webui.py:377-383  (normalise by max, weighted
 * sum) and the full Python sort webui.py:191-192 (score descending, ties by ascending id).
 * ---------------------------------------------------------------------------------------- */
/* out[q][d] = wa * A + (double)((float)wb * B) with A = a[q][d] / max(a[q]) if norm_a and that
 * max > 0 else a[q][d] (float64) and B likewise on the float32 b.  out may alias a. */
int hipts_combine(const double* a, const float* b, int nq, int64_t n, double wa, double wb,
                  int norm_a, int norm_b, double* out, int device, void* stream);
/* The same in two steps, for an index whose rows are sharded over ranks: every rank
 * takes the row maxima of its shard (device pointers in and out; a or b may be NULL), the ranks
 * all-reduce them with MAX, and combine_with_max normalises by the maxima it is GIVEN (NULL = do not
 * normalise that operand), with exactly the arithmetic of hipts_combine. */
int hipts_rowmax(const double* a, const float* b, int nq, int64_t n, double* max_a_out, float* max_b_out, int device,
                 void* stream);
int hipts_combine_with_max(const double* a, const float* b, int nq, int64_t n, double wa, double wb, const double* max_a,
                           const float* max_b, double* out, int device, void* stream);
/* per query the k best entries of vals[q][0..n): ids_out int32 [nq][k], vals_out float64
 * [nq][k], in rank order (value descending, ties by ascending index).  k <= 1024.
 * vals is device memory; outputs in out_memspace. */
int hipts_topk(const double* vals, int nq, int64_t n, int k, int32_t* ids_out, double* vals_out,
               int out_memspace, int device, void* stream);
/* The ranking continued past a prefix already in hand: the next k entries of one query's order (value descending, ties by ascending
 * index) AFTER the entry (after_val, after_id) -- i.e. among the scores with  v < after_val  or  v == after_val and index > after_id.
 * vals: device float64 [n]; outputs on the host, padded with (-1, -inf) when fewer than k entries remain.  webui.py:191-192 sorts all
 * documents and :63-80 then looks for its second cut point anywhere in that list; the device path ranks 1024 at a time and asks
 * for more only while the filter still needs them (hiptagsearch/search.py::_doc2vec_rerank). */
int hipts_topk_after(const double* vals, int64_t n, int k, double after_val, int64_t after_id, int32_t* ids_out, double* vals_out,
                     int device, void* stream);
/* the fused query of webui.py:352-383 for nq queries: BM25 + index product + normalise +
 * w_bm25/w_sim combine + top-k.  final_out (optional, device, float64 [nq][len]) receives the
 * combined scores for the rerank stage (webui.py:189-253). */
/* With nq == 1 -- the reference's only real usage (webui.py:586: one query, topn = 800) -- the call takes the one-query path:
 * the query travels in the kernel arguments, every step runs thread-per-document over the whole chip (BM25 in the reference's
 * own document-major form, the index product as the k-ordered fmaf chain = the same bits as the batched MFMA chain), and the
 * results are stored straight into pinned host memory.  Same results as the batched path, bit for bit. */
int hipts_search(hipts_bm25_t* bm25, hipts_index_t* index,
                 const int32_t* q_terms, const double* q_weights, const int32_t* q_ptr,
                 const float* q_vectors, int nq, double w_bm25, double w_sim, int k,
                 int32_t* ids_out, double* vals_out, double* final_out_device, void* stream);
/* The same call in two halves, for a host that serves a stream of query batches (the reference's webui.py:586 loop is one query at a
 * time; a batch is what a server front end accumulates): submit packs and launches a batch into slot 0 or 1 and returns without
 * waiting; collect waits for that slot's batch and unpacks it.  With two slots the host prepares batch i + 1 while the device runs
 * batch i -- same kernels, same results (hipts_search is submit + collect on slot 0).  All batches of a handle must use ONE stream. */
int hipts_search_submit(hipts_bm25_t* bm25, hipts_index_t* index, const int32_t* q_terms, const double* q_weights,
                        const int32_t* q_ptr, const float* q_vectors, int nq, double w_bm25, double w_sim, int k,
                        int slot, void* stream);
int hipts_search_collect(hipts_bm25_t* bm25, int slot, int32_t* ids_out, double* vals_out);

/* Per-kernel timing of the query path for roofline accounting (bench.py), as hipts_vit_profile_* above: while enabled, every
 * kernel hipts_search launches is bracketed by HIP events on the stream it is launched on.  read() resolves them (synchronises) and
 * returns, for one kernel category, the summed device time, the launches and the ALGORITHMIC bytes those launches stand for
 * (DESIGN.md section 4).  Categories 0-4: the batched path (bm25_postings, sim_mfma, rowmax, combine, topk); 5-8: the
 * one-query path (search1_score, search1_combine, search1_collect, topk on collected candidates). */
#define HIPTS_QUERY_PROF_CATEGORIES 9
int hipts_query_profile_enable(hipts_bm25_t* bm25, int enable);
int hipts_query_profile_read(hipts_bm25_t* bm25, int category, double* total_ms, int64_t* launches, double* total_bytes);
int hipts_query_profile_name(int category, char* buf, size_t n);

/* ------------------------------------------------------------------------------------------
 * Doc2Vec PV-DBOW inference.   Replaces gensim Doc2Vec.infer_vector      genmodel.py:169,
 * webui.py:106,185  (model built at genmodel.py:159: dm=0, vector_size=300, negative=5).
 * One wavefront per document; see DESIGN.md for the explicit (v0, seed) inputs.
 * ---------------------------------------------------------------------------------------- */
typedef struct hipts_d2v hipts_d2v_t;
int hipts_d2v_create(const float* syn1neg, const uint32_t* cum_table, const uint32_t* sample_int,
                     int64_t vocab, int dim, int negative, double exp_scale, int device, hipts_d2v_t** out);
int hipts_d2v_destroy(hipts_d2v_t* h);
/* documents in CSR form over vocabulary indices (-1 = out of vocabulary); v0 float32
 * [ndocs][dim] start vectors, seeds uint64 [ndocs]; out float32 [ndocs][dim]. */
int hipts_d2v_infer(hipts_d2v_t* h, const int64_t* doc_ptr, const int32_t* words, int64_t ndocs,
                    const float* v0, const uint64_t* seeds, int epochs, float alpha, float min_alpha,
                    float* out, int out_memspace, void* stream);

/* Doc2Vec PV-DM inference (dm=1, non-concatenative; dm_mean 0: sum, 1: mean of context word vectors + document vector).
 * BASELINE.json's north_star names this form; the reference's model is dm=0 (genmodel.py:159), so no call site of the reference
 * reaches it -- it replaces gensim Doc2Vec.infer_vector of a dm=1 model (doc2vec_inner.pyx::train_document_dm with
 * learn_doctags only).  word_vectors: float32 [vocab][dim] = the model's wv.vectors, frozen.  window: the model's window; the
 * reduced windows continue the document's explicit LCG stream (oracle/csrc/oracle.c::orc_d2v_infer_dm).  Documents of more
 * than 512 words are refused (HIPTS_ERR_INVALID). */
int hipts_d2v_set_word_vectors(hipts_d2v_t* h, const float* word_vectors);
int hipts_d2v_infer_dm(hipts_d2v_t* h, const int64_t* doc_ptr, const int32_t* words, int64_t ndocs,
                       const float* v0, const uint64_t* seeds, int epochs, float alpha, float min_alpha,
                       int window, int dm_mean, float* out, int out_memspace, void* stream);

/* Doc2Vec PV-DBOW TRAINING.   Replaces  Doc2Vec(vector_size=300, window=50, min_count=1, workers=1, dm=0) / build_vocab /
 * train(epochs=100)                                                                       genmodel.py:159-162.
 * The vocabulary statistics (build_vocab: cum_table with ns_exponent 0.75, sample_int for sample = 1e-3, both over the
 * vocabulary sorted by descending count) are prepared by the caller (hiptagsearch/d2v.py::Doc2Vec.build_vocab); documents
 * arrive in CSR form over vocabulary indices.  syn1neg float32 [vocab][dim] (gensim starts it at zero) and doc_vectors
 * float32 [ndocs][dim] (gensim: uniform in +-1/dim) are HOST arrays updated in place.  seed replaces the model's hidden
 * RandomState: the 48-bit LCG state of (epoch, document) is splitmix64(seed + epoch * ndocs + document).
 * mode 0: sequential -- every document of every epoch in corpus order on one wavefront (the reference's workers=1 semantics;
 *         reproducible, bit-identical to the CPU oracle); for parity and small corpora.
 * mode 1: parallel -- a wavefront per document, launches of 2048 documents (HIPTS_D2V_CHUNK): the documents of a launch train
 *         concurrently, their hidden-layer updates are float atomic adds (no update lost, order free); the throughput mode,
 *         not reproducible bit for bit. */
int hipts_d2v_train(const uint32_t* cum_table, const uint32_t* sample_int, int64_t vocab, int dim, int negative, double exp_scale,
                    const int64_t* doc_ptr, const int32_t* words, int64_t ndocs, float* doc_vectors, float* syn1neg, int epochs,
                    float alpha, float min_alpha, uint64_t seed, int batch_words, int mode, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIP_TAGSEARCH_H */
