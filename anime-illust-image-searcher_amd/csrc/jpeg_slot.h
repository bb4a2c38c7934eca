// The hand-over between the host half and the device half of the hybrid JPEG decode (round 4; SURVEY.md §8 f4, "or GPU decode").
//
//   host   (jpeg_host.c -> libhipts_jpeg_host.so, plain C: loaded by the decode worker PROCESSES, which must not touch the GPU)
//          parses a baseline / extended-sequential / progressive Huffman JPEG and entropy-decodes it into quantised DCT coefficients -- the part of
//          libjpeg's decoder that is a serial bit stream;
//   device (jpeg.hip, in libhip_tagsearch.so) dequantises, runs libjpeg's accurate integer inverse DCT (jidctint.c, JDCT_ISLOW: Pillow's
//          default), the "fancy" chroma upsampling (jdsample.c) and the YCbCr -> RGB conversion (jdcolor.c), then the tagger's pad +
//          resize -- everything that is per block / per pixel.
//
// A slot is a ring-buffer slot of hiptagsearch/pipeline.py: this header, then int16 coefficient blocks.  What the reference does at this
// point is PIL's Image.open(...).load() (tagging.py:234-252, gen_cfeatures.py:285-295): the device half reproduces libjpeg-turbo's output
// byte for byte (tests/test_gpu_jpeg.py compares with Pillow).
#pragma once
#include <stdint.h>

#define HIPTS_JPEG_MAGIC 0x4745504a       /* "JPEG" */
#define HIPTS_JPEG_HEADER_BYTES 1024      /* coefficients start here */

typedef struct {
    int32_t h, v;                 /* sampling factors (1 or 2) */
    int32_t blocks_w, blocks_h;   /* 8 x 8 blocks per row / rows of blocks, padded to whole MCUs */
    int32_t dw, dh;               /* real (downsampled) width and height in samples */
    int32_t offset;               /* first coefficient of the component, in int16 units from the start of the coefficient area */
    int32_t pad;
} hipts_jpeg_component;

typedef struct {
    int32_t magic;                /* HIPTS_JPEG_MAGIC */
    int32_t kind;                 /* 1: coefficient blocks follow */
    int32_t width, height;        /* image size in pixels */
    int32_t ncomp;                /* 1 (grey) or 3 (YCbCr) */
    int32_t hmax, vmax;
    int32_t reserved;
    hipts_jpeg_component comp[3];
    uint16_t quant[3][64];        /* per component, natural (row-major) order */
    int64_t total_bytes;          /* header + coefficients */
} hipts_jpeg_header;
