/* Host half of the hybrid JPEG decode (see jpeg_slot.h): marker parsing and Huffman entropy decoding of a baseline, extended-sequential
 * or progressive 8-bit JPEG into quantised DCT coefficient blocks.  Plain C, no GPU runtime: built into libhipts_jpeg_host.so, which the
 * decode worker processes of hiptagsearch/pipeline.py load (a worker must never initialise the GPU).
 *
 * What it follows: ITU-T T.81 (markers B.2, Huffman procedures F.2.2 / Annex C, progressive mode Annex G) -- the same stream
 * libjpeg-turbo's jdmarker.c / jdhuff.c / jdphuff.c read when the reference calls PIL's Image.open (tagging.py:234-252).  Anything this
 * file does not handle -- arithmetic coding, progressions that leave a coefficient out or below full precision, 12-bit samples,
 * CMYK / RGB-coded files, sampling other than 4:4:4 / 4:2:2 / 4:2:0, sequential files in several scans, tiny images -- and any
 * irregularity in the stream is reported (status 1 or 3) and the caller decodes that file with Pillow as before: the fast path never
 * has to guess what libjpeg's error recovery would have produced.
 *
 * That includes streams that parse but carry coefficients no 8-bit image produces (flipped bits): libjpeg-turbo's SIMD inverse DCT works on
 * 16-bit lanes (wrapping products and sums, saturating packs) where the C code and the device kernel compute in 32 bits, so beyond the
 * range of real data the two differ.  Per block and coefficient column the sum of |coefficient x quantiser| is held to COLSUM_LIMIT:
 * below it no 16-bit lane of either pass can wrap or saturate (pass 1 scales a column by at most 4 sqrt 8 / 2 = 5.66, pass 2 adds two such
 * values), so 16-bit and 32-bit arithmetic agree; an 8 x 8 block of 8-bit samples cannot exceed sqrt 8 x 1024 = 2896 (Parseval).  A block
 * above the limit sends the file to Pillow. */
#include <emmintrin.h> /* SSE2: part of the x86-64 baseline */
#include <string.h>

#include "jpeg_slot.h"

enum { JH_OK = 0, JH_UNSUPPORTED = 1, JH_TOO_SMALL = 2, JH_CORRUPT = 3 };
enum { COLSUM_LIMIT = 2850 };

/* zigzag position -> natural index; positions 64..79 (a run that overshoots the block: an irregular stream, detected after the block)
 * land in a spare row of the local block buffer */
static const uint8_t ZIGZAG[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                   6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79};

#define LOOK 11 /* bits of lookahead of the decoding tables */
typedef struct {
    uint16_t lut[1 << LOOK];  /* prefix -> (length << 8) | symbol; 0: the code is longer than LOOK bits */
    int32_t fast[1 << LOOK];  /* AC tables: prefix -> (value << 8) | (run << 4) | (code length + magnitude bits) when code AND magnitude
                                 bits fit in the prefix (the common small coefficients: one lookup per coefficient); 0: decode the
                                 symbol, then its bits */
    int32_t maxcode[17];      /* largest code of length l, -1 if there is none */
    int32_t mincode[17];
    int32_t valptr[17];       /* index of the first symbol of length l */
    uint8_t vals[256];
    int present;
} Huff;

/* T.81 Annex C: canonical codes from the list of code lengths */
static int huff_build(Huff* h, const uint8_t* counts, const uint8_t* symbols, int nsym, int is_ac) {
    int code = 0, p = 0;
    memset(h->lut, 0, sizeof(h->lut));
    memset(h->fast, 0, sizeof(h->fast));
    memcpy(h->vals, symbols, (size_t)nsym);
    for (int l = 1; l <= 16; ++l) {
        const int n = counts[l - 1];
        h->valptr[l] = p;
        h->mincode[l] = code;
        if (n == 0) {
            h->maxcode[l] = -1;
        } else {
            if (code + n > (1 << l)) return JH_CORRUPT; /* more codes than the length can hold */
            h->maxcode[l] = code + n - 1;
            if (l <= LOOK)
                for (int i = 0; i < n; ++i) {
                    const int first = (code + i) << (LOOK - l), span = 1 << (LOOK - l);
                    for (int j = 0; j < span; ++j) h->lut[first + j] = (uint16_t)((l << 8) | symbols[p + i]);
                }
        }
        p += n;
        code = (code + n) << 1;
    }
    if (is_ac)
        for (int i = 0; i < (1 << LOOK); ++i) {
            const int e = h->lut[i];
            if (!e) continue;
            const int len = e >> 8, run = (e >> 4) & 15, mag = e & 15;
            if (mag == 0 || len + mag > LOOK) continue;
            int k = ((i << len) & ((1 << LOOK) - 1)) >> (LOOK - mag);
            if (k < (1 << (mag - 1))) k += (int)((~0u) << mag) + 1;
            h->fast[i] = k * 256 + run * 16 + (len + mag);
        }
    h->present = 1;
    return JH_OK;
}

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t bits; /* left-justified */
    int nbits;
    int fake;      /* bits appended after the data ran into a marker (or the end of the file) */
} BitReader;

/* at least 32 valid bits afterwards: a symbol (<= 16 bits) and its magnitude bits (<= 15) */
static inline void refill(BitReader* b) {
    if (b->nbits > 32) return;
    if (b->fake == 0 && b->p + 4 <= b->end) {
        uint32_t v;
        memcpy(&v, b->p, 4);
        v = __builtin_bswap32(v);
        const uint32_t x = ~v;
        if (!((x - 0x01010101u) & ~x & 0x80808080u)) { /* no 0xFF among the four bytes: nothing stuffed, no marker */
            b->bits |= (uint64_t)v << (32 - b->nbits);
            b->nbits += 32;
            b->p += 4;
            return;
        }
    }
    while (b->nbits <= 56) {
        uint32_t c = 0;
        if (b->fake == 0 && b->p < b->end) {
            c = *b->p;
            if (c == 0xFF) {
                if (b->p + 1 < b->end && b->p[1] == 0) {
                    b->p += 2; /* stuffed zero byte */
                } else {
                    c = 0; /* a marker: stay on it */
                    b->fake += 8;
                }
            } else {
                b->p++;
            }
        } else {
            b->fake += 8;
        }
        b->bits |= (uint64_t)c << (56 - b->nbits);
        b->nbits += 8;
    }
}

static inline int huff_decode(BitReader* b, const Huff* h) {
    const uint32_t e = h->lut[b->bits >> (64 - LOOK)];
    if (e) {
        const int l = (int)(e >> 8);
        b->bits <<= l;
        b->nbits -= l;
        return (int)(e & 255u);
    }
    const int32_t code16 = (int32_t)(b->bits >> 48);
    for (int l = LOOK + 1; l <= 16; ++l) {
        const int32_t code = code16 >> (16 - l);
        if (code <= h->maxcode[l]) {
            b->bits <<= l;
            b->nbits -= l;
            return h->vals[h->valptr[l] + code - h->mincode[l]];
        }
    }
    return -1;
}

/* T.81 F.2.2.1: s more bits, sign-extended */
static inline int receive_extend(BitReader* b, int s) {
    const int v = (int)(b->bits >> (64 - s));
    b->bits <<= s;
    b->nbits -= s;
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

static inline unsigned be16(const uint8_t* p) { return ((unsigned)p[0] << 8) | p[1]; }

/* n raw bits (1..16); the caller has refilled */
static inline int receive(BitReader* b, int n) {
    const int v = (int)(b->bits >> (64 - n));
    b->bits <<= n;
    b->nbits -= n;
    return v;
}

/* The bits of an interval are used up (but for the padding of its last byte) and marker 0xFF `want` follows (want < 0: any marker: the
 * reader is left ON it).  The reader stays on a marker's 0xFF once it has met it; before that it points at the next unread byte. */
static int at_marker(BitReader* br, int want) {
    if (br->fake > br->nbits) return JH_CORRUPT;          /* read past the end of the data */
    if ((br->nbits - br->fake) >= 8) return JH_CORRUPT;   /* whole unread data bytes in front of the marker */
    const uint8_t* q = br->p;
    if (q + 2 > br->end || q[0] != 0xFF) return JH_CORRUPT;
    while (q + 1 < br->end && q[1] == 0xFF) ++q;
    if (q + 2 > br->end) return JH_CORRUPT;
    if (want >= 0) {
        if (q[1] != want) return JH_CORRUPT;
        br->p = q + 2;
    } else {
        if (q[1] == 0) return JH_CORRUPT;
        br->p = q;
    }
    br->bits = 0;
    br->nbits = 0;
    br->fake = 0;
    return JH_OK;
}

/* ---- progressive mode (T.81 Annex G; the procedures of libjpeg's jdphuff.c): one scan = one band of coefficients (Ss..Se) at one
 * precision (successive approximation Ah -> Al) of one component, or the DC coefficients of several.  The scans accumulate into the same
 * coefficient blocks the baseline path fills; when every coefficient of every component has arrived at full precision the blocks are what
 * a baseline file of the same image would carry, and libjpeg applies no block smoothing (jdcoefct.c smoothing_ok). */
typedef struct {
    const hipts_jpeg_header* hd;
    int16_t* coef;
    int restart;
    int coef_bits[3][64]; /* -1: not seen yet; else the Al of the last scan that carried the coefficient */
} Prog;

static inline int getbit(BitReader* br) {
    refill(br);
    return receive(br, 1);
}

static int prog_scan(Prog* P, BitReader* br, int ns, const int* sc, const Huff* const* hdc, const Huff* const* hac, int Ss, int Se, int Ah, int Al) {
    const hipts_jpeg_header* hd = P->hd;
    /* jdphuff.c start_pass_phuff_decoder: what a legal progression may ask for */
    if (Ss == 0) {
        if (Se != 0) return JH_CORRUPT;
    } else {
        if (ns != 1 || Se < Ss || Se > 63) return JH_CORRUPT;
    }
    if (Al > 13 || (Ah != 0 && Ah != Al + 1)) return JH_CORRUPT;
    for (int i = 0; i < ns; ++i) {
        int* cb = P->coef_bits[sc[i]];
        if (Ss > 0 && cb[0] < 0) return JH_CORRUPT; /* AC before DC */
        for (int k = Ss; k <= Se; ++k) {
            const int expected = cb[k] < 0 ? 0 : cb[k];
            if (Ah != expected) return JH_CORRUPT; /* libjpeg warns and decodes on; here: Pillow's business */
            cb[k] = Al;
        }
        if (Ss == 0 ? !hdc[i] && Ah == 0 : !hac[i]) return JH_CORRUPT;
    }
    const int hmax = hd->hmax, vmax = hd->vmax;
    int nx, ny; /* MCUs of the scan */
    if (ns == 1) {
        nx = (hd->comp[sc[0]].dw + 7) / 8;
        ny = (hd->comp[sc[0]].dh + 7) / 8;
    } else {
        nx = (hd->width + 8 * hmax - 1) / (8 * hmax);
        ny = (hd->height + 8 * vmax - 1) / (8 * vmax);
    }
    int pred[3] = {0, 0, 0}, eobrun = 0, until_restart = P->restart, next_rst = 0;
    const int p1 = 1 << Al, m1 = -(1 << Al);
    for (int my = 0; my < ny; ++my)
        for (int mx = 0; mx < nx; ++mx) {
            if (P->restart && until_restart == 0) {
                if (at_marker(br, 0xD0 + next_rst) != JH_OK) return JH_CORRUPT;
                next_rst = (next_rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                eobrun = 0;
                until_restart = P->restart;
            }
            for (int i = 0; i < ns; ++i) {
                const hipts_jpeg_component* k = &hd->comp[sc[i]];
                const int bh = ns == 1 ? 1 : k->h, bv = ns == 1 ? 1 : k->v;
                for (int v = 0; v < bv; ++v)
                    for (int h = 0; h < bh; ++h) {
                        int16_t* blk = P->coef + k->offset + ((int64_t)(my * bv + v) * k->blocks_w + (mx * bh + h)) * 64;
                        if (Ss == 0) {
                            if (Ah == 0) { /* DC, first pass */
                                refill(br);
                                const int s = huff_decode(br, hdc[i]);
                                if (s < 0 || s > 15) return JH_CORRUPT;
                                if (s) pred[i] += receive_extend(br, s);
                                const int val = pred[i] * p1;
                                if (val < -32768 || val > 32767) return JH_CORRUPT;
                                blk[0] = (int16_t)val;
                            } else if (getbit(br)) { /* DC, one more bit */
                                blk[0] = (int16_t)(blk[0] | p1);
                            }
                        } else if (Ah == 0) { /* AC band, first pass */
                            if (eobrun > 0) {
                                --eobrun;
                                continue;
                            }
                            for (int kk = Ss; kk <= Se; ++kk) {
                                refill(br);
                                const int f = hac[i]->fast[br->bits >> (64 - LOOK)];
                                if (f) { /* run, code and magnitude bits in one lookup (as in the sequential path) */
                                    kk += (f >> 4) & 15;
                                    if (kk > Se) return JH_CORRUPT;
                                    br->bits <<= (f & 15);
                                    br->nbits -= (f & 15);
                                    const int val = (f >> 8) * p1;
                                    if (val < -32768 || val > 32767) return JH_CORRUPT;
                                    blk[ZIGZAG[kk]] = (int16_t)val;
                                    continue;
                                }
                                const int rs = huff_decode(br, hac[i]);
                                if (rs < 0) return JH_CORRUPT;
                                const int r = rs >> 4, s = rs & 15;
                                if (s) {
                                    kk += r;
                                    if (kk > Se) return JH_CORRUPT;
                                    const int val = receive_extend(br, s) * p1;
                                    if (val < -32768 || val > 32767) return JH_CORRUPT;
                                    blk[ZIGZAG[kk]] = (int16_t)val;
                                } else if (r == 15) {
                                    kk += 15;
                                } else {
                                    eobrun = 1 << r;
                                    if (r) eobrun += receive(br, r);
                                    --eobrun; /* this block is the first of the run */
                                    break;
                                }
                            }
                        } else { /* AC band, refinement (jdphuff.c decode_mcu_AC_refine) */
                            int kk = Ss;
                            if (eobrun == 0) {
                                for (; kk <= Se; ++kk) {
                                    refill(br);
                                    const int rs = huff_decode(br, hac[i]);
                                    if (rs < 0) return JH_CORRUPT;
                                    int r = rs >> 4, s = rs & 15;
                                    if (s) {
                                        if (s != 1) return JH_CORRUPT;
                                        s = getbit(br) ? p1 : m1;
                                    } else if (r != 15) {
                                        eobrun = 1 << r;
                                        if (r) {
                                            refill(br);
                                            eobrun += receive(br, r);
                                        }
                                        break; /* the rest of the band is handled below */
                                    }
                                    /* past r still-zero coefficients, correcting the non-zero ones on the way */
                                    do {
                                        int16_t* c = blk + ZIGZAG[kk];
                                        if (*c != 0) {
                                            if (getbit(br) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                                        } else if (--r < 0) {
                                            break;
                                        }
                                        ++kk;
                                    } while (kk <= Se);
                                    if (s) {
                                        if (kk > Se) return JH_CORRUPT;
                                        blk[ZIGZAG[kk]] = (int16_t)s;
                                    }
                                }
                            }
                            if (eobrun > 0) {
                                for (; kk <= Se; ++kk) {
                                    int16_t* c = blk + ZIGZAG[kk];
                                    if (*c != 0 && getbit(br) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                                }
                                --eobrun;
                            }
                        }
                    }
            }
            if (br->fake > br->nbits) return JH_CORRUPT;
            if (P->restart) --until_restart;
        }
    return at_marker(br, -1); /* the next marker follows the scan's last byte */
}

/* every block's per-column sum of |coefficient x quantiser| against COLSUM_LIMIT (the baseline path checks while it decodes) */
static int prog_check_range(const hipts_jpeg_header* hd, const int16_t* coef) {
    for (int c = 0; c < hd->ncomp; ++c) {
        const hipts_jpeg_component* k = &hd->comp[c];
        const uint16_t* q = hd->quant[c];
        const int16_t* b = coef + k->offset;
        for (int64_t i = 0, nb = (int64_t)k->blocks_w * k->blocks_h; i < nb; ++i, b += 64) {
            int cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < 8; ++r)
                for (int x = 0; x < 8; ++x) {
                    const int v = b[r * 8 + x];
                    cs[x] += (v < 0 ? -v : v) * q[r * 8 + x];
                }
            for (int x = 0; x < 8; ++x)
                if (cs[x] > COLSUM_LIMIT) return JH_CORRUPT;
        }
    }
    return JH_OK;
}

int hipts_jpeg_entropy_decode(const uint8_t* data, int64_t n, void* slot, int64_t slot_bytes) {
    if (!data || !slot || n < 4 || slot_bytes < HIPTS_JPEG_HEADER_BYTES) return JH_TOO_SMALL;
    if (data[0] != 0xFF || data[1] != 0xD8) return JH_UNSUPPORTED;
    uint16_t qt[4][64];
    memset(qt, 0, sizeof(qt));
    int qt_present[4] = {0, 0, 0, 0};
    static _Thread_local Huff dc[4], ac[4];
    for (int i = 0; i < 4; ++i) dc[i].present = ac[i].present = 0;
    int width = 0, height = 0, ncomp = 0, have_sof = 0, restart = 0, saw_jfif = 0, saw_adobe = 0, adobe_transform = 0;
    int progressive = 0, prog_started = 0;
    static _Thread_local Prog P;
    int cid[3] = {0, 0, 0}, ch[3] = {1, 1, 1}, cv[3] = {1, 1, 1}, ctq[3] = {0, 0, 0};
    int64_t pos = 2;
    for (;;) {
        if (pos + 2 > n) return JH_CORRUPT;
        if (data[pos] != 0xFF) return JH_CORRUPT;
        while (pos < n && data[pos] == 0xFF) ++pos; /* fill bytes */
        if (pos >= n) return JH_CORRUPT;
        const int m = data[pos++];
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9 && progressive && prog_started) {
            /* end of a progressive file: complete only if every coefficient of every component arrived at full precision -- otherwise
             * libjpeg smooths the blocks with what it has (jdcoefct.c decompress_smooth_data): Pillow's business */
            for (int c = 0; c < ncomp; ++c)
                for (int k = 0; k < 64; ++k)
                    if (P.coef_bits[c][k] != 0) return JH_UNSUPPORTED;
            return prog_check_range(P.hd, P.coef);
        }
        if (m == 0xD8 || m == 0xD9 || m == 0x00) return JH_CORRUPT;
        if (pos + 2 > n) return JH_CORRUPT;
        const int len = (int)be16(data + pos);
        if (len < 2 || pos + len > n) return JH_CORRUPT;
        const uint8_t* seg = data + pos + 2;
        int sl = len - 2;
        if (m == 0xDB) {
            /* libjpeg latches a component's table at the first scan that contains it (jdinput.c latch_quant_tables); the slot header takes
             * every component's table at the FIRST scan, so a table (re)defined between the scans of a progressive file is Pillow's business */
            if (prog_started) return JH_UNSUPPORTED;
            while (sl > 0) {
                const int pq = seg[0] >> 4, tq = seg[0] & 15;
                if (tq > 3 || pq > 1 || sl < 1 + 64 * (pq + 1)) return JH_CORRUPT;
                for (int i = 0; i < 64; ++i) {
                    const unsigned q = pq ? be16(seg + 1 + 2 * i) : seg[1 + i];
                    /* 8-bit frames only (SOF below): an entry above 255 is legal with pq = 1 but beyond any real 8-bit file, and it would take
                     * the |coefficient x quantiser| column sums out of int (COLSUM_LIMIT guard) */
                    if (q > 255) return JH_UNSUPPORTED;
                    qt[tq][ZIGZAG[i]] = (uint16_t)q;
                }
                qt_present[tq] = 1;
                seg += 1 + 64 * (pq + 1);
                sl -= 1 + 64 * (pq + 1);
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (have_sof || sl < 6) return JH_CORRUPT;
            progressive = m == 0xC2;
            if (seg[0] != 8) return JH_UNSUPPORTED;
            height = (int)be16(seg + 1);
            width = (int)be16(seg + 3);
            ncomp = seg[5];
            if (ncomp != 1 && ncomp != 3) return JH_UNSUPPORTED;
            if (sl < 6 + 3 * ncomp) return JH_CORRUPT;
            for (int c = 0; c < ncomp; ++c) {
                cid[c] = seg[6 + 3 * c];
                ch[c] = seg[7 + 3 * c] >> 4;
                cv[c] = seg[7 + 3 * c] & 15;
                ctq[c] = seg[8 + 3 * c];
                if (ctq[c] > 3 || ch[c] < 1 || cv[c] < 1) return JH_CORRUPT;
            }
            have_sof = 1;
        } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JH_UNSUPPORTED; /* lossless, arithmetic, hierarchical */
        } else if (m == 0xCC) {
            return JH_UNSUPPORTED;
        } else if (m == 0xC4) {
            while (sl > 0) {
                if (sl < 17) return JH_CORRUPT;
                const int tc = seg[0] >> 4, th = seg[0] & 15;
                int nsym = 0;
                for (int i = 0; i < 16; ++i) nsym += seg[1 + i];
                if (tc > 1 || th > 3 || nsym > 256 || sl < 17 + nsym) return JH_CORRUPT;
                if (huff_build(tc ? &ac[th] : &dc[th], seg + 1, seg + 17, nsym, tc) != JH_OK) return JH_CORRUPT;
                seg += 17 + nsym;
                sl -= 17 + nsym;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return JH_CORRUPT;
            restart = (int)be16(seg);
        } else if (m == 0xE0) {
            if (sl >= 5 && memcmp(seg, "JFIF\0", 5) == 0) saw_jfif = 1;
        } else if (m == 0xEE) {
            if (sl >= 12 && memcmp(seg, "Adobe", 5) == 0) {
                saw_adobe = 1;
                adobe_transform = seg[11];
            }
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) return JH_CORRUPT;
            const int ns = seg[0];
            if (ns < 1 || ns > ncomp) return progressive ? JH_CORRUPT : JH_UNSUPPORTED;
            if (!progressive && ns != ncomp) return JH_UNSUPPORTED; /* a sequential file in several scans */
            if (sl < 1 + 2 * ns + 3) return JH_CORRUPT;
            int tdc[3], tac[3], sc[3];
            for (int c = 0; c < ns; ++c) {
                sc[c] = -1;
                for (int j = 0; j < ncomp; ++j)
                    if (seg[1 + 2 * c] == cid[j]) sc[c] = j;
                if (sc[c] < 0 || (c > 0 && sc[c] <= sc[c - 1])) return JH_CORRUPT; /* T.81 B.2.3: in frame order */
                if (!progressive && sc[c] != c) return JH_UNSUPPORTED;
                tdc[c] = seg[2 + 2 * c] >> 4;
                tac[c] = seg[2 + 2 * c] & 15;
                if (tdc[c] > 3 || tac[c] > 3) return JH_CORRUPT;
                if (!progressive && (!dc[tdc[c]].present || !ac[tac[c]].present)) return JH_CORRUPT;
                if (!qt_present[ctq[sc[c]]]) return JH_CORRUPT;
            }
            const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
            if (!progressive && (Ss != 0 || Se != 63 || Ah != 0 || Al != 0)) return JH_UNSUPPORTED;
            pos += len;
            if (progressive && prog_started) {
                /* a further scan of a progressive file */
                const Huff* hdcp[3];
                const Huff* hacp[3];
                for (int c = 0; c < ns; ++c) {
                    hdcp[c] = dc[tdc[c]].present ? &dc[tdc[c]] : 0;
                    hacp[c] = ac[tac[c]].present ? &ac[tac[c]] : 0;
                }
                P.restart = restart;
                BitReader pbr = {data + pos, data + n, 0, 0, 0};
                const int st = prog_scan(&P, &pbr, ns, sc, hdcp, hacp, Ss, Se, Ah, Al);
                if (st != JH_OK) return st;
                pos = pbr.p - data;
                continue;
            }
            /* the header below copies EVERY frame component's table now: all of them must have arrived (a first scan with Y only and the
             * chroma DQT after it is legal T.81 -- and decoded by libjpeg, i.e. Pillow) */
            for (int c = 0; c < ncomp; ++c)
                if (!qt_present[ctq[c]]) return JH_UNSUPPORTED;
            /* ---- colour model and sampling, by libjpeg's rules (jdapimin.c default_decompress_parms) */
            if (width < 16 || height < 16) return JH_UNSUPPORTED; /* (jdsample.c leaves the fancy upsampling below 3 chroma columns) */
            if (ncomp == 3) {
                int ycc = 1;
                if (saw_jfif) ycc = 1;
                else if (saw_adobe) ycc = adobe_transform != 0;
                else if (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B') ycc = 0;
                if (!ycc) return JH_UNSUPPORTED;
                if (ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1) return JH_UNSUPPORTED;
                if (!((ch[0] == 1 && cv[0] == 1) || (ch[0] == 2 && cv[0] == 1) || (ch[0] == 2 && cv[0] == 2))) return JH_UNSUPPORTED;
            } else {
                ch[0] = cv[0] = 1; /* a single-component scan is not interleaved: one block per MCU whatever the factors say */
            }
            const int hmax = ch[0], vmax = cv[0];
            const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
            hipts_jpeg_header* hd = (hipts_jpeg_header*)slot;
            memset(hd, 0, HIPTS_JPEG_HEADER_BYTES);
            hd->magic = HIPTS_JPEG_MAGIC;
            hd->kind = 1;
            hd->width = width;
            hd->height = height;
            hd->ncomp = ncomp;
            hd->hmax = hmax;
            hd->vmax = vmax;
            int64_t off = 0;
            for (int c = 0; c < ncomp; ++c) {
                hipts_jpeg_component* k = &hd->comp[c];
                k->h = ch[c];
                k->v = cv[c];
                k->blocks_w = mcux * ch[c];
                k->blocks_h = mcuy * cv[c];
                k->dw = (width * ch[c] + hmax - 1) / hmax;
                k->dh = (height * cv[c] + vmax - 1) / vmax;
                k->offset = (int32_t)off;
                off += (int64_t)k->blocks_w * k->blocks_h * 64;
                memcpy(hd->quant[c], qt[ctq[c]], sizeof(qt[0]));
            }
            if (off > 0x7fffffff / 2) return JH_UNSUPPORTED;
            hd->total_bytes = HIPTS_JPEG_HEADER_BYTES + off * 2;
            if (hd->total_bytes > slot_bytes) return JH_TOO_SMALL;
            int16_t* coef = (int16_t*)((char*)slot + HIPTS_JPEG_HEADER_BYTES);
            if (progressive) {
                /* the first scan of a progressive file: the blocks accumulate over the scans, so they start from zero */
                memset(coef, 0, (size_t)off * 2);
                P.hd = hd;
                P.coef = coef;
                P.restart = restart;
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 64; ++k) P.coef_bits[c][k] = -1;
                prog_started = 1;
                const Huff* hdcp[3];
                const Huff* hacp[3];
                for (int c = 0; c < ns; ++c) {
                    hdcp[c] = dc[tdc[c]].present ? &dc[tdc[c]] : 0;
                    hacp[c] = ac[tac[c]].present ? &ac[tac[c]] : 0;
                }
                BitReader pbr = {data + pos, data + n, 0, 0, 0};
                const int st = prog_scan(&P, &pbr, ns, sc, hdcp, hacp, Ss, Se, Ah, Al);
                if (st != JH_OK) return st;
                pos = pbr.p - data;
                continue;
            }
            /* every block is decoded into a local buffer and leaves as eight non-temporal 16-byte stores: the slot (megabytes of a
             * shared-memory ring, not in any cache) is neither cleared first nor read for ownership */
            _Alignas(16) int16_t local[80];
            /* ---- the entropy-coded segment (T.81 F.2.2): MCU by MCU, restart markers every `restart` MCUs */
            BitReader br = {data + pos, data + n, 0, 0, 0};
            int pred[3] = {0, 0, 0};
            int until_restart = restart, next_rst = 0;
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    if (restart && until_restart == 0) {
                        if (at_marker(&br, 0xD0 + next_rst) != JH_OK) return JH_CORRUPT;
                        next_rst = (next_rst + 1) & 7;
                        pred[0] = pred[1] = pred[2] = 0;
                        until_restart = restart;
                    }
                    for (int c = 0; c < ncomp; ++c) {
                        const hipts_jpeg_component* k = &hd->comp[c];
                        const Huff* hdc = &dc[tdc[c]];
                        const Huff* hac = &ac[tac[c]];
                        for (int v = 0; v < k->v; ++v)
                            for (int h = 0; h < k->h; ++h) {
                                int16_t* out = coef + k->offset + ((int64_t)(my * k->v + v) * k->blocks_w + (mx * k->h + h)) * 64;
                                int16_t* blk = local;
                                const uint16_t* qn = hd->quant[c];
                                int colsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                                for (int z = 0; z < 8; ++z) _mm_store_si128((__m128i*)local + z, _mm_setzero_si128());
                                refill(&br);
                                int s = huff_decode(&br, hdc);
                                if (s < 0 || s > 15) return JH_CORRUPT;
                                if (s) pred[c] += receive_extend(&br, s);
                                if (pred[c] < -32768 || pred[c] > 32767) return JH_CORRUPT;
                                blk[0] = (int16_t)pred[c];
                                colsum[0] = (pred[c] < 0 ? -pred[c] : pred[c]) * qn[0];
                                int kk = 1;
                                while (kk < 64) {
                                    refill(&br);
                                    const int f = hac->fast[br.bits >> (64 - LOOK)];
                                    if (f) { /* run, code and magnitude bits in one lookup */
                                        kk += (f >> 4) & 15;
                                        br.bits <<= (f & 15);
                                        br.nbits -= (f & 15);
                                        const int nat = ZIGZAG[kk++], v = f >> 8;
                                        blk[nat] = (int16_t)v;
                                        colsum[nat & 7] += (v < 0 ? -v : v) * qn[nat & 63];
                                        continue;
                                    }
                                    const int rs = huff_decode(&br, hac);
                                    if (rs < 0) return JH_CORRUPT;
                                    const int r = rs >> 4;
                                    s = rs & 15;
                                    if (s == 0) {
                                        if (r != 15) break; /* end of block */
                                        kk += 16;
                                        continue;
                                    }
                                    kk += r;
                                    const int nat = ZIGZAG[kk], v = receive_extend(&br, s);
                                    blk[nat] = (int16_t)v;
                                    colsum[nat & 7] += (v < 0 ? -v : v) * qn[nat & 63];
                                    ++kk;
                                }
                                /* a run that overshoots position 63 (kk <= 79: the spare row took any store).  libjpeg reads on
                                 * silently; here the file goes to Pillow, i.e. to libjpeg itself */
                                if (kk > 64) return JH_CORRUPT;
                                {
                                    int m01 = colsum[0] > colsum[1] ? colsum[0] : colsum[1], m23 = colsum[2] > colsum[3] ? colsum[2] : colsum[3];
                                    int m45 = colsum[4] > colsum[5] ? colsum[4] : colsum[5], m67 = colsum[6] > colsum[7] ? colsum[6] : colsum[7];
                                    m01 = m01 > m23 ? m01 : m23;
                                    m45 = m45 > m67 ? m45 : m67;
                                    if ((m01 > m45 ? m01 : m45) > COLSUM_LIMIT) return JH_CORRUPT; /* beyond any 8-bit image: Pillow's business */
                                }
                                for (int z = 0; z < 8; ++z) _mm_stream_si128((__m128i*)out + z, _mm_load_si128((const __m128i*)local + z));
                            }
                    }
                    if (br.fake > br.nbits) return JH_CORRUPT; /* read past the end of the segment */
                    if (restart) --until_restart;
                }
            _mm_sfence();
            return JH_OK;
        }
        pos += len;
    }
}

/* Size of the slot a width x height image needs at most (4:4:4: three full planes of int16, padded to whole MCUs of 16 x 16). */
int64_t hipts_jpeg_slot_bytes(int width, int height) {
    const int64_t bw = (width + 15) / 16 * 2, bh = (height + 15) / 16 * 2;
    return HIPTS_JPEG_HEADER_BYTES + 3 * bw * bh * 128;
}
