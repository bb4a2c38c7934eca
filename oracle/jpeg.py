"""CPU oracle -- TEST INFRASTRUCTURE ONLY -- for the device half of the hybrid JPEG decode (csrc/jpeg.hip).

A numpy restatement of what libjpeg-turbo does between entropy decoding and the RGB rows PIL hands to the reference's
`Image.open(path)` (tagging.py:234-252, gen_cfeatures.py:285-295; PIL's JpegImagePlugin -> jpeg_read_scanlines with libjpeg's
defaults: JDCT_ISLOW, do_fancy_upsampling):

  idct_islow        jidctint.c  jpeg_idct_islow: dequantise, accurate integer 8 x 8 inverse DCT (CONST_BITS 13, PASS1_BITS 2), range limit
  upsample_h2v1/2   jdsample.c  h2v1_fancy_upsample / h2v2_fancy_upsample: the triangle filter with libjpeg's alternating rounding
  ycc_to_rgb        jdcolor.c   build_ycc_rgb_table + ycc_rgb_convert: 16-bit fixed point

PINNED: libjpeg-turbo is a third-party dependency of the reference (through Pillow) and is not under /root/reference, but it IS in this
image as Pillow's decoder, so the restatement is checked byte for byte against `PIL.Image.open` on generated files
(tests/test_oracle_jpeg.py) -- the very call the reference makes.

Input: a slot as written by the host half (csrc/jpeg_slot.h): header + int16 coefficient blocks."""
import numpy as np

HEADER_BYTES = 1024
MAGIC = 0x4745504A

_F = dict(f0_298631336=2446, f0_390180644=3196, f0_541196100=4433, f0_765366865=6270, f0_899976223=7373, f1_175875602=9633,
          f1_501321110=12299, f1_847759065=15137, f1_961570560=16069, f2_053119869=16819, f2_562915447=20995, f3_072711026=25172)


def parse_slot(slot: np.ndarray):
    """-> dict(width, height, ncomp, hmax, vmax, comps=[dict(h, v, blocks_w, blocks_h, dw, dh, coef [blocks,64] int16, quant [64])])"""
    b = np.ascontiguousarray(slot).view(np.uint8)
    hd = b[:32].view(np.int32)
    assert hd[0] == MAGIC and hd[1] == 1, "not a coefficient slot"
    width, height, ncomp, hmax, vmax = (int(v) for v in hd[2:7])
    comps = []
    quant = b[32 + 3 * 32:32 + 3 * 32 + 3 * 128].view(np.uint16).reshape(3, 64)
    coef_all = b[HEADER_BYTES:].view(np.int16)
    for c in range(ncomp):
        k = b[32 + 32 * c:64 + 32 * c].view(np.int32)
        h, v, bw, bh, dw, dh, off = (int(x) for x in k[:7])
        comps.append(dict(h=h, v=v, blocks_w=bw, blocks_h=bh, dw=dw, dh=dh, quant=quant[c].astype(np.int64),
                          coef=coef_all[off:off + bw * bh * 64].reshape(bw * bh, 64)))
    return dict(width=width, height=height, ncomp=ncomp, hmax=hmax, vmax=vmax, comps=comps)


def _idct_1d(d, shift):
    """jidctint.c, one pass over d[0..7] (int64 arrays); DESCALE by `shift`."""
    F = _F
    z2, z3 = d[2], d[6]
    z1 = (z2 + z3) * F["f0_541196100"]
    tmp2 = z1 + z3 * (-F["f1_847759065"])
    tmp3 = z1 + z2 * F["f0_765366865"]
    z2, z3 = d[0], d[4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = d[7], d[5], d[3], d[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1_175875602"]
    tmp0 = tmp0 * F["f0_298631336"]
    tmp1 = tmp1 * F["f2_053119869"]
    tmp2 = tmp2 * F["f3_072711026"]
    tmp3 = tmp3 * F["f1_501321110"]
    z1 = z1 * (-F["f0_899976223"])
    z2 = z2 * (-F["f2_562915447"])
    z3 = z3 * (-F["f1_961570560"]) + z5
    z4 = z4 * (-F["f0_390180644"]) + z5
    tmp0 = tmp0 + z1 + z3
    tmp1 = tmp1 + z2 + z4
    tmp2 = tmp2 + z2 + z3
    tmp3 = tmp3 + z1 + z4
    r = 1 << (shift - 1)
    return [(tmp10 + tmp3 + r) >> shift, (tmp11 + tmp2 + r) >> shift, (tmp12 + tmp1 + r) >> shift, (tmp13 + tmp0 + r) >> shift,
            (tmp13 - tmp0 + r) >> shift, (tmp12 - tmp1 + r) >> shift, (tmp11 - tmp2 + r) >> shift, (tmp10 - tmp3 + r) >> shift]


def _range_limit(x):
    """+ CENTERJSAMPLE, limited to 0..255: the saturating form of libjpeg-turbo's SIMD code (what Pillow runs); jidctint.c's
    sample_range_limit[x & RANGE_MASK] agrees for |x| < 512, and the host half (csrc/jpeg_host.c COLSUM_LIMIT) lets no block through whose
    values come near either limit or near the 16-bit lanes of the SIMD passes."""
    return np.clip(x + 128, 0, 255).astype(np.uint8)


def idct_islow(coef: np.ndarray, quant: np.ndarray) -> np.ndarray:
    """coef int16 [n,64] (natural order), quant [64] -> samples uint8 [n,8,8]."""
    x = coef.astype(np.int64).reshape(-1, 8, 8) * quant.reshape(1, 8, 8)
    ws = np.stack(_idct_1d([x[:, k, :] for k in range(8)], 11), axis=1)          # pass 1: columns -> ws[:, k, col]
    out = np.stack(_idct_1d([ws[:, :, k] for k in range(8)], 18), axis=2)        # pass 2: rows
    return _range_limit(out)


def plane_of(comp) -> np.ndarray:
    """the component's sample plane, padded to whole blocks: uint8 [blocks_h * 8, blocks_w * 8]"""
    s = idct_islow(comp["coef"], comp["quant"]).reshape(comp["blocks_h"], comp["blocks_w"], 8, 8)
    return s.transpose(0, 2, 1, 3).reshape(comp["blocks_h"] * 8, comp["blocks_w"] * 8)


def _h_fancy_v2(cs):
    """the horizontal part of h2v2_fancy_upsample on column sums cs int [rows, dw] -> [rows, 2 dw]"""
    dw = cs.shape[1]
    out = np.empty((cs.shape[0], 2 * dw), dtype=np.int64)
    out[:, 0] = (cs[:, 0] * 4 + 8) >> 4
    out[:, 2::2] = (cs[:, 1:] * 3 + cs[:, :-1] + 8) >> 4
    out[:, 1:-1:2] = (cs[:, :-1] * 3 + cs[:, 1:] + 7) >> 4
    out[:, -1] = (cs[:, -1] * 4 + 7) >> 4
    return out


def upsample_h2v2(c: np.ndarray) -> np.ndarray:
    """c uint8 [dh, dw] (real samples) -> uint8 [2 dh, 2 dw]; the rows above the first / below the last are those rows themselves
    (jdmainct.c's context rows at the image edges)."""
    c = c.astype(np.int64)
    up = np.concatenate([c[:1], c[:-1]])          # row r - 1
    dn = np.concatenate([c[1:], c[-1:]])          # row r + 1
    out = np.empty((2 * c.shape[0], 2 * c.shape[1]), dtype=np.int64)
    out[0::2] = _h_fancy_v2(3 * c + up)
    out[1::2] = _h_fancy_v2(3 * c + dn)
    return out.astype(np.uint8)


def upsample_h2v1(c: np.ndarray) -> np.ndarray:
    c = c.astype(np.int64)
    dw = c.shape[1]
    out = np.empty((c.shape[0], 2 * dw), dtype=np.int64)
    out[:, 0] = c[:, 0]
    out[:, 2::2] = (c[:, 1:] * 3 + c[:, :-1] + 1) >> 2
    out[:, 1:-1:2] = (c[:, :-1] * 3 + c[:, 1:] + 2) >> 2
    out[:, -1] = c[:, -1]
    return out.astype(np.uint8)


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    y, cb, cr = (a.astype(np.int64) for a in (y, cb, cr))
    r = y + ((91881 * (cr - 128) + 32768) >> 16)
    g = y + ((-22554 * (cb - 128) + 32768 - 46802 * (cr - 128)) >> 16)
    b = y + ((116130 * (cb - 128) + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode_slot(slot: np.ndarray) -> np.ndarray:
    """slot -> uint8 [height, width, 3] RGB, the bytes of PIL.Image.open(file).convert('RGB')."""
    j = parse_slot(slot)
    W, H = j["width"], j["height"]
    planes = [plane_of(c) for c in j["comps"]]
    y = planes[0][:H, :W]
    if j["ncomp"] == 1:
        return np.repeat(y[:, :, None], 3, axis=2)
    out = []
    for c, p in zip(j["comps"][1:], planes[1:]):
        real = p[:c["dh"], :c["dw"]]
        if j["hmax"] == 2 and j["vmax"] == 2:
            real = upsample_h2v2(real)
        elif j["hmax"] == 2:
            real = upsample_h2v1(real)
        out.append(real[:H, :W])
    return ycc_to_rgb(y, out[0], out[1])
