"""CPU: the host half of the hybrid JPEG decode (csrc/jpeg_host.c -> libhipts_jpeg_host.so: markers + Huffman entropy decoding) followed by
the oracle's restatement of libjpeg's inverse DCT / fancy upsampling / colour conversion (oracle/jpeg.py) gives the bytes of
PIL.Image.open(file).convert('RGB') -- the decode the reference performs (tagging.py:234-252) -- on files Pillow writes here: three chroma
samplings, odd sizes, qualities 30..100 (quality 100: all-ones quantisation tables, long codes), optimised Huffman tables, restart
markers, greyscale, progressive mode.  Files the fast path does not take (incomplete progressions, CMYK, tiny, truncated) are refused with a status, never decoded
wrongly.  This pins oracle/jpeg.py (the checker of the GPU kernels, tests/test_gpu_jpeg.py) against libjpeg-turbo itself."""
import ctypes
import io
import os

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "anime-illust-image-searcher_amd", "libhipts_jpeg_host.so")


def host_lib():
    lib = ctypes.CDLL(HOST_LIB)
    lib.hipts_jpeg_entropy_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.hipts_jpeg_slot_bytes.restype = ctypes.c_int64
    return lib


def synth_image(rng, h, w, grey=False, noise=12):
    small = rng.integers(0, 256, (max(2, h // 32), max(2, w // 32), 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(small).resize((w, h), Image.BICUBIC), dtype=np.int16) + rng.integers(-noise, noise + 1, (h, w, 3), dtype=np.int16)
    im = Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))
    return im.convert("L") if grey else im


def jpeg_bytes(im, **kw):
    buf = io.BytesIO()
    im.save(buf, "JPEG", **kw)
    return buf.getvalue()


def entropy_decode(lib, data, slot_bytes=None):
    nb = slot_bytes or lib.hipts_jpeg_slot_bytes(*Image.open(io.BytesIO(data)).size)
    slot = np.zeros(nb, dtype=np.uint8)
    arr = np.frombuffer(data, dtype=np.uint8)
    return lib.hipts_jpeg_entropy_decode(arr.ctypes.data, len(data), slot.ctypes.data, nb), slot


def cases():
    out = []
    for (h, w) in [(768, 1024), (101, 77), (16, 16), (33, 250), (480, 641)]:
        for sub in (0, 1, 2):
            for q in (30, 90, 100):
                out.append(((h, w), False, dict(quality=q, subsampling=sub)))
    out.append(((300, 200), True, dict(quality=85)))
    out.append(((301, 203), False, dict(quality=85, subsampling=2, optimize=True)))
    out.append(((301, 203), False, dict(quality=85, subsampling=2, restart_marker_blocks=7)))
    out.append(((301, 203), False, dict(quality=75, subsampling=1, restart_marker_rows=1)))
    out.append(((64, 48), False, dict(quality=95, subsampling=0, optimize=True, restart_marker_blocks=1)))
    # progressive mode (Pillow's libjpeg writes the standard ten-scan script: spectral selection and successive approximation)
    for (h, w) in [(120, 160), (101, 77), (33, 250), (480, 641)]:
        for sub in (0, 1, 2):
            out.append(((h, w), False, dict(quality=85, subsampling=sub, progressive=True)))
    out.append(((300, 200), True, dict(quality=85, progressive=True)))
    out.append(((96, 128), False, dict(quality=30, subsampling=2, progressive=True, optimize=True)))
    out.append(((96, 128), False, dict(quality=100, subsampling=1, progressive=True, restart_marker_blocks=5)))
    return out


def test_entropy_decode_then_oracle_equals_pillow():
    from oracle import jpeg as oj
    lib = host_lib()
    rng = np.random.default_rng(0)
    for (h, w), grey, kw in cases():
        data = jpeg_bytes(synth_image(rng, h, w, grey), **kw)
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        st, slot = entropy_decode(lib, data)
        assert st == 0, ((h, w), grey, kw, st)
        got = oj.decode_slot(slot)
        assert got.shape == want.shape
        assert np.array_equal(got, want), ((h, w), grey, kw, int(np.abs(got.astype(int) - want).max()))


def test_flat_and_saturated_content():
    """all-zero AC blocks (both zero-coefficient short cuts of jidctint.c are value-identical to the full passes), saturated colours
    (the range limit), one-colour chroma"""
    from oracle import jpeg as oj
    lib = host_lib()
    imgs = [Image.new("RGB", (64, 40), (255, 255, 255)), Image.new("RGB", (64, 40), (0, 0, 0)), Image.new("RGB", (50, 70), (255, 0, 0)),
            Image.new("RGB", (50, 70), (0, 0, 255))]
    a = np.zeros((96, 96, 3), np.uint8)
    a[::2, ::2] = 255
    a[1::2, 1::2, 1] = 255
    imgs.append(Image.fromarray(a))
    for im in imgs:
        for sub in (0, 1, 2):
            data = jpeg_bytes(im, quality=97, subsampling=sub)
            st, slot = entropy_decode(lib, data)
            assert st == 0
            assert np.array_equal(oj.decode_slot(slot), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))


def test_files_outside_the_fast_path_are_refused():
    lib = host_lib()
    rng = np.random.default_rng(1)
    im = synth_image(rng, 120, 160)
    prog = jpeg_bytes(im, quality=80, progressive=True)
    last = prog.rindex(b"\xff\xda")
    assert entropy_decode(lib, prog)[0] == 0
    # a progression that stops short (the last refinement scan cut away): libjpeg would smooth the blocks -- refused
    assert entropy_decode(lib, prog[:last] + b"\xff\xd9", slot_bytes=lib.hipts_jpeg_slot_bytes(160, 120))[0] == 1
    assert entropy_decode(lib, jpeg_bytes(im.convert("CMYK"), quality=80))[0] == 1                  # four components
    assert entropy_decode(lib, jpeg_bytes(synth_image(rng, 12, 12), quality=80))[0] == 1            # below 16 x 16
    buf = io.BytesIO()
    im.save(buf, "PNG")
    arr = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    slot = np.zeros(1 << 20, dtype=np.uint8)
    assert lib.hipts_jpeg_entropy_decode(arr.ctypes.data, len(arr), slot.ctypes.data, len(slot)) == 1    # not a JPEG
    data = jpeg_bytes(im, quality=80)
    assert entropy_decode(lib, data, slot_bytes=4096)[0] == 2                                        # slot too small
    for cut in (len(data) // 2, len(data) - 40, 300):
        assert entropy_decode(lib, data[:cut], slot_bytes=lib.hipts_jpeg_slot_bytes(160, 120))[0] == 3    # truncated: Pillow's business
    broken = bytearray(jpeg_bytes(im, quality=80, restart_marker_blocks=5))
    i = broken.index(b"\xff\xd3")
    broken[i + 1] = 0xD5                                                                            # restart markers out of sequence
    assert entropy_decode(lib, bytes(broken), slot_bytes=lib.hipts_jpeg_slot_bytes(160, 120))[0] == 3


def test_mutated_files_never_crash_the_entropy_decoder():
    """Byte flips, truncations and spliced garbage: the host half returns one of its four statuses and writes inside its slot (a guard
    band behind the slot stays untouched).  Run in a child process so that a crash would be a test failure, not the end of the run."""
    import subprocess
    import sys
    code = r'''
import ctypes, io, sys
import numpy as np
sys.path.insert(0, %r)
from test_oracle_jpeg import host_lib, synth_image, jpeg_bytes
lib = host_lib()
rng = np.random.default_rng(11)
seeds = [jpeg_bytes(synth_image(rng, 120, 152), quality=q, subsampling=s, **kw) for q, s, kw in
         [(85, 2, {}), (60, 1, {}), (95, 0, {"optimize": True}), (80, 2, {"restart_marker_blocks": 3}), (85, 2, {"progressive": True}),
          (70, 0, {"progressive": True, "restart_marker_blocks": 4})]]
nb = int(lib.hipts_jpeg_slot_bytes(152, 120))
guard = 4096
buf = np.zeros(nb + guard, dtype=np.uint8)
counts = {0: 0, 1: 0, 2: 0, 3: 0}
for it in range(3000):
    d = bytearray(seeds[it %% len(seeds)])
    kind = it %% 5
    if kind == 0:
        for _ in range(int(rng.integers(1, 6))):
            d[int(rng.integers(2, len(d)))] = int(rng.integers(0, 256))
    elif kind == 1:
        d = d[:int(rng.integers(4, len(d)))]
    elif kind == 2:
        i = int(rng.integers(2, len(d)))
        d[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
    elif kind == 3:
        i = int(rng.integers(2, len(d) - 8))
        del d[i:i + int(rng.integers(1, 8))]
    else:
        i = int(rng.integers(2, len(d) - 2))
        d[i] = 0xFF
        d[i + 1] = int(rng.integers(0xC0, 0xFF))
    buf[nb:] = 0xA5
    src = np.frombuffer(bytes(d), dtype=np.uint8)
    st = lib.hipts_jpeg_entropy_decode(src.ctypes.data, len(src), buf.ctypes.data, nb)
    assert st in counts, st
    counts[st] += 1
    assert (buf[nb:] == 0xA5).all(), "wrote behind the slot"
print(counts)
assert counts[0] > 0 and counts[3] > 0
''' % os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])


def test_every_accepted_mutated_stream_still_equals_pillow():
    """Flipped bytes inside the entropy-coded segment often leave a stream that parses: what the host half ACCEPTS must still decode to
    Pillow's bytes.  libjpeg-turbo's SIMD inverse DCT (16-bit lanes) and 32-bit arithmetic part ways on coefficients no real image has;
    the host half refuses such blocks (COLSUM_LIMIT), the rest of the corrupted-but-plausible streams must match exactly."""
    import warnings
    from oracle import jpeg as oj
    lib = host_lib()
    rng = np.random.default_rng(12)
    seeds = [jpeg_bytes(synth_image(rng, 120, 152), quality=q, subsampling=s, **kw) for q, s, kw in
             [(85, 2, {}), (60, 1, {}), (95, 0, {"optimize": True}), (80, 2, {"restart_marker_blocks": 3}), (85, 2, {"progressive": True}),
              (70, 1, {"progressive": True})]]
    nb = int(lib.hipts_jpeg_slot_bytes(152, 120))
    accepted = refused = 0
    for it in range(1800):
        d = bytearray(seeds[it % 6])
        start = d.index(b"\xff\xda") + 14
        for _ in range(int(rng.integers(1, 4))):
            d[int(rng.integers(start, len(d) - 2))] = int(rng.integers(0, 256))
        data = bytes(d)
        st, slot = entropy_decode(lib, data, slot_bytes=nb)
        if st != 0:
            refused += 1
            continue
        accepted += 1
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        assert np.array_equal(oj.decode_slot(slot), want), it
    assert accepted > 200 and refused > 200, (accepted, refused)


def _segments(data):
    """(marker, start, end) of the marker segments up to and including the first SOS header (entropy data excluded)"""
    out, pos = [], 2
    while True:
        assert data[pos] == 0xFF
        m = data[pos + 1]
        ln = (data[pos + 2] << 8) | data[pos + 3]
        out.append((m, pos, pos + 2 + ln))
        pos += 2 + ln
        if m == 0xDA:
            return out


def test_quantisation_tables_between_progressive_scans_are_refused():
    """libjpeg latches a component's quantisation table at the first scan that contains the component; the slot header takes all of them
    at the first scan.  A table that arrives (or is redefined) after the first SOS therefore never reaches the header: such files --
    legal T.81 -- must go to Pillow (status 1), never be decoded with a missing table (advisor finding, round 4)."""
    from oracle import jpeg as oj
    lib = host_lib()
    rng = np.random.default_rng(5)
    im = synth_image(rng, 120, 160)
    prog = jpeg_bytes(im, quality=80, subsampling=2, progressive=True)
    nb = lib.hipts_jpeg_slot_bytes(160, 120)
    st, slot = entropy_decode(lib, prog)
    assert st == 0 and np.array_equal(oj.decode_slot(slot), np.asarray(Image.open(io.BytesIO(prog)).convert("RGB")))
    segs = _segments(prog)
    dqts = [(a, b) for (m, a, b) in segs if m == 0xDB]
    assert dqts
    second_sos = prog.index(b"\xff\xda", segs[-1][2])
    # (a) the same tables repeated between the first and the second scan: Pillow decodes the same picture, the fast path steps aside
    dup = b"".join(prog[a:b] for a, b in dqts)
    redefined = prog[:second_sos] + dup + prog[second_sos:]
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(redefined)).convert("RGB")), np.asarray(Image.open(io.BytesIO(prog)).convert("RGB")))
    assert entropy_decode(lib, redefined, slot_bytes=nb)[0] == 1
    # (b) every table moved behind the first scan: the first SOS names components whose tables have not arrived (corrupt: 3), and with
    # only the chroma table moved the frame is incomplete at the first scan (refused: 1 or 3, never 0)
    a0, b0 = dqts[0][0], dqts[-1][1]
    moved = prog[:a0] + prog[b0:second_sos] + prog[a0:b0] + prog[second_sos:]
    assert entropy_decode(lib, moved, slot_bytes=nb)[0] in (1, 3)
    # split a two-table DQT segment (Pillow writes one segment per table or one for both) and move the last table only
    tables = []
    for a, b in dqts:
        p = a + 4
        while p < b:
            n = 65 + 64 * (prog[p] >> 4)
            tables.append(prog[p:p + n])
            p += n
    assert len(tables) == 2
    seg = lambda t: b"\xff\xdb" + (len(t) + 2).to_bytes(2, "big") + t
    chroma_late = prog[:a0] + seg(tables[0]) + prog[b0:second_sos] + seg(tables[1]) + prog[second_sos:]
    assert entropy_decode(lib, chroma_late, slot_bytes=nb)[0] in (1, 3)


def test_sixteen_bit_quantisation_entries_are_refused():
    """A DQT with 16-bit entries (Pq = 1) above 255 is beyond any 8-bit file and would overflow the per-column |coefficient x quantiser|
    guard: refused (status 1).  The same table written with Pq = 1 and entries <= 255 decodes like its 8-bit form."""
    from oracle import jpeg as oj
    lib = host_lib()
    rng = np.random.default_rng(6)
    prog = jpeg_bytes(synth_image(rng, 64, 80), quality=85, subsampling=0, progressive=True)
    nb = lib.hipts_jpeg_slot_bytes(80, 64)
    segs = _segments(prog)
    dqts = [(a, b) for (m, a, b) in segs if m == 0xDB]
    wide, huge = b"", b""
    for a, b in dqts:
        p = a + 4
        while p < b:
            assert prog[p] >> 4 == 0
            vals = prog[p + 1:p + 65]
            body = bytes([0x10 | (prog[p] & 15)]) + b"".join(bytes([0, v]) for v in vals)
            wide += b"\xff\xdb" + (len(body) + 2).to_bytes(2, "big") + body
            body = bytes([0x10 | (prog[p] & 15)]) + b"".join(bytes([0x7f, v]) for v in vals)
            huge += b"\xff\xdb" + (len(body) + 2).to_bytes(2, "big") + body
            p += 65
    a0, b0 = dqts[0][0], dqts[-1][1]
    st, slot = entropy_decode(lib, prog[:a0] + wide + prog[b0:], slot_bytes=nb)
    st0, slot0 = entropy_decode(lib, prog, slot_bytes=nb)
    assert st == 0 and st0 == 0 and np.array_equal(oj.decode_slot(slot), oj.decode_slot(slot0))
    assert entropy_decode(lib, prog[:a0] + huge + prog[b0:], slot_bytes=nb)[0] == 1
