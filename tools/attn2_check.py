#!/usr/bin/env python3
"""Development aid (gpurun only): the round-3 attention kernel (csrc/attn2.hip) -- every geometry variant against a float64 softmax on
the shapes of tests/test_gpu_attention.py, then timed on the ViT-B/16 @448 sub-batch shape next to the round-2 kernel.
usage: attn2_check.py [check|time|both] [variants, e.g. 2,3,1]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hiptagsearch import _lib
from test_gpu_attention import _op, _bits, _from_bits, _reference, _spike, TOL
lib = _lib.load()
fn = lib.hiptsdbg_attention2
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]


def run2(q, k, v, tokens, f16, variant):
    BH = q.shape[0]
    tp = (tokens + 63) // 64 * 64
    qp = np.zeros((BH, tp, 64), np.float32); qp[:, :tokens] = q
    kp = np.zeros((BH, tp, 64), np.float32); kp[:, :tokens] = k
    vp = np.zeros((BH, tp, 64), np.float32); vp[:, :tokens] = v
    out = np.zeros((1, tokens, BH * 64), np.uint16)
    _lib.check(fn(_lib.ptr(_bits(qp, f16)), _lib.ptr(_bits(kp, f16)), _lib.ptr(_bits(vp, f16)), _lib.ptr(out), 1, BH, tokens, tp, int(f16), variant, 0, None))
    return _from_bits(out, f16).reshape(tokens, BH, 64).transpose(1, 0, 2)


def check(variants):
    ok = True
    for variant in variants:
        for f16 in (0, 1):
            for tokens in (784, 1025, 50, 16, 64, 96, 97, 200):
                rng = np.random.default_rng(tokens + 64)
                BH = 6
                q = _op(rng.standard_normal((BH, tokens, 64)) * 0.6, f16); k = _op(rng.standard_normal((BH, tokens, 64)), f16)
                v = _op(rng.standard_normal((BH, tokens, 64)), f16)
                got = run2(q, k, v, tokens, f16, variant)
                err = np.abs(got - _reference(q, k, v)).max()
                flag = "" if err <= TOL[f16] else "   <-- FAIL"
                ok &= err <= TOL[f16]
                print("variant %d f16=%d tokens %4d: max |error| %.3e%s" % (variant, f16, tokens, err, flag), flush=True)
            # forced fallback: late spike, all-low row, early spike
            rng = np.random.default_rng(7)
            BH, tokens = 4, 784
            q = _op(rng.standard_normal((BH, tokens, 64)) * 0.6, f16); k = _op(rng.standard_normal((BH, tokens, 64)), f16)
            v = _op(rng.standard_normal((BH, tokens, 64)), f16)
            _spike(q, k, 0, 5, 600, 300.0, f16)
            base = _op(np.ones(64) * 1.5, f16)
            k[1] = _op(-base[None, :] * (1.0 + 0.05 * rng.standard_normal((tokens, 1))), f16)
            q[1, 300:304] = base * 1.4
            _spike(q, k, 2, 700, 3, 200.0, f16)
            got = run2(q, k, v, tokens, f16, variant)
            want = _reference(q, k, v)
            err = np.abs(got - want).max()
            fin = bool(np.isfinite(got).all())
            flag = "" if (err <= TOL[f16] and fin) else "   <-- FAIL"
            ok &= err <= TOL[f16] and fin
            print("variant %d f16=%d forced fallback: finite %s, max |error| %.3e (row 0/5: %.3e)%s" % (variant, f16, fin, err, np.abs(got[0, 5] - want[0, 5]).max(), flag), flush=True)
    print("CHECK", "OK" if ok else "FAILED", flush=True)
    return ok


def timing(variants, T=784, B=32):
    H, TP = 12, (T + 63) // 64 * 64
    rng = np.random.default_rng(0)
    q = np.zeros((B * H, TP, 64), np.float32); k = np.zeros_like(q); v = np.zeros_like(q)
    q[:, :T] = rng.standard_normal((B * H, T, 64)) * 0.18 * 1.4427
    k[:, :T] = rng.standard_normal((B * H, T, 64)); v[:, :T] = rng.standard_normal((B * H, T, 64))
    flop = 4.0 * T * T * 64 * B * H
    old = lib.hiptsdbg_attention_time
    old.restype = ctypes.c_int
    for f16 in (0, 1):
        qb, kb, vb = _bits(q, f16), _bits(k, f16), _bits(v, f16)
        vT = np.ascontiguousarray(vb.transpose(0, 2, 1))
        us = ctypes.c_double()
        for rep in range(2):
            assert old(_lib.ptr(qb), _lib.ptr(kb), _lib.ptr(vT), B, H, T, TP, 64, f16, 30, ctypes.byref(us)) == 0
            print("round-2 kernel  f16=%d: %.1f us  %.0f TFLOP/s" % (f16, us.value, flop / us.value / 1e6), flush=True)
        for variant in variants:
            for rep in range(2):
                _lib.check(fn(_lib.ptr(qb), _lib.ptr(kb), _lib.ptr(vb), None, B, H, T, TP, f16, variant, 30, ctypes.byref(us)))
                print("attn2 variant %d f16=%d: %.1f us  %.0f TFLOP/s" % (variant, f16, us.value, flop / us.value / 1e6), flush=True)


def stamps(variant, f16=0, T=784, B=32):
    """HIPTS_X_STAMPS builds: one launch, then the cycle stamps of the stamped wave, per tile: wait+barrier | S | softmax | PV per half."""
    H, TP = 12, (T + 63) // 64 * 64
    rng = np.random.default_rng(0)
    q = np.zeros((B * H, TP, 64), np.float32); k = np.zeros_like(q); v = np.zeros_like(q)
    q[:, :T] = rng.standard_normal((B * H, T, 64)) * 0.18 * 1.4427
    k[:, :T] = rng.standard_normal((B * H, T, 64)); v[:, :T] = rng.standard_normal((B * H, T, 64))
    qb, kb, vb = _bits(q, f16), _bits(k, f16), _bits(v, f16)
    us = ctypes.c_double()
    _lib.check(fn(_lib.ptr(qb), _lib.ptr(kb), _lib.ptr(vb), None, B, H, T, TP, f16, variant, 5, ctypes.byref(us)))
    st = np.zeros(4096, np.uint64)
    g = lib.hiptsdbg_attention2_stamps
    g.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _lib.check(g(_lib.ptr(st), 4096))
    st = st.astype(np.int64)
    print("stamps variant %d f16=%d (cycles; per tile: top->barrier passed | S issued | softmax done | PV issued, both halves | tile total)" % (variant, f16))
    for t in range(TP // 64):
        s = st[t * 16: t * 16 + 16]
        nxt = st[(t + 1) * 16] if t + 1 < TP // 64 else s[12]
        print("  tile %2d: vmcnt %5d barrier %5d dma %5d | h0: S %5d sm %5d pv %5d | h1: S %5d sm %5d pv %5d | total %6d" % (
            t, s[5] - s[0], s[6] - s[5], s[1] - s[6], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[10] - s[4], s[11] - s[10], s[12] - s[11], nxt - s[0]))


def stamps3(f16=1, T=784, B=32):
    """HIPTS_A3_STAMPS builds (csrc/attn3.h): one launch of variant 6, then the stamped wave's cycles per tile:
    phase A | phase B of the odd half-step, vmcnt wait, barrier, phase A | phase B of the even half-step."""
    H, TP = 12, (T + 63) // 64 * 64
    rng = np.random.default_rng(0)
    q = np.zeros((B * H, TP, 64), np.float32); k = np.zeros_like(q); v = np.zeros_like(q)
    q[:, :T] = rng.standard_normal((B * H, T, 64)) * 0.18 * 1.4427
    k[:, :T] = rng.standard_normal((B * H, T, 64)); v[:, :T] = rng.standard_normal((B * H, T, 64))
    qb, kb, vb = _bits(q, f16), _bits(k, f16), _bits(v, f16)
    us = ctypes.c_double()
    _lib.check(fn(_lib.ptr(qb), _lib.ptr(kb), _lib.ptr(vb), None, B, H, T, TP, f16, 6, 5, ctypes.byref(us)))
    st = np.zeros(4096, np.uint64)
    g = lib.hiptsdbg_attention2_stamps
    g.argtypes = [ctypes.c_void_p, ctypes.c_int]
    _lib.check(g(_lib.ptr(st), 4096))
    st = st.astype(np.int64)
    print("attn3 stamps f16=%d (%.1f us per launch)" % (f16, us.value))
    for it in range(4):
        b = it * 128
        if st[b + 3] == 0:
            break
        print(" item %d: entry -> B_0 wait %d, prologue (B_0 .. first body) %d" % (it, st[b + 0] - st[b + 4], st[b + 1] - st[b + 0]))
        for t in range(TP // 64 - 1):
            s = st[b + 8 + t * 8: b + 16 + t * 8]
            nxt = st[b + 8 + (t + 1) * 8] if t + 2 < TP // 64 else s[7]
            if t in (0, 1, 5, TP // 64 - 2):
                print("  body %2d: A1 %5d B1 %5d | vmcnt %5d barrier %5d | A0 %5d mask %5d B0 %5d | to next %5d | total %6d" % (
                    t, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[7] - s[6], nxt - s[7], nxt - s[0]))
        last = st[b + 8 + (TP // 64 - 2) * 8 + 7]
        nxt_item = st[b + 128 + 4]
        print("  loop %d, last P V + sums %d, stores %d; item %d cycles; to the next item's entry %d" % (
            last - st[b + 1], st[b + 2] - last, st[b + 3] - st[b + 2], st[b + 3] - st[b + 4], nxt_item - st[b + 3] if nxt_item else -1))


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "both"
    variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3]
    ok = True
    if mode in ("check", "both"):
        ok = check(variants)
    if mode in ("time", "both") and ok:
        timing(variants)
    if mode == "stamps3":
        stamps3(1)
        stamps3(0)
    if mode == "stamps":
        for vnt in variants:
            stamps(vnt, 0)
    sys.exit(0 if ok else 1)
