#!/usr/bin/env python3
"""Development aid (gpurun only): end-to-end images/s of the tagging loop over a directory of JPEGs, by input pipeline --
the reference's 8 decode threads, the same with the resize on the device, N decode processes, N decode-only processes + device resize,
N entropy-decode-only processes + the rest of the JPEG decode and the resize on the device (hybrid decode, round 4).
usage: pipeline_e2e.py [n_images] [workers]"""
import concurrent.futures, io, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anime-illust-image-searcher_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 8)
import numpy as np
from PIL import Image


def make(args):
    d, i = args
    rng = np.random.default_rng(i)
    small = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).resize((1024, 768), Image.BICUBIC)
    a = np.asarray(small, dtype=np.int16) + rng.integers(-6, 7, (768, 1024, 3), dtype=np.int16)
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(os.path.join(d, "img%05d.jpg" % i), quality=90,
                                                               progressive=bool(int(os.environ.get("E2E_PROGRESSIVE", "0"))))


with tempfile.TemporaryDirectory() as tmp:
    d = os.path.join(tmp, "imgs")
    os.makedirs(d)
    t0 = time.perf_counter()
    with concurrent.futures.ProcessPoolExecutor(W) as ex:
        list(ex.map(make, [(d, i) for i in range(N)], chunksize=16))
    print("%d %sJPEGs 1024x768 q90 written in %.1f s (%d processes)" % (N, "progressive " if int(os.environ.get("E2E_PROGRESSIVE", "0")) else "", time.perf_counter() - t0, W), flush=True)
    ref = None
    only = os.environ.get("E2E_MODES")          # e.g. "3,4": run only those rows (0-based)
    for mode_i, (name, extra) in enumerate([("8 threads, host resize (the reference's structure)", []), ("8 threads, device resize", ["--gpu-resize"]),
                        ("%d processes, host resize" % W, ["--workers", str(W)]), ("%d decode-only processes, device resize" % W, ["--workers", str(W), "--gpu-resize"]),
                        ("%d entropy-decode processes, device IDCT + resize" % W, ["--workers", str(W), "--gpu-resize", "--gpu-jpeg"])]):
        if only and str(mode_i) not in only.split(","):
            continue
        out = os.path.join(tmp, "tags-wd-tagger.txt")
        if os.path.exists(out):
            os.remove(out)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(PKG, "tagging.py"), "--dir", "imgs", "--batch", "64"] + extra, cwd=tmp, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            print(name, "FAILED", r.stderr[-1500:]); continue
        # the CLI prints "<n> files processed / <t> seconds elapsed" from inside the loop: the last pair = loop time without model load
        lines = r.stdout.splitlines()
        el = [float(l.split()[0]) for l in lines if l.endswith("seconds elapsed")]
        cnt = [int(l.split()[0]) for l in lines if l.endswith("files processed")]
        for l in lines:
            if l.startswith("pipeline timing"):
                print("    " + l, flush=True)
        text = open(out, encoding="utf-8").read()
        same = ref is None or text == ref
        ref = ref or text
        # steady state: between the first and the last progress print (pool start-up, first-use allocations and the model's warm-up lie before the first)
        loop = ("%.0f images/s steady (images %d..%d in %.2f s), %.0f incl. start-up" % ((cnt[-1] - cnt[0]) / (el[-1] - el[0]), cnt[0], cnt[-1], el[-1] - el[0], cnt[-1] / el[-1])
                if len(el) >= 2 else "n/a")
        print("%-58s %s; whole process %.1f s; output identical: %s" % (name, loop, dt, same), flush=True)
