#!/usr/bin/env python3
"""Development aid (gpurun only): jpeg_overlap_bench.py made to look like the tagging CLI's loop step by step, to find what separates the
CLI's 15.3 ms per batch from the device-only 12.9 ms.  Variants (argv[1], cumulative letters):
  a  decode issued by a PRODUCER THREAD through a queue of depth 2 (events handed over), main thread only launches forwards
  b  + a fresh torch.empty output tensor per batch with record_stream
  c  + selection of each batch on a POST THREAD with its own stream (event from the main thread), main thread waits for batch k - 1's finish
  d  + hipHostRegister'ed shared-memory ring instead of torch pinned memory"""
import io, os, queue, sys, threading, time, concurrent.futures
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
import torch
from PIL import Image
from hiptagsearch import _lib, synth
from hiptagsearch.tagger import ViTTagger, TagSelector, format_lines

V = sys.argv[1] if len(sys.argv) > 1 else "abc"
PARTS = int(os.environ.get("RING_PARTS", "3"))
NODECODE = bool(os.environ.get("NODECODE"))
N, S, STEPS = 64, 448, 40
cfg = dict(synth.VIT_B16_448)
names, cats = synth.label_table(cfg["num_classes"]) if hasattr(synth, "label_table") else (None, None)
rng = np.random.default_rng(3)
small = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).resize((1024, 768), Image.BICUBIC)
a = np.asarray(small, dtype=np.int16) + rng.integers(-6, 7, (768, 1024, 3), dtype=np.int16)
buf = io.BytesIO()
Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90)
data = buf.getvalue()
lib = _lib.load()
stride = int(lib.hipts_jpeg_slot_bytes(1024, 768))
if "d" in V:
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(create=True, size=PARTS * N * stride)
    ring = np.ndarray((PARTS * N, stride), dtype=np.uint8, buffer=shm.buf)
    assert int(torch.cuda.cudart().cudaHostRegister(ring.ctypes.data, ring.nbytes, 0)) == 0
else:
    ring = torch.empty((PARTS * N, stride), dtype=torch.uint8).pin_memory().numpy()
src = np.frombuffer(data, dtype=np.uint8)
for i in range(PARTS * N):
    assert lib.hipts_jpeg_entropy_decode(src.ctypes.data, len(data), ring[i].ctypes.data, stride) == 0
kinds = np.ones(N, np.int32)
hw = np.ascontiguousarray(np.tile(np.asarray([[768, 1024]], np.int32), (N, 1)))
fixed = [torch.empty((N, S, S, 3), dtype=torch.uint8, device="cuda") for _ in range(3)]
dev_probs = [torch.empty((N, cfg["num_classes"]), dtype=torch.float32, device="cuda") for _ in range(2)]


def to_device(k, side):
    out = torch.empty((N, S, S, 3), dtype=torch.uint8, device="cuda") if "b" in V else fixed[k % 3]
    if NODECODE and k >= 3:
        ev = torch.cuda.Event()
        ev.record(side)
        return out, ev
    _lib.call("hipts_jpeg_batch_u8", ring[(k % PARTS) * N].ctypes.data, stride, _lib.ptr(kinds), _lib.ptr(hw), N, 1, _lib.ptr(out), S, 3, 0, side.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(side)
    return out, ev


def batches_e(steps):
    # variant e: the producer thread hands over only "batch k is decoded by the workers"; the CONSUMER issues the device half of batch k + 1
    # itself, on the side stream, right before it launches the forward of batch k
    q = queue.Queue(maxsize=2)

    def produce():
        for k in range(steps):
            q.put(k)
        q.put(None)
    threading.Thread(target=produce, daemon=True).start()
    side = torch.cuda.Stream()
    nxt = None
    k = q.get()
    with torch.cuda.stream(side):
        nxt = to_device(k, side)
    while nxt is not None:
        out, ev = nxt
        k = q.get()
        if k is not None:
            with torch.cuda.stream(side):
                nxt = to_device(k, side)
        else:
            nxt = None
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        if "b" in V:
            out.record_stream(cur)
        yield out


def batches(steps):
    if "e" in V:
        yield from batches_e(steps)
        return
    q = queue.Queue(maxsize=2)

    def produce():
        with torch.cuda.device(0):
            side = torch.cuda.Stream()
            for k in range(steps):
                with torch.cuda.stream(side):
                    q.put(to_device(k, side))
            q.put(None)
    threading.Thread(target=produce, daemon=True).start()
    while True:
        item = q.get()
        if item is None:
            return
        out, ev = item
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        if "b" in V:
            out.record_stream(cur)
        yield out


from hiptagsearch.tagger import Predictor
pr = Predictor(device=0, max_batch=N)
pr.load_model()
model = pr.tagger_model
post = concurrent.futures.ThreadPoolExecutor(1)
post_stream = torch.cuda.Stream()


def finish(probs, ev):
    with torch.cuda.device(0), torch.cuda.stream(post_stream):
        post_stream.wait_event(ev)
        return pr._select_lines([probs], 0.3, True, 0.3, True)


def run(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for k, out in enumerate(batches(steps)):
        p = dev_probs[k & 1]
        model.forward_u8(out, probs=p, want="probs")
        if "c" in V:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            fut = post.submit(finish, p, ev)
            if pending is not None:
                pending.result()
            pending = fut
    if pending is not None:
        pending.result()
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t0)


run(6)
print("variant %-5s %.0f images/s   %.0f images/s" % (V, run(STEPS), run(STEPS)), flush=True)
