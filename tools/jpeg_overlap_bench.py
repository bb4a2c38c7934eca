#!/usr/bin/env python3
"""Development aid (gpurun only): what the device alone sustains when the hybrid JPEG decode of batch k + 1 (side stream) runs beside the
ViT-B/16 forward of batch k (main stream) -- no worker processes, the coefficient slots are prepared once.  Against the forward alone."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
import torch
from PIL import Image
from hiptagsearch import _lib, synth
from hiptagsearch.tagger import ViTTagger

N, S = 64, 448
cfg = dict(synth.VIT_B16_448)
model = ViTTagger(cfg, synth.vit_weights(cfg, seed=0), max_batch=N)
rng = np.random.default_rng(3)
small = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).resize((1024, 768), Image.BICUBIC)
a = np.asarray(small, dtype=np.int16) + rng.integers(-6, 7, (768, 1024, 3), dtype=np.int16)
buf = io.BytesIO()
Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90)
data = buf.getvalue()
lib = _lib.load()
stride = int(lib.hipts_jpeg_slot_bytes(1024, 768))
REGISTERED = len(sys.argv) > 1 and sys.argv[1] == "registered"      # the pipeline's ring: shared memory pinned with hipHostRegister
if REGISTERED:
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(create=True, size=N * stride)
    slots_np = np.ndarray((N, stride), dtype=np.uint8, buffer=shm.buf)
    assert int(torch.cuda.cudart().cudaHostRegister(slots_np.ctypes.data, slots_np.nbytes, 0)) == 0
    slots = torch.from_numpy(slots_np)
else:
    slots = torch.empty((N, stride), dtype=torch.uint8).pin_memory()
src = np.frombuffer(data, dtype=np.uint8)
for i in range(N):
    assert lib.hipts_jpeg_entropy_decode(src.ctypes.data, len(data), slots[i].numpy().ctypes.data, stride) == 0
kinds = np.ones(N, np.int32)
hw = np.ascontiguousarray(np.tile(np.asarray([[768, 1024]], np.int32), (N, 1)))
bufs = [torch.empty((N, S, S, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
probs = torch.empty((N, cfg["num_classes"]), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()


def decode(into):
    _lib.call("hipts_jpeg_batch_u8", slots.data_ptr(), stride, _lib.ptr(kinds), _lib.ptr(hw), N, 1, _lib.ptr(into), S, 3, 0, side.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(side)
    return ev


def run(steps, with_decode):
    ev = decode(bufs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        cur = bufs[k & 1]
        if with_decode:
            torch.cuda.current_stream().wait_event(ev)
            side.wait_stream(torch.cuda.current_stream()) if False else None
            ev = decode(bufs[(k + 1) & 1])          # batch k + 1 decodes while batch k is in the forward
        model.forward_u8(cur, probs=probs, want="probs")
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t0)


run(5, True)
for name, wd in (("forward alone", False), ("forward + decode of the next batch beside it", True), ("forward alone", False), ("forward + decode of the next batch beside it", True)):
    print("%-46s %.0f images/s%s" % (name, run(30, wd), " (ring pinned with hipHostRegister)" if REGISTERED else ""), flush=True)
if REGISTERED:
    torch.cuda.synchronize()
    torch.cuda.cudart().cudaHostUnregister(slots_np.ctypes.data)
    del slots, slots_np
    shm.close()
    shm.unlink()
