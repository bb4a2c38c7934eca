#!/usr/bin/env python3
"""Development aid (gpurun only): device time of the hybrid JPEG decode per batch -- hipts_jpeg_batch_u8 over 64 coefficient slots of a
1024 x 768 4:2:0 JPEG against hipts_resize_batch_u8 over the same images already decoded (what the Pillow workers hand over)."""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "anime-illust-image-searcher_amd"))
import numpy as np
import torch
from PIL import Image
from hiptagsearch import _lib

N, S = 64, 448
rng = np.random.default_rng(3)
small = Image.fromarray(rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)).resize((1024, 768), Image.BICUBIC)
a = np.asarray(small, dtype=np.int16) + rng.integers(-6, 7, (768, 1024, 3), dtype=np.int16)
buf = io.BytesIO()
Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90)
data = buf.getvalue()
img = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
lib = _lib.load()
stride = int(lib.hipts_jpeg_slot_bytes(1024, 768))
slots = torch.empty((N, stride), dtype=torch.uint8).pin_memory()
raw = torch.empty((N, stride), dtype=torch.uint8).pin_memory()
src = np.frombuffer(data, dtype=np.uint8)
for i in range(N):
    assert lib.hipts_jpeg_entropy_decode(src.ctypes.data, len(data), slots[i].numpy().ctypes.data, stride) == 0
    raw[i, :img.size] = torch.from_numpy(img.reshape(-1).copy())
kinds = np.ones(N, np.int32)
hw = np.ascontiguousarray(np.tile(np.asarray([[768, 1024]], np.int32), (N, 1)))
out = torch.empty((N, S, S, 3), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream


def run_jpeg():
    _lib.call("hipts_jpeg_batch_u8", slots.data_ptr(), stride, _lib.ptr(kinds), _lib.ptr(hw), N, 1, _lib.ptr(out), S, 3, 0, s)


def run_raw():
    _lib.call("hipts_resize_batch_u8", raw.data_ptr(), _lib.HOST, stride, _lib.ptr(hw), N, 1, _lib.ptr(out), S, 3, 0, s)


for name, fn in (("hybrid decode + pad + resize", run_jpeg), ("pad + resize of decoded images", run_raw)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    print("%-34s %.2f ms per batch of %d on the device (%.0f images/s), %.2f ms of host time per call" % (name, e0.elapsed_time(e1) / 10, N, N / (e0.elapsed_time(e1) / 10) * 1e3, t_host * 1e3))
