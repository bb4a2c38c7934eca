"""Host input pipeline for the tagging / character-feature stages.

Once the forward runs at thousands of images per second the corpus walk is bound by image decode and
resize on the host, which the reference does on 8 threads inside the GIL (tagging.py:52,293-331).  Here:

  DecodePool     N worker PROCESSES decode + composite + pad + resize (exactly `Predictor.prepare_image` /
                 `gen_image_tensor`, tagging.py:100-120,234-252, or the CCIP variant gen_cfeatures.py:285-295
                 before its float normalisation) straight into a shared-memory ring of uint8 HWC slots; the
                 consumer gets [B,S,S,3] uint8 views, one batch ahead of the device, and hands them to the u8
                 entry points (`hipts_vit_forward_u8` / `hipts_ccip_forward_u8`), where /255, normalisation and the
                 BGR flip happen on the device.
  write_shards   the `utility/make_tensor_files.py:164-197` idea (decode once, tag many times) with a packed
  iter_shards    format: `shard-00000.npy` = uint8 [n,S,S,3] (memory-mappable) + `shard-00000.txt` = one path per
                 row; the reference stores one float32 torch tensor file per image (2.4 MB each, 4x the bytes).

  DecodePool(device_resize=True)   the workers only decode and composite: the image goes into its ring slot at its own size and the
                 consumer pads (tagger) and resizes it on the device with the Pillow-exact kernel (`hipts_resize_u8`), yielding
                 uint8 [B,S,S,3] CUDA tensors.  Decode alone is ~3.5x cheaper than decode + resize per core (bench.py
                 `input_pipeline`), so N workers feed ~3.5x the images; the ring is registered as pinned memory so the copies
                 are asynchronous DMA.  Images larger than a slot (`max_pixels`) are resized by the worker as before.

  DecodePool(device_resize=True, device_jpeg=True)   (round 4) hybrid JPEG decode: for a baseline or progressive JPEG the worker runs only the serial half
                 of libjpeg -- markers and Huffman decoding (csrc/jpeg_host.c in libhipts_jpeg_host.so, a library without any GPU runtime
                 behind it) -- and leaves quantised DCT coefficients in its ring slot; inverse DCT, chroma upsampling and YCbCr -> RGB
                 (libjpeg-turbo's arithmetic byte for byte: csrc/jpeg.hip) run on the device in front of the pad + resize
                 (`hipts_jpeg_batch_u8`).  Files that path does not take (PNG, CMYK JPEGs, alpha, anything irregular) are
                 decoded by Pillow in the same worker as before; the two kinds mix freely inside a batch.

Workers are started with the `forkserver` method so that no child is forked from a process that holds a GPU
context; create the pool before or after the model, either is safe.
"""
import multiprocessing as mp
import os
from multiprocessing import shared_memory
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

TAGGER = "tagger"      # white composite, centred pad to square, bicubic resize (tagging.py:100-120 + timm eval transform)
CCIP = "ccip"          # white composite, bilinear resize (gen_cfeatures.py:285-295, 106)


def decode_image(path: str, size: int, mode: str = TAGGER, raw_max_pixels: int = 0) -> Optional[np.ndarray]:
    """One image file -> uint8 [size,size,3] RGB, or None (error printed, like the reference's per-file handler).
    raw_max_pixels > 0: an image of at most that many pixels is returned composited but neither padded nor resized ([h,w,3])."""
    from PIL import Image
    img = None
    try:
        img = Image.open(path)
        img.load()
        if img.mode in ("RGBA", "LA"):
            bg = Image.new("RGB", img.size, (255, 255, 255))
            bg.paste(img, mask=img.split()[-1])
            img = bg
        elif raw_max_pixels > 0 and img.mode == "RGB":
            pass                                         # already what the composite would produce: no extra copies in the decode-only worker
        else:
            img = img.copy().convert("RGB") if mode == TAGGER else img.convert("RGB")
        if raw_max_pixels > 0 and img.size[0] * img.size[1] <= raw_max_pixels:
            return np.asarray(img, dtype=np.uint8)
        if mode == TAGGER:
            w, h = img.size
            m = max(w, h)
            padded = Image.new("RGB", (m, m), (255, 255, 255))
            padded.paste(img, ((m - w) // 2, (m - h) // 2))
            img = padded
            if img.size != (size, size):
                img = img.resize((size, size), Image.BICUBIC)
        else:
            img = img.resize((size, size), resample=Image.BILINEAR)
        return np.asarray(img, dtype=np.uint8)
    except Exception as e:
        if img is not None:
            img.close()
        print('%s: %s' % (type(e), str(e)))
        return None


_W = {}


JPEG_HOST_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libhipts_jpeg_host.so")


def load_jpeg_host_lib():
    """The host half of the hybrid JPEG decode as a ctypes library (no GPU runtime behind it: safe in worker processes)."""
    import ctypes
    if not os.path.exists(JPEG_HOST_LIB):
        raise ImportError("libhipts_jpeg_host.so not found at %s -- build it with `make -C anime-illust-image-searcher_amd/csrc`" % JPEG_HOST_LIB)
    lib = ctypes.CDLL(JPEG_HOST_LIB)
    lib.hipts_jpeg_entropy_decode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    lib.hipts_jpeg_entropy_decode.restype = ctypes.c_int
    lib.hipts_jpeg_slot_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.hipts_jpeg_slot_bytes.restype = ctypes.c_int64
    return lib


def _raw_slot_bytes(raw_max_pixels: int, size: int, jpeg: bool) -> int:
    """Bytes of a ring slot in raw mode: the decoded image, or -- device_jpeg -- the coefficient blocks of a 4:2:0 / 4:2:2 JPEG of that
    many pixels: 2 B per sample, at most 2 samples per pixel, planes padded to whole 16 x 16 MCUs, a 1 KB header (csrc/jpeg_slot.h).
    A JPEG that does not fit (4:4:4 above two thirds of max_pixels) is decoded by Pillow like any other file."""
    px = max(raw_max_pixels, size * size)
    return px * 3 if not jpeg else px * 4 + 2 * 16 * 4 * (int(px ** 0.5) + 16) + 4096


def _worker_init(shm_name: str, slots: int, size: int, mode: str, raw_max_pixels: int = 0, jpeg: bool = False) -> None:
    shm = shared_memory.SharedMemory(name=shm_name)
    _W["shm"] = shm
    if raw_max_pixels > 0:
        _W["ring"] = np.ndarray((slots, _raw_slot_bytes(raw_max_pixels, size, jpeg)), dtype=np.uint8, buffer=shm.buf)
    else:
        _W["ring"] = np.ndarray((slots, size, size, 3), dtype=np.uint8, buffer=shm.buf)
    _W["size"] = size
    _W["mode"] = mode
    _W["raw"] = raw_max_pixels
    _W["jpeg"] = load_jpeg_host_lib() if jpeg else None


def _worker_decode(task: Tuple[int, str]):
    """-> False (decode failed), True (model-input image in the slot), (h, w) (raw mode: the composited image at its own size) or
    (h, w, 1) (raw mode with device_jpeg: the slot holds the JPEG's coefficient blocks, csrc/jpeg_slot.h)."""
    slot, path = task
    if _W.get("jpeg") is not None:
        try:
            with open(path, "rb") as f:
                data = f.read(2)
                if data == b"\xff\xd8":            # a JPEG by its first marker: the whole file; anything else is left to Pillow unread
                    data += f.read()
        except OSError:
            data = b""
        if len(data) > 4:
            row = _W["ring"][slot]
            src = np.frombuffer(data, dtype=np.uint8)
            if _W["jpeg"].hipts_jpeg_entropy_decode(src.ctypes.data, len(data), row.ctypes.data, row.nbytes) == 0:
                hd = row[:16].view(np.int32)
                return (int(hd[3]), int(hd[2]), 1)
        # everything else -- and every file the fast path refused -- goes the way it always went
    a = decode_image(path, _W["size"], _W["mode"], _W["raw"])
    if a is None:
        return False
    if _W["raw"] > 0:
        _W["ring"][slot, :a.size] = a.reshape(-1)
        return (int(a.shape[0]), int(a.shape[1]))
    _W["ring"][slot] = a
    return True


class DecodePool:
    """Multi-process decode into shared memory, one batch ahead of the consumer.

        with DecodePool(workers=16, size=448, batch=64) as pool:
            for paths, images_u8 in pool.batches(file_list):      # images_u8: uint8 [len(paths),448,448,3]
                ...                                               # valid until the next iteration

    Files that fail to decode are dropped from `paths` (message printed by the worker), like the reference."""

    def __init__(self, workers: Optional[int] = None, size: int = 448, batch: int = 64, mode: str = TAGGER,
                 device_resize: bool = False, device: int = 0, max_pixels: int = 1600 * 1600, device_jpeg: bool = False):
        if device_jpeg and not device_resize:
            raise ValueError("device_jpeg needs device_resize: the decoded image only exists on the device")
        self.workers = max(1, workers or (os.cpu_count() or 1))
        self.size, self.batch, self.mode = size, batch, mode
        # ring parts: two batches (one being decoded, one being consumed); with the device resize three -- two batches at the workers, so
        # that they never idle at a batch boundary waiting for the stragglers of the batch or for the device to copy a part out
        self.parts = 3 if device_resize else 2
        self.slots = self.parts * batch
        self.raw = max(int(max_pixels), size * size) if device_resize else 0
        self.device = device
        self.jpeg = bool(device_jpeg)
        self._pinned = False
        self._events = [None] * 3
        slot_bytes = _raw_slot_bytes(self.raw, size, self.jpeg) if self.raw else size * size * 3
        self._shm = shared_memory.SharedMemory(create=True, size=self.slots * slot_bytes)
        if self.raw:
            self._ring = np.ndarray((self.slots, slot_bytes), dtype=np.uint8, buffer=self._shm.buf)
            import torch
            try:    # pinned ring: the per-image copies become asynchronous DMA (without it they are staged copies -- slower, still correct)
                self._pinned = int(torch.cuda.cudart().cudaHostRegister(self._ring.ctypes.data, self._ring.nbytes, 0)) == 0
            except Exception:
                self._pinned = False
        else:
            self._ring = np.ndarray((self.slots, size, size, 3), dtype=np.uint8, buffer=self._shm.buf)
        ctx = mp.get_context("forkserver")
        self._pool = ctx.Pool(self.workers, initializer=_worker_init, initargs=(self._shm.name, self.slots, size, mode, self.raw, self.jpeg))

    def _to_device(self, base: int, results, stream) -> "object":
        """Raw mode: the ring slots base .. of one decoded batch -> uint8 [n,S,S,3] CUDA tensor on `stream`: copied, padded (tagger: white,
        centred -- prepare_image, tagging.py:100-120) and resized (bicubic for the tagger's transform, bilinear for gen_cfeatures.py:101) by
        one library call (hipts_resize_batch_u8); nothing waits for the device."""
        import torch
        from . import _lib
        S = self.size
        out = torch.empty((len(results), S, S, 3), dtype=torch.uint8, device="cuda:%d" % self.device)
        if self.jpeg:       # coefficient slots and Pillow-decoded slots side by side: decode on the device, then the same pad + resize
            hw = np.ascontiguousarray(np.asarray([r[:2] for r in results], dtype=np.int32))
            kinds = np.ascontiguousarray(np.asarray([1 if len(r) > 2 else 0 for r in results], dtype=np.int32))
            _lib.call("hipts_jpeg_batch_u8", self._ring[base].ctypes.data, self._ring.shape[1], _lib.ptr(kinds), _lib.ptr(hw), len(results),
                      1 if self.mode == TAGGER else 0, _lib.ptr(out), S, 3 if self.mode == TAGGER else 2, self.device, stream.cuda_stream)
            return out
        hw = np.ascontiguousarray(np.asarray(results, dtype=np.int32).reshape(-1, 2))
        _lib.call("hipts_resize_batch_u8", self._ring[base].ctypes.data, _lib.HOST, self._ring.shape[1], _lib.ptr(hw), len(results),
                  1 if self.mode == TAGGER else 0, _lib.ptr(out), S, 3 if self.mode == TAGGER else 2, self.device, stream.cuda_stream)
        return out

    def batches(self, paths: Sequence[str]) -> Iterator[Tuple[List[str], np.ndarray]]:
        chunks = [list(paths[i:i + self.batch]) for i in range(0, len(paths), self.batch)]
        if not chunks:
            return
        if self.raw:
            yield from self._batches_raw(chunks)
            return

        def submit(k: int):
            base = (k & 1) * self.batch
            return self._pool.map_async(_worker_decode, [(base + i, p) for i, p in enumerate(chunks[k])],
                                        chunksize=max(1, len(chunks[k]) // (4 * self.workers)))
        pending = submit(0)
        for k, chunk in enumerate(chunks):
            ok = pending.get()
            if k + 1 < len(chunks):
                pending = submit(k + 1)          # the other half of the ring: batch k - 1 has been consumed
            base = (k & 1) * self.batch
            view = self._ring[base:base + len(chunk)]
            if all(ok):
                yield chunk, view
            else:
                keep = [i for i, good in enumerate(ok) if good]
                if keep:
                    yield [chunk[i] for i in keep], np.ascontiguousarray(view[keep])

    def _batches_raw(self, chunks):
        """decode (worker processes, batches k + 2 and k + 3: the ring has three parts)  ||  copy + JPEG device half + pad + resize (this
        producer thread, its own stream, batch k + 1)  ||  the consumer's forward (caller's stream, batch k).  The hand-over is a queue of
        (paths, tensor, event)."""
        import queue
        import threading
        import torch
        q: "queue.Queue" = queue.Queue(maxsize=2)
        stop = threading.Event()
        timing = os.environ.get("HIPTS_PIPELINE_TIMING")          # development aid: where the producer and the consumer wait (printed at the end)
        import time as _t
        T = {"wait_workers": 0.0, "submit": 0.0, "to_device": 0.0, "put": 0.0, "consumer_wait": 0.0, "consumer_busy": 0.0, "batches": 0}

        P = self.parts

        def submit(k: int):
            base = (k % P) * self.batch
            if self._events[k % P] is not None:
                self._events[k % P].synchronize()       # the copies out of this part of the ring (batch k - P) are done
            return self._pool.map_async(_worker_decode, [(base + i, p) for i, p in enumerate(chunks[k])],
                                        chunksize=max(1, len(chunks[k]) // (4 * self.workers)))

        def produce():
            try:
                with torch.cuda.device(self.device):
                    side = torch.cuda.Stream()
                    ahead = P - 1                           # batches at the workers
                    pend = {j: submit(j) for j in range(min(ahead, len(chunks)))}
                    for k, chunk in enumerate(chunks):
                        t0 = _t.perf_counter()
                        res = pend.pop(k).get()
                        t1 = _t.perf_counter()
                        if k + ahead < len(chunks):
                            pend[k + ahead] = submit(k + ahead)
                        t2 = _t.perf_counter()
                        warm = k >= 8                        # (timing aid: the first batches carry pool start-up and first-use allocations)
                        if warm:
                            T["wait_workers"] += t1 - t0
                            T["submit"] += t2 - t1
                        if stop.is_set():
                            return
                        keep = [i for i, r in enumerate(res) if r is not False]
                        if not keep:
                            continue
                        base = (k % P) * self.batch
                        with torch.cuda.stream(side):
                            if len(keep) == len(chunk):
                                out = self._to_device(base, res, side)
                            else:       # failed decodes leave holes: slot by slot
                                out = torch.cat([self._to_device(base + i, [res[i]], side) for i in keep])
                            ev = torch.cuda.Event()
                            ev.record(side)
                        self._events[k % P] = ev
                        t3 = _t.perf_counter()
                        q.put(([chunk[i] for i in keep], out, ev))
                        if warm:
                            T["to_device"] += t3 - t2
                            T["put"] += _t.perf_counter() - t3
                            T["batches"] += 1
                q.put(None)
            except BaseException as e:      # hand the error to the consumer
                q.put(e)

        th = threading.Thread(target=produce, name="hipts-decode-producer", daemon=True)
        th.start()
        try:
            seen = 0
            while True:
                t0 = _t.perf_counter()
                item = q.get()
                seen += 1
                if seen > 8:
                    T["consumer_wait"] += _t.perf_counter() - t0
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                kept, out, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                out.record_stream(cur)
                t0 = _t.perf_counter()
                yield kept, out
                if seen > 8:
                    T["consumer_busy"] += _t.perf_counter() - t0
        finally:
            if timing and T["batches"]:
                n = T["batches"]
                print("pipeline timing per batch (ms): producer waits for the workers %.2f, submits the next batch %.2f, copies + decodes + resizes "
                      "(host side of the calls) %.2f, waits for a free queue place %.2f | consumer waits for a batch %.2f, works on it %.2f"
                      % tuple(1e3 * T[k] / n for k in ("wait_workers", "submit", "to_device", "put", "consumer_wait", "consumer_busy")), flush=True)
            stop.set()
            while th.is_alive():            # unblock a producer that waits on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)

    def close(self) -> None:
        if getattr(self, "_pinned", False):
            try:
                import torch
                torch.cuda.synchronize()
                torch.cuda.cudart().cudaHostUnregister(self._ring.ctypes.data)
            except Exception:
                pass
            self._pinned = False
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None
        if self._shm is not None:
            self._ring = None
            self._shm.close()
            self._shm.unlink()
            self._shm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_shards(paths: Sequence[str], out_dir: str, size: int = 448, mode: str = TAGGER, workers: Optional[int] = None,
                 per_shard: int = 1024, batch: int = 64) -> int:
    """Decode once: shard-%05d.npy (uint8 [n,size,size,3]) + shard-%05d.txt (paths, row aligned).  Returns the
    number of images written."""
    os.makedirs(out_dir, exist_ok=True)
    total = 0
    shard = 0
    buf_paths: List[str] = []
    buf = np.empty((per_shard, size, size, 3), dtype=np.uint8)

    def flush():
        nonlocal shard, buf_paths
        if not buf_paths:
            return
        np.save(os.path.join(out_dir, "shard-%05d.npy" % shard), buf[:len(buf_paths)])
        with open(os.path.join(out_dir, "shard-%05d.txt" % shard), "w", encoding="utf-8") as f:
            f.write("".join(p + "\n" for p in buf_paths))
        shard += 1
        buf_paths = []
    with DecodePool(workers, size, batch, mode) as pool:
        for kept, images in pool.batches(paths):
            for p, img in zip(kept, images):
                buf[len(buf_paths)] = img
                buf_paths.append(p)
                total += 1
                if len(buf_paths) == per_shard:
                    flush()
    flush()
    return total


def iter_shards(shard_dir: str, batch: int = 64) -> Iterator[Tuple[List[str], np.ndarray]]:
    """Batches of (paths, uint8 [b,S,S,3]) from a directory written by write_shards; arrays are memory-mapped."""
    names = sorted(f for f in os.listdir(shard_dir) if f.startswith("shard-") and f.endswith(".npy"))
    for n in names:
        arr = np.load(os.path.join(shard_dir, n), mmap_mode="r")
        with open(os.path.join(shard_dir, n[:-4] + ".txt"), encoding="utf-8") as f:
            paths = [l.rstrip("\n") for l in f]
        if len(paths) != arr.shape[0]:
            raise ValueError("%s: %d rows but %d paths" % (n, arr.shape[0], len(paths)))
        for s in range(0, len(paths), batch):
            yield paths[s:s + batch], np.ascontiguousarray(arr[s:s + batch])


def index_shards(shard_dir: str) -> Tuple[List[Tuple[str, int, int]], List[str]]:
    """([(npy path, first global row, rows)], all paths in shard order) of a directory written by write_shards: what every rank needs to
    find its own slice of the rows (Predictor.process_directory_sharded)."""
    names = sorted(f for f in os.listdir(shard_dir) if f.startswith("shard-") and f.endswith(".npy"))
    index: List[Tuple[str, int, int]] = []
    paths: List[str] = []
    for n in names:
        with open(os.path.join(shard_dir, n[:-4] + ".txt"), encoding="utf-8") as f:
            p = [l.rstrip("\n") for l in f]
        index.append((os.path.join(shard_dir, n), len(paths), len(p)))
        paths.extend(p)
    return index, paths


def iter_shard_rows(index: Sequence[Tuple[str, int, int]], lo: int, hi: int, batch: int = 64) -> Iterator[Tuple[int, np.ndarray]]:
    """(first global row, uint8 [b,S,S,3]) batches of the global rows [lo, hi) across the shards of `index` (memory-mapped: a rank touches
    only the bytes of its own slice)."""
    for path, first, rows in index:
        a, b = max(lo, first), min(hi, first + rows)
        if a >= b:
            continue
        arr = np.load(path, mmap_mode="r")
        if arr.shape[0] != rows:
            raise ValueError("%s: %d rows but %d paths" % (path, arr.shape[0], rows))
        for s0 in range(a, b, batch):
            e = min(b, s0 + batch)
            yield s0, np.ascontiguousarray(arr[s0 - first:e - first])
