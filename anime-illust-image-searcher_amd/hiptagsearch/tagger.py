"""Tagging stage: host-side mirror of tagging.py's Predictor over the device kernels.

  ViTTagger.forward / .forward_u8     timm `model.forward` + `F.sigmoid`     tagging.py:174-176
  TagSelector                         per-image MCut selection                 tagging.py:61-66,185-227
  Predictor.predict(tensors, general_thresh, general_mcut_enabled, character_thresh,
                    character_mcut_enabled) -> List[str]                       tagging.py:156-229
  Predictor.prepare_image / gen_image_tensor / list_files_recursive / process_directory
                                                                               tagging.py:91-120,234-359
"""
import ctypes
import datetime
import os
import time
import concurrent.futures
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, synth
from ._lib import VitConfig, c_double, c_void_p

EXTENSIONS: List[str] = ['.png', '.jpg', '.jpeg', ".PNG", ".JPG", ".JPEG"]   # tagging.py:47
BATCH_SIZE: int = 10          # tagging.py:49 (reference default; the device path takes any batch <= max_batch)
WORKER_NUM: int = 8           # tagging.py:52
PROGRESS_INTERVAL: int = 1000  # tagging.py:50


class ViTTagger:
    """Device-resident ViT tagger.  `weights` uses timm state_dict keys (float32 numpy arrays).

    MFMA operands default to IEEE half (cfg["operand_f16"] = 1, round 3): a bf16 checkpoint's matrices are exact in half (down to
    its subnormal range), activations keep 11 instead of 8 significant bits at the same matrix rate.  With bf16 activations the
    logit error against the float32 oracle is 4-6e-4 on noise images but ~4e-3 on flat / cel-shaded ones -- every token then carries
    the SAME rounding error, which the mean pool does not average out -- i.e. outside the 1e-3 of BASELINE.json on the images the
    product is for (tests/test_gpu_vit.py::test_vit_default_config_structured_images).  cfg["operand_f16"] = 0 selects bf16."""

    def __init__(self, cfg: Dict, weights: Dict[str, np.ndarray], max_batch: int = 64, device: int = 0):
        self.cfg = dict(cfg)
        self.device = device
        self.max_batch = max_batch
        self.num_classes = cfg["num_classes"]
        c = VitConfig(cfg["image_size"], cfg["patch"], cfg["dim"], cfg["depth"], cfg["heads"], cfg["mlp_dim"],
                      cfg["num_classes"], cfg.get("ln_eps", 1e-6), cfg.get("gelu_tanh", 1), cfg.get("pool_then_norm", 0),
                      max_batch, cfg.get("operand_f16", 1))
        self._h = c_void_p()
        _lib.call("hipts_vit_create", ctypes.byref(c), device, ctypes.byref(self._h))
        for key, val in weights.items():
            arr = np.ascontiguousarray(val, dtype=np.float32)
            _lib.call("hipts_vit_set_tensor", self._h, key.encode(), _lib.ptr(arr), ctypes.c_int64(arr.size))

    @classmethod
    def from_safetensors(cls, path: str, cfg: Dict, **kw) -> "ViTTagger":
        """Load a timm-layout checkpoint (model.safetensors of a wd-vit-tagger style repo)."""
        from safetensors.numpy import load_file
        return cls(cfg, {k: v.astype(np.float32) for k, v in load_file(path).items()}, **kw)

    def flops_per_image(self) -> float:
        f = c_double()
        _lib.call("hipts_vit_flops_per_image", self._h, ctypes.byref(f))
        return f.value

    def _run(self, fn: str, x, batch: int, logits, probs):
        out_space = _lib.HOST
        for o in (logits, probs):
            if o is not None:
                out_space = _lib.memspace_of(o)
        _lib.call(fn, self._h, _lib.ptr(x), _lib.memspace_of(x), batch, _lib.ptr(logits), _lib.ptr(probs), out_space,
                  _lib.current_stream_ptr())

    def forward_u8(self, images, logits=None, probs=None, want: str = "both"):
        """images: uint8 [B,S,S,3] RGB (numpy or torch, host or device).  Returns (logits, probs)
        float32 [B,num_classes] in the same memory space as the outputs passed (host numpy by default)."""
        B = int(images.shape[0])
        if logits is None and want in ("both", "logits"):
            logits = np.empty((B, self.num_classes), dtype=np.float32)
        if probs is None and want in ("both", "probs"):
            probs = np.empty((B, self.num_classes), dtype=np.float32)
        if isinstance(images, np.ndarray):
            images = np.ascontiguousarray(images, dtype=np.uint8)
        self._run("hipts_vit_forward_u8", images, B, logits, probs)
        return logits, probs

    def forward(self, x, logits=None, probs=None):
        """x: float32 [B,3,S,S] normalised BGR -- the tensor tagging.py:174 feeds model.forward."""
        B = int(x.shape[0])
        if logits is None:
            logits = np.empty((B, self.num_classes), dtype=np.float32)
        if probs is None:
            probs = np.empty((B, self.num_classes), dtype=np.float32)
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float32)
        self._run("hipts_vit_forward_f32", x, B, logits, probs)
        return logits, probs

    def close(self):
        if self._h:
            _lib.call("hipts_vit_destroy", self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EvaTagger(ViTTagger):
    """Device-resident EVA02 tagger (the model tagging.py:45 names: wd-eva02-large-tagger-v3).  Same interface as
    ViTTagger; `weights` uses timm `Eva` state_dict keys.  MFMA operands default to IEEE half here (cfg["operand_f16"] = 1):
    the same matrix rate as bf16 with 8x smaller activation rounding -- max |dlogit| against the float32 oracle 2.1e-3
    instead of 1.7e-2 on EVA02-L/14 (tests/test_gpu_eva.py); operand_f16 = 0 selects bf16."""

    _PREFIX = "hipts_eva"

    def __init__(self, cfg: Dict, weights: Dict[str, np.ndarray], max_batch: int = 32, device: int = 0):
        self.cfg = dict(cfg)
        self.device = device
        self.max_batch = max_batch
        self.num_classes = cfg["num_classes"]
        c = _lib.EvaConfig(cfg["image_size"], cfg["patch"], cfg["dim"], cfg["depth"], cfg["heads"], cfg["mlp_hidden"], cfg["num_classes"],
                           cfg.get("ln_eps", 1e-6), cfg.get("rope_ref_grid", 16), max_batch, cfg.get("operand_f16", 1))
        self._h = c_void_p()
        _lib.call("hipts_eva_create", ctypes.byref(c), device, ctypes.byref(self._h))
        for key, val in weights.items():
            arr = np.ascontiguousarray(val, dtype=np.float32)
            _lib.call("hipts_eva_set_tensor", self._h, key.encode(), _lib.ptr(arr), ctypes.c_int64(arr.size))

    def flops_per_image(self) -> float:
        f = c_double()
        _lib.call("hipts_eva_flops_per_image", self._h, ctypes.byref(f))
        return f.value

    def _run(self, fn: str, x, batch: int, logits, probs):
        super()._run(fn.replace("hipts_vit", "hipts_eva"), x, batch, logits, probs)

    def close(self):
        if getattr(self, "_h", None):
            _lib.call("hipts_eva_destroy", self._h)
            self._h = c_void_p()


def device_resize_u8(image: np.ndarray, out_h: int, out_w: int, pil_filter: int = 3, device: int = 0, out=None):
    """uint8 [H,W,3] (host numpy or CUDA tensor) -> uint8 [out_h,out_w,3] CUDA tensor: PIL's Image.resize((out_w, out_h), filter) on the
    device, bit for bit (pil_filter 3 = BICUBIC: tagging.py:241's transform; 2 = BILINEAR: gen_cfeatures.py:101)."""
    import torch
    if isinstance(image, np.ndarray):
        image = np.ascontiguousarray(image, dtype=np.uint8)
    if out is None:
        out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device="cuda:%d" % device)
    _lib.call("hipts_resize_u8", _lib.ptr(image), _lib.memspace_of(image), int(image.shape[0]), int(image.shape[1]), _lib.ptr(out), out_h, out_w,
              pil_filter, device, _lib.current_stream_ptr())
    return out


class TagSelector:
    def __init__(self, category: np.ndarray, max_batch: int = 64, device: int = 0):
        self.category = np.ascontiguousarray(category, dtype=np.int32)
        self.num_classes = len(self.category)
        self._h = c_void_p()
        _lib.call("hipts_tagsel_create", _lib.ptr(self.category), self.num_classes, device, max_batch, ctypes.byref(self._h))

    def run(self, probs, general_thresh=0.3, general_mcut=True, character_thresh=0.3, character_mcut=True,
            row_cap: Optional[int] = None):
        """probs float32 [B,C] (host numpy or device tensor).  Returns (counts int32 [B,2], ids int32
        [B,row_cap], thresholds float64 [B,2]) on the host."""
        B = int(probs.shape[0])
        row_cap = row_cap or self.num_classes
        counts = np.empty((B, 2), dtype=np.int32)
        ids = np.empty((B, row_cap), dtype=np.int32)
        thr = np.empty((B, 2), dtype=np.float64)
        if isinstance(probs, np.ndarray):
            probs = np.ascontiguousarray(probs, dtype=np.float32)
        _lib.call("hipts_tagsel_run", self._h, _lib.ptr(probs), _lib.memspace_of(probs), B, c_double(general_thresh),
                  int(general_mcut), c_double(character_thresh), int(character_mcut), _lib.ptr(counts), _lib.ptr(ids),
                  row_cap, _lib.ptr(thr), _lib.HOST, _lib.current_stream_ptr())
        return counts, ids, thr

    def run_device(self, probs, rows, general_thresh=0.3, general_mcut=True, character_thresh=0.3, character_mcut=True):
        """Device-resident variant: `rows` is an int32 device tensor [B, 2 + row_cap]; row b receives
        {#general, #character, ids...} -- the fixed-width tag row that is all-gathered across ranks."""
        B, width = int(rows.shape[0]), int(rows.shape[1])
        assert rows.is_cuda and rows.is_contiguous() and width > 2
        _lib.call("hipts_tagsel_run_rows", self._h, _lib.ptr(probs), B, c_double(general_thresh), int(general_mcut),
                  c_double(character_thresh), int(character_mcut), _lib.ptr(rows), width, _lib.current_stream_ptr())
        return rows

    def close(self):
        if self._h:
            _lib.call("hipts_tagsel_destroy", self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_lines(names: Sequence[str], counts: np.ndarray, ids: np.ndarray) -> List[str]:
    """tagging.py:210-225: general tags then character tags, ' ' -> '_', joined by ','."""
    names_us = _underscored(names)
    out = []
    for r in range(len(counts)):
        ng, nc = int(counts[r, 0]), int(counts[r, 1])
        row = ids[r, :ng + nc].tolist()         # general tags, then character tags (hipts_tagsel_run's row layout)
        line = ",".join([names_us[i] for i in row])
        out.append("," + line if ng == 0 and nc > 0 else line)      # (tagging.py:212-225 appends "," + characters to an empty general string too)
    return out


_US_CACHE: Dict[int, Tuple[Sequence[str], List[str]]] = {}


def _underscored(names: Sequence[str]) -> List[str]:
    """names with ' ' -> '_' (tagging.py:212,219), computed once per label table: the per-tag str.replace was a third of the formatting time."""
    hit = _US_CACHE.get(id(names))
    if hit is None or hit[0] is not names:
        hit = (names, [n.replace(' ', '_') for n in names])
        _US_CACHE.clear()
        _US_CACHE[id(names)] = hit
    return hit[1]


class Predictor:
    """Same surface as tagging.py's Predictor.  `load_model` needs a checkpoint + label table, which
    the reference downloads from the HF hub (tagging.py:146-151); offline it takes local files or the
    seeded synthetic stand-ins."""

    def __init__(self, device: int = 0, max_batch: int = 64, compat: bool = False, gpu_resize: bool = False, gpu_jpeg: bool = False,
                 precise: bool = False) -> None:
        self.device = device
        # precise: operand_f16 |= 16 (HIPTS_OPERAND_SPLIT_ATT, include/hip_tagsearch.h) for ViT and EVA02 -- the attention output reaches the
        # output projection as a hi | lo pair of halves.  Flat pictures' logit error 1.7e-3 -> 2.1e-4 for about 5 % of the throughput
        # (profiles/r04_precision_vit.json; the bench line's `precise` block carries both)
        self.precise = precise
        self.max_batch = max_batch
        self.compat = compat            # reproduce the reference's dropped tail batch
        self.gpu_resize = gpu_resize    # the decode threads only decode and pad; Resize(bicubic) runs on the device (hipts_resize_u8)
        self.gpu_jpeg = gpu_jpeg        # worker processes only entropy-decode baseline JPEGs; the rest of libjpeg runs on the device (pipeline.DecodePool)
        self.tagger_model: Optional[ViTTagger] = None
        self.selector: Optional[TagSelector] = None
        self.tag_names: Optional[List[str]] = None
        self.rating_index = self.general_index = self.character_index = None
        self.cfg = None
        self.f = None

    # ---- tagging.py:91-98
    def list_files_recursive(self, dir_path: str) -> List[str]:
        file_list: List[str] = []
        for root, _, files in os.walk(dir_path):
            for file in files:
                file_path = os.path.join(root, file)
                if any(file_path.endswith(ext) for ext in EXTENSIONS):
                    file_list.append(file_path)
        return file_list

    # ---- tagging.py:100-120
    def prepare_image(self, image):
        from PIL import Image
        if image.mode in ('RGBA', 'LA'):
            background = Image.new("RGB", image.size, (255, 255, 255))
            background.paste(image, mask=image.split()[-1])
            image = background
        else:
            image = image.copy().convert("RGB")
        w, h = image.size
        max_dim = max(w, h)
        padded = Image.new("RGB", (max_dim, max_dim), (255, 255, 255))
        padded.paste(image, ((max_dim - w) // 2, (max_dim - h) // 2))
        return padded

    # ---- tagging.py:122-154
    def load_labels(self, names: Sequence[str], category: np.ndarray):
        category = np.asarray(category)
        self.rating_index = list(np.where(category == 9)[0])
        self.general_index = list(np.where(category == 0)[0])
        self.character_index = list(np.where(category == 4)[0])
        self.tag_names = list(names)
        self.selector = TagSelector(category, self.max_batch, self.device)

    def load_model(self, checkpoint: Optional[str] = None, labels_csv: Optional[str] = None, cfg: Optional[Dict] = None,
                   seed: int = 0) -> None:
        if self.tagger_model is not None:
            return
        self.cfg = dict(cfg or synth.VIT_B16_448)
        if self.precise:
            self.cfg["operand_f16"] = int(self.cfg.get("operand_f16", 1)) | 16
        # an EVA02 configuration (the reference's MODEL_REPO, tagging.py:45) is recognised by its SwiGLU width
        eva = "mlp_hidden" in self.cfg
        cls_ = EvaTagger if eva else ViTTagger
        if checkpoint:
            self.tagger_model = cls_.from_safetensors(checkpoint, self.cfg, max_batch=self.max_batch, device=self.device)
        else:
            print("No checkpoint given: using the seeded synthetic %s weights (no network in this environment)." % ("EVA02" if eva else "ViT"))
            # the trained-like variant (peaked attention, sparse probabilities: tens of labels per image, as a real tagger selects)
            weights = synth.eva_weights(self.cfg, seed, trained_like=True) if eva else synth.vit_weights(self.cfg, seed, trained_like=True)
            self.tagger_model = cls_(self.cfg, weights, self.max_batch, self.device)
        if labels_csv:
            import pandas as pd
            df = pd.read_csv(labels_csv, usecols=["name", "category"])
            self.load_labels(df["name"].tolist(), df["category"].to_numpy())
        else:
            names, cat = synth.label_table(self.cfg["num_classes"])
            self.load_labels(names, cat)

    # ---- tagging.py:234-252: returns the uint8 HWC image (the device kernel applies the transform)
    def gen_image_tensor(self, file_path: str, gpu_resize: bool = False):
        """gpu_resize: decode and pad on the host, Resize(bicubic) on the device (hipts_resize_u8: Pillow's resample bit for bit) --
        returns a uint8 [S,S,3] CUDA tensor instead of a numpy array."""
        from PIL import Image
        img = None
        try:
            img = Image.open(file_path)
            img.load()
            img_tmp = self.prepare_image(img)
            size = self.cfg["image_size"]
            if gpu_resize:
                return device_resize_u8(np.asarray(img_tmp, dtype=np.uint8), size, size, 3, self.device)
            if img_tmp.size != (size, size):
                img_tmp = img_tmp.resize((size, size), Image.BICUBIC)   # timm eval transform: Resize(bicubic) + CenterCrop
            return np.asarray(img_tmp, dtype=np.uint8)
        except Exception as e:
            if img is not None:
                img.close()
            print('%s: %s' % (type(e), str(e)))
            return None

    # ---- tagging.py:156-229
    def predict(self, tensors: List, general_thresh: float, general_mcut_enabled: bool, character_thresh: float,
                character_mcut_enabled: bool) -> List[str]:
        return self._select_lines(self._forward_probs(tensors), general_thresh, general_mcut_enabled, character_thresh, character_mcut_enabled)

    def _forward_probs(self, tensors: List) -> List[np.ndarray]:
        """tagging.py:164-176: the forward + sigmoid of predict, per chunk of max_batch images."""
        first = tensors[0]
        out: List[np.ndarray] = []
        packed = isinstance(tensors, np.ndarray) and tensors.dtype == np.uint8 and tensors.ndim == 4      # [B,S,S,3] from the decode pool / shards
        for s in range(0, len(tensors), self.max_batch):
            chunk = tensors[s:s + self.max_batch]
            if packed or (hasattr(tensors, "is_cuda") and tensors.is_cuda and tensors.dim() == 4):      # ... or [B,S,S,3] on the device (decode pool with device resize)
                _, probs = self.tagger_model.forward_u8(chunk, want="probs")
            elif hasattr(first, "is_cuda") and first.is_cuda:                     # uint8 [S,S,3] device tensors (gpu_resize)
                import torch
                _, probs = self.tagger_model.forward_u8(torch.stack(list(chunk)), want="probs")
            elif hasattr(first, "dtype") and str(first.dtype) in ("uint8", "torch.uint8"):
                batch = np.stack([np.asarray(t) for t in chunk])
                _, probs = self.tagger_model.forward_u8(batch, want="probs")
            else:   # float32 CHW tensors exactly as the reference's transform produces (tagging.py:241-243)
                batch = np.stack([np.asarray(t, dtype=np.float32) for t in chunk])
                _, probs = self.tagger_model.forward(batch)
            out.append(probs)
        return out

    def _select_lines(self, probs_list: List[np.ndarray], general_thresh: float, general_mcut_enabled: bool, character_thresh: float,
                      character_mcut_enabled: bool) -> List[str]:
        """tagging.py:185-227: MCut thresholds, selection and the tag strings of predict."""
        out: List[str] = []
        for probs in probs_list:
            # 512 label ids per image come back first (128 KB per batch of 64 instead of the 2.8 MB of full-width rows, whose read-back
            # into pageable memory was the largest transfer of a batch); a batch in which an image selects more is read again in full
            counts, ids, _ = self.selector.run(probs, general_thresh, general_mcut_enabled, character_thresh,
                                               character_mcut_enabled, row_cap=min(512, self.selector.num_classes))
            if int(counts.sum(axis=1).max()) > ids.shape[1]:
                counts, ids, _ = self.selector.run(probs, general_thresh, general_mcut_enabled, character_thresh, character_mcut_enabled)
            out.extend(format_lines(self.tag_names, counts, ids))
        return out

    # ---- multi-GPU form of tagging.py:276-359 --------------------------------------------------
    ROW_WIDTH = int(os.environ.get("HIPTS_ROW_WIDTH", 2 + 254))     # int32 {n_general, n_character, ids[254]}: 1 KiB per image on the wire
    #                                                                  (HIPTS_ROW_WIDTH: tests force every row through the second gather)

    def process_directory_sharded(self, dir_path: str, added_date: Optional[datetime.date], batch_size: int, dist, rank: int,
                                  world: int, workers: int = 0, shards: Optional[str] = None, synthetic: int = 0,
                                  synthetic_seed: int = 1234) -> None:
        """One process per GPU (launched by torch.distributed.run; hiptagsearch.dist.init_from_env ran first).  Rank r tags
        the contiguous block shard_range(n, r, world) of the corpus with its own copy of the model, keeps fixed-width
        tag rows on its device (hipts_tagsel_run_rows), ONE all-gather puts them in rank order == file order, rank 0 formats
        and appends the lines.  The file written is the file the single-process loop writes.

        Where a rank's images come from (round 3: every source is cut by rank, none is rank-0-only):
          files      its block of the directory listing, decoded on 8 threads (the reference's pool, tagging.py:52) or, with
                     `workers`, by its own pipeline.DecodePool;
          shards     its slice of the rows of the packed uint8 shards `tagging.py --write-shards` wrote (memory-mapped);
          synthetic  `synthetic` images of the benchmark corpus generated on its own GPU (hipts_synth_images_u8, keyed by
                     the global image index: BASELINE.json configs[3], no host I/O).
        A row that selected more labels than the 254 a wire row holds is completed by the rank that OWNS it -- the image is
        fetched again from the rank's source, selected without a cap -- and those few full lines travel in a second gather
        (round 2 redid them all on rank 0, i.e. serially)."""
        import torch
        from . import dist as hdist
        from .shard import gather_rows, padded_rows_per_rank, rows_to_lines, shard_range
        size = None
        shard_index = None                      # [(npy path, first global row, rows)]
        file_list = None
        if rank == 0:
            if synthetic:
                file_list = synthetic           # names are a function of the index: nothing to broadcast but the count
            elif shards:
                from . import pipeline
                shard_index, file_list = pipeline.index_shards(shards)
                print(f'{len(file_list)} images in {len(shard_index)} shards')
            else:
                file_list = self.list_files_recursive(dir_path)
                print(f'{len(file_list)} files found')
                if added_date is not None:
                    file_list = self.filter_files_by_date(file_list, added_date)
                    print(f'{len(file_list)} files found after {added_date}')
            if added_date is not None:
                if os.path.exists('tags-wd-tagger.txt'):
                    with open('tags-wd-tagger.txt', 'r', encoding='utf-8') as f, \
                            open('tags-wd-tagger.txt.bak', 'w', encoding='utf-8') as f_bak:
                        f_bak.write(f.read())
                else:
                    print('tags-wd-tagger.txt not found')
                    file_list = 1                                   # every rank leaves with the reference's exit code
        file_list, shard_index = hdist.broadcast_object((file_list, shard_index), dist)
        if file_list == 1:
            raise SystemExit(1)
        if synthetic:
            file_list = ["synthetic/%07d.png" % i for i in range(int(file_list))]
        if self.compat and file_list:
            file_list = file_list[:(max(0, (len(file_list) + 9) // 10 - 1)) * 10]     # the reference's dropped tail (tagging.py:309), batch 10
        self.load_model()
        size = self.cfg["image_size"]
        n = len(file_list)
        lo, hi = shard_range(n, rank, world)
        per = padded_rows_per_rank(n, world)
        dev = torch.device("cuda", self.device)
        W = self.ROW_WIDTH
        rows = torch.full((max(per, 1), W), -1, dtype=torch.int32, device=dev)        # sentinel rows: padding and failed loads
        bs = min(batch_size, self.max_batch)
        probs = torch.empty((bs, self.tagger_model.num_classes), dtype=torch.float32, device=dev)
        tmp = torch.empty((bs, W), dtype=torch.int32, device=dev)
        start = time.perf_counter()

        # ---- this rank's image source: batches of (block positions, uint8 images [k,S,S,3] on the host or on the device) + a fetch-one-again
        def synthetic_images(first: int, k: int):
            buf = torch.empty((k, size, size, 3), dtype=torch.uint8, device=dev)
            _lib.call("hipts_synth_images_u8", _lib.ptr(buf), ctypes.c_int64(first), ctypes.c_int64(k), size, ctypes.c_uint64(synthetic_seed),
                      self.device, _lib.current_stream_ptr())
            return buf
        pool = None
        if synthetic:
            def batches():
                for s0 in range(lo, hi, bs):
                    k = min(bs, hi - s0)
                    yield list(range(s0 - lo, s0 - lo + k)), synthetic_images(s0, k)
            fetch = lambda gi: synthetic_images(gi, 1).cpu().numpy()
        elif shards:
            from . import pipeline
            def batches():
                for first, arr in pipeline.iter_shard_rows(shard_index, lo, hi, bs):
                    yield list(range(first - lo, first - lo + arr.shape[0])), arr
            fetch = lambda gi: next(pipeline.iter_shard_rows(shard_index, gi, gi + 1, 1))[1]
        elif workers > 0 and not self.compat:
            from . import pipeline
            mine = file_list[lo:hi]
            where = {p: i for i, p in enumerate(mine)}
            pool = pipeline.DecodePool(workers, size, bs, pipeline.TAGGER, device_resize=self.gpu_resize, device=self.device, device_jpeg=self.gpu_jpeg)
            def batches():
                for kept, images in pool.batches(mine):
                    yield [where[p] for p in kept], images
            fetch = lambda gi: (lambda t: None if t is None else t[None])(self.gen_image_tensor(file_list[gi]))
        else:
            mine = file_list[lo:hi]
            def batches():
                chunks = [mine[i:i + bs] for i in range(0, len(mine), bs)]
                with concurrent.futures.ThreadPoolExecutor(max_workers=WORKER_NUM) as ex:
                    nxt = [ex.submit(self.gen_image_tensor, p, self.gpu_resize) for p in chunks[0]] if chunks else []
                    for bi in range(len(chunks)):
                        futs = nxt
                        nxt = [ex.submit(self.gen_image_tensor, p, self.gpu_resize) for p in chunks[bi + 1]] if bi + 1 < len(chunks) else []
                        imgs, pos = [], []
                        for j, fu in enumerate(futs):
                            t = fu.result()
                            if t is not None:
                                imgs.append(t)
                                pos.append(bi * bs + j)
                        if imgs:
                            yield pos, (torch.stack(imgs) if self.gpu_resize else np.stack(imgs))
            fetch = lambda gi: (lambda t: None if t is None else t[None])(self.gen_image_tensor(file_list[gi]))

        done = 0
        try:
            for pos, images in batches():
                k = len(pos)
                self.tagger_model.forward_u8(images, probs=probs[:k], want="probs")
                self.selector.run_device(probs[:k], tmp[:k], 0.3, True, 0.3, True)     # tagging.py:333
                rows[torch.as_tensor(pos, device=dev)] = tmp[:k]
                done += k
                if rank == 0 and done // PROGRESS_INTERVAL != (done - k) // PROGRESS_INTERVAL:
                    diff = time.perf_counter() - start
                    print(f'{done * world} files processed (all ranks)\n{diff:.2f} seconds elapsed\n', flush=True)
        finally:
            if pool is not None:
                pool.close()
        # rows that did not fit the wire width: completed HERE, by their owner
        mine_rows = rows[:max(hi - lo, 0)]
        over = torch.nonzero((mine_rows[:, 0] >= 0) & (mine_rows[:, 0] + mine_rows[:, 1] > W - 2)).flatten().cpu().numpy()
        wide = {}
        for i in over:
            img = fetch(lo + int(i))
            if img is not None:
                wide[lo + int(i)] = self.predict(img, 0.3, True, 0.3, True)[0]
        full = gather_rows(rows[:per] if per else rows[:0], n, dist)                   # [n, W] in file order, on every rank
        wide_all = [None] * world
        if dist is not None and dist.is_initialized() and world > 1:
            dist.gather_object(wide, wide_all if rank == 0 else None, dst=0)           # the second, variable-width gather: over-wide rows only
        else:
            wide_all = [wide]
        if rank == 0:
            redo = {}
            for w_ in wide_all:
                redo.update(w_ or {})
            full = full.cpu().numpy()
            ok = full[:, 0] >= 0
            with open('tags-wd-tagger.txt', 'a', encoding='utf-8') as f:              # tagging.py:293
                lines = rows_to_lines(full, self.tag_names, file_list)
                for i in range(n):
                    if ok[i]:
                        f.write((file_list[i] + ',' + redo[i] if i in redo else lines[i]) + '\n')
            print(f'{int(ok.sum())} of {n} files tagged by {world} ranks in {time.perf_counter() - start:.2f} seconds '
                  f'({len(redo)} rows wider than {W - 2} labels completed by their ranks)', flush=True)

    def write_to_file(self, csv_line: str) -> None:
        self.f.write(csv_line + '\n')

    def filter_files_by_date(self, file_list: List[str], added_date: datetime.date) -> List[str]:
        return [p for p in file_list if datetime.date.fromtimestamp(os.stat(p).st_ctime) >= added_date]   # tagging.py:266-274

    # ---- tagging.py:276-359
    def process_directory(self, dir_path: str, added_date: Optional[datetime.date] = None, batch_size: int = BATCH_SIZE,
                          workers: int = 0, shards: Optional[str] = None) -> None:
        """workers > 0: decode in that many processes (pipeline.DecodePool) instead of the reference's 8 threads;
        shards: tag the pre-decoded shards of pipeline.write_shards in that directory instead of walking dir_path."""
        file_list = [] if shards else self.list_files_recursive(dir_path)
        print(f'{len(file_list)} files found')
        if added_date is not None:
            file_list = self.filter_files_by_date(file_list, added_date)
            print(f'{len(file_list)} files found after {added_date}')
            if os.path.exists('tags-wd-tagger.txt'):
                with open('tags-wd-tagger.txt', 'r', encoding='utf-8') as f, \
                        open('tags-wd-tagger.txt.bak', 'w', encoding='utf-8') as f_bak:
                    f_bak.write(f.read())
            else:
                print('tags-wd-tagger.txt not found')
                raise SystemExit(1)
        self.f = open('tags-wd-tagger.txt', 'a', encoding='utf-8')
        self.load_model()
        start = time.perf_counter()
        done = 0
        last = 0
        if (workers > 0 or shards) and not self.compat:
            from . import pipeline
            size = self.cfg["image_size"]
            pool = None
            if shards:
                source = pipeline.iter_shards(shards, min(batch_size, self.max_batch))
            else:
                pool = pipeline.DecodePool(workers, size, min(batch_size, self.max_batch), pipeline.TAGGER, device_resize=self.gpu_resize,
                                           device=self.device, device_jpeg=self.gpu_jpeg)
                source = pool.batches(file_list)
            # predict() in two halves: while one thread selects, formats and writes the lines of batch k (the reference does that part on
            # the CPU too, tagging.py:185-232), this one is already in the forward of batch k + 1 -- one worker, so the file keeps its order
            post = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            # the formatting thread is pure Python: with the interpreter's default 5 ms switch interval the main thread, back from the
            # forward, waited that long for the GIL before it could launch the next one
            import sys
            old_switch = sys.getswitchinterval()
            sys.setswitchinterval(2e-4)

            def finish(kept, probs_list, ev=None):
                if ev is not None:
                    # device-resident probabilities of a forward that may still be running: the selection kernel is ordered behind it on
                    # this thread's own stream, nothing here holds up the main thread's next launch
                    import torch
                    with torch.cuda.device(self.device), torch.cuda.stream(post_stream[0]):
                        post_stream[0].wait_event(ev)
                        lines = self._select_lines(probs_list, 0.3, True, 0.3, True)
                else:
                    lines = self._select_lines(probs_list, 0.3, True, 0.3, True)
                for p, line in zip(kept, lines):
                    self.write_to_file(p + ',' + line)
                self.f.flush()
            pending = None
            post_stream = [None]
            dev_probs = []          # two device buffers: batch k + 1's forward is launched while batch k's probabilities are being read
            step = 0
            try:
                for kept, images in source:
                    on_device = hasattr(images, "is_cuda") and images.is_cuda and len(images) <= self.max_batch
                    if on_device:
                        # the forward is only LAUNCHED here (outputs stay on the device): the next iteration launches the next one at once,
                        # so the device never waits for this thread between batches
                        import torch
                        if post_stream[0] is None:
                            post_stream[0] = torch.cuda.Stream(device=self.device)
                            dev_probs = [torch.empty((self.max_batch, self.tagger_model.num_classes), dtype=torch.float32, device=images.device)
                                         for _ in range(2)]
                        pr = dev_probs[step & 1][:len(images)]       # last read by the finish of batch step - 2, which was awaited one iteration ago
                        self.tagger_model.forward_u8(images, probs=pr, want="probs")
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(self.device))
                        fut = post.submit(finish, list(kept), [pr], ev)
                    else:
                        fut = post.submit(finish, list(kept), self._forward_probs(images))
                    if pending is not None:
                        pending.result()
                    pending = fut
                    step += 1
                    done += len(kept)
                    if done - last >= PROGRESS_INTERVAL:
                        diff = time.perf_counter() - start
                        print(f'{done} files processed\n{diff:.2f} seconds elapsed\n{diff / done:.4f} seconds per file\n', flush=True)
                        last = done
                if pending is not None:
                    pending.result()
            finally:
                post.shutdown(wait=True)
                sys.setswitchinterval(old_switch)
                if pool is not None:
                    pool.close()
            self.f.close()
            return
        batches = [file_list[i:i + batch_size] for i in range(0, len(file_list), batch_size)]
        if self.compat and batches:
            batches = batches[:-1]      # the reference never consumes its last submitted batch (tagging.py:309)
        with concurrent.futures.ThreadPoolExecutor(max_workers=WORKER_NUM) as ex:
            nxt = [ex.submit(self.gen_image_tensor, p, self.gpu_resize) for p in batches[0]] if batches else []
            for bi, paths in enumerate(batches):
                futs = nxt
                nxt = [ex.submit(self.gen_image_tensor, p, self.gpu_resize) for p in batches[bi + 1]] if bi + 1 < len(batches) else []   # prefetch
                tensors, kept = [], []
                for p, fu in zip(paths, futs):
                    t = fu.result()
                    if t is not None:
                        tensors.append(t)
                        kept.append(p)
                if tensors:
                    for p, line in zip(kept, self.predict(tensors, 0.3, True, 0.3, True)):      # tagging.py:333
                        self.write_to_file(p + ',' + line)
                    self.f.flush()
                done += len(paths)
                if done - last >= PROGRESS_INTERVAL:
                    diff = time.perf_counter() - start
                    print(f'{done} files processed\n{diff:.2f} seconds elapsed\n{diff / done:.4f} seconds per file\n', flush=True)
                    last = done
        self.f.close()
