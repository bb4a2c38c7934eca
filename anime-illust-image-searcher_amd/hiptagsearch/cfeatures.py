"""Character-feature stage: host-side mirror of gen_cfeatures.py around the pieces that exist here.

  _normalize / _preprocess_image / gen_image_ndarray     gen_cfeatures.py:100-110,285-295  (pinned: g9)
  write_vecs_to_index                                     gen_cfeatures.py:307-315         (index.Similarity)
  character-oriented rerank                               webui.py:255-342, restated per BASELINE.json
                                                          configs[4] as cosine over the feature index
  CCIPEncoder                                             gen_cfeatures.py:112-118,133-159 (hipts_ccip_*)
The CCIP encoder's ONNX file (fetched from the HF hub, gen_cfeatures.py:112-118) cannot be obtained in
this environment; CCIPEncoder runs the CAFormer graph it contains (oracle/ccip.py) on the
device from a timm-layout state_dict and offers the onnxruntime call shape `run(['output'], {'input': x})`,
so it plugs in as `encoder`.  The metric model (gen_cfeatures.py:124-130) is likewise opaque; BASELINE.json
restates the rerank as cosine similarity, which is what runs on the device here:
difference := 1 - cos(feature, query).
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .index import Similarity

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)          # gen_cfeatures.py:100
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
BATCH_SIZE = 20                                          # gen_cfeatures.py:50
DEFAULT_THRESHOLD = 0.17847511429108218                  # docstring value, gen_cfeatures.py:195-196 (metric model's scale)
# The reference cuts at metrics.json's threshold / 1.5 (gen_cfeatures.py:298-299) on the output of its METRIC MODEL.  That
# constant is calibrated for that model's difference scale and does not transfer to 1 - cosine of the encoder features,
# which is what BASELINE.json configs[4] restates the rerank as.  The cosine cut is therefore its own parameter
# (CharacterFeatureIndex.cosine_diff_threshold); the default below only reuses the reference's number as a placeholder:
# calibrate_threshold() derives the cut from labelled features the way the reference's constant was derived (best F1 over
# same- / different-character pairs; tests/test_gpu_flows.py::test_ccip_metric_and_calibrated_threshold).
DEFAULT_COSINE_DIFF_THRESHOLD = DEFAULT_THRESHOLD / 1.5
INDEX_PREFIX = 'charactor-featues-idx'                   # gen_cfeatures.py:311 (the reference's spelling)


def get_current_cfeature_number(dirpath: str = '.') -> int:
    """gen_cfeatures.py:317-335 / webui.py:272-277: the highest revision N among `charactor-featues-idx` (N = 0) and
    `charactor-featues-idxN`; ValueError (max of an empty list) when there is none, like the reference."""
    import os
    import re
    pattern = re.compile(r'^charactor-featues-idx(\d*)$')
    numbers = []
    for file in os.listdir(dirpath):
        m = pattern.match(file)
        if m:
            numbers.append(int(m.group(1)) if m.group(1) else 0)
    return max(numbers)


def revision_name(number: int) -> str:
    return INDEX_PREFIX if number == 0 else INDEX_PREFIX + str(number)


def backup_index_files(dirpath: str = '.') -> str:
    """gen_cfeatures.py:346-352: copy every `charactor-featues-idx*` file into a directory named by the current time."""
    import datetime
    import os
    import shutil
    from pathlib import Path
    backup_dir = os.path.join(dirpath, datetime.datetime.now().strftime('%Y%m%d_%H%M%S'))
    os.makedirs(backup_dir, exist_ok=True)
    for file in Path(dirpath).glob('charactor-featues-idx*'):
        if file.is_file():
            shutil.copy2(file, Path(backup_dir) / file.name)
            print(f'Backed up {file} to {backup_dir}')
    return backup_dir


def _normalize(data, mean=CLIP_MEAN, std=CLIP_STD):
    mean, std = np.asarray(mean), np.asarray(std)
    return (data - mean[:, None, None]) / std[:, None, None]           # float64, like the reference


def _preprocess_image(image, size: int = 384):
    from PIL import Image
    image = image.resize((size, size), resample=Image.BILINEAR)
    data = np.array(image).transpose(2, 0, 1).astype(np.float32) / 255.0
    return _normalize(data)


def gen_image_ndarray(file_path: str, size: int = 384, gpu_resize: bool = False, device: int = 0):
    """gen_cfeatures.py:285-295 (imgutils.load_images(mode='RGB') = alpha composited on white).
    gpu_resize: the host only decodes; the bilinear resize of gen_cfeatures.py:101 runs on the device (hipts_resize_u8, Pillow's resample
    bit for bit) and a uint8 [size,size,3] CUDA tensor is returned for CCIPEncoder.forward_u8, which applies :102-110 in its first kernel."""
    from PIL import Image
    try:
        img = Image.open(file_path)
        img.load()
        if img.mode in ("RGBA", "LA"):
            bg = Image.new("RGB", img.size, (255, 255, 255))
            bg.paste(img, mask=img.split()[-1])
            img = bg
        else:
            img = img.convert("RGB")
        if gpu_resize:
            from .tagger import device_resize_u8
            return device_resize_u8(np.asarray(img, dtype=np.uint8), size, size, 2, device)
        return _preprocess_image(img, size)
    except Exception as e:
        print('%s: %s' % (type(e), str(e)))
        return None


class CCIPEncoder:
    """Device-resident CCIP feature encoder (CAFormer forward, libhip_tagsearch `hipts_ccip_*`).

    Stands where gen_cfeatures.py:112-118 opens the onnxruntime session: `run(['output'], {'input': x})`
    (gen_cfeatures.py:158) and plain calls `encoder(x)` both take float32 [B,3,S,S] normalised images and
    return float32 [B, dims[3]] features.  `weights`: timm MetaFormer state_dict keys -> float32 arrays."""

    def __init__(self, cfg: Dict, weights: Dict[str, np.ndarray], max_batch: int = BATCH_SIZE, device: int = 0):
        import ctypes
        from . import _lib
        self._lib = _lib
        self.cfg = dict(cfg)
        self.max_batch = max_batch
        self.out_dim = int(cfg["dims"][3])
        c = _lib.CcipConfig(cfg["image_size"], (ctypes.c_int32 * 4)(*cfg["dims"]), (ctypes.c_int32 * 4)(*cfg["depths"]),
                            cfg.get("head_dim", 32), cfg.get("attn_from_stage", 2), cfg.get("ln_eps", 1e-6), max_batch,
                            cfg.get("operand_f16", 1))     # IEEE-half operands by default, as ViTTagger / EvaTagger
        self._h = ctypes.c_void_p()
        _lib.call("hipts_ccip_create", ctypes.byref(c), device, ctypes.byref(self._h))
        for key, val in weights.items():
            arr = np.ascontiguousarray(val, dtype=np.float32)
            _lib.call("hipts_ccip_set_tensor", self._h, key.encode(), _lib.ptr(arr), ctypes.c_int64(arr.size))

    @classmethod
    def from_safetensors(cls, path: str, cfg: Dict, **kw) -> "CCIPEncoder":
        from safetensors.numpy import load_file
        return cls(cfg, {k: v.astype(np.float32) for k, v in load_file(path).items()}, **kw)

    def flops_per_image(self) -> float:
        import ctypes
        f = ctypes.c_double()
        self._lib.call("hipts_ccip_flops_per_image", self._h, ctypes.byref(f))
        return f.value

    def _forward(self, fn: str, x, out):
        _lib = self._lib
        B = int(x.shape[0])
        if out is None:
            out = np.empty((B, self.out_dim), dtype=np.float32)
        for s in range(0, B, self.max_batch):
            xs, os_ = x[s:s + self.max_batch], out[s:s + self.max_batch]
            _lib.call(fn, self._h, _lib.ptr(xs), _lib.memspace_of(xs), int(xs.shape[0]), _lib.ptr(os_), _lib.memspace_of(os_),
                      _lib.current_stream_ptr())
        return out

    def __call__(self, x, out=None):
        """x: float32 [B,3,S,S] (numpy or torch, host or device) -> float32 [B, out_dim]."""
        if isinstance(x, np.ndarray):
            x = np.ascontiguousarray(x, dtype=np.float32)
        return self._forward("hipts_ccip_forward_f32", x, out)

    def forward_u8(self, images, out=None):
        """images: uint8 [B,S,S,3] RGB, already S x S: /255 and the CLIP normalisation run on the device."""
        if isinstance(images, np.ndarray):
            images = np.ascontiguousarray(images, dtype=np.uint8)
        return self._forward("hipts_ccip_forward_u8", images, out)

    def run(self, output_names, feed):                                              # onnxruntime call shape, :158
        assert list(output_names) == ["output"] and list(feed.keys()) == ["input"]
        return [self(np.asarray(feed["input"], dtype=np.float32))]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.call("hipts_ccip_destroy", self._h)
                self._h = None
        except Exception:
            pass


class CharacterFeatureIndex:
    """charactor-featues-idx (+ .csv of paths): feature rows unit-normalised on add (gensim's
    behaviour for sparse documents, gen_cfeatures.py:310-314), device resident."""

    def __init__(self, encoder: Callable[[np.ndarray], np.ndarray], device: int = 0, prefix: str = "charactor-featues-idx"):
        self.encoder = encoder
        self.image_size = int(getattr(encoder, "cfg", {}).get("image_size", 384)) if hasattr(encoder, "cfg") else 384   # gen_cfeatures.py:101
        self.index = Similarity(prefix, None, 768, device)
        self.paths: List[str] = []
        self.threshold = DEFAULT_THRESHOLD                       # the reference's metric-model threshold (kept for completeness)
        self.cosine_diff_threshold = DEFAULT_COSINE_DIFF_THRESHOLD  # the cut actually applied to 1 - cosine (see the note above)

    @classmethod
    def load_latest(cls, encoder, device: int = 0, dirpath: str = '.') -> "CharacterFeatureIndex":
        """webui.py:265-277: the paths csv and the latest index revision.  The csv is appended per batch while the index is
        saved at the end of a run (gen_cfeatures.py:419,459): after a crash the csv is longer than the index.  Rows and paths
        must stay aligned, so surplus csv lines are dropped here (with a warning); a csv SHORTER than the index is an error."""
        import os
        self = cls(encoder, device)
        name = revision_name(get_current_cfeature_number(dirpath))
        self.index = Similarity.load(os.path.join(dirpath, name), device)
        csv = os.path.join(dirpath, INDEX_PREFIX + '.csv')
        paths = [l.rstrip('\n') for l in open(csv, encoding='utf-8')] if os.path.exists(csv) else []
        n = len(self.index)
        if len(paths) < n:
            raise ValueError('%s lists %d paths but %s holds %d rows' % (csv, len(paths), name, n))
        if len(paths) > n:
            print('warning: %s lists %d paths, %s holds %d rows: ignoring the last %d paths (an interrupted run?)'
                  % (csv, len(paths), name, n, len(paths) - n))
        self.paths = paths[:n]
        return self

    def ccip_batch_extract_features(self, images: Sequence[np.ndarray]) -> np.ndarray:      # :133-159
        data = np.stack(images).astype(np.float32)
        return np.asarray(self.encoder(data), dtype=np.float32)

    def add_files(self, paths: Sequence[str]):
        for s in range(0, len(paths), BATCH_SIZE):
            chunk = [(p, gen_image_ndarray(p)) for p in paths[s:s + BATCH_SIZE]]
            chunk = [(p, a) for p, a in chunk if a is not None]
            if not chunk:
                continue
            feats = self.ccip_batch_extract_features([a for _, a in chunk])
            self.add_features([p for p, _ in chunk], feats)

    def add_features(self, paths: Sequence[str], feats: np.ndarray):
        feats = np.asarray(feats, dtype=np.float32)
        n = np.sqrt((feats.astype(np.float32) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
        self.index.add_matrix(np.where(n > 0, feats / np.where(n > 0, n, 1), feats).astype(np.float32))
        self.paths.extend(paths)

    def differences(self, query_feature: np.ndarray) -> np.ndarray:
        """1 - cosine(row, query) for every indexed feature (rows are unit vectors)."""
        q = np.asarray(query_feature, dtype=np.float32)
        nq = np.float32(np.sqrt(np.sum(q * q)))
        if nq > 0:
            q = q / nq
        return np.float32(1.0) - self.index.query(q)[0]


def ccip_batch_differences(features: np.ndarray, device: int = 0) -> np.ndarray:
    """gen_cfeatures.py:257-274: the matrix of pairwise differences of a list of feature vectors -- float32 [n, n].  The reference runs its
    metric model here; this is the cosine restatement (hipts_ccip_metric, kind 0): 1 - cosine of the unit-normalised rows."""
    from . import _lib
    f = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.float32) for x in features]), dtype=np.float32)
    out = np.empty((f.shape[0], f.shape[0]), dtype=np.float32)
    _lib.call("hipts_ccip_metric", _lib.ptr(f), _lib.HOST, int(f.shape[0]), int(f.shape[1]), 0, _lib.ptr(out), _lib.HOST, device, _lib.current_stream_ptr())
    return out


def ccip_difference(x: np.ndarray, y: np.ndarray, device: int = 0) -> float:
    """gen_cfeatures.py:212-244."""
    return float(ccip_batch_differences([x, y], device)[0, 1])


def calibrate_threshold(features: np.ndarray, labels: Sequence[int], device: int = 0) -> Tuple[float, float]:
    """The cut for `difference < threshold` = "same character", chosen the way the reference's constant was (the metric model's
    metrics.json carries the threshold with the best F1 on labelled pairs; gen_cfeatures.py:186-196 only reads it): all pairwise
    differences of labelled features, every midpoint between consecutive distinct values as a candidate, the one with the highest
    F1 over pairs.  Returns (threshold, its F1).  webui.py:298-299 then cuts at threshold / 1.5; CharacterFeatureIndex keeps that
    ratio (set `cindex.cosine_diff_threshold = calibrate_threshold(...)[0] / 1.5`)."""
    labels = np.asarray(labels)
    d = ccip_batch_differences(features, device).astype(np.float64)
    iu = np.triu_indices(len(labels), 1)
    diffs, same = d[iu], (labels[:, None] == labels[None, :])[iu]
    order = np.argsort(diffs, kind="stable")
    diffs, same = diffs[order], same[order]
    tp = np.cumsum(same)                                   # pairs called "same" when the cut sits right after position i
    fp = np.cumsum(~same)
    fn = same.sum() - tp
    f1 = 2 * tp / np.maximum(2 * tp + fp + fn, 1)
    valid = np.ones(len(diffs), dtype=bool)
    valid[:-1] = diffs[1:] > diffs[:-1]                    # a cut is only possible between distinct values
    i = int(np.argmax(np.where(valid, f1, -1.0)))
    hi = diffs[i + 1] if i + 1 < len(diffs) else diffs[i] + 1e-3
    return float((diffs[i] + hi) / 2), float(f1[i])


def cfeatures_rerank(final_scores_top10: Sequence[Tuple[int, float]], top10_features: Sequence[np.ndarray],
                     cindex: CharacterFeatureIndex, file_tag_index: Dict[str, Dict[str, bool]],
                     filepath_docid: Dict[str, int], required_tags: Sequence[str], exclude_tags: Sequence[str],
                     threshold: Optional[float] = None) -> List[Tuple[int, float]]:
    """webui.py:283-335: mean feature of the top-10 images, difference to every indexed image,
    keep those below the threshold that carry all required and no excluded tags, best first; the
    original top-10 lead the list (returned without topn cut or gap filter, like the reference)."""
    threshold = cindex.cosine_diff_threshold if threshold is None else threshold
    mean = np.average(np.stack(top10_features), axis=0)                               # :303
    diffs = cindex.differences(mean)                                                  # :306-309 on the device
    out: List[Tuple[int, float]] = []
    for idx, path in enumerate(cindex.paths):
        if path not in file_tag_index:
            continue                                                                  # :312-323 (not found: ignored)
        tags = file_tag_index[path]
        if diffs[idx] < threshold and all(t in tags for t in required_tags) and all(t not in tags for t in exclude_tags):
            out.append((filepath_docid[path], float(np.float32(1.0) - diffs[idx])))   # :325-328
    out = sorted(out, key=lambda it: -it[1])                                          # :330
    return list(final_scores_top10) + out                                            # :332-335
