"""Seeded synthetic inputs of the shapes BASELINE.json names.

No dataset, checkpoint or label file exists in the build or GPU environment, so benchmarks and
parity tests run on these.  Pure numpy; shared by tests, bench.py and the CLIs' --synthetic mode.
"""
from typing import Dict, List, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------
# ViT tagger (config[1]: wd-tagger ViT-B/16 448px bf16)
# ---------------------------------------------------------------------------------------------
VIT_B16_448 = dict(image_size=448, patch=16, dim=768, depth=12, heads=12, mlp_dim=3072, num_classes=10861,
                   ln_eps=1e-6, gelu_tanh=1, pool_then_norm=0)
VIT_TINY = dict(image_size=64, patch=16, dim=128, depth=2, heads=2, mlp_dim=256, num_classes=200,
                ln_eps=1e-6, gelu_tanh=1, pool_then_norm=0)
# The model the reference really loads (tagging.py:45): EVA02-L/14 @448; hidden = int(1024 * 8 / 3)
EVA02_L14_448 = dict(image_size=448, patch=14, dim=1024, depth=24, heads=16, mlp_hidden=2730, num_classes=10861, ln_eps=1e-6,
                     rope_ref_grid=16)
EVA02_TINY = dict(image_size=56, patch=14, dim=128, depth=2, heads=2, mlp_hidden=340, num_classes=200, ln_eps=1e-6,
                  rope_ref_grid=16)
# CCIP feature encoder (gen_cfeatures.py): CAFormer-B36 widths at 384 px, 768-d feature
CCIP_B36_384 = dict(image_size=384, dims=(128, 256, 512, 768), depths=(3, 12, 18, 3), head_dim=32, ln_eps=1e-6)
CCIP_TINY = dict(image_size=64, dims=(64, 64, 128, 128), depths=(1, 1, 2, 1), head_dim=32, ln_eps=1e-6)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16 -> float32 (values exactly representable in bf16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape, dtype=np.float32)
    np.clip(x, -2.0, 2.0, out=x)
    x *= np.float32(std)
    return x


def _trained_like_qk(rng, w: Dict[str, np.ndarray], key_q: str, key_k: str, row0_q: int, row0_k: int, li: int, heads: int, hd: int,
                      bias_q: str = None, bias_k: str = None) -> None:
    """Scale the q and k projection rows of every head by powers of two (exact in bf16) so that attention scores look like a
    trained model's instead of a random init's (score std 0.33: a near-uniform softmax): ordinary heads x4 / x2 (natural-log score
    std ~2.6, row maxima ~9), and in every fourth layer head 0 x8 / x4 (std ~10, maxima 30-40: a heavy tail that exceeds an
    IEEE-half 2^S outright and, for a good part of its rows, the fixed-reference window of attn.hip, so the classic fallback runs)."""
    for h in range(heads):
        sq, sk = (8.0, 4.0) if (h == 0 and li % 4 == 1) else (4.0, 2.0)
        w[key_q][row0_q + h * hd:row0_q + (h + 1) * hd] *= np.float32(sq)
        w[key_k][row0_k + h * hd:row0_k + (h + 1) * hd] *= np.float32(sk)
        if bias_q is not None:
            w[bias_q][row0_q + h * hd:row0_q + (h + 1) * hd] *= np.float32(sq)
        if bias_k is not None:
            w[bias_k][row0_k + h * hd:row0_k + (h + 1) * hd] *= np.float32(sk)


def _trained_like_head(rng, w: Dict[str, np.ndarray], num_classes: int) -> None:
    """A tagger head whose outputs look like a trained one's: weights x4 (exact), biases bimodal -- about 28 general and 3
    character labels (a fixed random set per checkpoint) sit at +5, everything else at -10, one rating at +3.  Probabilities are
    then sparse: the MCut gap falls between the two clusters and 10-40 labels are selected per image, logit rms ~10."""
    _, cat = label_table(num_classes)
    w["head.weight"] *= np.float32(4.0)
    b = (-10.0 + rng.standard_normal(num_classes)).astype(np.float32)
    gen = np.flatnonzero(cat == 0)
    ch = np.flatnonzero(cat == 4)
    n_gen = min(28, max(1, len(gen) // 8))
    n_ch = min(3, max(1, len(ch) // 8)) if len(ch) else 0
    hi = list(rng.choice(gen, size=n_gen, replace=False)) + (list(rng.choice(ch, size=n_ch, replace=False)) if n_ch else [])
    b[hi] = (5.0 + 0.5 * rng.standard_normal(len(hi))).astype(np.float32)
    rating = np.flatnonzero(cat == 9)
    if len(rating):
        b[rating] = -3.0
        b[rating[0]] = 3.0
    w["head.bias"] = b


def vit_weights(cfg: Dict, seed: int = 0, bf16_matrices: bool = True, trained_like: bool = False) -> Dict[str, np.ndarray]:
    """Random-init checkpoint with timm state_dict keys.  W ~ N(0,0.02^2) truncated at 2 sigma,
    biases ~ N(0,0.02^2), LN gamma ~ U(0.5,1.5), LN beta ~ N(0,0.02^2), pos_embed ~ N(0,0.02^2).
    With bf16_matrices the GEMM weight matrices are bf16-representable (a bf16 checkpoint, as
    config[1] names), so the float32 oracle and the bf16 MFMA path see identical weights.
    trained_like: the same tensors with the q / k projections and the head rescaled (by powers of two) and the head bias made
    bimodal, see _trained_like_qk / _trained_like_head: peaked attention with a heavy tail, logit rms ~10, sparse probabilities,
    tens of labels selected per image -- the regime a trained wd-tagger runs in, which a plain random init (logit rms 0.35, every
    probability near 0.5, ~3000 labels "selected") never visits."""
    rng = np.random.default_rng(seed)
    D, P, M, C = cfg["dim"], cfg["patch"], cfg["mlp_dim"], cfg["num_classes"]
    N = (cfg["image_size"] // P) ** 2
    rb = round_to_bf16 if bf16_matrices else (lambda a: a)
    w: Dict[str, np.ndarray] = {}
    w["patch_embed.proj.weight"] = rb(_trunc_normal(rng, (D, 3, P, P), 0.02))
    w["patch_embed.proj.bias"] = _trunc_normal(rng, (D,), 0.02)
    w["pos_embed"] = _trunc_normal(rng, (1, N, D), 0.02)
    for i in range(cfg["depth"]):
        p = "blocks.%d." % i
        for ln in ("norm1", "norm2"):
            w[p + ln + ".weight"] = rng.uniform(0.5, 1.5, D).astype(np.float32)
            w[p + ln + ".bias"] = _trunc_normal(rng, (D,), 0.02)
        w[p + "attn.qkv.weight"] = rb(_trunc_normal(rng, (3 * D, D), 0.02))
        w[p + "attn.qkv.bias"] = _trunc_normal(rng, (3 * D,), 0.02)
        w[p + "attn.proj.weight"] = rb(_trunc_normal(rng, (D, D), 0.02))
        w[p + "attn.proj.bias"] = _trunc_normal(rng, (D,), 0.02)
        w[p + "mlp.fc1.weight"] = rb(_trunc_normal(rng, (M, D), 0.02))
        w[p + "mlp.fc1.bias"] = _trunc_normal(rng, (M,), 0.02)
        w[p + "mlp.fc2.weight"] = rb(_trunc_normal(rng, (D, M), 0.02))
        w[p + "mlp.fc2.bias"] = _trunc_normal(rng, (D,), 0.02)
    w["norm.weight"] = rng.uniform(0.5, 1.5, D).astype(np.float32)
    w["norm.bias"] = _trunc_normal(rng, (D,), 0.02)
    w["head.weight"] = rb(_trunc_normal(rng, (C, D), 0.02))
    w["head.bias"] = _trunc_normal(rng, (C,), 0.02)
    if trained_like:
        rng2 = np.random.default_rng(seed + 7919)
        H = cfg["heads"]
        for i in range(cfg["depth"]):
            p = "blocks.%d." % i
            _trained_like_qk(rng2, w, p + "attn.qkv.weight", p + "attn.qkv.weight", 0, D, i, H, D // H, p + "attn.qkv.bias", p + "attn.qkv.bias")
        _trained_like_head(rng2, w, C)
    return w


def eva_weights(cfg: Dict, seed: int = 0, bf16_matrices: bool = True, trained_like: bool = False) -> Dict[str, np.ndarray]:
    """Random-init EVA02 checkpoint with timm `Eva` state_dict keys (q/k/v separate, SwiGLU with inner norm),
    same distributions as vit_weights (and the same trained_like variant)."""
    rng = np.random.default_rng(seed)
    D, P, Hd, C = cfg["dim"], cfg["patch"], cfg["mlp_hidden"], cfg["num_classes"]
    N = (cfg["image_size"] // P) ** 2
    rb = round_to_bf16 if bf16_matrices else (lambda a: a)
    w: Dict[str, np.ndarray] = {}
    w["patch_embed.proj.weight"] = rb(_trunc_normal(rng, (D, 3, P, P), 0.02))
    w["patch_embed.proj.bias"] = _trunc_normal(rng, (D,), 0.02)
    w["cls_token"] = _trunc_normal(rng, (1, 1, D), 0.02)
    w["pos_embed"] = _trunc_normal(rng, (1, N + 1, D), 0.02)
    for i in range(cfg["depth"]):
        p = "blocks.%d." % i
        for ln in ("norm1", "norm2"):
            w[p + ln + ".weight"] = rng.uniform(0.5, 1.5, D).astype(np.float32)
            w[p + ln + ".bias"] = _trunc_normal(rng, (D,), 0.02)
        for nm in ("q_proj", "k_proj", "v_proj"):
            w[p + "attn." + nm + ".weight"] = rb(_trunc_normal(rng, (D, D), 0.02))
        w[p + "attn.q_proj.bias"] = _trunc_normal(rng, (D,), 0.02)
        w[p + "attn.v_proj.bias"] = _trunc_normal(rng, (D,), 0.02)
        w[p + "attn.proj.weight"] = rb(_trunc_normal(rng, (D, D), 0.02))
        w[p + "attn.proj.bias"] = _trunc_normal(rng, (D,), 0.02)
        w[p + "mlp.fc1_g.weight"] = rb(_trunc_normal(rng, (Hd, D), 0.02))
        w[p + "mlp.fc1_g.bias"] = _trunc_normal(rng, (Hd,), 0.02)
        w[p + "mlp.fc1_x.weight"] = rb(_trunc_normal(rng, (Hd, D), 0.02))
        w[p + "mlp.fc1_x.bias"] = _trunc_normal(rng, (Hd,), 0.02)
        w[p + "mlp.norm.weight"] = rng.uniform(0.5, 1.5, Hd).astype(np.float32)
        w[p + "mlp.norm.bias"] = _trunc_normal(rng, (Hd,), 0.02)
        w[p + "mlp.fc2.weight"] = rb(_trunc_normal(rng, (D, Hd), 0.02))
        w[p + "mlp.fc2.bias"] = _trunc_normal(rng, (D,), 0.02)
    w["fc_norm.weight"] = rng.uniform(0.5, 1.5, D).astype(np.float32)
    w["fc_norm.bias"] = _trunc_normal(rng, (D,), 0.02)
    w["head.weight"] = rb(_trunc_normal(rng, (C, D), 0.02))
    w["head.bias"] = _trunc_normal(rng, (C,), 0.02)
    if trained_like:
        rng2 = np.random.default_rng(seed + 7919)
        H = cfg["heads"]
        for i in range(cfg["depth"]):
            p = "blocks.%d." % i
            _trained_like_qk(rng2, w, p + "attn.q_proj.weight", p + "attn.k_proj.weight", 0, 0, i, H, D // H, p + "attn.q_proj.bias", None)
        _trained_like_head(rng2, w, C)
    return w


def ccip_weights(cfg: Dict, seed: int = 46, bf16_matrices: bool = True) -> Dict[str, np.ndarray]:
    """Random-init CAFormer checkpoint with timm `MetaFormer` state_dict keys.  Linear / 1x1 weights
    ~ N(0, 1/fan_in) truncated at 2 sigma (activations stay O(1) through the StarReLU blocks), StarReLU
    (scale, bias) near the variance-preserving (0.8944, -0.4472), depthwise 7x7 ~ N(0, 1/49), LN gamma
    ~ U(0.5,1.5), res_scale ~ U(0.8,1.2).  Matrices are bf16-representable with bf16_matrices (see vit_weights)."""
    rng = np.random.default_rng(seed)
    dims, depths = cfg["dims"], cfg["depths"]
    rb = round_to_bf16 if bf16_matrices else (lambda a: a)
    w: Dict[str, np.ndarray] = {}

    def lin(out_f, in_f, shape=None):
        m = rb(_trunc_normal(rng, (out_f, in_f), 1.0 / np.sqrt(in_f)))
        return m.reshape(shape) if shape else m

    w["stem.conv.weight"] = rb(_trunc_normal(rng, (dims[0], 3, 7, 7), 1.0 / np.sqrt(147.0)))
    w["stem.conv.bias"] = _trunc_normal(rng, (dims[0],), 0.02)
    w["stem.norm.weight"] = rng.uniform(0.5, 1.5, dims[0]).astype(np.float32)
    for s in range(4):
        C = dims[s]
        if s > 0:
            Cp = dims[s - 1]
            w["stages.%d.downsample.norm.weight" % s] = rng.uniform(0.5, 1.5, Cp).astype(np.float32)
            w["stages.%d.downsample.conv.weight" % s] = rb(_trunc_normal(rng, (C, Cp, 3, 3), 1.0 / np.sqrt(9.0 * Cp)))
            w["stages.%d.downsample.conv.bias" % s] = _trunc_normal(rng, (C,), 0.02)
        for i in range(depths[s]):
            p = "stages.%d.blocks.%d." % (s, i)
            w[p + "norm1.weight"] = rng.uniform(0.5, 1.5, C).astype(np.float32)
            w[p + "norm2.weight"] = rng.uniform(0.5, 1.5, C).astype(np.float32)
            if s < 2:
                w[p + "token_mixer.pwconv1.weight"] = lin(2 * C, C, (2 * C, C, 1, 1))
                w[p + "token_mixer.act1.scale"] = np.float32([0.8944 * rng.uniform(0.9, 1.1)])
                w[p + "token_mixer.act1.bias"] = np.float32([-0.4472 + rng.normal(0, 0.02)])
                w[p + "token_mixer.dwconv.weight"] = _trunc_normal(rng, (2 * C, 1, 7, 7), 1.0 / 7.0)
                w[p + "token_mixer.pwconv2.weight"] = lin(C, 2 * C, (C, 2 * C, 1, 1))
            else:
                w[p + "token_mixer.qkv.weight"] = lin(3 * C, C)
                w[p + "token_mixer.proj.weight"] = lin(C, C)
                w[p + "res_scale1.scale"] = rng.uniform(0.8, 1.2, C).astype(np.float32)
                w[p + "res_scale2.scale"] = rng.uniform(0.8, 1.2, C).astype(np.float32)
            shape4 = s < 2
            w[p + "mlp.fc1.weight"] = lin(4 * C, C, (4 * C, C, 1, 1) if shape4 else None)
            w[p + "mlp.act.scale"] = np.float32([0.8944 * rng.uniform(0.9, 1.1)])
            w[p + "mlp.act.bias"] = np.float32([-0.4472 + rng.normal(0, 0.02)])
            w[p + "mlp.fc2.weight"] = lin(C, 4 * C, (C, 4 * C, 1, 1) if shape4 else None)
    w["head.norm.weight"] = rng.uniform(0.5, 1.5, dims[3]).astype(np.float32)
    w["head.norm.bias"] = _trunc_normal(rng, (dims[3],), 0.02)
    return w


def images_u8(n: int, size: int = 448, seed: int = 1234) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, (n, size, size, 3), dtype=np.uint8)


STRUCTURED_KINDS = ("flat", "posterised", "gradient", "lineart", "halfflat", "blocks")


def structured_images_u8(size: int = 448, seed: int = 77, kinds=STRUCTURED_KINDS) -> np.ndarray:
    """One uint8 [len(kinds), size, size, 3] image per kind -- what illustrations are made of, and what uniform noise never shows a
    16-bit datapath: large regions in which every patch is IDENTICAL, so that an operand's rounding error is the same on every token
    and does not average out in the mean pool (the case that takes bf16 operands to ~4e-3 logit error).
      flat        one colour                         posterised  noise quantised to 4 levels per channel
      gradient    smooth diagonal colour ramp        lineart     black strokes, 1-3 px wide, on white
      halfflat    left half one colour, right noise  blocks      64 x 64 px tiles of flat colour (cel shading)"""
    rng = np.random.default_rng(seed)
    out = np.zeros((len(kinds), size, size, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    for i, kind in enumerate(kinds):
        if kind == "flat":
            out[i] = rng.integers(0, 256, 3, dtype=np.uint8)
        elif kind == "posterised":
            out[i] = (rng.integers(0, 256, (size, size, 3), dtype=np.uint8) // 64) * 64
        elif kind == "gradient":
            c0, c1 = rng.integers(0, 256, 3).astype(np.float64), rng.integers(0, 256, 3).astype(np.float64)
            t = ((xx + yy) / (2.0 * (size - 1)))[..., None]
            out[i] = np.rint(c0 * (1 - t) + c1 * t).astype(np.uint8)
        elif kind == "lineart":
            img = np.full((size, size, 3), 255, dtype=np.uint8)
            for _ in range(24):
                x0, y0, x1, y1 = rng.integers(0, size, 4)
                wd = int(rng.integers(1, 4))
                n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
                xs = np.rint(np.linspace(x0, x1, n)).astype(int)
                ys = np.rint(np.linspace(y0, y1, n)).astype(int)
                for d in range(wd):
                    img[np.clip(ys + d, 0, size - 1), xs] = 0
                    img[ys, np.clip(xs + d, 0, size - 1)] = 0
            out[i] = img
        elif kind == "halfflat":
            out[i] = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
            out[i, :, :size // 2] = rng.integers(0, 256, 3, dtype=np.uint8)
        elif kind == "blocks":
            bs = max(8, size // 7)
            nb = (size + bs - 1) // bs
            cols = rng.integers(0, 256, (nb, nb, 3), dtype=np.uint8)
            out[i] = np.repeat(np.repeat(cols, bs, axis=0), bs, axis=1)[:size, :size]
        else:
            raise ValueError(kind)
    return out


def label_table(num_classes: int = 10861) -> Tuple[List[str], np.ndarray]:
    """Synthetic selected_tags.csv: first 4 rating (9), then general (0), last ~24 % character (4)."""
    n_char = int(round(num_classes * 2600 / 10861))
    cat = np.zeros(num_classes, dtype=np.int32)
    cat[:min(4, num_classes)] = 9
    cat[num_classes - n_char:] = 4
    names = ["tag_%05d" % i for i in range(num_classes)]
    return names, cat


# ---------------------------------------------------------------------------------------------
# Tag documents / queries (config[2]: 100k random tag-docs)
# ---------------------------------------------------------------------------------------------
def tag_corpus(D: int = 100_000, V: int = 10_000, seed: int = 42, mean_len: int = 20) -> Tuple[np.ndarray, np.ndarray]:
    """CSR (doc_ptr int64[D+1], term_ids int32[nnz]) of token ids per document: Zipf(1.1) tag
    popularity, clip(Poisson(mean_len),3,60) distinct tags, 1 % of documents repeat a tag (tf up
    to 3).  Token id == popularity rank (id 0 most frequent)."""
    rng = np.random.default_rng(seed)
    p = np.arange(1, V + 1, dtype=np.float64) ** -1.1
    cdf = np.cumsum(p / p.sum())
    lens = np.clip(rng.poisson(mean_len, D), 3, 60)
    ptr = np.zeros(D + 1, dtype=np.int64)
    out: List[np.ndarray] = []
    for d in range(D):
        n = int(min(lens[d], V))
        draws = np.searchsorted(cdf, rng.random(4 * n + 8)).astype(np.int32)
        np.minimum(draws, V - 1, out=draws)
        _, first = np.unique(draws, return_index=True)
        ids = draws[np.sort(first)][:n]
        if len(ids) < 3:   # cannot happen for V >= 3 in practice; keep the >= 3 tags invariant anyway
            ids = np.unique(np.concatenate([ids, np.arange(3, dtype=np.int32)]))[:3]
        if rng.random() < 0.01:
            ids = np.concatenate([ids, np.repeat(ids[:1], int(rng.integers(1, 3)))])
        out.append(ids.astype(np.int32))
        ptr[d + 1] = ptr[d] + len(ids)
    return ptr, np.concatenate(out)


def vocab_tokens(V: int) -> List[str]:
    return ["t%05d" % i for i in range(V)]


def queries(nq: int = 1000, V: int = 10_000, seed: int = 43, head: int = 2000) -> List[List[Tuple[int, int]]]:
    """nq queries of 1-4 distinct terms drawn from the `head` most popular ids; each term plain
    (weight 1) 70 %, required (+w) 15 %, excluded (-w) 15 %, w in {1,2,3}.  Returned as
    [(term_id, signed weight)] where required terms carry 1000 + w (webui.py:364)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nq):
        nt = int(rng.integers(1, 5))
        ids = rng.choice(min(head, V), size=nt, replace=False)
        q = []
        for t in ids:
            r = rng.random()
            w = int(rng.integers(1, 4))
            q.append((int(t), 1 if r < 0.7 else (1000 + w if r < 0.85 else -w)))
        out.append(q)
    return out


def query_string(q: List[Tuple[int, int]], tokens: List[str]) -> str:
    """Render a synthetic query in the reference's syntax (README.md:87-99): tag, tag:+w, tag:-w."""
    parts = []
    for t, w in q:
        if w > 1000:
            parts.append("%s:+%d" % (tokens[t], w - 1000))
        elif w < 0:
            parts.append("%s:%d" % (tokens[t], w))
        else:
            parts.append(tokens[t])
    return " ".join(parts)


# ---------------------------------------------------------------------------------------------
# Doc2Vec model stand-in (training is out of scope: genmodel.py:159-162)
# ---------------------------------------------------------------------------------------------
def d2v_model(counts: np.ndarray, dim: int = 300, seed: int = 44, sample: float = 1e-3,
              ns_exponent: float = 0.75) -> Dict[str, np.ndarray]:
    """syn1neg ~ N(0,0.1^2) [V,dim]; cum_table / sample_int derived from the corpus counts the way
    gensim's Word2Vec.make_cum_table / prepare_vocab do [published algorithm]."""
    rng = np.random.default_rng(seed)
    V = len(counts)
    syn1neg = (rng.standard_normal((V, dim)) * 0.1).astype(np.float32)
    counts = np.maximum(np.asarray(counts, dtype=np.float64), 1.0)
    pw = counts ** ns_exponent
    cum = np.cumsum(pw)
    domain = 2 ** 31 - 1
    cum_table = np.round(cum / cum[-1] * domain).astype(np.uint32)
    cum_table[-1] = domain
    retain_total = counts.sum()
    threshold = sample * retain_total
    prob = (np.sqrt(counts / threshold) + 1) * (threshold / counts)
    prob = np.minimum(prob, 1.0)
    sample_int = (prob * (2 ** 32 - 1)).astype(np.uint32)
    return {"syn1neg": syn1neg, "cum_table": cum_table, "sample_int": sample_int}


def d2v_inputs(ndocs: int, dim: int = 300, seed: int = 44) -> Tuple[np.ndarray, np.ndarray]:
    """(v0 float32 [ndocs,dim] = (U(0,1)-0.5)/dim as gensim's pseudorandom_weak_vector, seeds uint64)."""
    rng = np.random.default_rng(seed + 1)
    v0 = ((rng.random((ndocs, dim), dtype=np.float32) - np.float32(0.5)) / np.float32(dim)).astype(np.float32)
    seeds = rng.integers(0, 2 ** 63, ndocs, dtype=np.uint64)
    return v0, seeds


def term_counts(ptr: np.ndarray, terms: np.ndarray, V: int) -> np.ndarray:
    return np.bincount(terms[terms >= 0], minlength=V).astype(np.int64)


def index_vectors(D: int = 100_000, dim: int = 300, seed: int = 46, unit: bool = False) -> np.ndarray:
    x = np.random.default_rng(seed).standard_normal((D, dim), dtype=np.float32)
    if unit:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    else:
        x *= np.float32(0.05)
    return x
