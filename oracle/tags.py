"""Tag-selection oracle (numpy float64) -- test infrastructure, see oracle/__init__.py.

Follows tagging.py:61-66 (mcut_threshold) and tagging.py:185-227 (per-image
selection inside Predictor.predict).  Pinned by tests/golden/g3_mcut.json and
g4_predict.{npz,json}.
"""
from typing import List, Sequence, Tuple

import numpy as np


def sigmoid_f32(logits: np.ndarray) -> np.ndarray:
    """F.sigmoid on float32 (tagging.py:176).  torch CPU evaluates 1/(1+exp(-x)) in
    float32; the product's device kernel is compared with a tolerance (float
    transcendental), the *selection* below is then exact given the probabilities."""
    x = logits.astype(np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def mcut_threshold(probs: np.ndarray) -> float:
    """tagging.py:61-66."""
    sorted_probs = probs[probs.argsort()[::-1]]
    difs = sorted_probs[:-1] - sorted_probs[1:]
    t = difs.argmax()
    return (sorted_probs[t] + sorted_probs[t + 1]) / 2


def select_indices(probs_row: np.ndarray, general_index: Sequence[int], character_index: Sequence[int],
                   general_thresh: float = 0.3, general_mcut: bool = True,
                   character_thresh: float = 0.3, character_mcut: bool = True) -> Tuple[List[int], List[int], float, float]:
    """One image: returns (general label ids in output order, character label ids in
    output order, general threshold, character threshold).  tagging.py:186-225:
    probabilities widened to float64 (:186), threshold by MCut, strict '>' filter in
    label order, then a *stable* descending sort (Python sorted(reverse=True))."""
    p = probs_row.astype(float)                                      # :186
    gi = np.asarray(general_index, dtype=np.int64)
    ci = np.asarray(character_index, dtype=np.int64)
    if general_mcut:
        general_thresh = mcut_threshold(p[gi])                       # :190-192
    g_sel = [int(i) for i in gi if p[i] > general_thresh]            # :194
    if character_mcut:
        character_thresh = mcut_threshold(p[ci])                     # :198-200
        character_thresh = max(0.15, character_thresh)               # :201
    c_sel = [int(i) for i in ci if p[i] > character_thresh]          # :203
    g_sorted = sorted(g_sel, key=lambda i: p[i], reverse=True)       # :205-209 (stable)
    c_sorted = sorted(c_sel, key=lambda i: p[i], reverse=True)       # :217-221
    return g_sorted, c_sorted, float(general_thresh), float(character_thresh)


def format_line(names: Sequence[str], g_sorted: Sequence[int], c_sorted: Sequence[int]) -> str:
    """tagging.py:210-225."""
    s = ",".join(names[i].replace(" ", "_") for i in g_sorted)
    if len(c_sorted) > 0:
        s += ","
        s += ",".join(names[i].replace(" ", "_") for i in c_sorted)
    return s


def predict_lines(probs: np.ndarray, names: Sequence[str], category: np.ndarray) -> List[str]:
    """All images of a batch.  category: 9 rating / 0 general / 4 character
    (tagging.py:137-139)."""
    gi = list(np.where(category == 0)[0])
    ci = list(np.where(category == 4)[0])
    out = []
    for r in range(probs.shape[0]):
        g, c, _, _ = select_indices(probs[r], gi, ci)
        out.append(format_line(names, g, c))
    return out
