"""CPU: the C-ABI library loads, exports every symbol include/hip_tagsearch.h declares, and fails
loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hip_tagsearch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hipts_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from hiptagsearch import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes binding and header disagree"
    assert lib.hipts_abi_version() == 1


def test_no_cpu_fallback_when_gpu_missing():
    import hiptagsearch
    from hiptagsearch.bm25 import BM25Index
    if hiptagsearch.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(hiptagsearch.HipTagSearchError) as e:
        BM25Index(np.array([0, 2]), np.array([0, 1], dtype=np.int32), 2)
    assert e.value.status == -2 and "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    """The product package must not reference the oracle (ADVICE: parity claims depend on it)."""
    pkg = os.path.join(ROOT, "anime-illust-image-searcher_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
