"""CPU: the C-ABI library loads, exports every symbol include/hip_tagsearch.h declares, and fails
loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="hip_tagsearch.h", prefix="hipts_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def _header_struct_fields(name):
    """[(ctype, field, array length)] of `typedef struct <name> {...}` as the header declares it."""
    text = open(os.path.join(ROOT, "include", "hip_tagsearch.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\}" % name, text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for typ, field, arr in re.findall(r"\b(int32_t|float)\s+([a-z_0-9]+)(?:\[(\d+)\])?\s*;", body):
        out.append((typ, field, int(arr) if arr else 1))
    return out


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


def test_config_struct_layouts_match_header_library_and_docs():
    """The three configuration structures are passed by pointer: the ctypes structures of the binding, the stubs printed in
    INTEGRATION.md and the header must agree field by field, and their size must be what the library was compiled with
    (hipts_sizeof_config) -- VERDICT r1: a stub one field short hands the library a struct it reads past."""
    from hiptagsearch import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.hipts_sizeof_config.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]
    cmap = {"int32_t": ctypes.c_int32, "float": ctypes.c_float}
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ns = {"ctypes": ctypes}
    for cls_name in ("VitCfg", "EvaCfg", "CcipCfg"):                   # execute the stubs' class definitions as printed
        m = re.search(r"^class %s\(ctypes\.Structure\):.*?\n(?=\S)" % cls_name, doc, flags=re.S | re.M)
        assert m, cls_name
        exec(m.group(0), ns)
    for kind, (hname, binding, stub) in enumerate((("hipts_vit_config", _lib.VitConfig, ns["VitCfg"]),
                                                   ("hipts_eva_config", _lib.EvaConfig, ns["EvaCfg"]),
                                                   ("hipts_ccip_config", _lib.CcipConfig, ns["CcipCfg"]))):
        fields = _header_struct_fields(hname)
        want = [(f, cmap[t] if n == 1 else cmap[t] * n) for t, f, n in fields]
        for st in (binding, stub):
            got = [(f, t) for f, t in st._fields_]
            assert [f for f, _ in got] == [f for f, _ in want], (hname, st)
            assert all(ctypes.sizeof(a) == ctypes.sizeof(b) and a._type_ == b._type_ for (_, a), (_, b) in zip(got, want)), (hname, st)
        n = ctypes.c_size_t(0)
        assert lib.hipts_sizeof_config(kind, ctypes.byref(n)) == 0
        assert n.value == ctypes.sizeof(binding) == ctypes.sizeof(stub) == sum(4 * k for _, _, k in fields), hname
    assert lib.hipts_sizeof_config(7, ctypes.byref(n)) != 0


def test_debug_entry_points_are_declared():
    """Every hiptsdbg_* symbol the library exports is declared in include/hip_tagsearch_debug.h (none undeclared), and
    nothing else leaks out of the library with C linkage."""
    import subprocess
    from hiptagsearch import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("_")})
    declared = set(_declared_symbols()) | set(_declared_symbols("hip_tagsearch_debug.h", "hiptsdbg_"))
    assert [e for e in exported if e not in declared] == []
    assert all(d in exported for d in declared)
