"""ctypes binding of libhip_tagsearch.so (the C ABI declared in include/hip_tagsearch.h).

There is no CPU fallback: if the shared library is missing, or a compute entry point reports
an error (for instance "no HIP device"), this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import numpy as np

HOST = 0
DEVICE = 1

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_PKG_DIR), "libhip_tagsearch.so")


class HipTagSearchError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("libhip_tagsearch status %d: %s" % (status, message))
        self.status = status


class VitConfig(ctypes.Structure):
    _fields_ = [("image_size", c_int32), ("patch", c_int32), ("dim", c_int32), ("depth", c_int32),
                ("heads", c_int32), ("mlp_dim", c_int32), ("num_classes", c_int32), ("ln_eps", c_float),
                ("gelu_tanh", c_int32), ("pool_then_norm", c_int32), ("max_batch", c_int32), ("operand_f16", c_int32)]


class EvaConfig(ctypes.Structure):
    _fields_ = [("image_size", c_int32), ("patch", c_int32), ("dim", c_int32), ("depth", c_int32), ("heads", c_int32),
                ("mlp_hidden", c_int32), ("num_classes", c_int32), ("ln_eps", c_float), ("rope_ref_grid", c_int32),
                ("max_batch", c_int32), ("operand_f16", c_int32)]


class CcipConfig(ctypes.Structure):
    _fields_ = [("image_size", c_int32), ("dims", c_int32 * 4), ("depths", c_int32 * 4), ("head_dim", c_int32),
                ("attn_from_stage", c_int32), ("ln_eps", c_float), ("max_batch", c_int32), ("operand_f16", c_int32)]


# name -> (argtypes); every function returns int status unless listed in _PLAIN
_SIGNATURES = {
    "hipts_abi_version": [],
    "hipts_last_error": [c_char_p, c_size_t],
    "hipts_device_count": [POINTER(c_int)],
    "hipts_sizeof_config": [c_int, POINTER(c_size_t)],
    "hipts_vit_create": [POINTER(VitConfig), c_int, POINTER(c_void_p)],
    "hipts_vit_destroy": [c_void_p],
    "hipts_vit_set_tensor": [c_void_p, c_char_p, c_void_p, c_int64],
    "hipts_vit_forward_u8": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_vit_forward_f32": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_vit_flops_per_image": [c_void_p, POINTER(c_double)],
    "hipts_vit_set_sub_batches": [c_void_p, c_int],
    "hipts_vit_set_deferred_join": [c_void_p, c_int],
    "hipts_vit_join": [c_void_p, c_void_p],
    "hipts_vit_profile_enable": [c_void_p, c_int],
    "hipts_vit_profile_select": [c_void_p, ctypes.c_uint32],
    "hipts_vit_profile_read": [c_void_p, c_int, POINTER(c_double), POINTER(c_int64), POINTER(c_double), POINTER(c_double)],
    "hipts_vit_profile_name": [c_int, c_char_p, c_size_t],
    "hipts_eva_create": [POINTER(EvaConfig), c_int, POINTER(c_void_p)],
    "hipts_eva_destroy": [c_void_p],
    "hipts_eva_set_tensor": [c_void_p, c_char_p, c_void_p, c_int64],
    "hipts_eva_forward_u8": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_eva_forward_f32": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_eva_flops_per_image": [c_void_p, POINTER(c_double)],
    "hipts_ccip_create": [POINTER(CcipConfig), c_int, POINTER(c_void_p)],
    "hipts_ccip_destroy": [c_void_p],
    "hipts_ccip_set_tensor": [c_void_p, c_char_p, c_void_p, c_int64],
    "hipts_ccip_forward_u8": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "hipts_ccip_forward_f32": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "hipts_ccip_flops_per_image": [c_void_p, POINTER(c_double)],
    "hipts_tagsel_create": [c_void_p, c_int, c_int, c_int, POINTER(c_void_p)],
    "hipts_tagsel_destroy": [c_void_p],
    "hipts_tagsel_run": [c_void_p, c_void_p, c_int, c_int, c_double, c_int, c_double, c_int, c_void_p, c_void_p, c_int,
                         c_void_p, c_int, c_void_p],
    "hipts_tagsel_run_rows": [c_void_p, c_void_p, c_int, c_double, c_int, c_double, c_int, c_void_p, c_int, c_void_p],
    "hipts_bm25_build": [c_void_p, c_void_p, c_int64, c_int32, c_int, POINTER(c_void_p)],
    "hipts_bm25_destroy": [c_void_p],
    "hipts_bm25_info": [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), POINTER(c_double)],
    "hipts_bm25_export": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "hipts_bm25_set_idf": [c_void_p, c_void_p],
    "hipts_bm25_set_avgdl": [c_void_p, c_double],
    "hipts_bm25_score": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p],
    "hipts_index_create": [c_int, c_int64, c_int, POINTER(c_void_p)],
    "hipts_index_destroy": [c_void_p],
    "hipts_index_add": [c_void_p, c_void_p, c_int64, c_int],
    "hipts_index_len": [c_void_p, POINTER(c_int64)],
    "hipts_index_vector_by_id": [c_void_p, c_int64, c_void_p],
    "hipts_index_data": [c_void_p, POINTER(c_void_p)],
    "hipts_index_export": [c_void_p, c_int64, c_int64, c_void_p],
    "hipts_index_query": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p],
    "hipts_combine": [c_void_p, c_void_p, c_int, c_int64, c_double, c_double, c_int, c_int, c_void_p, c_int, c_void_p],
    "hipts_rowmax": [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_combine_with_max": [c_void_p, c_void_p, c_int, c_int64, c_double, c_double, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_topk": [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "hipts_synth_images_u8": [c_void_p, c_int64, c_int64, c_int, ctypes.c_uint64, c_int, c_void_p],
    "hipts_resize_u8": [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "hipts_resize_batch_u8": [c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "hipts_jpeg_entropy_decode": [c_void_p, c_int64, c_void_p, c_int64],
    "hipts_jpeg_slot_bytes": [c_int, c_int],
    "hipts_jpeg_decode_rgb": [c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_void_p],
    "hipts_jpeg_batch_u8": [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "hipts_ccip_metric": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "hipts_comm_unique_id": [c_void_p, c_size_t],
    "hipts_comm_create": [c_void_p, c_size_t, c_int, c_int, c_int, POINTER(c_void_p)],
    "hipts_comm_destroy": [c_void_p],
    "hipts_allgather_rows": [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    "hipts_topk_after": [c_void_p, c_int64, c_int, c_double, c_int64, c_void_p, c_void_p, c_int, c_void_p],
    "hipts_search": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_int,
                     c_void_p, c_void_p, c_void_p, c_void_p],
    "hipts_search_submit": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_int, c_int, c_void_p],
    "hipts_search_collect": [c_void_p, c_int, c_void_p, c_void_p],
    "hipts_query_profile_enable": [c_void_p, c_int],
    "hipts_query_profile_read": [c_void_p, c_int, POINTER(c_double), POINTER(c_int64), POINTER(c_double)],
    "hipts_query_profile_name": [c_int, c_char_p, c_size_t],
    "hipts_d2v_create": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_double, c_int, POINTER(c_void_p)],
    "hipts_d2v_destroy": [c_void_p],
    "hipts_d2v_train": [c_void_p, c_void_p, c_int64, c_int, c_int, c_double, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int,
                        c_float, c_float, ctypes.c_uint64, c_int, c_int, c_int, c_void_p],
    "hipts_d2v_infer": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p,
                        c_int, c_void_p],
    "hipts_d2v_set_word_vectors": [c_void_p, c_void_p],
    "hipts_d2v_infer_dm": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_int, c_void_p,
                           c_int, c_void_p],
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

_lib = None


def load():
    """Load (once) and return the ctypes library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libhip_tagsearch.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C anime-illust-image-searcher_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
    # Device pointers are only meaningful inside ONE runtime instance, so torch must be imported
    # first: the dynamic linker then binds this library's libamdhip64.so.7 dependency to the copy
    # torch already loaded instead of opening /opt/rocm's as a second runtime.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    _check_single_hip_runtime()
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the header and the library disagree
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.hipts_jpeg_slot_bytes.restype = ctypes.c_int64      # the one entry point that returns a size, not a status
    # the three configuration structures are passed by pointer: a layout that differs from the library's would be read past
    for kind, st in enumerate((VitConfig, EvaConfig, CcipConfig)):
        n = c_size_t(0)
        if lib.hipts_sizeof_config(kind, ctypes.byref(n)) != 0 or n.value != ctypes.sizeof(st):
            raise ImportError("%s is %d bytes here but %d in libhip_tagsearch.so: binding and library are out of step"
                              % (st.__name__, ctypes.sizeof(st), n.value))
    _lib = lib
    return lib


def _check_single_hip_runtime():
    try:
        paths = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    except OSError:
        return
    if len(paths) > 1:
        raise ImportError("two HIP runtimes are loaded in this process (%s): import torch before anything that "
                          "links /opt/rocm's libamdhip64" % ", ".join(sorted(paths)))


def last_error() -> str:
    buf = ctypes.create_string_buffer(1024)
    load().hipts_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


def check(status: int):
    if status != 0:
        raise HipTagSearchError(status, last_error())


def call(name: str, *args):
    check(getattr(load(), name)(*args))


def device_count() -> int:
    n = c_int(0)
    call("hipts_device_count", ctypes.byref(n))
    return n.value


def ptr(x):
    """void* of a numpy array (host), a torch tensor (host or device), an int address or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return x.ctypes.data_as(c_void_p)
    if isinstance(x, int):
        return c_void_p(x)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous(), "tensor must be contiguous"
        return c_void_p(x.data_ptr())
    raise TypeError("cannot take the address of %r" % type(x))


def memspace_of(x) -> int:
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):
        return DEVICE if x.is_cuda else HOST
    return HOST


_raw_stream = None


def current_stream_ptr():
    """hipStream_t of torch's current stream as void* (None = null stream when torch is absent
    or has no device).  On the one-query path this is called per query: the raw-stream accessor
    (0.2 us) is used when torch offers it, torch.cuda.current_stream() (several us) otherwise."""
    global _raw_stream
    try:
        if _raw_stream is None:
            import torch
            if not torch.cuda.is_available():
                _raw_stream = False
            else:
                get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
                cur = torch.cuda.current_device
                _raw_stream = (lambda: get(cur())) if get is not None else (lambda: torch.cuda.current_stream().cuda_stream)
        if _raw_stream is False:
            return None
        return c_void_p(_raw_stream())
    except Exception:
        pass
    return None
