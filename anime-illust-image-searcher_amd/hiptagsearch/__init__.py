"""hiptagsearch -- MI355X-native indexing / query-scoring hot path of
ryogrid/anime-illust-image-searcher behind the reference's own Python call sites.

All arithmetic runs in libhip_tagsearch.so (hand-written HIP for gfx950, C ABI in
include/hip_tagsearch.h).  Importing this package loads that library and raises if it has not
been built; there is no CPU fallback.
"""
from . import _lib

_lib.load()

from ._lib import HipTagSearchError, device_count  # noqa: E402,F401

__all__ = ["HipTagSearchError", "device_count"]
