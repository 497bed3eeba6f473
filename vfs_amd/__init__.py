"""vfs_amd -- MI355X-native exact search behind the VFS Ranker API (hot path only).

The compute lives in vfs_amd/lib/libvfsx.so (HIP, gfx950; C ABI in include/vfsx.h).
"""
from ._lib import Hit, VfsxError, load  # noqa: F401
from .ranker import Corpus, DistanceMethod, Ranker, SearchType  # noqa: F401

__all__ = ["Corpus", "DistanceMethod", "Ranker", "SearchType", "VfsxError", "Hit", "load"]
