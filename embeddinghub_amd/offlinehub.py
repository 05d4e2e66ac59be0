"""Drop-in for embeddinghub/sdk/python/offlinehub.py `Index` (offlinehub.py:27-141) on the MI355X
engine: same class surface and semantics, with hnswlib.Index replaced by an engine space reached
through the C ABI (exhaustive MFMA scan + canonical re-rank, i.e. exact neighbours).

Divergences, all deliberate:
  * nearest_neighbor(embedding=...) returns KEYS; the reference returns raw hnswlib labels in that
    branch (offlinehub.py:120-121 skips the mapper unless `key` was given) — an upstream bug;
  * asking for more neighbours than stored returns what exists instead of raising.
"""
try:  # py3.10+: collections.Mapping is gone (the reference still imports it, offlinehub.py:24)
    from collections.abc import Mapping
except ImportError:  # pragma: no cover
    from collections import Mapping

import numpy as np

from . import _lib
from .space import Space


class Index:
    def __init__(self, key_emb_iter, dims):
        self._data = {}
        self._orig = {}  # str(key) -> the caller's key object (keys travel through the C ABI as strings)
        self._dims = dims
        self._space = Space.unique("offline-index", dims, metric=_lib.METRIC_L2SQ,
                                   initial_capacity=1024)  # offlinehub.py:34-35 (cap 1024)
        self._size = 0
        self.multiset(key_emb_iter)

    def set(self, key, embedding):
        if key not in self._data:
            self._size += 1
        self._data[key] = embedding
        self._orig[str(key)] = key
        self._space.set(str(key), embedding)

    def get(self, key):
        return self._data[key]

    def multiset(self, embedding_tuples):
        if isinstance(embedding_tuples, Mapping):
            embedding_tuples = embedding_tuples.items()
        keys, embs = [], []
        for key, embedding in embedding_tuples:
            keys.append(str(key))
            embs.append(embedding)
            if key not in self._data:
                self._size += 1
            self._data[key] = embedding
            self._orig[str(key)] = key
        if not keys:
            return
        self._space.set_batch(keys, np.asarray(embs, dtype=np.float32).reshape(len(keys), self._dims))

    def multiget(self, keys):
        return [self._data[key] for key in keys]

    def nearest_neighbor(self, num, key=None, embedding=None):
        if key is not None:
            ids, _ = self._space.knn_by_key(str(key), num)
            return [self._orig_key(self._space.key_of(i)) for i in ids]
        keys = self._space.knn_keys(np.asarray(embedding, dtype=np.float32), num)[0]
        return [self._orig_key(k) for k in keys]

    def _orig_key(self, skey):
        # hand back the caller's original key object: one dict lookup per neighbour (it was a scan over every key
        # of the index for non-string keys — the reference's own tests use integer keys, offlinehub_test.py:68-86)
        return self._orig.get(skey, skey)

    def size(self):
        return self._size

    def close(self):
        self._space.drop()
