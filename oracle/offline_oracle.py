"""ORACLE — TEST INFRASTRUCTURE ONLY.

Restatement of embeddinghub/sdk/python/offlinehub.py:27-183 (`Index`,
`HnswlibIndexMapper`) on top of the restated HNSW, i.e. what the reference class does
when `hnswlib.Index("l2", dims)` is hnswlib==0.5.2 with its defaults
(init_index(max_elements): M=16, ef_construction=200, random_seed=100; ef=10) and
single-threaded insertion order (add_items inserts serially when rows <= 4*threads, and
one-at-a-time `set` is always serial — SURVEY Appendix A.7).
"""
from . import pyoracle


class HnswlibIndexMapper:  # offlinehub.py:144-183
    def __init__(self):
        self._key_to_idx = {}
        self._idx_to_key = {}
        self._next_idx = 0

    def to_idx(self, key):
        if key in self._key_to_idx:
            return self._key_to_idx[key]
        idx = self._next_idx
        self._next_idx += 1
        self._key_to_idx[key] = idx
        self._idx_to_key[idx] = key
        return idx

    def to_key(self, idx):
        return self._idx_to_key[idx]


class Index:  # offlinehub.py:27-141
    def __init__(self, key_emb_iter, dims):
        self._data = {}
        self._dims = dims
        self._mapper = HnswlibIndexMapper()
        self._size = 0
        self._cap = 1024
        self._idx = pyoracle.Hnsw(dims, pyoracle.METRIC_L2, self._cap)
        self.multiset(key_emb_iter)

    def set(self, key, embedding):
        if key not in self._data:
            self._size += 1
        self._data[key] = embedding
        idx = self._mapper.to_idx(key)
        self._idx.add(embedding, idx)
        self._add_capacity_to_fit(1)

    def get(self, key):
        return self._data[key]

    def multiset(self, embedding_tuples):
        embeddings, idxs = [], []
        if hasattr(embedding_tuples, "items"):
            embedding_tuples = embedding_tuples.items()
        for key, embedding in embedding_tuples:
            embeddings.append(embedding)
            idxs.append(self._mapper.to_idx(key))
            if key not in self._data:
                self._size += 1
            self._data[key] = embedding
        if len(idxs) == 0:
            return
        self._add_capacity_to_fit(len(idxs))
        for e, i in zip(embeddings, idxs):
            self._idx.add(e, i)

    def multiget(self, keys):
        return [self._data[key] for key in keys]

    def nearest_neighbor(self, num, key=None, embedding=None):
        has_key = key is not None
        if has_key:
            embedding = self._data[key]
            num_retrieve = num + 1
        else:
            num_retrieve = num
        labels, _ = self._idx.search(embedding, num_retrieve)
        if len(labels) != num_retrieve:  # hnswlib 0.5.2 knn_query raises here
            raise RuntimeError("Cannot return the results in a contigious 2D array. Probably ef or M is too small")
        results = [int(x) for x in labels]
        if has_key:
            idx = self._mapper.to_idx(key)
            results = [self._mapper.to_key(r) for r in results if r != idx]
            if len(results) > num:
                results = results[:-1]
        return results

    def size(self):
        return self._size

    def _add_capacity_to_fit(self, size):
        min_cap = self._size + size
        if min_cap > self._cap:
            self._cap *= 2
            self._idx.resize(self._cap)
