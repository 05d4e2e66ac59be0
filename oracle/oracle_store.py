"""ORACLE — TEST INFRASTRUCTURE ONLY.  An embeddingstore `store` (the interface embeddinghub_amd/rpc/server.py serves)
backed by the CPU oracle's ANNIndex restatement: the CPU leg of tests/test_grpc_shim.py runs the gRPC shim over it, and
bench.py's reference-benchmark leg times it beside the engine-backed store (the reference's own benchmark shape,
embeddinghub/test/benchmark.py:217-272) and compares the key lists.  Like the reference's server (server.cc:68-215: one
std::mutex around every RPC body) every call is serialised."""
import threading

import numpy as np

from . import pyoracle


def _errors():
    from embeddinghub_amd.rpc import server as srv   # the servicer's own exception types (status mapping)
    return srv.SpaceNotWritable, srv.KeyNotFound


class OracleSpace:
    def __init__(self, dims):
        self.dims = dims
        self._idx = pyoracle.AnnIndex(dims)
        self._vals = {}
        self._frozen = False
        self._mu = threading.Lock()    # server.cc: mtx_

    def set(self, key, vec):
        self.set_batch([key], [vec])

    def set_batch(self, keys, vecs):
        with self._mu:
            if self._frozen:
                raise _errors()[0]()
            for k, v in zip(keys, vecs):
                v = np.asarray(v, dtype=np.float32)
                self._idx.set(k, v)
                self._vals[k] = v

    def get(self, key):
        return self._vals.get(key)

    def freeze(self):
        self._frozen = True

    def __len__(self):
        return len(self._vals)

    def keys_sorted(self):
        return sorted(self._vals, key=lambda k: k.encode())

    def nearest(self, num, key="", embedding=None):
        with self._mu:
            if key:
                if key not in self._vals:
                    raise _errors()[1]()
                got = self._idx.approx_nearest(self._vals[key], min(num + 1, len(self._vals)))
                if key in got:
                    got.remove(key)
                else:
                    got = got[:-1]
                return got[:num]
            return self._idx.approx_nearest(np.asarray(embedding, dtype=np.float32), min(num, len(self._vals)))

    def nearest_many(self, num, embeddings):  # the servicer's batched branch (EngineSpace: one ehx_knn_keys call)
        self.batched_calls = getattr(self, "batched_calls", 0) + 1
        return [self.nearest(num, embedding=e) for e in np.asarray(embeddings, dtype=np.float32)]


class OracleStore:
    def __init__(self):
        self._spaces = {}

    def create_space(self, name, dims):
        return self._spaces.setdefault(name, OracleSpace(dims))

    def get_space(self, name):
        return self._spaces.get(name)

    def delete_space(self, name):
        self._spaces.pop(name, None)
