"""ctypes restatement of integration/go/mi355x/mi355x.go, method for method: the SAME sequence of ehx_* calls with the same
arguments, and the same mapping of return codes to error types — the fferr constructors of the Go file appear here as
exception classes of the same name.  No Go toolchain exists in this image, so the Go file cannot be compiled; what CAN be
executed is its call pattern against the real library, which is what the reference's provider conformance suites
(provider/vectorstore_test.go:22-46, provider/online_test.go:28-70) exercise.  tests/test_go_conformance.py checks that
this file and the Go file issue the same C calls per method (parsed from both sources) and runs the suites on the GPU.

TEST INFRASTRUCTURE — not part of the product."""
import ctypes as C

from embeddinghub_amd import _lib


class FfErr(Exception):
    pass


class DatasetNotFoundError(FfErr):          # fferr.NewDatasetNotFoundError
    pass


class DatasetAlreadyExistsError(FfErr):     # fferr.NewDatasetAlreadyExistsError
    pass


class EntityNotFoundError(FfErr):           # fferr.NewEntityNotFoundError
    pass


class DataTypeNotFoundError(FfErr):         # fferr.NewDataTypeNotFoundErrorf
    pass


class InvalidArgumentError(FfErr):          # fferr.NewInvalidArgumentError
    pass


class ResourceExecutionError(FfErr):        # fferr.NewResourceExecutionError
    pass


class ConnectionError_(FfErr):              # fferr.NewConnectionError
    pass


class VectorType:                           # provider/types/value_type.go:96-100
    def __init__(self, dimension, is_embedding=True, scalar="float32"):
        self.Dimension, self.IsEmbedding, self.ScalarType = dimension, is_embedding, scalar


def _last_error(L):
    return (L.ehx_last_error() or b"").decode(errors="replace")


def space_name(feature, variant):           # mi355x.go: spaceName
    return "Featureform_table__%s__%s" % (feature, variant)


class Mi355xOnlineStore:
    """mi355x.go: mi355xOnlineStore (OnlineStore + VectorStore, provider/online.go:42-59)"""

    def __init__(self, devices=(0,), shards=0, metric="cosine", mode="flat", ef=0, build_batch=0, search_width=0):
        # mi355xOnlineStoreFactory
        self.cfg = dict(Devices=list(devices), Shards=shards, Metric=metric, Mode=mode, EF=ef, BuildBatch=build_batch,
                        SearchWidth=search_width)
        L = self.L = _lib.load()
        devs = (C.c_int * len(devices))(*devices)
        rc = L.ehx_init(devs if len(devices) else None, len(devices))
        if rc != _lib.OK:
            raise ConnectionError_(_last_error(L))

    def Close(self):
        return None

    def Type(self):
        return "MI355X_ONLINE"

    def metric(self):
        return {"l2": _lib.METRIC_L2SQ, "ip": _lib.METRIC_IP}.get(self.cfg["Metric"], _lib.METRIC_COSINE)

    def open(self, feature, variant):
        L = self.L
        name = space_name(feature, variant).encode()
        sp = C.c_void_p()
        rc = L.ehx_space_open(name, len(name), C.byref(sp))
        if rc != _lib.OK:
            raise DatasetNotFoundError(feature, variant)
        dims = C.c_uint32()
        L.ehx_space_dims(sp, C.byref(dims))
        return Mi355xTable(L, sp, dims.value, feature, variant)

    def create(self, feature, variant, dims):
        L = self.L
        name = space_name(feature, variant).encode()
        p = _lib.Params()
        p.shards = self.cfg["Shards"]
        if self.cfg["Mode"] == "graph":
            p.mode = _lib.MODE_GRAPH
        p.ef = self.cfg["EF"]
        p.build_batch = self.cfg["BuildBatch"]
        p.search_width = self.cfg["SearchWidth"]
        sp = C.c_void_p()
        rc = L.ehx_space_create(name, len(name), dims, self.metric(), _lib.DTYPE_F32, C.byref(p), C.byref(sp))
        if rc == _lib.OK:
            return Mi355xTable(L, sp, dims, feature, variant)
        if rc == _lib.EEXISTS:
            raise DatasetAlreadyExistsError(feature, variant)
        raise ResourceExecutionError(_last_error(L))

    # VectorStore
    def CreateIndex(self, feature, variant, vector_type):
        try:
            return self.open(feature, variant)
        except DatasetNotFoundError:
            return self.create(feature, variant, vector_type.Dimension)

    def DeleteIndex(self, feature, variant):
        return self.DeleteTable(feature, variant)

    # OnlineStore
    def GetTable(self, feature, variant):
        return self.open(feature, variant)

    def CreateTable(self, feature, variant, value_type):
        if not isinstance(value_type, VectorType):
            raise DataTypeNotFoundError("the MI355X store holds embedding vectors only")
        try:
            t = self.open(feature, variant)
        except DatasetNotFoundError:
            return self.create(feature, variant, value_type.Dimension)
        n = C.c_uint64(0)
        self.L.ehx_space_size(t.sp, C.byref(n))
        if n.value > 0 or not value_type.IsEmbedding:
            raise DatasetAlreadyExistsError(feature, variant)
        return t

    def DeleteTable(self, feature, variant):
        t = self.open(feature, variant)
        self.L.ehx_space_drop(t.sp)
        return None


class Mi355xTable:
    """mi355x.go: mi355xTable (VectorStoreTable + BatchOnlineTable, provider/online.go:50-70)"""

    def __init__(self, L, sp, dims, feature, variant):
        self.L, self.sp, self.dims, self.feature, self.variant = L, sp, dims, feature, variant

    def fail(self):
        return ResourceExecutionError(_last_error(self.L))

    @staticmethod
    def _is_f32_slice(value, dims):
        import numpy as np
        return isinstance(value, np.ndarray) and value.dtype == np.float32 and value.ndim == 1 and value.shape[0] == dims

    def Set(self, entity, value):
        if not self._is_f32_slice(value, self.dims):
            raise DataTypeNotFoundError("expected []float32 of length %d" % self.dims)
        key = entity.encode()
        rc = self.L.ehx_set(self.sp, key, len(key), value.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != _lib.OK:
            raise self.fail()
        return None

    def MaxBatchSize(self):
        return 65536

    def BatchSet(self, items):
        import numpy as np
        n = len(items)
        if n == 0:
            return None
        flat = np.empty((n, self.dims), dtype=np.float32)
        keys = (C.c_char_p * n)()
        lens = (C.c_size_t * n)()
        for i, (entity, value) in enumerate(items):
            if not self._is_f32_slice(value, self.dims):
                raise DataTypeNotFoundError("expected []float32 of length %d" % self.dims)
            flat[i] = value
            keys[i] = entity.encode()
            lens[i] = len(entity.encode())
        rc = self.L.ehx_set_batch(self.sp, n, keys, lens, flat.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != _lib.OK:
            raise self.fail()
        return None

    def Get(self, entity):
        import numpy as np
        out = np.empty(self.dims, dtype=np.float32)
        key = entity.encode()
        rc = self.L.ehx_get(self.sp, key, len(key), out.ctypes.data_as(C.POINTER(C.c_float)))
        if rc == _lib.OK:
            return out
        if rc == _lib.ENOTFOUND:
            raise EntityNotFoundError(self.feature, self.variant, entity)
        raise self.fail()

    def Nearest(self, feature, variant, vector, k):
        import numpy as np
        if not self._is_f32_slice(vector, self.dims) or k <= 0:
            raise InvalidArgumentError("expected a %d-dim vector and k > 0" % self.dims)
        ids = (C.c_uint64 * k)()
        dist = (C.c_float * k)()
        off = (C.c_uint64 * (k + 1))()
        count = C.c_uint32()
        arena = C.create_string_buffer(64 * k + 4096)
        while True:
            rc = self.L.ehx_knn_keys(self.sp, 1, vector.ctypes.data_as(C.POINTER(C.c_float)), k, ids, dist,
                                     C.byref(count), arena, len(arena), off)
            if rc == _lib.ERANGE:
                arena = C.create_string_buffer(4 * len(arena))
                continue
            if rc != _lib.OK:
                raise self.fail()
            break
        raw = arena.raw
        return [raw[off[j]:off[j + 1]].decode() for j in range(count.value)]
