"""The Go provider boundary at the C ABI.  integration/go/mi355x/mi355x.go cannot be compiled here (no
Go toolchain), so:

  * CPU: the Go file and its ctypes restatement tests/go_mirror.py must issue the SAME ehx_* calls, in the same order, in
    every method (both sources are parsed) and every C symbol / constant the Go file names must exist in include/ehx.h;
  * GPU: the reference's provider conformance suites, restated against the mirror — VectorStoreTest
    (provider/vectorstore_test.go:27-46: TypeAssertion, CreateIndex, GetSet with DeepEqual, Nearest on the reference's own
    fixture) and OnlineStoreTest (provider/online_test.go:37-70: CreateGetTable, TableAlreadyExists, TableNotFound,
    SetGetEntity, EntityNotFound, MassTableWrite, type errors, FloatVecValues, BatchSetGetEntity, DeleteTable) — asserting
    the error TYPE each Go method maps the engine's return code to."""
import json
import os
import re
import threading
import uuid

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "integration", "go", "mi355x", "mi355x.go")
MIRROR = os.path.join(ROOT, "tests", "go_mirror.py")

# Go method -> mirror method
METHODS = {"mi355xOnlineStoreFactory": "__init__", "open": "open", "create": "create", "CreateIndex": "CreateIndex",
           "DeleteIndex": "DeleteIndex", "GetTable": "GetTable", "CreateTable": "CreateTable",
           "DeleteTable": "DeleteTable", "fail": "fail", "Set": "Set", "BatchSet": "BatchSet", "Get": "Get",
           "Nearest": "Nearest"}


def _go_funcs():
    src = open(GO).read()
    out = {}
    for m in re.finditer(r"^func (?:\([^)]*\) )?(\w+)\(", src, flags=re.M):
        name, start = m.group(1), m.end()
        nxt = re.search(r"^func ", src[start:], flags=re.M)
        body = src[start:start + nxt.start()] if nxt else src[start:]
        out[name] = body
    return out


def _py_funcs():
    src = open(MIRROR).read()
    out = {}
    for m in re.finditer(r"^    def (\w+)\(", src, flags=re.M):
        name, start = m.group(1), m.end()
        nxt = re.search(r"^(    def |class |def )", src[start:], flags=re.M)
        body = src[start:start + nxt.start()] if nxt else src[start:]
        if name not in out or ("L.ehx_" in body and "L.ehx_" not in out[name]):  # (several classes have an __init__)
            out[name] = body
    return out


def test_mirror_issues_the_go_files_c_calls_method_by_method():
    go, py = _go_funcs(), _py_funcs()
    for gname, pname in METHODS.items():
        assert gname in go, gname
        g_calls = [c for c in re.findall(r"C\.(ehx_\w+)\(", go[gname]) if c != "ehx_last_error"]
        p_calls = [c for c in re.findall(r"L\.(ehx_\w+)\(", py[pname]) if c != "ehx_last_error"]
        assert g_calls == p_calls, "%s: Go calls %s, mirror calls %s" % (gname, g_calls, p_calls)
    # nothing in the Go file calls the engine outside the mirrored methods
    for gname, body in go.items():
        if gname not in METHODS:
            assert not re.findall(r"C\.ehx_\w+\(", body), gname


def test_every_c_name_the_go_file_uses_is_declared_in_the_header():
    header = open(os.path.join(ROOT, "include", "ehx.h")).read()
    src = open(GO).read()
    names = set(re.findall(r"C\.((?:ehx|EHX)_\w+)", src))
    assert {"ehx_init", "ehx_space_create", "ehx_space_open", "ehx_space_drop", "ehx_set", "ehx_set_batch", "ehx_get",
            "ehx_knn_keys", "ehx_last_error", "EHX_EEXISTS", "EHX_ENOTFOUND", "EHX_ERANGE"} <= names
    for n in names:
        assert re.search(r"\b%s\b" % n, header), "%s is not in include/ehx.h" % n
    # the struct fields the Go file sets
    for field in re.findall(r"\bp\.(\w+) =", src):
        assert re.search(r"\buint\d+_t %s;" % field, header), field


# ---------------------------------------------------------------------------------------------------------------------
def _fv():
    return str(uuid.uuid4()), str(uuid.uuid4())   # randomFeatureVariant (online_test.go:75-77)


@pytest.fixture(scope="module")
def store():
    pytest.importorskip("embeddinghub_amd")
    import go_mirror
    s = go_mirror.Mi355xOnlineStore()
    yield s
    assert s.Close() is None


@pytest.fixture(scope="module")
def fixture_rows():
    with open(os.path.join(ROOT, "tests", "golden", "go_vectorstore.json")) as f:
        g = json.load(f)
    return g


VT768 = dict(dimension=768, is_embedding=True)


@pytest.mark.gpu
def test_vectorstore_type_assertion_and_create_index(store):   # vectorstore_test.go:50-73
    import go_mirror
    for method in ("CreateIndex", "DeleteIndex", "GetTable", "CreateTable", "DeleteTable", "Close"):
        assert callable(getattr(store, method))
    f, v = _fv()
    tbl = store.CreateIndex(f, v, go_mirror.VectorType(**VT768))
    assert tbl is not None
    for method in ("Set", "Get", "Nearest", "BatchSet", "MaxBatchSize"):
        assert callable(getattr(tbl, method))
    assert store.DeleteIndex(f, v) is None
    with pytest.raises(go_mirror.DatasetNotFoundError):
        store.GetTable(f, v)
    with pytest.raises(go_mirror.DatasetNotFoundError):    # DeleteIndex of what does not exist: DeleteTable's error
        store.DeleteIndex(f, v)


def _index_table(store):
    import go_mirror
    f, v = _fv()
    vt = go_mirror.VectorType(**VT768)
    assert store.CreateIndex(f, v, vt) is not None          # vectorstore_test.go:87-90
    assert store.CreateTable(f, v, vt) is not None          # :91-94  (the index a moment ago: same table, no error)
    tbl = store.GetTable(f, v)                              # :95-98
    assert tbl is not None and hasattr(tbl, "Nearest")      # :99-102 (VectorStoreTable)
    return f, v, tbl


@pytest.mark.gpu
def test_vectorstore_get_set_deep_equal(store, fixture_rows):   # vectorstore_test.go:76-119
    f, v, tbl = _index_table(store)
    for entity, vec in zip(fixture_rows["entities"], fixture_rows["vectors"]):
        x = np.asarray(vec, dtype=np.float32)
        assert tbl.Set(entity, x) is None
        got = tbl.Get(entity)
        assert got.dtype == np.float32 and got.tobytes() == x.tobytes()   # reflect.DeepEqual on []float32
    assert store.DeleteIndex(f, v) is None


@pytest.mark.gpu
def test_vectorstore_nearest(store, fixture_rows):   # vectorstore_test.go:121-168 (+ the derived golden)
    f, v, tbl = _index_table(store)
    for entity, vec in zip(fixture_rows["entities"], fixture_rows["vectors"]):
        tbl.Set(entity, np.asarray(vec, dtype=np.float32))
    q = np.asarray(fixture_rows["query"], dtype=np.float32)
    results = tbl.Nearest(f, v, q, 2)
    assert len(results) == 2                                               # what the reference asserts
    assert results == ["Investor's Business Daily", "Seeking Alpha"]        # what the fixture's arithmetic gives
    assert len(tbl.Nearest(f, v, q, 50)) == 5                              # <= k entities, like Redis (redis.go:463-471)
    import go_mirror
    with pytest.raises(go_mirror.InvalidArgumentError):
        tbl.Nearest(f, v, q[:10], 2)
    with pytest.raises(go_mirror.InvalidArgumentError):
        tbl.Nearest(f, v, q, 0)
    assert store.DeleteIndex(f, v) is None


@pytest.mark.gpu
def test_online_create_get_table_already_exists_not_found(store):   # online_test.go:79-116
    import go_mirror
    f, v = _fv()
    vt = go_mirror.VectorType(3, is_embedding=False)
    assert store.CreateTable(f, v, vt) is not None
    assert store.GetTable(f, v) is not None
    with pytest.raises(go_mirror.DatasetAlreadyExistsError) as e:   # testTableAlreadyExists
        store.CreateTable(f, v, vt)
    assert str(e.value) != ""
    store.DeleteTable(f, v)
    f2, v2 = _fv()
    with pytest.raises(go_mirror.DatasetNotFoundError) as e:        # testTableNotFound
        store.GetTable(f2, v2)
    assert str(e.value) != ""
    # an embedding table that already holds rows is not handed out a second time either
    f3, v3 = _fv()
    et = go_mirror.VectorType(3, is_embedding=True)
    t = store.CreateTable(f3, v3, et)
    t.Set("e", np.array([1, 2, 3], dtype=np.float32))
    with pytest.raises(go_mirror.DatasetAlreadyExistsError):
        store.CreateTable(f3, v3, et)
    store.DeleteTable(f3, v3)


@pytest.mark.gpu
def test_online_set_get_entity_and_entity_not_found(store):   # online_test.go:118-198, testFloatVecValues :309-345
    import go_mirror
    for ent, val, emb in (("c", [1, 2, 3], True), ("d", [4, 5, 6], False)):
        f, v = _fv()
        tab = store.CreateTable(f, v, go_mirror.VectorType(3, is_embedding=emb))
        x = np.asarray(val, dtype=np.float32)
        assert tab.Set(ent, x) is None
        assert tab.Get(ent).tobytes() == x.tobytes()
        with pytest.raises(go_mirror.EntityNotFoundError) as e:    # testEntityNotFound
            tab.Get("no-such-entity")
        assert str(e.value) != ""
        store.DeleteTable(f, v)


@pytest.mark.gpu
def test_online_type_errors(store):   # testTypeCasting (online_test.go) / redis.go:408-413
    import go_mirror
    f, v = _fv()
    with pytest.raises(go_mirror.DataTypeNotFoundError):           # a scalar table: this store holds vectors only
        store.CreateTable(f, v, "string")
    tab = store.CreateTable(f, v, go_mirror.VectorType(3))
    for bad in ("val", 1, None, [1.0, 2.0, 3.0], np.array([1, 2, 3], dtype=np.float64),
                np.array([1, 2], dtype=np.float32)):
        with pytest.raises(go_mirror.DataTypeNotFoundError):
            tab.Set("e", bad)
        with pytest.raises(go_mirror.DataTypeNotFoundError):
            tab.BatchSet([("e", bad)])
    with pytest.raises(go_mirror.EntityNotFoundError):              # nothing was written by the refused Sets
        tab.Get("e")
    store.DeleteTable(f, v)


@pytest.mark.gpu
def test_online_mass_table_write(store):   # online_test.go:200-236: 10 tables x 10 entities, read back after all writes
    import go_mirror
    tables = [_fv() for _ in range(10)]
    entities = [str(uuid.uuid4()) for _ in range(10)]
    rng = np.random.default_rng(8)
    vals = rng.standard_normal((10, 10, 8)).astype(np.float32)
    for i, (f, v) in enumerate(tables):
        tab = store.CreateTable(f, v, go_mirror.VectorType(8))
        for j, e in enumerate(entities):
            tab.Set(e, vals[i, j])
    for i, (f, v) in enumerate(tables):
        tab = store.GetTable(f, v)
        for j, e in enumerate(entities):
            assert tab.Get(e).tobytes() == vals[i, j].tobytes()
    for f, v in tables:
        store.DeleteTable(f, v)
        with pytest.raises(go_mirror.DatasetNotFoundError):
            store.GetTable(f, v)


@pytest.mark.gpu
def test_online_batch_set_get_entity(store):   # online_test.go:137-183 (testBatchSetGetEntity; copy.go:99-144's branch)
    import go_mirror
    f, v = _fv()
    tab = store.CreateTable(f, v, go_mirror.VectorType(16))
    max_num = tab.MaxBatchSize()
    assert max_num >= 1000
    rng = np.random.default_rng(9)
    one = rng.standard_normal(16).astype(np.float32)
    assert tab.BatchSet([("e", one)]) is None
    assert tab.Get("e").tobytes() == one.tobytes()
    X = rng.standard_normal((max_num, 16)).astype(np.float32)
    items = [("entity_%d" % i, X[i]) for i in range(max_num)]
    assert tab.BatchSet(items) is None
    for i in rng.integers(0, max_num, size=200):
        assert tab.Get("entity_%d" % i).tobytes() == X[i].tobytes()
    assert tab.BatchSet([]) is None
    store.DeleteTable(f, v)


@pytest.mark.gpu
def test_a_fresh_provider_object_per_request_sees_the_same_tables(store):
    """serving/serving.go:779-794 builds a new provider per request: engine state lives in the process-global registry"""
    import go_mirror
    f, v = _fv()
    tab = store.CreateIndex(f, v, go_mirror.VectorType(4))
    tab.Set("a", np.array([1, 0, 0, 0], dtype=np.float32))
    tab.Set("b", np.array([0, 1, 0, 0], dtype=np.float32))
    got = []

    def request():
        s2 = go_mirror.Mi355xOnlineStore()          # a fresh provider object (ehx_init is idempotent)
        t2 = s2.GetTable(f, v)
        got.append(t2.Nearest(f, v, np.array([0.9, 0.1, 0, 0], dtype=np.float32), 1))
    ts = [threading.Thread(target=request) for _ in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert got == [["a"]] * 8
    store.DeleteIndex(f, v)
