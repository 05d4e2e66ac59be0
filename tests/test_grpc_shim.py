"""The embeddingstore gRPC shim (embeddinghub_amd/rpc): the reference's own end-to-end tests
(embeddinghub/test/integration.py:44-123) and the RPC semantics of server.cc:65-268, restated against
the shim.

CPU leg (`not gpu`): the servicer, wire contract and client run over a store backed by the ORACLE
(tests may use the oracle; the product never does) — this pins the gRPC layer itself.  GPU leg: the
same suite over the engine-backed store, plus parity of NearestNeighbor with the oracle and the
coalescing of concurrent single-query RPCs into device batches."""
import os
import threading
import uuid

import grpc
import numpy as np
import pytest

from embeddinghub_amd.rpc import embedding_store_pb2 as pb
from embeddinghub_amd.rpc import server as srv
from embeddinghub_amd.rpc.client import EmbeddingHubClient
from oracle import pyoracle


# ---- an oracle-backed store (test double for the CPU leg): oracle/oracle_store.py -----------------------
from oracle.oracle_store import OracleSpace, OracleStore  # noqa: E402,F401


def _serve(store):
    server, port = srv.make_server(store, "127.0.0.1:0", max_workers=48)
    server.start()
    client = EmbeddingHubClient(host="127.0.0.1", port=port)
    return server, client


@pytest.fixture
def oracle_client():
    server, client = _serve(OracleStore())
    yield client
    client.close()
    server.stop(0)


@pytest.fixture
def engine_client():
    pytest.importorskip("embeddinghub_amd")
    server, client = _serve(srv.EngineStore())
    yield client
    client.close()
    server.stop(0)


# ---- the suite (runs on both legs) -------------------------------------------------------------------
def check_set_get(c):  # integration.py:44-49
    space = uuid.uuid4()
    c.create_space(space, 3)
    c.set(space, "a", [1, 2, 3])
    assert c.get(space, "a") == [1, 2, 3]


def check_immutable_set(c):  # integration.py:52-61
    space = uuid.uuid4()
    c.create_space(space, 3)
    c.set(space, "a", [1, 2, 3])
    assert c.get(space, "a") == [1, 2, 3]
    c.freeze_space(space)
    with pytest.raises(TypeError):
        c.set(space, "a", [1, 2, 3])
    with pytest.raises(grpc.RpcError) as e:  # MultiSet maps to the same status (server.cc:144-146)
        c.multiset(space, {"b": [1, 1, 1]})
    assert e.value.code() == grpc.StatusCode.FAILED_PRECONDITION
    assert e.value.details() == "Cannot write to immutable space"


def check_multiset_get_multiget_download(c):  # integration.py:64-123
    space = uuid.uuid4()
    embs = {"a": [1, 2, 3], "b": [3, 2, 1]}
    c.create_space(space, 3)
    c.multiset(space, embs)
    for key, emb in embs.items():
        assert c.get(space, key) == emb
    assert dict(zip(embs.keys(), c.multiget(space, embs.keys()))) == embs
    assert {k: v for k, v in c.download(space)} == embs
    assert [k for k, _ in c.download(space)] == sorted(embs)  # RocksDB iteration order


def check_multi_space(c):  # integration.py:91-106
    tag = str(uuid.uuid4())
    embs = {"a" + tag: [1, 2, 3], "b" + tag: [3, 2, 1]}
    for space in embs:
        c.create_space(space, 3)
    for space, emb in embs.items():
        c.set(space, "key", emb)
    for space, emb in embs.items():
        assert c.get(space, "key") == emb


def check_status_codes(c):  # server.cc:88, 178, 183-189; storage.cc:28-36; embedding_store.cc:33-36, 64-69
    space = str(uuid.uuid4())
    for call in (lambda: c.get("no-such-space", "a"), lambda: c.set("no-such-space", "a", [1.0]),
                 lambda: c.freeze_space("no-such-space"), lambda: c.nearest_neighbor("no-such-space", 1, key="a"),
                 lambda: list(c.download("no-such-space")), lambda: list(c.multiget("no-such-space", ["a"]))):
        with pytest.raises(grpc.RpcError) as e:
            call()
        assert e.value.code() == grpc.StatusCode.NOT_FOUND and e.value.details() == "Not found"
    c.create_space(space, 3)
    c.create_space(space, 3)  # idempotent: returns the existing space
    c.set(space, "a", [0, 1, 0])
    assert list(c.get(space, "missing")) == []  # OK with an empty embedding
    with pytest.raises(grpc.RpcError) as e:
        c.nearest_neighbor(space, 1, key="a", embedding=[0, 1, 0])
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
    assert e.value.details() == "Key and embedding cannot both be set"
    with pytest.raises(grpc.RpcError) as e:
        c.nearest_neighbor(space, 1)
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT and e.value.details() == "Key or embedding must be set"
    with pytest.raises(grpc.RpcError) as e:
        c.set(space, "b", [1, 2])  # wrong length: undefined behaviour in the reference, an error here
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
    c.delete_space(space)
    c.delete_space(space)  # unknown name: still OK
    with pytest.raises(grpc.RpcError) as e:
        c.get(space, "a")
    assert e.value.code() == grpc.StatusCode.NOT_FOUND


def check_nearest_neighbor(c):  # index_test.cc:17-49 data through the RPC; server.cc:193-207 by key
    space = uuid.uuid4()
    c.create_space(space, 3)
    c.multiset(space, {"a": [0, 1, 0], "b": [1, 1, 0], "c": [1, 0, 0]})
    assert list(c.nearest_neighbor(space, 1, embedding=[0, 1, 0])) == ["a"]
    assert list(c.nearest_neighbor(space, 2, embedding=[0, 1, 0])) == ["a", "b"]
    assert list(c.nearest_neighbor(space, 2, key="a")) == ["b", "c"]  # the key itself is removed
    fut = c.nearest_neighbor(space, 1, key="c", wait=False)
    assert list(fut.result()) == ["b"]
    c.set(space, "a", [0, -1, 0])  # update in place (index_test.cc:39-49)
    assert list(c.nearest_neighbor(space, 1, embedding=[0, 1, 0])) == ["b"]


def check_multi_nearest_neighbor(c):  # docs/inference.md:17-22 (`multi_nearest_neighbor`): the additive stream RPC
    a, b = uuid.uuid4(), uuid.uuid4()
    rng = np.random.default_rng(4)
    c.create_space(a, 6)
    c.create_space(b, 3)
    XA = rng.standard_normal((300, 6)).astype(np.float32)
    c.multiset(a, (("a%d" % i, XA[i].tolist()) for i in range(300)))
    c.multiset(b, {"a": [0, 1, 0], "b": [1, 1, 0], "c": [1, 0, 0]})
    Q = rng.standard_normal((70, 6)).astype(np.float32)
    want = [list(c.nearest_neighbor(a, 4, embedding=q.tolist())) for q in Q]
    assert c.multi_nearest_neighbor(a, 4, embeddings=[q.tolist() for q in Q]) == want     # answers in request order
    assert c.multi_nearest_neighbor(a, 4, vectors=[q.tolist() for q in Q[:3]]) == want[:3]  # the documented keyword
    assert c.multi_nearest_neighbor(b, 2, keys=["a", "c"]) == [["b", "c"], ["b", "a"]]      # by key: itself removed
    assert c.multi_nearest_neighbor(a, 0, embeddings=[Q[0].tolist()]) == [[]]
    assert c.multi_nearest_neighbor(a, 3, embeddings=[]) == []
    # one stream may mix spaces, by-key and by-embedding requests, and different num
    mixed = [pb.NearestNeighborRequest(space=str(a), num=4, embedding=pb.Embedding(values=Q[0].tolist())),
             pb.NearestNeighborRequest(space=str(b), num=1, key="c"),
             pb.NearestNeighborRequest(space=str(a), num=2, embedding=pb.Embedding(values=Q[1].tolist())),
             pb.NearestNeighborRequest(space=str(b), num=2, embedding=pb.Embedding(values=[0, 1, 0]))]
    got = [list(r.keys) for r in c._stub.MultiNearestNeighbor(iter(mixed))]
    assert got == [want[0], ["b"], want[1][:2], ["a", "b"]]
    # a client that waits for each answer before sending the next request is served at once (no window to fill)
    import queue as _q
    outbox, answers = _q.Queue(), []

    def requests():
        while True:
            r = outbox.get()
            if r is None:
                return
            yield r
    stream = c._stub.MultiNearestNeighbor(requests())
    for i in range(5):
        outbox.put(pb.NearestNeighborRequest(space=str(a), num=4, embedding=pb.Embedding(values=Q[i].tolist())))
        answers.append(list(next(stream).keys))
    outbox.put(None)
    assert list(stream) == [] and answers == want[:5]
    # the unary RPC's checks and status codes, per request
    for bad, code in ((pb.NearestNeighborRequest(space=str(a), num=1), grpc.StatusCode.INVALID_ARGUMENT),
                      (pb.NearestNeighborRequest(space="no such space", num=1, key="x"), grpc.StatusCode.NOT_FOUND),
                      (pb.NearestNeighborRequest(space=str(a), num=1, embedding=pb.Embedding(values=[1.0])),
                       grpc.StatusCode.INVALID_ARGUMENT)):
        with pytest.raises(grpc.RpcError) as e:
            list(c._stub.MultiNearestNeighbor(iter([mixed[0], bad])))
        assert e.value.code() == code
        # the requests before the failing one are answered before the stream ends with its status
        got_before = []
        with pytest.raises(grpc.RpcError) as e:
            for r in c._stub.MultiNearestNeighbor(iter([mixed[0], mixed[2], bad, mixed[0]])):
                got_before.append(list(r.keys))
        assert e.value.code() == code and got_before == [want[0], want[1][:2]]
    # a long stream that fails early must not leave its reader thread blocked on a full queue
    before = threading.active_count()
    many = [mixed[0]] * 6000
    with pytest.raises(grpc.RpcError):
        list(c._stub.MultiNearestNeighbor(iter([pb.NearestNeighborRequest(space=str(a), num=1)] + many)))
    import time as _t
    deadline = _t.time() + 10
    while threading.active_count() > before and _t.time() < deadline:
        _t.sleep(0.05)
    assert threading.active_count() <= before, "a MultiNearestNeighbor reader thread outlived its failed stream"


SUITE = [check_set_get, check_immutable_set, check_multiset_get_multiget_download, check_multi_space,
         check_status_codes, check_nearest_neighbor, check_multi_nearest_neighbor]


@pytest.mark.parametrize("check", SUITE, ids=lambda f: f.__name__)
def test_shim_over_oracle_store(oracle_client, check):
    check(oracle_client)


def test_wire_contract_matches_the_reference_proto():
    # field numbers / types of embedding_store.proto:84-112 as bytes on the wire
    r = pb.NearestNeighborRequest(num=5, space="s", key="k", embedding=pb.Embedding(values=[1.0, 2.0]))
    assert r.SerializeToString() == (b"\x08\x05" b"\x12\x01s" b"\x1a\x01k" b"\x22\x0a\x0a\x08" +
                                     np.array([1.0, 2.0], dtype="<f4").tobytes())
    assert pb.Embedding(values=[1.5]).SerializeToString() == b"\x0a\x04" + np.float32(1.5).tobytes()  # packed
    assert pb.DESCRIPTOR.package == "featureform.embedding.proto"
    assert sorted(m.name for m in pb.DESCRIPTOR.services_by_name["EmbeddingHub"].methods) == sorted(pb.METHODS)
    # the reference's nine RPCs (embedding_store.proto:9-19) plus ONE additive stream built from its own messages
    assert sorted(pb.REFERENCE_METHODS) == ["CreateSpace", "DeleteSpace", "Download", "FreezeSpace", "Get", "MultiGet",
                                            "MultiSet", "NearestNeighbor", "Set"]
    assert pb.ADDITIVE_METHODS == {"MultiNearestNeighbor": ("NearestNeighborRequest", "NearestNeighborResponse",
                                                            True, True)}


@pytest.mark.gpu
@pytest.mark.parametrize("check", SUITE, ids=lambda f: f.__name__)
def test_shim_over_engine_store(engine_client, check):
    check(engine_client)


@pytest.mark.gpu
def test_engine_nearest_neighbor_matches_oracle_and_coalesces(engine_client):
    import embeddinghub_amd as ehx
    c = engine_client
    rng = np.random.default_rng(11)
    n, d = 3000, 64
    X = rng.standard_normal((n, d)).astype(np.float32)
    space = "rpc-parity-%s" % uuid.uuid4()
    c.create_space(space, d)
    c.multiset(space, ((("k%d" % i), X[i].tolist()) for i in range(n)))
    Q = rng.standard_normal((96, d)).astype(np.float32)
    oids, _, _ = pyoracle.exhaustive(X, Q, 10, pyoracle.METRIC_L2)
    # 96 single-query RPCs in flight at once
    futs = [c.nearest_neighbor(space, 10, embedding=q.tolist(), wait=False) for q in Q]
    got = [list(f.result()) for f in futs]
    assert got == [["k%d" % i for i in row] for row in oids]
    by_key = list(c.nearest_neighbor(space, 5, key="k7"))
    o7, _, _ = pyoracle.exhaustive(X, X[7:8], 6, pyoracle.METRIC_L2)
    assert by_key == ["k%d" % i for i in o7[0] if i != 7][:5]
    st = ehx.Space.open(space).stats()
    assert st["n_queries"] >= 96


# ---- durable store + rebuild-on-load (embeddinghub_amd/rpc/durable.py) -------------------------------
def _durable_roundtrip(make_inner, tmp_path, n=3000, d=24):
    from embeddinghub_amd.rpc.durable import DurableStore, value_header
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d)).astype(np.float32)
    st = DurableStore(make_inner(), str(tmp_path))
    server, port = srv.make_server(st, "127.0.0.1:0", max_workers=8)
    server.start()
    c = EmbeddingHubClient(host="127.0.0.1", port=port)
    c.create_space("dur", d)
    c.multiset("dur", ((("k%d" % i), X[i].tolist()) for i in range(n)))
    X[7] = rng.standard_normal(d).astype(np.float32)
    c.set("dur", "k7", X[7].tolist())  # upsert: the later record wins on replay
    c.create_space("gone", d)
    c.set("gone", "x", X[0].tolist())
    c.delete_space("gone")
    c.create_space("ice", d)
    c.set("ice", "only", X[1].tolist())
    c.freeze_space("ice")
    before = [list(c.nearest_neighbor("dur", 5, embedding=X[i].tolist())) for i in (3, 7, 11)]
    c.close()
    server.stop(0)
    st.close()
    # the stored value is the reference's serialized Embedding (serializer.cc:19-26)
    rec = open(os.path.join(st._space_dir("dur"), "values.dat"), "rb").read(len(value_header(d)) + 4 * d)
    assert rec == pb.Embedding(values=X[0].tolist()).SerializeToString()
    # a crash mid-append leaves a torn tail: cut back to the last complete record
    with open(os.path.join(st._space_dir("dur"), "values.dat"), "ab") as f:
        f.write(b"\x0a\x60garbage")
    st2 = DurableStore(make_inner(), str(tmp_path))  # restart: rebuild through the bulk write path
    assert st2.rebuilt_rows == n + 1 + 1
    server, port = srv.make_server(st2, "127.0.0.1:0", max_workers=8)
    server.start()
    c = EmbeddingHubClient(host="127.0.0.1", port=port)
    assert list(c.get("dur", "k7")) == X[7].tolist() and list(c.get("dur", "k2999")) == X[2999].tolist()
    assert [list(c.nearest_neighbor("dur", 5, embedding=X[i].tolist())) for i in (3, 7, 11)] == before
    with pytest.raises(grpc.RpcError) as e:
        c.get("gone", "x")
    assert e.value.code() == grpc.StatusCode.NOT_FOUND
    with pytest.raises(TypeError):
        c.set("ice", "only", X[2].tolist())  # still frozen after the restart
    c.close()
    server.stop(0)
    st2.close()


def test_durable_store_rebuilds_on_load_over_oracle_store(tmp_path):
    _durable_roundtrip(OracleStore, tmp_path)


@pytest.mark.gpu
def test_durable_store_rebuilds_on_load_over_engine(tmp_path):
    pytest.importorskip("embeddinghub_amd")
    made = []

    def make():
        for s_ in made:  # "restart": the previous process's spaces are gone
            for name in ("dur", "ice"):
                s_.delete_space(name)
        made.append(srv.EngineStore())
        return made[-1]
    _durable_roundtrip(make, tmp_path, n=20000, d=64)


def test_durable_store_crash_consistency(tmp_path):
    """(1) a directory left behind by a crash between the DELETE record and the removal of the files is
    not adopted by a later CreateSpace of the same name; (2) a write the engine refuses is not left in the log."""
    import shutil
    from embeddinghub_amd.rpc.durable import DurableStore
    from embeddinghub_amd.rpc.server import SpaceNotWritable
    rng = np.random.default_rng(3)
    st = DurableStore(OracleStore(), str(tmp_path))
    sp = st.create_space("a", 8)
    sp.set_batch(["x", "y"], rng.standard_normal((2, 8)).astype(np.float32))
    keep = str(tmp_path / "saved")
    shutil.copytree(st._space_dir("a"), keep)
    st.delete_space("a")
    shutil.copytree(keep, st._space_dir("a"))          # "the crash": the files survive the catalog's DELETE record
    sp = st.create_space("a", 8)                        # same name, fresh space
    assert len(sp) == 0
    sp.set("z", rng.standard_normal(8).astype(np.float32))
    sp.freeze()
    with pytest.raises(SpaceNotWritable):
        sp.set("w", rng.standard_normal(8).astype(np.float32))
    st.close()
    st2 = DurableStore(OracleStore(), str(tmp_path))    # restart: only "z" comes back, neither x / y nor the refused w
    sp2 = st2.get_space("a")
    assert len(sp2) == 1 and sp2.keys_sorted() == ["z"]
    st2.close()


def test_durable_freeze_and_delete_do_not_deadlock(tmp_path):
    """FreezeSpace (space lock, then the store's for the catalog record) against DeleteSpace (store lock,
    then the space's) on the same space was a lock-order inversion; both now take the store's lock first.  A stale
    handle's freeze after the delete records nothing, so a later space of the same name does not come back frozen."""
    from embeddinghub_amd.rpc.durable import DurableStore
    rng = np.random.default_rng(5)
    st = DurableStore(OracleStore(), str(tmp_path))
    for rnd in range(40):
        sp = st.create_space("race", 8)
        sp.set("k", rng.standard_normal(8).astype(np.float32))
        ts = [threading.Thread(target=sp.freeze), threading.Thread(target=st.delete_space, args=("race",))]
        for t in (ts if rnd % 2 else ts[::-1]):
            t.start()
        for t in ts:
            t.join(timeout=20)
            assert not t.is_alive(), "FreezeSpace / DeleteSpace deadlocked"
    sp = st.create_space("race", 8)
    stale = sp
    st.delete_space("race")
    stale.freeze()                               # a handle that outlived its space
    sp = st.create_space("race", 8)
    sp.set("fresh", rng.standard_normal(8).astype(np.float32))
    st.close()
    st2 = DurableStore(OracleStore(), str(tmp_path))
    sp2 = st2.get_space("race")
    sp2.set("still-writable", rng.standard_normal(8).astype(np.float32))   # not frozen by the stale handle's record
    assert sp2.keys_sorted() == ["fresh", "still-writable"]
    st2.close()
