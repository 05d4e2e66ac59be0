"""The embeddingstore gRPC service over the MI355X engine.

Replaces embeddinghub/embeddingstore/server.cc:65-268 (EmbeddingHubService) — same nine RPCs, same
status codes and messages — with the engine's spaces behind it instead of RocksDB + hnswlib (plus ONE additive RPC,
MultiNearestNeighbor: the batched lookup the reference documents, docs/inference.md:17-22, and never shipped):

  * unknown space -> NOT_FOUND "Not found" (server.cc:88, 103, 122, 140, 160, 178, 222);
  * NearestNeighbor: exactly one of key / embedding, else INVALID_ARGUMENT with the reference's texts
    (server.cc:183-189); by key: ask for num+1, drop the key itself, else drop the last (198-207);
  * writes to a frozen space -> FAILED_PRECONDITION "Cannot write to immutable space" (125-127, 144-146);
  * Get / MultiGet of a key that is not there -> OK with an empty embedding (storage.cc:28-36 ignores the
    lookup status); CreateSpace of an existing name returns the existing space (embedding_store.cc:33-36);
    DeleteSpace of an unknown name is OK (64-69); Download streams in key order (RocksDB iteration order).
Divergences, all where the reference has undefined behaviour: an embedding whose length is not the
space's dims -> INVALID_ARGUMENT; NearestNeighbor by a key that is not stored -> NOT_FOUND; fewer
stored rows than `num` -> a shorter list.

The reference serialises every handler with one process-wide mutex (server.cc:67 etc.); here handlers
run concurrently on the gRPC thread pool and concurrent NearestNeighbor calls are coalesced into one
device batch by the engine's micro-batcher (ehx_knn), which is where batch=1024 comes from when the
callers are single-query RPCs.

The servicer talks to a *store* (create_space / get_space / delete_space returning space objects with
set / set_batch / get / freeze / len / key_of / nearest); `EngineStore` is the engine-backed one.
"""
import argparse
import queue
import sys
import threading
from concurrent import futures

import grpc
import numpy as np

from . import embedding_store_pb2 as pb
from . import embedding_store_pb2_grpc as pb_grpc


class SpaceNotWritable(Exception):
    pass


class KeyNotFound(Exception):
    pass


class EngineSpace:
    """One embeddingstore space (its single 'initial' version, server.cc:28) on the engine."""

    def __init__(self, space):
        self._s = space
        self.dims = space.dims

    def set(self, key, vec):
        self.set_batch([key], [vec])

    def set_batch(self, keys, vecs):
        from .. import _lib
        try:
            self._s.set_batch(keys, np.asarray(vecs, dtype=np.float32).reshape(len(keys), self.dims))
        except _lib.EhxError as e:
            if e.code == _lib.EIMMUTABLE:
                raise SpaceNotWritable()
            raise

    def get(self, key):
        from .. import _lib
        try:
            return self._s.get(key)
        except _lib.EhxError as e:
            if e.code == _lib.ENOTFOUND:
                return None
            raise

    def freeze(self):
        self._s.freeze()

    def __len__(self):
        return len(self._s)

    def keys_sorted(self):
        ks = [self._s.key_of(i) for i in range(len(self._s))]
        return sorted(ks, key=lambda k: k.encode())

    def nearest(self, num, key="", embedding=None):
        from ..space import nearest_neighbor_rpc
        code, keys = nearest_neighbor_rpc(self._s, num, key=key, embedding=embedding)
        if code == 5:
            raise KeyNotFound()
        return keys

    def nearest_many(self, num, embeddings):
        """by-embedding lookups of one stream window as ONE engine call (n x dims matrix -> n key lists)"""
        return self._s.knn_keys(np.asarray(embeddings, dtype=np.float32).reshape(-1, self.dims), num)

    def drop(self):
        self._s.drop()


class EngineStore:
    """name -> space, in the engine's process-global registry (ehx_space_create / open / drop)."""

    def __init__(self, metric=None, **space_kw):
        import embeddinghub_amd as ehx
        self._ehx = ehx
        self._metric = ehx.METRIC_L2SQ if metric is None else metric  # index.cc:13: the embeddingstore is L2
        self._kw = space_kw
        self._mu = threading.Lock()
        self._spaces = {}

    def create_space(self, name, dims):
        with self._mu:
            if name in self._spaces:
                return self._spaces[name]
            sp = EngineSpace(self._ehx.Space(name, dims, metric=self._metric, **self._kw))
            self._spaces[name] = sp
            return sp

    def get_space(self, name):
        with self._mu:
            return self._spaces.get(name)

    def delete_space(self, name):
        with self._mu:
            sp = self._spaces.pop(name, None)
        if sp is not None:
            sp.drop()


def _values(embedding):
    return np.fromiter(embedding.values, dtype=np.float32, count=len(embedding.values))


class EmbeddingHubService(pb_grpc.EmbeddingHubServicer):
    MULTISET_CHUNK = 4096  # rows handed to the engine per set_batch while a MultiSet stream is consumed

    def __init__(self, store):
        self._store = store

    # ---- helpers -------------------------------------------------------------------------------
    def _space(self, name, context):
        sp = self._store.get_space(name)
        if sp is None:
            context.abort(grpc.StatusCode.NOT_FOUND, "Not found")
        return sp

    def _checked(self, sp, embedding, context):
        v = _values(embedding)
        if v.shape[0] != sp.dims:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT,
                          "embedding has %d values, space has %d dims" % (v.shape[0], sp.dims))
        return v

    @staticmethod
    def _write(fn, context):
        try:
            fn()
        except SpaceNotWritable:
            context.abort(grpc.StatusCode.FAILED_PRECONDITION, "Cannot write to immutable space")

    # ---- the nine RPCs (server.h:24-59) ----------------------------------------------------------
    def CreateSpace(self, request, context):
        if request.dims == 0:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "dims must be positive")
        self._store.create_space(request.name, request.dims)
        return pb.CreateSpaceResponse()

    def DeleteSpace(self, request, context):
        self._store.delete_space(request.name)
        return pb.DeleteSpaceResponse()

    def FreezeSpace(self, request, context):
        self._space(request.name, context).freeze()
        return pb.FreezeSpaceResponse()

    def Set(self, request, context):
        sp = self._space(request.space, context)
        v = self._checked(sp, request.embedding, context)
        self._write(lambda: sp.set(request.key, v), context)
        return pb.SetResponse()

    def Get(self, request, context):
        sp = self._space(request.space, context)
        v = sp.get(request.key)
        return pb.GetResponse(embedding=pb.Embedding(values=[] if v is None else v.tolist()))

    def MultiSet(self, request_iterator, context):
        # the stream may interleave spaces (server.cc:134-150 looks the space up per message); rows are
        # handed to the engine in chunks per space, in arrival order, so a repeated key keeps its last value
        pending = {}

        def flush(name):
            sp, keys, vecs = pending.pop(name)
            if keys:
                self._write(lambda: sp.set_batch(keys, vecs), context)

        for req in request_iterator:
            entry = pending.get(req.space)
            if entry is None:
                entry = pending[req.space] = (self._space(req.space, context), [], [])
            sp, keys, vecs = entry
            vecs.append(self._checked(sp, req.embedding, context))
            keys.append(req.key)
            if len(keys) >= self.MULTISET_CHUNK:
                flush(req.space)
        for name in list(pending):
            flush(name)
        return pb.MultiSetResponse()

    def MultiGet(self, request_iterator, context):
        for req in request_iterator:
            v = self._space(req.space, context).get(req.key)
            yield pb.MultiGetResponse(embedding=pb.Embedding(values=[] if v is None else v.tolist()))

    def NearestNeighbor(self, request, context):
        sp, has_key = self._nn_validate(request, context)
        try:
            if has_key:
                keys = sp.nearest(request.num, key=request.key)
            else:
                keys = sp.nearest(request.num, embedding=self._checked(sp, request.embedding, context))
        except KeyNotFound:
            context.abort(grpc.StatusCode.NOT_FOUND, "Not found")
        return pb.NearestNeighborResponse(keys=keys)

    MULTI_NN_WINDOW = 1024  # requests of one stream answered by one engine call (the bench's batch size)

    def _nn_validate(self, request, context):
        """the checks of NearestNeighbor (server.cc:178-189), shared by the unary and the streamed form"""
        sp = self._space(request.space, context)
        has_key = request.key != ""
        has_vec = len(request.embedding.values) != 0
        if has_key and has_vec:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "Key and embedding cannot both be set")
        if not has_key and not has_vec:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "Key or embedding must be set")
        if request.num < 0:
            context.abort(grpc.StatusCode.INVALID_ARGUMENT, "num must not be negative")
        return sp, has_key

    def MultiNearestNeighbor(self, request_iterator, context):
        """Additive RPC (embedding_store_pb2.ADDITIVE_METHODS): NearestNeighbor over a stream, answers in request
        order.  Requests are taken off the stream as they arrive — a reader thread feeds a queue, so a client that
        waits for an answer before sending its next request is served at once — and whatever has accumulated (up to
        MULTI_NN_WINDOW) is answered together: by-embedding requests of one (space, num) go to the engine as ONE batch
        (`nearest_many`), by-key requests one by one.  Every request gets the unary RPC's checks; the first failing one
        ends the stream with the unary RPC's status, after the requests before it have been answered.  The reader
        thread ends with the handler (abort, cancelled client, normal end): it never blocks on a queue nobody drains."""
        inbox = queue.Queue(maxsize=4 * self.MULTI_NN_WINDOW)
        end = object()
        stop = threading.Event()  # set when the handler is done (normally, by an abort, or by a cancelled client)

        def put(item):
            while not stop.is_set():
                try:
                    inbox.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    if not context.is_active():
                        return False
            return False

        def reader():
            try:
                for req in request_iterator:
                    if not put(req):
                        return  # nobody drains the queue any more: do not pile requests up, do not block for ever
            except Exception as exc:  # noqa: BLE001  (a cancelled stream: hand the error to the consumer)
                put(exc)
            put(end)
        threading.Thread(target=reader, daemon=True).start()

        class _Failed(Exception):
            def __init__(self, code, text):
                self.code, self.text = code, text

        class _Deferring:  # the unary checks, with the abort handed back instead of raised through the window
            @staticmethod
            def abort(code, text):
                raise _Failed(code, text)
        try:
            done = False
            while not done:
                window = [inbox.get()]
                while len(window) < self.MULTI_NN_WINDOW:
                    try:
                        window.append(inbox.get_nowait())
                    except queue.Empty:
                        break
                if window[-1] is end:
                    done = True
                    window.pop()
                failed = None
                answers = []
                groups = {}  # (space name, num) -> (space, [positions], [vectors])
                for pos, req in enumerate(window):
                    try:
                        if isinstance(req, Exception):
                            raise _Failed(grpc.StatusCode.CANCELLED, "request stream failed: %r" % (req,))
                        sp, has_key = self._nn_validate(req, _Deferring)
                        if has_key:
                            try:
                                answers.append(sp.nearest(req.num, key=req.key))
                            except KeyNotFound:
                                raise _Failed(grpc.StatusCode.NOT_FOUND, "Not found")
                        else:
                            v = self._checked(sp, req.embedding, _Deferring)
                            g = groups.setdefault((req.space, req.num), (sp, [], []))
                            g[1].append(pos)
                            g[2].append(v)
                            answers.append(None)
                    except _Failed as f:  # the requests before the failing one are still answered, then the stream ends
                        failed = f
                        break
                for (_, num), (sp, positions, vecs) in groups.items():
                    many = getattr(sp, "nearest_many", None)
                    if many is not None and len(vecs) > 1:
                        for pos, keys in zip(positions, many(num, np.stack(vecs))):
                            answers[pos] = keys
                    else:
                        for pos, v in zip(positions, vecs):
                            answers[pos] = sp.nearest(num, embedding=v)
                for keys in answers:
                    yield pb.NearestNeighborResponse(keys=keys)
                if failed is not None:
                    context.abort(failed.code, failed.text)
        finally:
            stop.set()

    def Download(self, request, context):
        sp = self._space(request.space, context)
        for key in sp.keys_sorted():
            v = sp.get(key)
            if v is not None:
                yield pb.DownloadResponse(key=key, embedding=pb.Embedding(values=v.tolist()))


def make_server(store, address, max_workers=64):
    """gRPC sync server with `max_workers` handler threads: that many NearestNeighbor calls can be inside the
    engine at once and get coalesced into one device batch."""
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers),
                         options=(("grpc.max_receive_message_length", 64 << 20),
                                  ("grpc.max_send_message_length", 64 << 20)))
    pb_grpc.add_EmbeddingHubServicer_to_server(EmbeddingHubService(store), server)
    port = server.add_insecure_port(address)  # no authentication, like the reference (server.cc:258)
    return server, port


def main(argv=None):
    ap = argparse.ArgumentParser(description="embeddingstore gRPC service on the MI355X engine")
    ap.add_argument("address", nargs="?", default="0.0.0.0:7462")  # main.cc: default port of the reference
    ap.add_argument("--metric", choices=["l2", "ip", "cosine"], default="l2")
    ap.add_argument("--workers", type=int, default=64)
    ap.add_argument("--data-dir", default=None,
                    help="keep every space in append-only logs under this directory and rebuild them on start "
                         "(the reference persists to RocksDB under ./embedding_store.dat, server.cc:249)")
    ap.add_argument("--sync", action="store_true", help="fsync the logs on every write")
    ap.add_argument("--mode", choices=["flat", "graph"], default="flat",
                    help="flat: exact answers (default); graph: the reference's own index (HNSW M=16 efC=200, index.cc:14-15)")
    ap.add_argument("--ef", type=int, default=0, help="graph mode: search ef (0 = the reference's 10)")
    ap.add_argument("--build-batch", type=int, default=0,
                    help="graph mode: rows per concurrent insertion round of MultiSet (0 = one by one, hnswlib's order)")
    ap.add_argument("--search-width", type=int, choices=[0, 1, 2, 4], default=0,
                    help="graph mode: expansions per search step (0 / 1 = hnswlib's order; 2 / 4 = the wide walk, a "
                         "throughput mode — INTEGRATION.md section 5)")
    args = ap.parse_args(argv)
    import embeddinghub_amd as ehx
    metric = {"l2": ehx.METRIC_L2SQ, "ip": ehx.METRIC_IP, "cosine": ehx.METRIC_COSINE}[args.metric]
    kw = {}
    if args.mode == "graph":
        kw = dict(mode=ehx.MODE_GRAPH, ef=args.ef, build_batch=args.build_batch, search_width=args.search_width)
    store = EngineStore(metric=metric, **kw)
    if args.data_dir:
        from .durable import DurableStore
        store = DurableStore(store, args.data_dir, sync=args.sync)
        print("rebuilt %d rows from %s" % (store.rebuilt_rows, args.data_dir), flush=True)
    server, port = make_server(store, args.address, args.workers)
    server.start()
    print("Server listening on %s" % args.address, flush=True)
    server.wait_for_termination()


if __name__ == "__main__":
    sys.exit(main())
