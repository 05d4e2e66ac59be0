"""Client for the embeddingstore service with the surface of the reference SDK's `EmbeddingHubClient`
(embeddinghub/sdk/python/embeddinghub.py:27-275): create_space / freeze_space / set / get / multiset /
multiget / nearest_neighbor / download / close, `wait=False` returning a future, a write to a frozen
space raising TypeError (embeddinghub.py:113-116).  Works against the reference server as well — the
wire contract is the same (embedding_store_pb2)."""
import concurrent.futures
from collections.abc import Mapping

import grpc

from . import embedding_store_pb2 as pb
from . import embedding_store_pb2_grpc as pb_grpc


class _Mapped:
    """A gRPC future whose result() is passed through `fn` (everything else is the wrapped future's)."""

    def __init__(self, future, fn):
        self._future, self._fn = future, fn

    def __getattr__(self, name):
        return getattr(self._future, name)

    def result(self, timeout=None):
        return self._fn(self._future.result(timeout))

    def add_done_callback(self, callback):
        def relay(done):
            out = concurrent.futures.Future()
            try:
                out.set_result(self._fn(done.result()))
            except Exception as exc:  # deliver the RPC error through the future, as gRPC does
                out.set_exception(exc)
            callback(out)
        self._future.add_done_callback(relay)


def _finish(future, wait, fn=None):
    if fn is not None:
        future = _Mapped(future, fn)
    return future.result() if wait else future


class EmbeddingHubClient:
    @staticmethod
    def grpc_channel(host="0.0.0.0", port=7462):
        return grpc.insecure_channel("%s:%d" % (host, port), options=(("grpc.enable_http_proxy", 0),))

    def __init__(self, grpc_channel=None, host="0.0.0.0", port=7462):
        self._channel = grpc_channel if grpc_channel is not None else self.grpc_channel(host, port)
        self._stub = pb_grpc.EmbeddingHubStub(self._channel)

    def close(self):
        return self._channel.close()

    def create_space(self, name, dims, wait=True):
        return _finish(self._stub.CreateSpace.future(pb.CreateSpaceRequest(name=str(name), dims=dims)), wait)

    def delete_space(self, name, wait=True):  # (the RPC exists in the proto; the reference SDK never wrapped it)
        return _finish(self._stub.DeleteSpace.future(pb.DeleteSpaceRequest(name=str(name))), wait)

    def freeze_space(self, name, wait=True):
        return _finish(self._stub.FreezeSpace.future(pb.FreezeSpaceRequest(name=str(name))), wait)

    def set(self, space, key, embedding, wait=True):
        req = pb.SetRequest(space=str(space), key=str(key), embedding=pb.Embedding(values=embedding))
        future = self._stub.Set.future(req)
        if wait:
            try:
                future.result()
            except grpc.RpcError as e:
                if e.code() == grpc.StatusCode.FAILED_PRECONDITION:
                    raise TypeError(e.details())
                raise
        return future

    def get(self, space, key, wait=True):
        future = self._stub.Get.future(pb.GetRequest(space=str(space), key=str(key)))
        return _finish(future, wait, lambda r: r.embedding.values)

    def multiset(self, space, embedding_tuples):
        items = embedding_tuples.items() if isinstance(embedding_tuples, Mapping) else embedding_tuples
        self._stub.MultiSet(pb.MultiSetRequest(space=str(space), key=str(k), embedding=pb.Embedding(values=v))
                            for k, v in items)

    def multiget(self, space, keys):
        reqs = (pb.MultiGetRequest(space=str(space), key=str(k)) for k in keys)
        return (r.embedding.values for r in self._stub.MultiGet(reqs))

    def nearest_neighbor(self, space, num, key=None, embedding=None, wait=True):
        req = pb.NearestNeighborRequest(space=str(space), num=num, key=None if key is None else str(key),
                                        embedding=None if embedding is None else pb.Embedding(values=embedding))
        return _finish(self._stub.NearestNeighbor.future(req), wait, lambda r: r.keys)

    def multi_nearest_neighbor(self, space, num, embeddings=None, keys=None, vectors=None):
        """The batched lookup of the reference's docs (`multi_nearest_neighbor(10, vectors=user_vecs)`,
        docs/inference.md:17-22; never implemented there): one key list per embedding (or per key), in order, over
        ONE MultiNearestNeighbor stream — the server answers what has accumulated as one device batch.  Needs this
        package's server (the RPC is additive; the reference server answers UNIMPLEMENTED)."""
        if vectors is not None and embeddings is None:
            embeddings = vectors
        if (embeddings is None) == (keys is None):
            raise ValueError("exactly one of embeddings / keys")
        if embeddings is not None:
            reqs = (pb.NearestNeighborRequest(space=str(space), num=num, embedding=pb.Embedding(values=e))
                    for e in embeddings)
        else:
            reqs = (pb.NearestNeighborRequest(space=str(space), num=num, key=str(k)) for k in keys)
        return [list(r.keys) for r in self._stub.MultiNearestNeighbor(reqs)]

    def download(self, space):
        return ((r.key, r.embedding.values) for r in self._stub.Download(pb.DownloadRequest(space=str(space))))
