"""gRPC shim: the embeddingstore service (embeddinghub/embeddingstore/embedding_store.proto, server.cc)
served by the MI355X engine — SURVEY.md §8f rank 1.

    python -m embeddinghub_amd.rpc.server 0.0.0.0:7462

`embedding_store_pb2` / `embedding_store_pb2_grpc` carry the reference's wire contract (package, service,
method and field names and numbers) built at import time with the protobuf runtime — the reference
generates these modules with protoc inside Bazel, which this environment does not have.  `client` is
the reference SDK's `EmbeddingHubClient` surface on top of them.
"""
