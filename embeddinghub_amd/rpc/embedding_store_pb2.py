"""Message classes of embeddinghub/embeddingstore/embedding_store.proto, built with the protobuf
runtime (no protoc here).  Wire-compatible with the reference: package
`featureform.embedding.proto`, the same message names, field names, numbers and types
(embedding_store.proto:21-112)."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
PACKAGE = "featureform.embedding.proto"
SERVICE = "EmbeddingHub"

_STRING, _UINT32, _INT32, _FLOAT, _MESSAGE = _F.TYPE_STRING, _F.TYPE_UINT32, _F.TYPE_INT32, _F.TYPE_FLOAT, _F.TYPE_MESSAGE
_EMB = "." + PACKAGE + ".Embedding"

# message -> [(field, number, type, repeated, type_name)]
_MESSAGES = {
    "DeleteSpaceRequest": [("name", 1, _STRING, False, None)],
    "DeleteSpaceResponse": [],
    "CreateSpaceRequest": [("name", 1, _STRING, False, None), ("dims", 2, _UINT32, False, None)],
    "CreateSpaceResponse": [],
    "FreezeSpaceRequest": [("name", 1, _STRING, False, None)],
    "FreezeSpaceResponse": [],
    "SetRequest": [("key", 1, _STRING, False, None), ("embedding", 2, _MESSAGE, False, _EMB),
                   ("space", 3, _STRING, False, None)],
    "SetResponse": [],
    "GetRequest": [("key", 1, _STRING, False, None), ("space", 2, _STRING, False, None)],
    "GetResponse": [("embedding", 1, _MESSAGE, False, _EMB)],
    "MultiSetRequest": [("key", 1, _STRING, False, None), ("embedding", 2, _MESSAGE, False, _EMB),
                        ("space", 3, _STRING, False, None)],
    "MultiSetResponse": [],
    "MultiGetRequest": [("key", 1, _STRING, False, None), ("space", 2, _STRING, False, None)],
    "MultiGetResponse": [("embedding", 1, _MESSAGE, False, _EMB)],
    "NearestNeighborRequest": [("num", 1, _INT32, False, None), ("space", 2, _STRING, False, None),
                               ("key", 3, _STRING, False, None), ("embedding", 4, _MESSAGE, False, _EMB)],
    "NearestNeighborResponse": [("keys", 1, _STRING, True, None)],
    "DownloadRequest": [("space", 1, _STRING, False, None)],
    "DownloadResponse": [("key", 1, _STRING, False, None), ("embedding", 2, _MESSAGE, False, _EMB)],
    "Embedding": [("values", 1, _FLOAT, True, None)],
    # catalog entries of the reference's metadata store (embedding_store_meta.proto:9-19), used by the durable log
    "SpaceEntry": [("path", 1, _STRING, False, None), ("name", 2, _STRING, False, None)],
    "VersionEntry": [("path", 1, _STRING, False, None), ("space", 2, _STRING, False, None),
                     ("name", 3, _STRING, False, None), ("dims", 4, _INT32, False, None)],
}

# rpc -> (request, response, client streaming, server streaming)   embedding_store.proto:9-19
METHODS = {
    "CreateSpace": ("CreateSpaceRequest", "CreateSpaceResponse", False, False),
    "DeleteSpace": ("DeleteSpaceRequest", "DeleteSpaceResponse", False, False),
    "FreezeSpace": ("FreezeSpaceRequest", "FreezeSpaceResponse", False, False),
    "Set": ("SetRequest", "SetResponse", False, False),
    "Get": ("GetRequest", "GetResponse", False, False),
    "MultiSet": ("MultiSetRequest", "MultiSetResponse", True, False),
    "MultiGet": ("MultiGetRequest", "MultiGetResponse", True, True),
    "NearestNeighbor": ("NearestNeighborRequest", "NearestNeighborResponse", False, False),
    "Download": ("DownloadRequest", "DownloadResponse", False, True),
}
# Additive (NOT in the reference's proto): the batched lookup its docs promise (`multi_nearest_neighbor`,
# embeddinghub/docs/inference.md:17-22) but the SDK and the server never got — a stream of NearestNeighbor requests
# answered by a stream of responses in request order, built from the reference's own messages only.
ADDITIVE_METHODS = {
    "MultiNearestNeighbor": ("NearestNeighborRequest", "NearestNeighborResponse", True, True),
}
REFERENCE_METHODS = dict(METHODS)
METHODS.update(ADDITIVE_METHODS)


def _file_descriptor():
    f = descriptor_pb2.FileDescriptorProto(name="embeddinghub_amd/embedding_store.proto", package=PACKAGE,
                                           syntax="proto3")
    for name, fields in _MESSAGES.items():
        m = f.message_type.add(name=name)
        for fname, number, ftype, repeated, type_name in fields:
            fd = m.field.add(name=fname, number=number, type=ftype,
                             label=_F.LABEL_REPEATED if repeated else _F.LABEL_OPTIONAL)
            if type_name:
                fd.type_name = type_name
    svc = f.service.add(name=SERVICE)
    for rpc, (req, resp, cs, ss) in METHODS.items():
        svc.method.add(name=rpc, input_type=".%s.%s" % (PACKAGE, req), output_type=".%s.%s" % (PACKAGE, resp),
                       client_streaming=cs, server_streaming=ss)
    return f


_pool = descriptor_pool.DescriptorPool()
DESCRIPTOR = _pool.Add(_file_descriptor()) or _pool.FindFileByName("embeddinghub_amd/embedding_store.proto")
for _name in _MESSAGES:
    globals()[_name] = message_factory.GetMessageClass(_pool.FindMessageTypeByName(PACKAGE + "." + _name))
del _name
