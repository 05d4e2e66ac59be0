"""Client stub and server registration for the EmbeddingHub service (embedding_store.proto:9-19) — the
hand-written counterpart of what `grpc_tools.protoc` would generate for the reference."""
import grpc

from . import embedding_store_pb2 as pb

_PREFIX = "/%s.%s/" % (pb.PACKAGE, pb.SERVICE)


def _kind(cs, ss):
    return {(False, False): "unary_unary", (True, False): "stream_unary", (False, True): "unary_stream",
            (True, True): "stream_stream"}[(cs, ss)]


class EmbeddingHubStub:
    def __init__(self, channel):
        for rpc, (req, resp, cs, ss) in pb.METHODS.items():
            factory = getattr(channel, _kind(cs, ss))
            setattr(self, rpc, factory(_PREFIX + rpc, request_serializer=getattr(pb, req).SerializeToString,
                                       response_deserializer=getattr(pb, resp).FromString))


class EmbeddingHubServicer:
    """Subclass and implement the nine methods (server.h:24-59) and the additive MultiNearestNeighbor."""


def add_EmbeddingHubServicer_to_server(servicer, server):
    handlers = {}
    for rpc, (req, resp, cs, ss) in pb.METHODS.items():
        make = getattr(grpc, _kind(cs, ss) + "_rpc_method_handler")
        handlers[rpc] = make(getattr(servicer, rpc), request_deserializer=getattr(pb, req).FromString,
                             response_serializer=getattr(pb, resp).SerializeToString)
    server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler("%s.%s" % (pb.PACKAGE, pb.SERVICE),
                                                                          handlers),))
