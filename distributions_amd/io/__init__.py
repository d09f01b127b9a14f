"""Data formats either side of the row-update path: the reference's protobuf
messages (schema_pb2) and its length-prefixed / json streams (stream)."""
