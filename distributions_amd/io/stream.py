"""Streams of records on disk, in the reference's file formats
(distributions/io/stream.py): files ending in .gz / .bz2 are compressed
transparently;

- protobuf streams: every record is a little-endian u32 byte count followed
  by that many bytes of a serialized message (stream.py:139-172);
- json streams: '[' on the first line, one compact json document per line
  separated by ',' at the line end, ']' on the last line (stream.py:68-136),
  so that a reader can parse line by line.
"""
import bz2
import gzip
import json
import os
import struct

_LEN = struct.Struct('<I')


def open_compressed(filename, mode='rb'):
    """binary file object; creates the directory when writing"""
    if 'w' in mode or 'a' in mode:
        dirname = os.path.dirname(filename)
        if dirname:
            os.makedirs(dirname, exist_ok=True)
    if 'b' not in mode:
        mode += 'b'
    if filename.endswith('.bz2'):
        return bz2.BZ2File(filename, mode)
    if filename.endswith('.gz'):
        return gzip.GzipFile(filename, mode)
    return open(filename, mode)


# ---------------------------------------------------------------------------
# json

def json_dump(data, filename, **kwargs):
    with open_compressed(filename, 'wb') as f:
        f.write(json.dumps(data, **kwargs).encode('utf-8'))


def json_load(filename):
    with open_compressed(filename, 'rb') as f:
        return json.loads(f.read().decode('utf-8'))


def _compact(item, kwargs):
    kwargs = dict(kwargs, separators=(',', ':'))
    return json.dumps(item, **kwargs).encode('utf-8')


def json_stream_dump(stream, filename, **kwargs):
    with open_compressed(filename, 'wb') as f:
        f.write(b'[')
        for i, item in enumerate(stream):
            f.write(b',\n' if i else b'\n')
            f.write(_compact(item, kwargs))
        f.write(b'\n]')


def json_costream_dump(filename, **kwargs):
    """coroutine form: send(item) per record, close() to finish"""
    with open_compressed(filename, 'wb') as f:
        f.write(b'[')
        count = 0
        try:
            while True:
                item = (yield)
                f.write(b',\n' if count else b'\n')
                f.write(_compact(item, kwargs))
                count += 1
        except GeneratorExit:
            pass
        f.write(b'\n]')


class json_stream_load(object):
    """iterates the records of a file written by json_stream_dump /
    json_costream_dump, one line at a time"""

    def __init__(self, filename):
        self.fd = open_compressed(filename, 'rb')
        if self.fd.readline(2) != b'[\n':
            self.fd.close()
            raise IOError('not a json stream (expected "[" on the first '
                          'line): %s' % filename)

    def __iter__(self):
        return self

    def __next__(self):
        line = self.fd.readline().rstrip(b',\n')
        if line == b']' or not line:
            self.close()
            raise StopIteration
        return json.loads(line.decode('utf-8'))

    next = __next__

    def close(self):
        self.fd.close()


# ---------------------------------------------------------------------------
# protobuf

def protobuf_stream_write(item, fd):
    assert isinstance(item, (bytes, bytearray)), type(item)
    fd.write(_LEN.pack(len(item)))
    fd.write(item)


def protobuf_stream_read(fd):
    """-> the next record's bytes; StopIteration at end of file"""
    head = fd.read(4)
    if len(head) < 4:
        raise StopIteration
    size, = _LEN.unpack(head)
    data = fd.read(size)
    if len(data) < size:
        raise IOError('truncated protobuf stream')
    return data


def protobuf_stream_dump(stream, filename):
    with open_compressed(filename, 'wb') as f:
        for item in stream:
            protobuf_stream_write(item, f)


class protobuf_stream_load(object):
    def __init__(self, filename):
        self.fd = open_compressed(filename, 'rb')

    def __iter__(self):
        return self

    def __next__(self):
        try:
            return protobuf_stream_read(self.fd)
        except StopIteration:
            self.close()
            raise

    next = __next__

    def close(self):
        self.fd.close()
