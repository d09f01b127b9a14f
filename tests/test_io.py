"""The data formats either side of the row-update path:
the reference's protobuf messages and record streams.

Pinned against the reference: tests/golden/schema_fields.json and
protobuf_messages.json come from the FileDescriptorProto embedded in the
reference's generated distributions/io/schema_pb2.py (make_goldens.py)."""
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def fill(message, content):
    for name, value in content.items():
        if isinstance(value, dict):
            fill(getattr(message, name), value)
        elif isinstance(value, list):
            getattr(message, name).extend(value)
        else:
            setattr(message, name, value)


def message_class(schema_pb2, full):
    obj = schema_pb2
    for part in full.split("."):
        obj = getattr(obj, part)
    return obj


def test_schema_table_is_the_reference_schema():
    from distributions_amd.io import schema_pb2
    ref = golden("schema_fields.json")
    assert schema_pb2.PACKAGE == ref["package"]
    mine = {full: [[label, typ, name, number, False]
                   for label, typ, name, number in fields]
            for full, fields in schema_pb2.SCHEMA.items()}
    assert mine == ref["messages"]      # no field of the reference is packed


def test_messages_serialize_to_the_reference_bytes():
    from distributions_amd.io import schema_pb2
    samples = golden("protobuf_messages.json")
    assert len(samples) >= 16
    for sample in samples:
        cls = message_class(schema_pb2, sample["message"])
        message = cls()
        fill(message, sample["content"])
        assert message.SerializeToString().hex() == sample["hex"], sample
        back = cls()
        back.ParseFromString(bytes.fromhex(sample["hex"]))
        assert back == message


def lp_modules():
    from distributions_amd.lp.models import bb, bnb, dd, dpd, gp, nich
    return [bb, bnb, dd, dpd, gp, nich]


def assert_close(a, b):
    if isinstance(a, dict):
        assert set(map(str, a)) == set(map(str, b)), (a, b)
        bs = {str(k): v for k, v in b.items()}
        for k, v in a.items():
            assert_close(v, bs[str(k)])
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert_close(x, y)
    else:
        assert abs(float(a) - float(b)) <= 1e-6 * (1 + abs(float(a))), (a, b)


def test_protobuf_round_trips_of_every_model():
    """distributions/tests/test_models.py:193-226 (test_protobuf)"""
    from distributions_amd.io import schema_pb2
    for module in lp_modules():
        for EXAMPLE in module.EXAMPLES:
            shared = module.Shared.from_dict(EXAMPLE['shared'])
            values = EXAMPLE['values']
            Message = getattr(schema_pb2, module.NAME)

            message = Message.Shared()
            shared.protobuf_dump(message)
            shared2 = module.Shared()
            shared2.protobuf_load(message)
            assert_close(shared2.dump(), shared.dump())
            # through the wire
            wire = Message.Shared()
            wire.ParseFromString(message.SerializeToString())
            shared3 = module.Shared()
            shared3.protobuf_load(wire)
            assert_close(shared3.dump(), shared.dump())

            message.Clear()
            dumped = shared.dump()
            module.Shared.to_protobuf(dumped, message)
            assert_close(module.Shared.from_protobuf(message), dumped)

            for value in values:
                shared.add_value(value)
            group = module.Group.from_values(shared, values)
            message = Message.Group()
            group.protobuf_dump(message)
            group2 = module.Group()
            group2.protobuf_load(message)
            assert_close(group2.dump(), group.dump())
            # the loaded group scores like the original
            probe = values[0]
            assert np.float32(group2.score_value(shared, probe)) == np.float32(
                group.score_value(shared, probe))

            message.Clear()
            dumped = group.dump()
            module.Group.to_protobuf(dumped, message)
            assert_close(module.Group.from_protobuf(message), dumped)


def test_clustering_message():
    from distributions_amd.io import schema_pb2
    from distributions_amd.lp.clustering import PitmanYor
    model = PitmanYor(alpha=1.5, d=0.25)
    message = schema_pb2.Clustering()
    model.protobuf_dump(message.pitman_yor)
    data = message.SerializeToString()
    back = schema_pb2.Clustering()
    back.ParseFromString(data)
    assert back.HasField("pitman_yor") and not back.HasField("low_entropy")
    model2 = PitmanYor()
    model2.protobuf_load(back.pitman_yor)
    assert model2.dump() == model.dump()


@pytest.mark.parametrize("suffix", ["", ".gz", ".bz2"])
def test_protobuf_stream(tmp_path, suffix):
    """stream.py:139-172: u32 little-endian length, then the bytes"""
    from distributions_amd.io import schema_pb2, stream
    rng = np.random.default_rng(0)
    records = []
    for _ in range(20):
        g = schema_pb2.DirichletDiscrete.Group()
        g.counts.extend(int(c) for c in rng.integers(0, 1000, rng.integers(0, 9)))
        records.append(g.SerializeToString())
    name = str(tmp_path / "sub" / ("groups.pbs" + suffix))
    stream.protobuf_stream_dump(records, name)
    assert list(stream.protobuf_stream_load(name)) == records
    if not suffix:
        raw = open(name, "rb").read()
        expect = b"".join(struct.pack("<I", len(r)) + r for r in records)
        assert raw == expect


@pytest.mark.parametrize("suffix", ["", ".gz"])
def test_json_stream(tmp_path, suffix):
    """stream.py:68-136: '[', one compact document per line, ']'"""
    from distributions_amd.io import stream
    items = [{"a": 1, "b": [1, 2, 3]}, {"c": "x"}, [1, 2], 7]
    name = str(tmp_path / ("rows.json" + suffix))
    stream.json_stream_dump(iter(items), name)
    assert list(stream.json_stream_load(name)) == items
    assert stream.json_load(name) == items         # also plain json
    if not suffix:
        text = open(name).read()
        assert text == '[\n{"a":1,"b":[1,2,3]},\n{"c":"x"},\n[1,2],\n7\n]'
    co = stream.json_costream_dump(str(tmp_path / ("co.json" + suffix)))
    next(co)
    for item in items:
        co.send(item)
    co.close()
    assert list(stream.json_stream_load(
        str(tmp_path / ("co.json" + suffix)))) == items
    stream.json_stream_dump([], str(tmp_path / "empty.json"))
    assert list(stream.json_stream_load(str(tmp_path / "empty.json"))) == []
    stream.json_dump({"k": [1.5]}, str(tmp_path / "d.json.gz"))
    assert stream.json_load(str(tmp_path / "d.json.gz")) == {"k": [1.5]}


# ---------------------------------------------------------------------------
# the C-ABI wire codec (dist_*_protobuf_*), no libprotobuf involved

def _kind_params(full, content):
    """-> (SharedParams for the model of message `full`, dense keys or None)"""
    from distributions_amd import _core
    model = full.split(".")[0]
    if model == "DirichletDiscrete":
        dim = len(content.get("alphas", content.get("counts", [0.5])))
        alphas = content.get("alphas", [0.5] * max(dim, 1))
        return _core.SharedParams.make(_core.KIND_DD, alphas=alphas), None
    if model == "BetaBernoulli":
        return _core.SharedParams.make(
            _core.KIND_BB, p=(content.get("alpha", 1.0),
                              content.get("beta", 1.0))), None
    if model == "GammaPoisson":
        return _core.SharedParams.make(
            _core.KIND_GP, p=(content.get("alpha", 1.0),
                              content.get("inv_beta", 1.0))), None
    if model == "BetaNegativeBinomial":
        return _core.SharedParams.make(
            _core.KIND_BNB, p=(content.get("alpha", 1.0),
                               content.get("beta", 1.0),
                               float(content.get("r", 1)))), None
    if model == "NormalInverseChiSq":
        return _core.SharedParams.make(
            _core.KIND_NICH, p=(content.get("mu", 0.0),
                                content.get("kappa", 1.0),
                                content.get("sigmasq", 1.0),
                                content.get("nu", 1.0))), None
    if model == "DirichletProcessDiscrete":
        keys = [0, 1, 7, 300]
        return _core.SharedParams.make(
            _core.KIND_DPD, p=(0.5, 0.25), betas=[0.25, 0.25, 0.125, 0.125]), keys
    return None, None


def test_c_codec_writes_and_reads_the_reference_bytes():
    from distributions_amd.io import schema_pb2
    from distributions_amd import _core
    checked = 0
    for sample in golden("protobuf_messages.json"):
        full, content = sample["message"], sample["content"]
        data = bytes.fromhex(sample["hex"])
        params, keys = _kind_params(full, content)
        if params is None:
            continue
        if full.endswith(".Shared"):
            if params.kind == _core.KIND_DPD:
                with pytest.raises(RuntimeError):
                    params.protobuf_dump()
                continue
            assert params.protobuf_dump() == data, full
            back = _core.SharedParams.protobuf_load(params.kind, data)
            assert back.kind == params.kind and back.dim == params.dim
            assert np.array_equal(np.float32(back.p), np.float32(params.p))
            assert np.array_equal(np.float32(back.alphas),
                                  np.float32(params.alphas))
        else:
            if full == "DirichletDiscrete.Group" and any(
                    c >= 2 ** 32 for c in content["counts"]):
                continue    # beyond the 32-bit group words of this build
            if full == "DirichletDiscrete.Group" and not content["counts"]:
                continue    # dim 0 is not a DirichletDiscrete
            words = params.group_protobuf_load(data, keys)
            assert params.group_protobuf_dump(words, keys) == data, full
            # the decoded statistics are what the message says
            cls = message_class(schema_pb2, full)
            message = cls()
            message.ParseFromString(params.group_protobuf_dump(words, keys))
            want = cls()
            fill(want, content)
            assert message == want
        checked += 1
    assert checked >= 11


def test_c_codec_round_trips_lp_groups_through_python_protobuf():
    """bytes written by the C codec parse with the message classes and load
    into lp groups; bytes written by the message classes load in C"""
    from distributions_amd.io import schema_pb2
    for module in lp_modules():
        for EXAMPLE in module.EXAMPLES:
            shared = module.Shared.from_dict(EXAMPLE['shared'])
            group = module.Group.from_values(shared, EXAMPLE['values'])
            keys = getattr(shared, "values", None)
            Message = getattr(schema_pb2, module.NAME)
            data = shared.params.group_protobuf_dump(group.words, keys)
            message = Message.Group()
            message.ParseFromString(data)
            group2 = module.Group()
            group2.protobuf_load(message)
            assert_close(group2.dump(), group.dump())
            message2 = Message.Group()
            group.protobuf_dump(message2)
            words = shared.params.group_protobuf_load(
                message2.SerializeToString(), keys)
            assert np.array_equal(words, group.words)
            # packed repeated fields (what a proto3 writer emits) load too
            if module.NAME == "DirichletDiscrete":
                counts = [int(c) for c in group.dump()['counts']]
                body = b"".join(_varint(c) for c in counts)
                packed = b"\x0a" + _varint(len(body)) + body
                assert np.array_equal(
                    shared.params.group_protobuf_load(packed), group.words)


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append(v & 0x7f | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def test_c_codec_rejects_malformed_input():
    from distributions_amd import _core
    dd = _core.SharedParams.make(_core.KIND_DD, alphas=[0.5] * 3)
    with pytest.raises(RuntimeError):
        dd.group_protobuf_load(b"\x08\x01\x08\x02")          # 2 counts, dim 3
    with pytest.raises(RuntimeError):
        dd.group_protobuf_load(b"\x08\x01\x08\x02\x08\x03\x08")  # truncated
    with pytest.raises(RuntimeError):
        dd.group_protobuf_load(b"\x08\x01\x08\x02\x08\x03\x08\x04")  # 4 counts
    ok = dd.group_protobuf_load(b"\x08\x01\x10\x63\x08\x02\x08\x03")  # unknown field 2
    assert list(ok) == [6, 1, 2, 3]
