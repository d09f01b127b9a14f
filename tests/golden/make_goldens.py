#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python3 tests/golden/make_goldens.py

Sources of the expected outputs:
  special_functions.npz  oracle/_ref/libref.so = the reference's own
      src/special.cc + include/distributions/special.hpp + vendor/fmath.hpp,
      compiled with the reference's release flags (g++ 11.4.0, glibc 2.35).
  vector_math.npz        the reference's src/vector_math.cc (same build).
  driver_tracker.npz     the reference's MixtureDriver / MixtureIdTracker
      templates (include/distributions/mixture.hpp) driven through a seeded
      random add/remove script.
  rng_libstdcxx.npz      libstdc++'s std::default_random_engine and
      std::uniform_real_distribution<float>, i.e. what rng_t/sample_unif01
      (random_fwd.hpp:34, random.hpp:47-50) resolve to; via
      oracle/_ref/check_libstdcxx.
  schema_fields.json, protobuf_messages.json   the FileDescriptorProto that
      protoc embedded in the reference's generated distributions/io/
      schema_pb2.py (its `serialized_pb` literal): the field table of every
      message, and sample messages serialized by classes built from THAT
      descriptor (hex of the wire bytes).
The fixtures are data (inputs + expected outputs); no reference source.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402


def special(R):
    rng = np.random.default_rng(1)
    f32 = np.float32
    edges = np.array([2.0 ** e for e in range(-20, 33)], f32)
    near = np.concatenate([np.nextafter(edges, f32(0)), edges,
                           np.nextafter(edges, f32(np.inf))])
    x_log = np.concatenate([
        np.exp(rng.uniform(-80, 80, 3000)).astype(f32), near,
        np.array([1e-38, 1.0, 0.5, 3.4e38], f32)])
    x_exp = np.concatenate([
        rng.uniform(-100, 5, 3000).astype(f32),
        np.array([0, -1e-8, -0.5, -0.49999997, -88, -87.99999, -88.00001,
                  -87.33655, -87.5, -103.97, -200, 1, 88, 10], f32)])
    x_lg = np.concatenate([
        np.exp(rng.uniform(np.log(2.5), np.log(4e9), 3000)).astype(f32),
        near[near >= 2.5], np.array([2.5, 3, 4, 64, 65, 1e6], f32),
        # libm lgammaf branch, exact cases only (integers)
        np.array([1.0, 2.0], f32)])
    x_nu = np.concatenate([
        np.exp(rng.uniform(np.log(0.0625), np.log(4e9), 3000)).astype(f32),
        near[near >= 0.0625], np.array([0.0625, 1, 2, 3, 4, 1e4], f32)])
    n_lf = np.concatenate([np.arange(0, 130), rng.integers(0, 2 ** 31, 300),
                           [2 ** 32 - 2]]).astype(np.uint32)
    out = {}
    for name, x in [("fast_log", x_log), ("fast_exp", x_exp),
                    ("fast_lgamma", x_lg), ("fast_lgamma_nu", x_nu)]:
        x = np.ascontiguousarray(x, f32)
        y = np.zeros_like(x)
        getattr(R, "ref_" + name)(x.size, x, y)
        out[name + "_in"] = x
        out[name + "_out"] = y.view(np.uint32)
    y = np.zeros(n_lf.size, f32)
    R.ref_fast_log_factorial(n_lf.size, n_lf, y)
    out["fast_log_factorial_in"] = n_lf
    out["fast_log_factorial_out"] = y.view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "special_functions.npz"), **out)


def vector_math(R):
    rng = np.random.default_rng(2)
    out = {}
    for n in [1, 3, 4, 7, 64, 1000]:
        io = (rng.normal(size=n) * 3).astype(np.float32)
        a = (rng.normal(size=n) * 5).astype(np.float32)
        b = (rng.normal(size=n) * 7).astype(np.float32)
        r = io.copy()
        R.ref_vector_add_subtract(n, r, a, b)
        r2 = io.copy()
        R.ref_vector_add_subtract_scalar(n, r2, ctypes.c_float(1.2345), b)
        r3 = io.copy()
        R.ref_vector_add(n, r3, a)
        out["n%d_io" % n] = io
        out["n%d_a" % n] = a
        out["n%d_b" % n] = b
        out["n%d_add_subtract" % n] = r.view(np.uint32)
        out["n%d_add_subtract_scalar" % n] = r2.view(np.uint32)
        out["n%d_add" % n] = r3.view(np.uint32)
        out["n%d_max" % n] = np.array([R.ref_vector_max(n, io)], np.float32)
    np.savez_compressed(os.path.join(HERE, "vector_math.npz"), **out)


def vector_sum(R):
    """vector_sum (vector_math.cc:85-93) of the release build: its
    association is the vectoriser's, so the expected bits are recorded"""
    R.ref_vector_sum.restype = ctypes.c_float
    R.ref_vector_sum.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
    rng = np.random.default_rng(6)
    out = {}
    sizes = list(range(0, 20)) + [63, 64, 65, 257, 1000, 1001, 1002, 1003]
    for n in sizes:
        x = (rng.normal(size=max(n, 1)) * 10 ** rng.uniform(-3, 3)).astype(
            np.float32)
        out["n%d_x" % n] = x
        out["n%d_sum" % n] = np.array(
            [R.ref_vector_sum(n, x.ctypes.data)], np.float32).view(np.uint32)
    out["sizes"] = np.array(sizes)
    np.savez_compressed(os.path.join(HERE, "vector_sum.npz"), **out)


def driver_tracker(R):
    """Random add/remove script; records every return flag and the state."""
    rng = np.random.default_rng(3)
    out = {}
    for case, empties in enumerate([1, 4]):
        counts = np.concatenate([rng.integers(1, 4, 6),
                                 np.zeros(empties)]).astype(np.int32)
        rng.shuffle(counts)
        d = R.ref_driver_new()
        t = R.ref_tracker_new()
        R.ref_driver_init(d, counts, counts.size)
        R.ref_tracker_init(t, counts.size)
        script, trace = [], []
        for step in range(400):
            size = R.ref_driver_size(d)
            cur = np.zeros(size, np.int32)
            R.ref_driver_counts(d, cur)
            if rng.random() < 0.5:
                g = int(rng.integers(0, size))
                flag = R.ref_driver_add_value(d, g)
                if flag:
                    R.ref_tracker_add_group(t)
                op = 1
            else:
                nz = np.nonzero(cur)[0]
                if nz.size == 0:
                    continue
                g = int(rng.choice(nz))
                flag = R.ref_driver_remove_value(d, g)
                if flag:
                    R.ref_tracker_remove_group(t, g)
                op = 0
            size = R.ref_driver_size(d)
            cur = np.zeros(size, np.int32)
            R.ref_driver_counts(d, cur)
            p2g = [R.ref_tracker_packed_to_global(t, i) for i in range(size)]
            script.append((op, g))
            trace.append((flag, size, R.ref_driver_sample_size(d),
                          R.ref_driver_empty_count(d),
                          int(np.dot(cur, np.arange(1, size + 1)) % 1000003),
                          int(np.dot(p2g, np.arange(1, size + 1)) % 1000003)))
        out["case%d_counts" % case] = counts
        out["case%d_script" % case] = np.array(script, np.int32)
        out["case%d_trace" % case] = np.array(trace, np.int64)
        R.ref_driver_delete(d)
        R.ref_tracker_delete(t)
    np.savez_compressed(os.path.join(HERE, "driver_tracker.npz"), **out)


def variates_libstdcxx():
    """sample_gamma / sample_beta_safe (random.hpp:87-119) from libstdc++
    itself: oracle/check_libstdcxx.cc variates"""
    import json
    exe = os.path.join(ROOT, "oracle", "_ref", "check_libstdcxx")
    cases = []
    for alpha, beta in [(1.0, 0.5), (1.0, 2.0), (1.0, 5.0), (0.3, 0.01),
                        (7.5, 1.0), (1.0, 0.01)]:
        seeds = [1, 7, 12345, 987654321]
        txt = subprocess.check_output(
            [exe, "variates", "40", repr(alpha), repr(beta)]
            + [str(s) for s in seeds], text=True)
        cur = None
        for line in txt.splitlines():
            if line.startswith("seed"):
                cur = {"alpha": alpha, "beta": beta,
                       "seed": int(line.split()[1]), "gamma_bits": [],
                       "gamma_next": [], "beta_safe_bits": [],
                       "beta_safe_next": []}
                cases.append(cur)
            else:
                g, gn, b, bn = line.split()
                cur["gamma_bits"].append(int(g, 16))
                cur["gamma_next"].append(int(gn))
                cur["beta_safe_bits"].append(int(b, 16))
                cur["beta_safe_next"].append(int(bn))
    with open(os.path.join(HERE, "variates_libstdcxx.json"), "w") as f:
        json.dump({"min_value": 1e-6, "cases": cases}, f,
                  separators=(",", ":"))
    print("variates_libstdcxx.json: %d cases" % len(cases))


def rng_libstdcxx():
    exe = os.path.join(ROOT, "oracle", "_ref", "check_libstdcxx")
    seeds = [1, 0, 12345, 2147483647, 2147483646, 987654321]
    txt = subprocess.check_output([exe, "64"] + [str(s) for s in seeds],
                                  text=True)
    out = {"seeds": np.array(seeds, np.uint64)}
    cur = None
    for line in txt.splitlines():
        if line.startswith("seed"):
            cur = int(line.split()[1])
            out["raw_%d" % cur] = []
            out["u_%d" % cur] = []
        else:
            raw, bits = line.split()
            out["raw_%d" % cur].append(int(raw))
            out["u_%d" % cur].append(int(bits, 16))
    for key in list(out):
        if key.startswith("raw_"):
            out[key] = np.array(out[key], np.uint32)
        if key.startswith("u_"):
            out[key] = np.array(out[key], np.uint32)
    np.savez_compressed(os.path.join(HERE, "rng_libstdcxx.npz"), **out)


PROTOBUF_SAMPLES = [
    ("Clustering", {"pitman_yor": {"alpha": 1.0, "d": 0.2}}),
    ("Clustering", {"low_entropy": {"dataset_size": 123456789012}}),
    ("BetaBernoulli.Shared", {"alpha": 0.5, "beta": 2.0}),
    ("BetaBernoulli.Group", {"heads": 3, "tails": 70000}),
    ("DirichletDiscrete.Shared", {"alphas": [0.5, 0.25, 4.0]}),
    ("DirichletDiscrete.Group", {"counts": [0, 1, 300, 2 ** 33]}),
    ("DirichletDiscrete.Group", {"counts": []}),
    ("DirichletDiscrete.Group", {"counts": [5, 0, 17, 4000000000, 1]}),
    ("DirichletProcessDiscrete.Shared",
     {"gamma": 0.5, "alpha": 0.5, "values": [0, 1, 7, 300],
      "betas": [0.25, 0.25, 0.125, 0.125], "counts": [1, 2, 4, 1]}),
    ("DirichletProcessDiscrete.Group",
     {"keys": [0, 7, 300], "values": [4, 1, 129]}),
    ("GammaPoisson.Shared", {"alpha": 1.0, "inv_beta": 0.7}),
    ("GammaPoisson.Group", {"count": 5, "sum": 31, "log_prod": 17.502307}),
    ("BetaNegativeBinomial.Shared", {"alpha": 1.5, "beta": 2.5, "r": 9}),
    ("BetaNegativeBinomial.Group", {"count": 4, "sum": 40}),
    ("NormalInverseChiSq.Shared",
     {"mu": -0.25, "kappa": 1.0, "sigmasq": 2.0, "nu": 3.5}),
    ("NormalInverseChiSq.Group",
     {"count": 3, "mean": 0.5, "count_times_variance": 2.0}),
]


def protobuf_schema():
    """field table + sample wire bytes from the reference's own descriptor"""
    import ast
    import json
    import re
    from google.protobuf import (descriptor_pb2, descriptor_pool,
                                 message_factory)
    src = open("/root/reference/distributions/io/schema_pb2.py",
               encoding="latin-1").read()
    literal = re.search(r"serialized_pb='((?:[^'\\]|\\.)*)'", src).group(1)
    raw = ast.literal_eval("b'" + literal + "'")
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.ParseFromString(raw)
    F = descriptor_pb2.FieldDescriptorProto
    type_names = {F.TYPE_FLOAT: "float", F.TYPE_UINT64: "uint64",
                  F.TYPE_UINT32: "uint32", F.TYPE_INT32: "int32"}
    label_names = {F.LABEL_REQUIRED: "required", F.LABEL_OPTIONAL: "optional",
                   F.LABEL_REPEATED: "repeated"}
    table = {}

    def walk(prefix, msg):
        full = prefix + msg.name
        if msg.field:
            table[full] = [
                [label_names[f.label],
                 type_names.get(f.type) or f.type_name.split(".")[-1],
                 f.name, f.number, bool(f.options.packed)]
                for f in msg.field]
        for nested in msg.nested_type:
            walk(full + ".", nested)

    for msg in fdp.message_type:
        walk("", msg)
    with open(os.path.join(HERE, "schema_fields.json"), "w") as f:
        json.dump({"package": fdp.package, "messages": table}, f, indent=1,
                  sort_keys=True)

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)

    def fill(message, content):
        for name, value in content.items():
            if isinstance(value, dict):
                fill(getattr(message, name), value)
            elif isinstance(value, list):
                getattr(message, name).extend(value)
            else:
                setattr(message, name, value)

    samples = []
    for full, content in PROTOBUF_SAMPLES:
        cls = message_factory.GetMessageClass(
            pool.FindMessageTypeByName(fdp.package + "." + full))
        message = cls()
        fill(message, content)
        samples.append({"message": full, "content": content,
                        "hex": message.SerializeToString().hex()})
    with open(os.path.join(HERE, "protobuf_messages.json"), "w") as f:
        json.dump(samples, f, indent=1)
    print("schema_fields.json: %d messages; protobuf_messages.json: %d samples"
          % (len(table), len(samples)))


def low_entropy_table():
    """the 48 log Z(n) values the reference ships for its low-entropy
    clustering model (src/clustering.cc:189-202), as data"""
    import json
    import re
    src = open("/root/reference/src/clustering.cc").read()
    body = re.search(r"log_partition_function_table\[48\] = \{(.*?)\};", src,
                     re.S).group(1)
    values = [float(x) for x in body.replace("\n", " ").split(",")]
    assert len(values) == 48
    bits = [int(np.float32(v).view(np.uint32)) for v in values]
    with open(os.path.join(HERE, "low_entropy.json"), "w") as f:
        json.dump({"log_partition_function_table": values,
                   "float32_bits": bits}, f, indent=1)
    print("low_entropy.json: %d table entries" % len(values))


if __name__ == "__main__":
    protobuf_schema()
    low_entropy_table()
    if "--variates-only" in sys.argv:
        variates_libstdcxx()
        sys.exit(0)
    if "--schema-only" in sys.argv:
        sys.exit(0)
    if "--vector-sum-only" in sys.argv:
        R = ol.ref()
        assert R is not None, "build oracle/_ref first (make -C oracle ref)"
        vector_sum(R)
        sys.exit(0)
    R = ol.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    special(R)
    vector_math(R)
    driver_tracker(R)
    rng_libstdcxx()
    variates_libstdcxx()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
