"""The C-ABI library loads and exports every symbol include/*.h declares; the
product does not reach into oracle/ (no GPU needed: no compute calls)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "distributions_amd", "libdistributions_hip.so")
HDR = os.path.join(ROOT, "include", "distributions_hip.h")


def declared_functions():
    text = open(HDR).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dist_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python __graft_entry__.py"
    lib = ctypes.CDLL(LIB)
    names = declared_functions()
    assert len(names) > 70
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.dist_abi_version.restype = ctypes.c_int
    assert lib.dist_abi_version() == 1


def test_host_only_entry_points_work_without_a_gpu():
    """entropy and the id tracker are host logic"""
    lib = ctypes.CDLL(LIB)
    lib.dist_rng_seed.restype = ctypes.c_uint32
    lib.dist_rng_seed.argtypes = [ctypes.c_uint64]
    st = ctypes.c_uint32(lib.dist_rng_seed(1))
    lib.dist_rng_next.restype = ctypes.c_uint32
    assert [lib.dist_rng_next(ctypes.byref(st)) for _ in range(4)] == [
        16807, 282475249, 1622650073, 984943658]
    lib.dist_rng_jump.restype = ctypes.c_uint32
    lib.dist_rng_jump.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
    assert lib.dist_rng_jump(1, 4) == 984943658
    lib.dist_id_tracker_create.restype = ctypes.c_void_p
    t = ctypes.c_void_p(lib.dist_id_tracker_create())
    lib.dist_id_tracker_init(t, ctypes.c_size_t(3))
    lib.dist_id_tracker_remove_group(t, ctypes.c_uint32(0))
    out = ctypes.c_uint32()
    lib.dist_id_tracker_packed_to_global(t, ctypes.c_uint32(0),
                                         ctypes.byref(out))
    assert out.value == 2
    assert lib.dist_id_tracker_global_to_packed(
        t, ctypes.c_uint32(0), ctypes.byref(out)) != 0   # stale id -> error
    lib.dist_last_error.restype = ctypes.c_char_p
    assert b"stale global id" in lib.dist_last_error()
    lib.dist_id_tracker_destroy(t)


def test_compute_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    lib = ctypes.CDLL(LIB)
    lib.dist_py_mixture_create.restype = ctypes.c_void_p
    m = ctypes.c_void_p(lib.dist_py_mixture_create())
    counts = (ctypes.c_int * 2)(3, 0)
    rc = lib.dist_py_mixture_init(m, ctypes.c_float(1), ctypes.c_float(0),
                                  counts, ctypes.c_size_t(2))
    lib.dist_last_error.restype = ctypes.c_char_p
    assert rc != 0 and b"no HIP device" in lib.dist_last_error()


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "distributions_amd")
    hits = subprocess.run(
        ["grep", "-rIl", "-e", "oracle", "--include=*.py", "--include=*.pyx",
         "--include=*.h", "--include=*.hip", "--include=*.hpp",
         "--include=Makefile", pkg, os.path.join(ROOT, "include")],
        capture_output=True, text=True).stdout.split()
    # ref_tables.h only names the script that generated it
    hits = [h for h in hits if not h.endswith("ref_tables.h")]
    assert hits == [], hits
    needed = subprocess.run(["ldd", LIB], capture_output=True,
                            text=True).stdout
    assert "liboracle" not in needed and "libref" not in needed


def test_sample_assignments_is_host_logic_and_matches_the_probe():
    """PitmanYor::sample_assignments (clustering.cc:67-142) needs no GPU; the
    recorded output of the compiled reference and the oracle"""
    import numpy as np
    import oracle_lib as ol
    lib = ctypes.CDLL(LIB)
    out = (ctypes.c_int * 20)()
    st = ctypes.c_uint32(1)
    rc = lib.dist_py_sample_assignments(ctypes.c_float(1.0),
                                        ctypes.c_float(0.2), 20,
                                        ctypes.byref(st), out)
    assert rc == 0
    assert list(out) == [0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 2, 0, 0, 2, 0, 0, 0, 0,
                         0, 0]
    L = ol.oracle()
    for alpha, d, n, seed in [(1.0, 0.0, 500, 3), (10.0, 0.1, 2000, 9),
                              (0.1, 0.9, 300, 77)]:
        a = (ctypes.c_int * n)()
        st = ctypes.c_uint32(L.orc_rng_seed(seed))
        lib.dist_py_sample_assignments(ctypes.c_float(alpha),
                                       ctypes.c_float(d), n,
                                       ctypes.byref(st), a)
        b = np.zeros(n, np.int32)
        st2 = ctypes.c_uint32(L.orc_rng_seed(seed))
        L.orc_py_sample_assignments(alpha, d, n, ctypes.byref(st2), b)
        assert list(a) == list(b) and st.value == st2.value
