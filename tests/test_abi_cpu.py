"""CPU-side checks of the product library: it loads, exports every symbol include/meshclust2_hip.h declares, refuses to
run without a GPU (no CPU fallback), and its host-only entry points (sequence encoding, weights parsing errors) behave
like the reference. No compute call is made here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from meshclust2_amd import _capi, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "meshclust2_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load_library()
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "%s declared in include/meshclust2_hip.h but not exported" % n
    # and the ctypes table covers the whole header
    assert set(names) == set(_capi.PROTOTYPES), set(names) ^ set(_capi.PROTOTYPES)
    assert lib.msc_abi_version() == 1


def _has_gpu():
    lib = _capi.load_library()
    h = C.c_void_p()
    rc = lib.msc_create(0, C.byref(h))
    if rc == 0:
        lib.msc_destroy(h)
    return rc == 0


def test_no_cpu_fallback_without_a_device():
    if _has_gpu():
        pytest.skip("a GPU is present")
    lib = _capi.load_library()
    h = C.c_void_p()
    assert lib.msc_create(0, C.byref(h)) == -2           # MSC_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.msc_last_error(None)
    with pytest.raises(api.MscError):
        api.Context(0)


NASTY = [
    b"ACGTNNNNACGTACGTACGTACGTAACCGGTTNNNNNNNNNNNNACGATCGATCGATCGATCGACTAGCTAGCTAGCATCGAT",
    b"acgtacgtnnacgtRYMKSWHBVDacgtacgtacgtagctagcatcgatcgatcgatcagctagcat",
    b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNN", b"", b"A", b"ACG",
    b"ACGTACGTACGTACGTACGTACGTNNNNNNNNNNNNNNNNNNNNNNA",
    b"ACGTACGTACGTACGTACGTNNNNNNNNNNNNNNNACGTACGTACGTAC",
    b"ACGTACGTACGTACGTACGTACGNNNNNNNNNACGTACGTACGTACGTACGTACGT",
    b"ACGTNNACGTNNACGTNN",
    b"ACGT" * 5 + b"N" + b"TTGCA" * 7 + b"N" * 10 + b"GATTACA" * 9,
]


@pytest.mark.parametrize("seq", NASTY)
def test_host_encoder_matches_oracle(oracle, seq):
    """msc_encode is host byte work and needs no device"""
    assert api.encode(seq) == oracle.encode(seq)


def test_host_encoder_random_and_invalid(oracle):
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(1, 400))
        s = bytes(rng.choice(np.frombuffer(b"ACGTNNNacgtnRYK", dtype=np.uint8), size=n))
        assert api.encode(s) == oracle.encode(s)
    with pytest.raises(api.MscError):
        api.encode(b"ACGTACGTACGTACGTACGT-ACGTACGTACGTACGTACGT")
    with pytest.raises(ValueError):
        oracle.encode(b"ACGTACGTACGTACGTACGT-ACGTACGTACGTACGTACGT")


def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    host = os.path.join(ROOT, "meshclust2_amd", "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    if _has_gpu():
        pytest.skip("a GPU is present")
    for exe, args in (("host_example", [os.path.join(ROOT, "tests", "golden", "weights_k5_u16.txt")]),
                      ("msc_cluster", [os.path.join(ROOT, "tests", "golden", "weights_k5_u16.txt"), "--recover", os.path.join(ROOT, "tests", "golden", "weights_k5_u16.txt")])):
        r = subprocess.run([os.path.join(host, exe)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 3 and b"no CPU fallback" in r.stdout
