"""CPU tests of libmsc_driver.so (include/meshclust2_driver.h): the clustering LOGIC in host C++.
  - it exports every symbol the header declares;
  - its length-binned store against the reference's own bvec (oracle/_ref) on random length multisets and random operation
    sequences: ranges, scoring windows, mark + remove, pop, erase -- the layouts must stay identical step by step;
  - its matrix inverse against Matrix::gaussJordanInverse of the reference, bit for bit, on random <= 5 x 5 matrices
    (well conditioned, zero pivots that need a row swap, singular);
  - the whole mean-shift run with the CPU oracle as the scoring backend: cfg1.clstr (the reference CLI's own output) byte for byte.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from golden_util import GOLDEN, weights_text
from meshclust2_amd import _driver, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_library_exports_every_declared_symbol():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "meshclust2_driver.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(msc_[a-z0-9_]+)\s*\(", text)))
    lib = _driver.load_library()
    for n in names:
        assert hasattr(lib, n), "%s declared in include/meshclust2_driver.h but not exported" % n
    assert set(names) == set(_driver.PROTOTYPES), set(names) ^ set(_driver.PROTOTYPES)


class _Bins:
    """the same operations on either library (prefix msc_ = libmsc_driver.so, ref_ = the compiled reference)"""

    def __init__(self, lib, prefix, lengths, per_bin):
        self.lib, self.p, self.n = lib, prefix, len(lengths)
        arr = (C.c_uint64 * max(self.n, 1))(*lengths)
        f = getattr(lib, prefix + "bins_create")
        f.restype, f.argtypes = C.c_void_p, [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]
        self.h = C.c_void_p(f(arr, self.n, per_bin))
        for name, res, args in (("count", C.c_uint64, [C.c_void_p]), ("layout", C.c_uint64, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
                                ("range", None, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
                                ("window", C.c_int64, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.c_uint64]),
                                ("mark", None, [C.c_void_p, C.c_uint64, C.c_uint64]),
                                ("take_marked", C.c_uint64, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]),
                                ("take_first", C.c_int64, [C.c_void_p]), ("erase", None, [C.c_void_p, C.c_uint64, C.c_uint64]), ("destroy", None, [C.c_void_p])):
            fn = getattr(lib, prefix + "bins_" + name)
            fn.restype, fn.argtypes = res, args

    def call(self, name, *a):
        return getattr(self.lib, self.p + "bins_" + name)(self.h, *a)

    def layout(self):
        nb = self.call("count")
        ids = (C.c_uint32 * max(self.n, 1))()
        sizes = (C.c_uint64 * max(nb, 1))()
        left = self.call("layout", ids, sizes)
        return list(ids[:left]), list(sizes[:nb])

    def range(self, a, b):
        out = (C.c_uint64 * 5)()
        self.call("range", a, b, out)
        return list(out)

    def window(self, a, b):
        ids = (C.c_uint32 * max(self.n, 1))()
        trips = self.call("window", a, b, ids, self.n)
        return trips, list(ids[:max(0, min(trips, self.n))])

    def take_marked(self, a, b):
        ids = (C.c_uint32 * max(self.n, 1))()
        k = self.call("take_marked", a, b, ids)
        return list(ids[:k])


@pytest.mark.ref
@pytest.mark.parametrize("seed", range(12))
def test_length_bins_match_the_reference_bvec(ref, seed):
    """LengthBins (msc_driver.hpp) states bvec's lookups with std::upper_bound / std::equal_range; the reference scans and bisects
    by hand (cluster/bvec.cpp:52-147). Same ranges, same windows, same removal order, same layouts -- including lengths outside
    every bin, empty bins in the middle and at the ends, and runs of equal lengths."""
    rng = np.random.default_rng(1000 + seed)
    ref.lib().ref_set_threads(1)          # remove_available pushes under `omp critical`: one thread = the canonical order
    n = int(rng.integers(1, 400))
    style = seed % 4
    if style == 0:
        lengths = rng.integers(900, 1100, size=n)               # many equal lengths
    elif style == 1:
        lengths = np.full(n, 1000)                              # all equal: every record lands in the last bin
    elif style == 2:
        lengths = np.exp(rng.uniform(np.log(50), np.log(50000), size=n)).astype(np.int64)
    else:
        lengths = rng.choice([10, 500, 501, 502, 9000], size=n)
    lengths = [int(x) for x in lengths]
    per_bin = int(rng.choice([1, 2, 3, 7, 50, 1000]))
    mine, theirs = _Bins(_driver.load_library(), "msc_", lengths, per_bin), _Bins(ref.lib(), "ref_", lengths, per_bin)
    assert mine.layout() == theirs.layout()
    lo, hi = min(lengths), max(lengths)
    for step in range(120):
        left = mine.layout()[0]
        if not left:
            break
        centre = int(rng.choice(lengths)) if rng.random() < 0.8 else int(rng.integers(max(1, lo // 2), hi * 2))
        sim = float(rng.choice([0.6, 0.8, 0.9, 0.95]))
        a, b = int(centre * sim), int(centre / sim)
        assert mine.range(a, b) == theirs.range(a, b), (step, a, b)
        wm, wt = mine.window(a, b), theirs.window(a, b)
        assert wm == wt, (step, a, b)
        op = rng.random()
        if op < 0.5 and wm[0] > 0:
            # mark some records of the window (as get_close does), then remove_available over the same range
            r = mine.range(a, b)
            ids, sizes = mine.layout()
            pos = {}
            at = 0
            for bi, sz in enumerate(sizes):
                for c in range(sz):
                    pos[ids[at]] = (bi, c)
                    at += 1
            for rid in wm[1]:
                if rng.random() < 0.4:
                    mine.call("mark", *pos[rid])
                    theirs.call("mark", *pos[rid])
            assert mine.take_marked(a, b) == theirs.take_marked(a, b), (step, a, b, r)
        elif op < 0.7:
            assert mine.call("take_first") == theirs.call("take_first")
        elif op < 0.85 and wm[0] > 0:
            ids, sizes = mine.layout()
            bi = int(rng.choice([i for i, s_ in enumerate(sizes) if s_]))
            c = int(rng.integers(0, sizes[bi]))
            mine.call("erase", bi, c)
            theirs.call("erase", bi, c)
        assert mine.layout() == theirs.layout(), step
    mine.call("destroy")
    theirs.call("destroy")


@pytest.mark.ref
def test_host_inverse_matches_the_reference_matrix(ref):
    """msc::hostmath::inverse (Gauss-Jordan on the augmented matrix) == Matrix::gaussJordanInverse (predict/Matrix.cpp:109-207),
    bit for bit: random well-conditioned matrices, normal-equation matrices X^T X, exact-zero pivots that need a row swap,
    unit pivots, and singular matrices (both hand the input back)."""
    rng = np.random.default_rng(77)
    mine, theirs = _driver.load_library().msc_host_inverse, ref.lib().ref_host_inverse
    theirs.restype, theirs.argtypes = None, [C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    swapped = singular = 0
    for trial in range(600):
        n = int(rng.integers(1, 6))
        kind = trial % 6
        if kind == 0:
            a = rng.normal(size=(n, n))
        elif kind == 1:
            x = np.hstack([np.ones((40, 1)), rng.uniform(0, 1, size=(40, n - 1))]) if n > 1 else np.ones((40, 1))
            a = x.T @ x
        elif kind == 2:
            a = rng.integers(-3, 4, size=(n, n)).astype(np.float64)          # small integers: exact zeros and ones on the diagonal
        elif kind == 3:
            a = rng.normal(size=(n, n))
            a[0, 0] = 0.0                                                    # zero pivot -> row swap (or singular for n == 1)
        elif kind == 4:
            a = rng.normal(size=(n, n))
            a[-1] = a[0] * 2.0 if n > 1 else 0.0                             # exactly dependent rows
        else:
            a = np.eye(n) + np.triu(rng.integers(0, 3, size=(n, n)), 1).astype(np.float64)
        a = np.ascontiguousarray(a, dtype=np.float64)
        o1, o2 = np.zeros_like(a), np.zeros_like(a)
        mine(n, a.ctypes.data_as(C.POINTER(C.c_double)), o1.ctypes.data_as(C.POINTER(C.c_double)))
        theirs(n, a.ctypes.data_as(C.POINTER(C.c_double)), o2.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(o1.view(np.uint64), o2.view(np.uint64)), (trial, n, kind, a)
        swapped += kind == 3 and n > 1
        singular += bool(np.array_equal(o1, a) and not np.allclose(a @ a, np.eye(n)))
    assert swapped > 50 and singular > 20


class OracleBackend:
    """msc::ClusterBackend with the CPU oracle as the scorer (test infrastructure): point handle = index into `hists`"""

    def __init__(self, oracle, hists, pred, cutoff):
        self.o, self.h, self.pred, self.cutoff = oracle, hists, pred, cutoff
        self.centres = []

    def get_close(self, q, window):
        f, pos, _, im = self.o.get_close(self.pred, self.cutoff, self.h[q], [self.h[i] for i in window])
        return f, pos, im

    def closest(self, members):
        return self.o.mean_nearest([self.h[i] for i in members])[2]

    def centre_new(self, point):
        c = self.o.Hist()
        self.o.lib().orc_hist_clone(C.byref(self.h[point]), C.byref(c))
        self.centres.append(c)
        return len(self.centres) - 1

    def centre_set(self, centre, point):
        self.o.lib().orc_hist_set(C.byref(self.centres[centre]), C.byref(self.h[point]))

    def filter(self, centre, points):
        return self.o.filter_(self.pred, self.cutoff, self.centres[centre], [self.h[i] for i in points])

    def merge(self, centres, current, begin, last):
        return self.o.merge(self.pred, self.cutoff, [self.centres[c] for c in centres], current, begin, last)


class RangedOracleBackend(OracleBackend):
    """the same with the window kept on the backend's side (set_order / get_close_range / kill: what msc_window does on the GPU)"""

    def set_order(self, order):
        self.order = order.astype(np.int64)
        self.alive = np.ones(order.size, dtype=bool)
        self.ranges = 0

    def kill(self, pos):
        assert self.alive[pos]
        self.alive[pos] = False

    def get_close_range(self, q, first, end):
        self.ranges += 1
        pos = first + np.flatnonzero(self.alive[first:end])
        f, best, _, im = self.o.get_close(self.pred, self.cutoff, self.h[q], [self.h[i] for i in self.order[pos]])
        close = pos[np.flatnonzero(f)]
        self.alive[close] = False
        return close, (int(pos[best]) if best >= 0 else -1), im

    def get_close(self, q, window):
        raise AssertionError("a backend that keeps the window is given ranges")


def test_driver_logic_with_the_oracle_reproduces_cfg1(oracle, tmp_path):
    """BASELINE cfg1 (1000 x 1 kb, --id 0.9 --kmer 5 --datatype 16) through libmsc_driver.so with the CPU oracle as the backend: the
    reference CLI's own cfg1.clstr byte for byte -- the clustering logic is right independently of any GPU (the GPU suite runs the
    same logic over the C ABI)."""
    seqs, hdrs = synth.families(20260001, 1000, 1000)
    oracle.lib().orc_set_threads(os.cpu_count() or 1)
    hists = [oracle.hist(s_, 5, 16) for s_ in seqs]
    pred = oracle.predictor(weights_text("weights_k5_u16.txt"))
    out = str(tmp_path / "cfg1.clstr")
    _driver.run(OracleBackend(oracle, hists, pred, 0.9), hdrs, [h.length for h in hists], 0.9, output=out, log=str(tmp_path / "log.txt"), batch_update=False)
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "cfg1.clstr"), "rb").read()
    log = open(str(tmp_path / "log.txt")).read()
    assert "timestamp accumulate" in log and "Number of clusters:" in log


def test_driver_run_reports_a_failing_callback(oracle):
    class Broken(OracleBackend):
        def closest(self, members):
            raise ValueError("backend failure")

        def get_close(self, q, window):
            f, pos, im = OracleBackend.get_close(self, q, window)
            f[:] = 1          # everything close -> get_mean is reached
            return f, pos, False
    seqs, hdrs = synth.families(5, 40, 300)
    hists = [oracle.hist(s_, 4, 16) for s_ in seqs]
    with pytest.raises(ValueError, match="backend failure"):
        _driver.run(Broken(oracle, hists, oracle.predictor(weights_text("weights_k5_u16.txt")), 0.9), hdrs, [h.length for h in hists], 0.9, log=os.devnull)


@pytest.mark.parametrize("case", ["cfg1", "mixed"])
def test_ranged_windows_give_the_same_clusters(oracle, tmp_path, case):
    """The accumulate loop over a backend that keeps the length-sorted order itself (positions of the sealed store instead of a slot
    list per step: msc_window on the GPU) writes the same bytes as the loop that rebuilds the window on the host -- cfg1 (the reference
    CLI's own .clstr) and a mixed-length set whose windows are real length neighbourhoods (cluster/ClusterFactory.cpp:553-610)."""
    if case == "cfg1":
        seqs, hdrs = synth.families(20260001, 1000, 1000)
        k, dt, w, sim = 5, 16, "weights_k5_u16.txt", 0.9
    else:
        seqs, hdrs = synth.families(777, 900, 1000, length_jitter=150)
        k, dt, w, sim = 5, 16, "weights_k5_u16.txt", 0.9
    oracle.lib().orc_set_threads(os.cpu_count() or 1)
    hists = [oracle.hist(s_, k, dt) for s_ in seqs]
    pred = oracle.predictor(weights_text(w))
    lens = [h.length for h in hists]
    a, b = str(tmp_path / "listed.clstr"), str(tmp_path / "ranged.clstr")
    _driver.run(OracleBackend(oracle, hists, pred, sim), hdrs, lens, sim, output=a, log=os.devnull, batch_update=False)
    rb = RangedOracleBackend(oracle, hists, pred, sim)
    _driver.run(rb, hdrs, lens, sim, output=b, log=os.devnull, batch_update=False)
    assert rb.ranges > 10 and not rb.alive.any()          # every point left the store through a range or a kill
    assert open(a, "rb").read() == open(b, "rb").read()
    if case == "cfg1":
        assert open(b, "rb").read() == open(os.path.join(GOLDEN, "cfg1.clstr"), "rb").read()
