"""Host-side mirror of the reference's interface for the hot path, over the C ABI.

Names follow the reference: Loader::get_point -> HistogramSet.build, Feature::compute ->
Feature.compute / raw, Trainer::get_close / filter / merge / closest -> Trainer.*, Predictor::close /
similarity -> Predictor.*. Everything here is plumbing (ctypes marshalling); the work happens in
libmeshclust2_hip.so on the GPU.
"""
import ctypes as C
import weakref

import numpy as np

from . import _capi
from ._capi import FEAT, FEAT_FAST, FEAT_SLOW, MscError, ORDER_CAND_FIRST, ORDER_QUERY_FIRST  # noqa: F401

NP_T = {8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP device + stream (msc_ctx). Not thread-safe: one per host thread."""

    def __init__(self, device=0):
        self.lib = _capi.load_library()
        h = C.c_void_p()
        rc = self.lib.msc_create(int(device), C.byref(h))
        if rc != 0:
            raise MscError(rc, self.lib.msc_last_error(None).decode())
        self.h = h
        self.device = device
        self._children = []          # weakrefs to sets / models so that close() can release them before the ctx
        self._pinned = []            # page-locked host arrays handed out by pinned_array

    def _adopt(self, obj):
        self._children.append(weakref.ref(obj))

    def check(self, rc):
        if rc != 0:
            raise MscError(rc, self.lib.msc_last_error(self.h).decode())

    def device_name(self):
        buf = C.create_string_buffer(256)
        self.check(self.lib.msc_device_name(self.h, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        self.check(self.lib.msc_synchronize(self.h))

    def last_kernel_ms(self):
        """(pair_tiles ms, whole device pipeline ms) of the last scoring call, from HIP events on the ctx stream."""
        a, b = C.c_float(), C.c_float()
        self.check(self.lib.msc_last_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_kernel_timing(self, on):
        """the HIP events behind last_kernel_ms: off for step-serial loops (four event records per call)"""
        self.check(self.lib.msc_set_kernel_timing(self.h, 1 if on else 0))

    def set_mirror_pass(self, on):
        """dense sets: 1 x M passes over the sparse mirror (default) or over the bins with the streaming kernel; same results"""
        self.check(self.lib.msc_set_mirror_pass(self.h, 1 if on else 0))

    def set_block_pipe(self, on):
        """msc_score_multi's blocks on three streams (default) or every kernel of a block on one: same results, unstretched kernel timings"""
        self.check(self.lib.msc_set_block_pipe(self.h, 1 if on else 0))

    def last_kernel_launches(self):
        return self.lib.msc_last_kernel_launches(self.h)

    def last_kernel_info(self):
        """-> (name of the streaming kernel the last scoring call ran, queries served per HBM read of a candidate tile)"""
        buf = C.create_string_buffer(128)
        n = C.c_int()
        self.check(self.lib.msc_last_kernel_info(self.h, buf, 128, C.byref(n)))
        return buf.value.decode(), n.value

    def close(self):
        if getattr(self, "h", None):
            for ref in self._children:
                obj = ref()
                if obj is not None:
                    obj.close()
            self._children = []
            for p in getattr(self, "_pinned", []):
                self.lib.msc_host_free(self.h, C.c_void_p(p))
            self._pinned = []
            self.lib.msc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode(seq):
    """Chromosome::help + ChromosomeOneDigit::encode -> (codes, [(s, e)...], effective length)."""
    lib = _capi.load_library()
    seq = seq if isinstance(seq, bytes) else seq.encode()
    codes = np.zeros(max(len(seq), 1), dtype=np.uint8)
    max_segs = len(seq) // 2 + 2
    segs = np.zeros(2 * max_segs, dtype=np.int64)
    n, eff = C.c_size_t(), C.c_uint64()
    rc = lib.msc_encode(seq, len(seq), codes.ctypes.data_as(C.POINTER(C.c_uint8)), segs.ctypes.data_as(C.POINTER(C.c_int64)),
                        max_segs, C.byref(n), C.byref(eff))
    if rc != 0:
        raise MscError(rc, "invalid nucleotide")
    return codes[:len(seq)].tobytes(), [(int(segs[2 * i]), int(segs[2 * i + 1])) for i in range(n.value)], eff.value


class HistogramSet:
    """Device-resident k-mer histograms (the `points` vector of DivergencePoint<T>*)."""

    def __init__(self, ctx, k, dtype, capacity, sparse_entries=0):
        """sparse_entries > 0 -> sparse layout able to hold that many stored bins in total (msc_hist_set_create_sparse)"""
        self.ctx, self.k, self.dtype, self.capacity = ctx, int(k), int(dtype), int(capacity)
        h = C.c_void_p()
        if sparse_entries:
            ctx.check(ctx.lib.msc_hist_set_create_sparse(ctx.h, self.k, self.dtype, self.capacity, int(sparse_entries), C.byref(h)))
        else:
            ctx.check(ctx.lib.msc_hist_set_create(ctx.h, self.k, self.dtype, self.capacity, C.byref(h)))
        self.h = h
        self.nbins = 4 ** self.k
        ctx._adopt(self)

    def nbytes(self):
        return self.ctx.lib.msc_hist_set_bytes(self.h)

    def entries(self, slot):
        return self.ctx.lib.msc_hist_set_entries(self.h, slot)

    def clear(self):
        """msc_hist_set_clear: a sparse set back to empty, its whole entry arena free (compaction target of a centre store)."""
        self.ctx.check(self.ctx.lib.msc_hist_set_clear(self.ctx.h, self.h))

    def build(self, seqs, first_slot=0, strip=False):
        """Loader<T>::get_point for a batch (clutil/Loader.cpp:112-179)."""
        seqs = [s if isinstance(s, bytes) else s.encode() for s in seqs]
        n = len(seqs)
        arr = (C.c_char_p * max(n, 1))(*seqs)
        lens = (C.c_uint64 * max(n, 1))(*[len(s) for s in seqs])
        self.ctx.check(self.ctx.lib.msc_hist_build(self.ctx.h, self.h, first_slot, n, arr, lens, 1 if strip else 0))

    def build_packed(self, first_slot, n_seqs, packed, n_bases, seg_seq, seg_start, seg_end, eff_len, one_mers):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        seg_seq = np.ascontiguousarray(seg_seq, dtype=np.uint32)
        seg_start = np.ascontiguousarray(seg_start, dtype=np.uint64)
        seg_end = np.ascontiguousarray(seg_end, dtype=np.uint64)
        eff_len = np.ascontiguousarray(eff_len, dtype=np.uint64)
        one_mers = np.ascontiguousarray(one_mers, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.msc_hist_build_packed(self.ctx.h, self.h, first_slot, n_seqs, _ptr(packed), int(n_bases), _ptr(seg_seq),
                                                          _ptr(seg_start), _ptr(seg_end), len(seg_seq), _ptr(eff_len), _ptr(one_mers)))

    def build_packed_dev(self, first_slot, n_seqs, packed_dev_ptr, n_bases, seg_seq, seg_start, seg_end, eff_len, one_mers):
        """build_packed with the 2-bit stream already in this device's memory (packed_dev_ptr: an integer device address, e.g. a torch
        tensor's data_ptr()); the segment table and the per-sequence scalars stay host arrays"""
        seg_seq = np.ascontiguousarray(seg_seq, dtype=np.uint32)
        seg_start = np.ascontiguousarray(seg_start, dtype=np.uint64)
        seg_end = np.ascontiguousarray(seg_end, dtype=np.uint64)
        eff_len = np.ascontiguousarray(eff_len, dtype=np.uint64)
        one_mers = np.ascontiguousarray(one_mers, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.msc_hist_build_packed_dev(self.ctx.h, self.h, first_slot, n_seqs, C.c_void_p(int(packed_dev_ptr)), int(n_bases), _ptr(seg_seq),
                                                              _ptr(seg_start), _ptr(seg_end), len(seg_seq), _ptr(eff_len), _ptr(one_mers)))

    def download(self, slot):
        out = np.zeros(self.nbins, dtype=NP_T[self.dtype])
        self.ctx.check(self.ctx.lib.msc_hist_download(self.ctx.h, self.h, slot, _ptr(out)))
        return out

    def upload(self, slot, bins, length, one_mers=None):
        bins = np.ascontiguousarray(bins, dtype=NP_T[self.dtype])
        assert bins.size == self.nbins
        om = None if one_mers is None else (C.c_uint64 * 4)(*one_mers)
        self.ctx.check(self.ctx.lib.msc_hist_upload(self.ctx.h, self.h, slot, _ptr(bins), int(length), om))

    def lengths(self, first=0, n=None):
        """effective lengths of n consecutive slots (one call)"""
        n = self.capacity - first if n is None else n
        out = np.zeros(n, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.msc_hist_lengths(self.ctx.h, self.h, first, n, _ptr(out)))
        return out

    def info(self, slot):
        hi = _capi.HistInfo()
        self.ctx.check(self.ctx.lib.msc_hist_info_get(self.ctx.h, self.h, slot, C.byref(hi)))
        return dict(mag=hi.mag, length=hi.length, sum=hi.sum, sum_sq=hi.sum_sq, max_count=hi.max_count,
                    one_mers=list(hi.one_mers), stddev=hi.stddev, overflow=hi.overflow, id=hi.id)

    def set_id(self, slot, id_):
        self.ctx.check(self.ctx.lib.msc_hist_set_id(self.ctx.h, self.h, slot, id_))

    def clone_from(self, dst_slot, src, src_slot):
        """DivergencePoint::clone"""
        self.ctx.check(self.ctx.lib.msc_hist_clone(self.ctx.h, self.h, dst_slot, src.h, src_slot))

    def assign_from(self, dst_slot, src, src_slot):
        """DivergencePoint::set (mag is NOT copied)"""
        self.ctx.check(self.ctx.lib.msc_hist_assign(self.ctx.h, self.h, dst_slot, src.h, src_slot))

    def copy_from(self, dst_slot, src, src_slot):
        """exact copy of a slot (stale magnitude included)"""
        self.ctx.check(self.ctx.lib.msc_hist_copy(self.ctx.h, self.h, dst_slot, src.h, src_slot))

    def clone_batch(self, dst_slots, src, src_slots):
        """DivergencePoint::clone of many slots in one launch per region"""
        d = np.ascontiguousarray(dst_slots, dtype=np.uint32)
        s_ = np.ascontiguousarray(src_slots, dtype=np.uint32)
        assert d.size == s_.size
        self.ctx.check(self.ctx.lib.msc_hist_clone_batch(self.ctx.h, self.h, _ptr(d), src.h, _ptr(s_), d.size))

    def copy_batch(self, dst_slots, src, src_slots):
        """exact copies of many slots in one launch per region"""
        d = np.ascontiguousarray(dst_slots, dtype=np.uint32)
        s_ = np.ascontiguousarray(src_slots, dtype=np.uint32)
        assert d.size == s_.size
        self.ctx.check(self.ctx.lib.msc_hist_copy_batch(self.ctx.h, self.h, _ptr(d), src.h, _ptr(s_), d.size))

    def device_view(self):
        b, s = C.c_void_p(), C.c_void_p()
        sb, ss = C.c_uint64(), C.c_uint64()
        self.ctx.check(self.ctx.lib.msc_hist_set_device_view(self.h, C.byref(b), C.byref(sb), C.byref(s), C.byref(ss)))
        return b.value, sb.value, s.value, ss.value

    def import_done(self, first_slot, n):
        self.ctx.check(self.ctx.lib.msc_hist_import_done(self.ctx.h, self.h, first_slot, n))

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.msc_hist_set_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _slots(slots, m=None):
    if slots is None:
        return None, int(m)
    a = np.ascontiguousarray(slots, dtype=np.uint32)
    return a, a.size


class Feature:
    """Feature<T> + GLM weights of one weights-file block (msc_model)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        self.k = ctx.lib.msc_model_k(handle)
        self.n_singles = ctx.lib.msc_model_n_singles(handle)
        self.n_combos = ctx.lib.msc_model_n_combos(handle)
        ctx._adopt(self)

    @classmethod
    def from_file(cls, ctx, path, block=0):
        h = C.c_void_p()
        ctx.check(ctx.lib.msc_model_load(ctx.h, path.encode(), block, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_text(cls, ctx, text, block=0):
        h = C.c_void_p()
        ctx.check(ctx.lib.msc_model_parse(ctx.h, text.encode() if isinstance(text, str) else text, block, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def create(cls, ctx, k, combos, weights, singles, bias=0.0):
        """combos: [(kind, flags)], weights: [w0, w1..], singles: [(flag, min, max)]"""
        nc, ns = len(combos), len(singles)
        kinds = (C.c_int * max(nc, 1))(*[c[0] for c in combos])
        flags = (C.c_uint64 * max(nc, 1))(*[c[1] for c in combos])
        w = (C.c_double * (nc + 1))(*weights)
        sf = (C.c_uint64 * max(ns, 1))(*[s[0] for s in singles])
        mn = (C.c_double * max(ns, 1))(*[s[1] for s in singles])
        mx = (C.c_double * max(ns, 1))(*[s[2] for s in singles])
        h = C.c_void_p()
        ctx.check(ctx.lib.msc_model_create(ctx.h, k, nc, kinds, flags, w, ns, sf, mn, mx, bias, C.byref(h)))
        return cls(ctx, h)

    def single_flags(self):
        out = (C.c_uint64 * max(self.n_singles, 1))()
        self.ctx.check(self.ctx.lib.msc_model_single_flags(self.h, out))
        return list(out[:self.n_singles])

    def set_bias(self, b):
        self.ctx.lib.msc_model_set_bias(self.h, b)

    def compute(self, cands, cand_slots, qset, q_slot, order=ORDER_CAND_FIRST, m=None):
        """Feature::compute + operator() + weighted sum for 1 query x m candidates.
        -> dict(singles [m,n_singles], combos [m,n_combos], sum [m], csum [m])"""
        sl, m = _slots(cand_slots, m)
        singles = np.zeros((m, self.n_singles))
        combos = np.zeros((m, self.n_combos))
        s = np.zeros(m)
        cs = np.zeros(m)
        self.ctx.check(self.ctx.lib.msc_score(self.ctx.h, self.h, cands.h, _ptr(sl), m, qset.h, q_slot, order,
                                              _ptr(singles), _ptr(combos), _ptr(s), _ptr(cs)))
        return dict(singles=singles, combos=combos, sum=s, csum=cs)

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.msc_model_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pair_features_raw(ctx, cands, cand_slots, qset, q_slot, feat_mask=FEAT_FAST, order=ORDER_CAND_FIRST, m=None):
    """The raw statistics of predict/Feature.cpp for 1 query x m candidates -> [m, popcount(mask)], ascending bit order."""
    sl, m = _slots(cand_slots, m)
    nf = bin(feat_mask).count("1")
    out = np.zeros((m, nf))
    ctx.check(ctx.lib.msc_pair_features_raw(ctx.h, cands.h, _ptr(sl), m, qset.h, q_slot, order, feat_mask, _ptr(out)))
    return out


def pinned_array(ctx, shape, dtype):
    """a numpy array over page-locked host memory (msc_host_alloc; lives as long as the context): device-to-host copies into it are
    not staged by the runtime. For result arrays that are filled call after call (score_multi's `out`)."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    ctx.check(ctx.lib.msc_host_alloc(ctx.h, n, C.byref(p)))
    ctx._pinned.append(p.value)          # released by Context.close (msc_host_free); the array must not be used after that
    buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def score_multi(ctx, feat, cands, cand_slots, qset, q_slots, order=ORDER_CAND_FIRST, m=None, feat_mask=0, want=("sum", "csum", "close"), out=None):
    """n_q queries x m candidates in one pass over the candidates (all-pairs shape).
    -> dict(sum [n_q,m], csum [n_q,m], close [n_q,m], raw [n_q,m,nf] or None); `want` limits what is copied back;
    out: dict of arrays of those shapes to fill instead of new ones (pinned_array: no allocation, no staged copy per call)"""
    sl, m = _slots(cand_slots, m)
    qs = np.ascontiguousarray(q_slots, dtype=np.uint32)
    nq = qs.size
    nf = bin(feat_mask).count("1")

    def arr(name, shape, dtype):
        a = (out or {}).get(name)
        if a is not None:
            if a.shape != shape or a.dtype != np.dtype(dtype) or not a.flags["C_CONTIGUOUS"]:
                raise ValueError("out[%r] must be a C-contiguous %s array of shape %r" % (name, np.dtype(dtype), shape))
            return a
        return np.zeros(shape, dtype=dtype)
    s = arr("sum", (nq, m), np.float64) if feat is not None and "sum" in want else None
    cs = arr("csum", (nq, m), np.float64) if feat is not None and "csum" in want else None
    close = arr("close", (nq, m), np.uint8) if feat is not None and "close" in want else None
    raw = arr("raw", (nq, m, nf), np.float64) if nf else None
    ctx.check(ctx.lib.msc_score_multi(ctx.h, feat.h if feat is not None else None, cands.h, _ptr(sl), m, qset.h, _ptr(qs), nq, order,
                                      _ptr(s), _ptr(cs), _ptr(close), feat_mask, _ptr(raw)))
    counts = None
    if close is not None and "counts" in want:          # row sums of `close`, from the device where the route keeps them
        counts = np.zeros(nq, dtype=np.uint64)
        if ctx.lib.msc_last_close_counts(ctx.h, _ptr(counts), nq) != 0:
            counts = np.einsum("ij->i", close, dtype=np.uint64)
    return dict(sum=s, csum=cs, close=close, raw=raw, counts=counts)


class Trainer:
    """The pair-scoring operators of cluster/Trainer.{h,cpp} (scoring half only; training is host work out of scope)."""

    def __init__(self, ctx, feat, cutoff):
        self.ctx, self.feat, self.cutoff = ctx, feat, float(cutoff)

    def get_close(self, points, cand_slots, qset, q_slot, m=None):
        """-> (close_flags[m], best_pos, best_sim, is_min)"""
        sl, m = _slots(cand_slots, m)
        flags = np.zeros(max(m, 1), dtype=np.uint8)
        bp, bs, im = C.c_int64(), C.c_double(), C.c_int()
        self.ctx.check(self.ctx.lib.msc_get_close(self.ctx.h, self.feat.h, self.cutoff, points.h, _ptr(sl), m, qset.h, q_slot,
                                                  _ptr(flags), C.byref(bp), C.byref(bs), C.byref(im)))
        return flags[:m], bp.value, bs.value, bool(im.value)

    def filter(self, centre_set, centre_slot, points, pt_slots, m=None):
        """-> keep[m] (1 = survives Trainer::filter)"""
        sl, m = _slots(pt_slots, m)
        keep = np.zeros(max(m, 1), dtype=np.uint8)
        n = C.c_uint64()
        self.ctx.check(self.ctx.lib.msc_filter(self.ctx.h, self.feat.h, self.cutoff, centre_set.h, centre_slot, points.h, _ptr(sl), m,
                                               _ptr(keep), C.byref(n)))
        return keep[:m]

    def merge(self, centres, centre_slots, current, begin, last, n=None):
        sl, n = _slots(centre_slots, n)
        out = C.c_int64()
        self.ctx.check(self.ctx.lib.msc_merge(self.ctx.h, self.feat.h, self.cutoff, centres.h, _ptr(sl), n, current, begin, last, C.byref(out)))
        return out.value

    def merge_all(self, centres, centre_slots, delta):
        """every Trainer::merge call of ClusterFactory's merge loop at once -> best[n] (what merge(i, i+1, min(n-1, i+delta)) returns)"""
        sl = np.ascontiguousarray(centre_slots, dtype=np.uint32)
        best = np.zeros(sl.size, dtype=np.int64)
        self.ctx.check(self.ctx.lib.msc_merge_all(self.ctx.h, self.feat.h, self.cutoff, centres.h, _ptr(sl), sl.size, int(delta), _ptr(best)))
        return best

    def merge_some(self, centres, centre_slots, delta, which):
        """... for the centres which[w] only -> best[len(which)]"""
        sl = np.ascontiguousarray(centre_slots, dtype=np.uint32)
        wh = np.ascontiguousarray(which, dtype=np.uint64)
        best = np.zeros(wh.size, dtype=np.int64)
        self.ctx.check(self.ctx.lib.msc_merge_some(self.ctx.h, self.feat.h, self.cutoff, centres.h, _ptr(sl), sl.size, int(delta), _ptr(wh), wh.size, _ptr(best)))
        return best

    def update_centres(self, centres, centre_slots, points, lists):
        """mean_shift_update for many centres: lists[c] = point slots of centre c's neighbourhood.
        -> (nearest_pos[n] (position inside lists[c], -1 if nothing survives the filter), n_kept[n])"""
        cs = np.ascontiguousarray(centre_slots, dtype=np.uint32)
        offsets = np.zeros(cs.size + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(x) for x in lists])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(x, dtype=np.uint32) for x in lists]) if len(lists) else np.zeros(0), dtype=np.uint32)
        nearest = np.zeros(cs.size, dtype=np.int64)
        kept = np.zeros(cs.size, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.msc_update_centres(self.ctx.h, self.feat.h, self.cutoff, centres.h, _ptr(cs), cs.size, points.h, _ptr(flat), _ptr(offsets),
                                                       _ptr(nearest), _ptr(kept)))
        return nearest, kept

    def closest(self, points, member_slots, m=None, want_mean=False):
        """get_mean / closest: -> (nearest_pos, dists[m], mean or None)"""
        sl, m = _slots(member_slots, m)
        d = np.zeros(m)
        mean = np.zeros(points.nbins) if want_mean else None
        pos = C.c_int64()
        self.ctx.check(self.ctx.lib.msc_mean_nearest(self.ctx.h, points.h, _ptr(sl), m, C.byref(pos), _ptr(d), _ptr(mean)))
        return pos.value, d, mean


class Window:
    """The length-sorted store of the accumulate loop kept on the device (msc_window): position i = slot slots[i] of `points`,
    alive until a get_close marks it or kill() removes it."""

    def __init__(self, ctx, points, slots):
        self.ctx, self.points = ctx, points
        sl = np.ascontiguousarray(slots, dtype=np.uint32)
        h = C.c_void_p()
        ctx.check(ctx.lib.msc_window_create(ctx.h, points.h, _ptr(sl), sl.size, C.byref(h)))
        self.h, self.n = h, sl.size
        ctx._adopt(self)

    def alive(self, first=0, end=None):
        return self.ctx.lib.msc_window_alive(self.h, first, self.n if end is None else end)

    def kill(self, positions):
        p = np.ascontiguousarray(positions, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.msc_window_kill(self.ctx.h, self.h, _ptr(p), p.size))

    def get_close(self, trainer, first, end, qset, q_slot):
        """Trainer::get_close over the alive positions of [first, end) -> (close positions (ascending; they die), best position, best_sim, is_min)"""
        lst = C.POINTER(C.c_uint32)()
        n, bp, bs, im = C.c_uint64(), C.c_int64(), C.c_double(), C.c_int()
        self.ctx.check(self.ctx.lib.msc_get_close_window(self.ctx.h, trainer.feat.h, trainer.cutoff, self.h, first, end, qset.h, q_slot, C.byref(lst), C.byref(n),
                                                         C.byref(bp), C.byref(bs), C.byref(im)))
        close = np.ctypeslib.as_array(lst, shape=(n.value,)).astype(np.int64) if n.value else np.zeros(0, dtype=np.int64)
        return close, bp.value, bs.value, bool(im.value)

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx.lib.msc_window_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_class(ctx, points, first_slots, second_slots, vals, n_train, feat_flags, min_feat, max_feat, ident):
    """Predictor::train's selection + GLM on labelled pairs -> (weights file text, training accuracy, testing accuracy)"""
    fs = np.ascontiguousarray(first_slots, dtype=np.uint32)
    ss = np.ascontiguousarray(second_slots, dtype=np.uint32)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    buf = C.create_string_buffer(1 << 16)
    a, b = C.c_double(), C.c_double()
    ctx.check(ctx.lib.msc_train_class(ctx.h, points.h, _ptr(fs), _ptr(ss), _ptr(va), n_train, fs.size - n_train, feat_flags, min_feat, max_feat, ident,
                                      buf, len(buf), C.byref(a), C.byref(b)))
    return buf.value.decode(), a.value, b.value


def train_regr(ctx, points, first_slots, second_slots, vals, n_train, feat_flags, max_feat, ident):
    """Predictor::train_regr's greedy selection + GLM on labelled pairs -> (weights file text (mode 2), training error, testing error)"""
    fs = np.ascontiguousarray(first_slots, dtype=np.uint32)
    ss = np.ascontiguousarray(second_slots, dtype=np.uint32)
    va = np.ascontiguousarray(vals, dtype=np.float64)
    buf = C.create_string_buffer(1 << 16)
    a, b = C.c_double(), C.c_double()
    ctx.check(ctx.lib.msc_train_regr(ctx.h, points.h, _ptr(fs), _ptr(ss), _ptr(va), n_train, fs.size - n_train, feat_flags, max_feat, ident, buf, len(buf),
                                     C.byref(a), C.byref(b)))
    return buf.value.decode(), a.value, b.value


def mean_nearest(ctx, points, member_slots, m=None, want_mean=False):
    return Trainer(ctx, None, 1.0).closest(points, member_slots, m, want_mean)


class Predictor:
    """Predictor::close / similarity (predict/Predictor.cpp:255-333) for one query against a database. Like fastcar's work()
    (fastcar/FC_Runner.cpp:432,446-458) it follows the weights file's mode: no classification block -> every entry is close,
    no regression block -> every similarity is 1."""

    def __init__(self, ctx, cls_feat, reg_feat=None):
        self.ctx, self.cls, self.reg = ctx, cls_feat, reg_feat

    @classmethod
    def from_text(cls, ctx, text):
        mode = 0
        for ln in text.splitlines():
            if ln.startswith("mode:"):
                mode = int(ln.split()[1])
                break
        c = Feature.from_text(ctx, text, 0) if mode & 1 else None
        r = Feature.from_text(ctx, text, 1) if mode & 2 else None
        return cls(ctx, c, r)

    @classmethod
    def from_file(cls, ctx, path):
        return cls.from_text(ctx, open(path).read())

    def search(self, db, db_slots, qset, q_slot, m=None):
        sl, m = _slots(db_slots, m)
        close = np.zeros(max(m, 1), dtype=np.uint8)
        sim = np.zeros(max(m, 1))
        self.ctx.check(self.ctx.lib.msc_search(self.ctx.h, self.cls.h if self.cls else None, self.reg.h if self.reg else None, db.h, _ptr(sl), m,
                                               qset.h, q_slot, _ptr(close), _ptr(sim)))
        return close[:m], sim[:m]
