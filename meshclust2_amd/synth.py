"""Seeded synthetic DNA families: byte-identical on every box.

Families of `family` sequences around uniform-ACGT templates; each member carries `sub_rate` substitutions and
`indel_rate` single-base insertions/deletions (indels are mandatory: with substitution-only data
`length_difference` is constant and Feature::normalize throws, predict/Feature.cpp:248-253).
Randomness is a counter-based SplitMix64 stream, so any (seed, sequence index) can be produced independently.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(x):
    """vectorised SplitMix64 finaliser over uint64 counters"""
    with np.errstate(over="ignore"):
        z = (x + _GOLD).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(seed, stream_id, n):
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream_id)], dtype=np.uint64))[0]
        return splitmix64(base + np.arange(n, dtype=np.uint64) * _GOLD)


def _unit(u):
    return (u >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def template(seed, t, length):
    return (_stream(seed, 2 * t, length) >> np.uint64(62)).astype(np.uint8)


def member(seed, t, j, tmpl, sub_rate=0.03, indel_rate=0.005):
    """member j of template t as 0..3 codes"""
    n = tmpl.size
    r = _stream(seed, (np.uint64(1) << np.uint64(40)) + np.uint64(t) * np.uint64(4096) + np.uint64(2 * j + 1), 3 * n)
    u = _unit(r[:n])
    shift = (r[n:2 * n] >> np.uint64(62)).astype(np.uint8) % 3 + 1
    ins_base = (r[2 * n:] >> np.uint64(62)).astype(np.uint8)
    codes = tmpl.copy()
    sub = u < sub_rate
    codes[sub] = (codes[sub] + shift[sub]) & 3
    dele = (u >= sub_rate) & (u < sub_rate + indel_rate / 2)
    ins = (u >= sub_rate + indel_rate / 2) & (u < sub_rate + indel_rate)
    counts = np.ones(n, dtype=np.int64)
    counts[dele] = 0
    counts[ins] = 2
    out = np.repeat(codes, counts)
    ends = np.cumsum(counts)
    out[ends[ins] - 1] = ins_base[ins]
    return out


def family_codes(seed, n_seqs, length, family=20, sub_rate=0.03, indel_rate=0.005, length_jitter=0):
    """-> list of uint8 code arrays (0..3), headers"""
    seqs, headers = [], []
    n_templates = (n_seqs + family - 1) // family
    i = 0
    for t in range(n_templates):
        ln = length
        if length_jitter:
            ln = int(length + (int(_stream(seed, 2 * t + 1, 1)[0] % np.uint64(2 * length_jitter + 1)) - length_jitter))
        tmpl = template(seed, t, ln)
        for j in range(family):
            if i >= n_seqs:
                break
            seqs.append(member(seed, t, j, tmpl, sub_rate, indel_rate))
            headers.append(">seq%d template_%d" % (i, t))
            i += 1
    return seqs, headers


def to_ascii(codes):
    return _BASES[codes].tobytes()


def families(seed, n_seqs, length, **kw):
    """-> (list of ASCII sequences (bytes), headers)"""
    seqs, headers = family_codes(seed, n_seqs, length, **kw)
    return [to_ascii(s) for s in seqs], headers


def write_fasta(path, seqs, headers, width=70):
    with open(path, "wb") as f:
        for h, s in zip(headers, seqs):
            f.write(h.encode() + b"\n")
            for o in range(0, len(s), width):
                f.write(s[o:o + width] + b"\n")


def pack_batch(code_seqs):
    """Packed 2-bit input of msc_hist_build_packed for pure-ACGT sequences (one segment per sequence, the whole
    sequence; sequences shorter than 21 bases are encoded the same way here because no N is present).
    -> dict(packed, n_bases, seg_seq, seg_start, seg_end, eff_len, one_mers)"""
    lens = np.array([s.size for s in code_seqs], dtype=np.uint64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    n_bases = int(lens.sum())
    allc = np.concatenate(code_seqs).astype(np.uint8) if code_seqs else np.zeros(0, np.uint8)
    pad = (-n_bases) % 4
    if pad:
        allc = np.concatenate([allc, np.zeros(pad, np.uint8)])
    q = allc.reshape(-1, 4)
    packed = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
    one = np.ones((len(code_seqs), 4), dtype=np.uint64)
    for i, s in enumerate(code_seqs):
        one[i] += np.bincount(s, minlength=4).astype(np.uint64)
    nonempty = lens > 0
    idx = np.nonzero(nonempty)[0].astype(np.uint32)
    return dict(packed=packed, n_bases=n_bases, seg_seq=idx, seg_start=starts[nonempty],
                seg_end=(starts + lens - np.uint64(1))[nonempty], eff_len=lens, one_mers=one.reshape(-1))
