"""The sharded mean-shift driver (meshclust2_amd/host/msc_sharded.hpp: what msc_cluster runs with one GPU per rank) on CPU: 1, 2 and 3
processes shard the points, every rank runs the clustering logic of msc_driver.hpp on replicated bookkeeping, the operators exchange
through msc::TcpComm (plain sockets; RCCL takes its place between GPUs), and the rank-local scorer is the CPU oracle (test
infrastructure: tests/sharded_oracle_main.cpp, built by oracle/Makefile). Rank 0's .clstr must be the reference CLI's own output byte
for byte (cfg1: k = 5 / 16-bit; k9_u8: k = 9 with the 8-bit type the reference chose by itself; mixed_slow: mixed lengths, a
`--feat slow` model, --id 0.6), and the collectives of a get_close step are counted: at most one broadcast and one all-gather."""
import os
import re
import socket
import subprocess

import pytest

from golden_util import GOLDEN
from meshclust2_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "sharded_oracle")


def _mixed_slow():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    seqs, hdrs = [], []
    for gi, (n, length, seed) in enumerate(((800, 600, 31), (800, 1000, 32), (800, 1500, 33))):      # gen_golden.mixed_length_set
        s_, h = synth.families(seed, n, length, length_jitter=120)
        seqs += s_
        hdrs += [">m%d_%s" % (gi, x[1:]) for x in h]
    return seqs[::3], hdrs[::3]


# block: the ownership block (1000 in production: one bvec bin; smaller here so that every rank owns points of every window)
CASES = {
    "cfg1": dict(make=lambda: synth.families(20260001, 1000, 1000), k=5, dtype=16, weights="weights_k5_u16.txt", sim=0.9, clstr="cfg1.clstr", block=100),
    "k9_u8": dict(make=lambda: synth.families(61, 320, 1000, family=16), k=9, dtype=8, weights="weights_k9_u8.txt", sim=0.9, clstr="k9_u8.clstr", block=40),
    "mixed": dict(make=lambda: synth.families(777, 700, 1000, family=20, length_jitter=100), k=5, dtype=16, weights="weights_k5_u16.txt", sim=0.9, clstr=None, block=100),
    "mixed_slow": dict(make=_mixed_slow, k=6, dtype=16, weights="weights_mixed_slow_k6_u16.txt", sim=0.6, clstr="mixed_slow.clstr", block=64),
}


def _free_port():
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    return port


def run_world(tmp_path, case, world, name="out.clstr", extra_env=None):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle", "sharded_oracle"])
    c = CASES[case]
    fa = str(tmp_path / (case + ".fa"))
    if not os.path.exists(fa):
        seqs, hdrs = c["make"]()
        synth.write_fasta(fa, seqs, hdrs)
    out = str(tmp_path / name)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([BIN, fa, os.path.join(GOLDEN, c["weights"]), str(c["k"]), str(c["dtype"]), str(c["sim"]), out, str(c["block"])],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=900)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    m = re.search(r"collectives: broadcast (\d+) all_gather (\d+) all_reduce (\d+) bytes (\d+) \| get_close steps (\d+) collectives (\d+) overflow (\d+) "
                  r"\| closest (\d+) update chunks (\d+) set chunks (\d+)", logs[0])
    assert m, logs[0][-2000:]
    keys = ("broadcast", "all_gather", "all_reduce", "bytes", "steps", "step_collectives", "overflow", "closest", "update_chunks", "set_chunks")
    return open(out, "rb").read(), dict(zip(keys, (int(x) for x in m.groups()))), logs[0]


@pytest.mark.parametrize("case,world", [("cfg1", 1), ("cfg1", 2), ("cfg1", 3), ("k9_u8", 2), ("mixed_slow", 2), ("mixed_slow", 3)])
def test_sharded_mean_shift_writes_the_reference_clstr(oracle, tmp_path, case, world):
    got, calls, _ = run_world(tmp_path, case, world)
    assert got == open(os.path.join(GOLDEN, CASES[case]["clstr"]), "rb").read()
    if world > 1:
        # a get_close step: at most one broadcast (the query) and one all-gather (the records), a second all-gather only on overflow
        assert calls["step_collectives"] <= 2 * calls["steps"] + calls["overflow"]
        assert calls["all_reduce"] > 0          # get_mean went through the column-sum reduction


def test_sharded_mean_shift_mixed_lengths(oracle, tmp_path):
    """real windows at every step (lengths 900-1100): 1, 2 and 3 ranks write the same bytes, batched and centre by centre, with
    thousands of exchanges; the batched update round costs collectives per CHUNK of centres, not per centre"""
    one, _, _ = run_world(tmp_path, "mixed", 1, "w1.clstr")
    two, calls, log = run_world(tmp_path, "mixed", 2, "w2.clstr")
    three, _, _ = run_world(tmp_path, "mixed", 3, "w3.clstr")
    serial, scalls, _ = run_world(tmp_path, "mixed", 2, "w2s.clstr", extra_env={"MSC_SERIAL_UPDATE": "1"})
    n_clusters = one.count(b">Cluster")
    assert n_clusters > 50 and calls["steps"] > 150
    assert one == two == three == serial
    assert calls["step_collectives"] <= 2 * calls["steps"] + calls["overflow"]
    # the update stage: a handful of chunks per round (15 rounds at most, + the final one), each 1 all-reduce + 1 all-gather, against
    # several collectives per centre and round in the serial order
    assert 0 < calls["update_chunks"] <= 17 and calls["set_chunks"] <= 17
    assert scalls["all_gather"] + scalls["all_reduce"] > 3 * (calls["all_gather"] + calls["all_reduce"])
    # the merge loop asks only about the centres whose delta + 1 histograms changed since the last round (msc_driver.hpp: merge_round ->
    # ShardedBackend::merge_some -> the engine's subset form; here the oracle engine's): taken, and for fewer centres than there are
    m = re.search(r"merge rounds through merge_some: (\d+) asked (\d+) of (\d+)", log)
    assert m, log[-500:]
    rounds, asked, of = (int(x) for x in m.groups())
    assert rounds >= 2 and 0 < asked < of


def test_fasta_line_ends_and_skipped_lines(oracle, tmp_path):
    """msc::read_fasta (host/msc_fasta.hpp; nonltr/ChromListMaker.cpp:24-48,117-165): LF, CR LF and lone CR line ends, a last line without
    one, lines that start with a blank skipped, text in front of the first header dropped -- the same records, hence the same .clstr bytes,
    as the plain file (r05: the file is read in one piece and cut in memory, by several threads when it is long)."""
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle", "sharded_oracle"])
    c = CASES["k9_u8"]
    seqs, hdrs = synth.families(61, 80, 1000, family=16)
    plain = str(tmp_path / "plain.fa")
    synth.write_fasta(plain, seqs, hdrs)
    odd = str(tmp_path / "odd.fa")
    ends = (b"\n", b"\r\n", b"\r")
    with open(odd, "wb") as f:
        f.write(b"text in front of the first header\nACGTACGT\r\n")
        for i, (h, s_) in enumerate(zip(hdrs, seqs)):
            s_ = bytes(s_)
            e = ends[i % 3]
            f.write(h.encode() + e)
            if i % 4 == 1:
                f.write(b" a line that starts with a blank" + e + b"\tand one with a tab" + e)
            for o in range(0, len(s_), 61 + i % 7):
                f.write(s_[o:o + 61 + i % 7] + (e if i % 5 else ends[(i + o) % 3]))
            if i % 6 == 2:
                f.write(e)          # an empty line
        last = bytes(seqs[0])[:300]
        f.write(b">last_one" + b"\r\n" + last)          # no line end behind the last line
    with open(plain, "ab") as f:
        f.write(b">last_one\n" + last + b"\n")
    outs = []
    # (MSC_FASTA_SHARE: the bytes a reading thread takes at least -- 32 MiB in production; here the file is cut into as many shares as the
    # box has threads, up to sixteen, so that shares begin inside records, inside CR LF pairs and in the text in front of the first header)
    for fa, share in ((plain, None), (odd, None), (odd, 1), (odd, 4099), (plain, 257)):
        out = str(tmp_path / (os.path.basename(fa) + ".clstr"))
        env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        if share:
            env["MSC_FASTA_SHARE"] = str(share)
        r = subprocess.run([BIN, fa, os.path.join(GOLDEN, c["weights"]), str(c["k"]), str(c["dtype"]), str(c["sim"]), out, str(c["block"])], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
        outs.append(open(out, "rb").read())
    assert all(o == outs[0] for o in outs) and outs[0].count(b"last_one") == 1 and outs[0].count(b">Cluster") >= 2


@pytest.mark.parametrize("dead", [1, 0])
def test_a_rank_that_dies_takes_the_others_down(oracle, tmp_path, dead):
    """no test killed a peer. A rank whose peer is gone must END with an error, not wait for ever: msc::TcpComm
    (host/msc_comm.hpp) throws from the first send / receive that meets the closed socket. The dead rank is played by this test itself --
    it keeps the rendezvous (rank 1 says its number, rank 0 listens and accepts) and then closes the connection, which is what the kernel
    does to the sockets of a process that was killed -- so that the moment of death does not depend on a race with a real process."""
    import struct
    import time
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle", "sharded_oracle"])
    c = CASES["cfg1"]
    fa = str(tmp_path / "cfg1.fa")
    seqs, hdrs = c["make"]()
    synth.write_fasta(fa, seqs, hdrs)
    port = _free_port()
    sock_port = port + 1 + 16          # CommEnv::from_environment: MASTER_PORT + 1 + MSC_PORT_OFFSET (16 by default)
    alive = 1 - dead
    listener = None
    if dead == 0:
        listener = socket.socket()
        listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        listener.bind(("127.0.0.1", sock_port))
        listener.listen(2)
        listener.settimeout(60)
    env = dict(os.environ, RANK=str(alive), WORLD_SIZE="2", LOCAL_RANK=str(alive), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.Popen([BIN, fa, os.path.join(GOLDEN, c["weights"]), str(c["k"]), str(c["dtype"]), str(c["sim"]), str(tmp_path / "dead.clstr"), str(c["block"])],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        if dead == 0:
            conn, _ = listener.accept()
            assert struct.unpack("<i", conn.recv(4))[0] == 1
            conn.close()
            listener.close()
        else:
            t0 = time.time()
            while True:
                try:
                    conn = socket.create_connection(("127.0.0.1", sock_port), timeout=5)
                    break
                except OSError:
                    assert time.time() - t0 < 60 and p.poll() is None, "rank 0 never listened"
                    time.sleep(0.05)
            conn.sendall(struct.pack("<i", 1))
            conn.close()
        t0 = time.time()
        log = p.communicate(timeout=120)[0].decode(errors="replace")
    except subprocess.TimeoutExpired:
        p.kill()
        pytest.fail("the surviving rank hung")
    assert p.returncode != 0, log[-2000:]
    assert "a peer went away" in log, log[-2000:]
    assert time.time() - t0 < 60
