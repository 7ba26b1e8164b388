"""the front of the SAVAGE overlap-graph assembler.  PARITY UNPINNED - the reference
(tools/HaploConduct/src) needs Boost and ships no vectors; the oracle restates its text (oracle/vq.py) and these tests
hold the library to the oracle and the oracle to hand-made cases."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vq as OV  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _savage_lines(rng, n, n_reads=40):
    rows = []
    for _ in range(n):
        a, b = rng.randrange(n_reads), rng.randrange(n_reads)
        kind = rng.random()
        if kind < 0.7:
            rows.append(f"{a}\t{b}\t{rng.randrange(3000)}\t-\t-\t{rng.choice('+-')}\t{rng.choice('+-')}\t{rng.randrange(80, 101)}\t-\t{rng.randrange(40, 400)}\t-\ts\ts")
        else:
            t1 = rng.choice("sp")                     # (an 's' on either side goes with order '-': Overlap.h:127-135)
            rows.append(f"{a}\t{b}\t{rng.randrange(300)}\t{rng.randrange(300)}\t{'-' if t1 == 's' else rng.choice('12')}\t+\t-\t{rng.randrange(80, 101)}\t{rng.randrange(80, 101)}\t{rng.randrange(40, 200)}\t{rng.randrange(40, 200)}\t{t1}\tp")
    return rows


def test_oracle_parser_on_hand_made_lines(tmp_path):
    p = tmp_path / "ov.txt"
    p.write_text("\n".join([
        "1\t2\t100\t-\t-\t+\t+\t99\t-\t200\t-\ts\ts",            # edge candidate
        "  3\t4\t5\t-\t-\t+\t-\t99\t-\t149\t-\ts\ts\t ",          # outer blanks trimmed; too short: non-edge
        "5\t5\t0\t-\t-\t+\t+\t100\t-\t500\t-\ts\ts",              # self overlap: skipped
        "6\t7\t10\t20\t1\t+\t-\t90\t80\t80\t90\tp\tp",            # paired: both halves >= 75, perc (90+80)/2 = 85
        "8\t9\t1\t2\t2\t+\t+\t90\t90\t100\t60\tp\ts",            # 'p' with an 's' needs ord '-': invalid in the reference ...
    ][:4] + ["short line", "", "0x10\t011\t1\t-\t-\t-\t-\t70\t-\t150\t-\ts\ts"]) + "\n")
    kept, nonedge, skipped = OV.parse_overlaps(str(p), 150, 80)
    assert [(o["id1"], o["id2"]) for o in kept] == [(1, 2), (6, 7)]        # the last row: ids 16, 9 (strtoul base 0), perc 70 < 80
    assert nonedge == 1 and skipped == 4
    kept, _, _ = OV.parse_overlaps(str(p), 150, 0)
    assert (kept[-1]["id1"], kept[-1]["id2"]) == (16, 9)
    with pytest.raises(ValueError):
        q = tmp_path / "bad.txt"
        q.write_text("8\t9\t1\t2\t2\t+\t+\t90\t90\t100\t60\tp\ts\n")
        OV.parse_overlaps(str(q))


def test_oracle_transitive_edges_on_hand_made_graphs():
    # a -> b -> c with the shortcut a -> c: only the shortcut is transitive
    flags, n = OV.transitive_edges(3, [0, 1, 0], [1, 2, 2])
    assert flags == [0, 0, 1] and n == 1
    # a tournament on 4 vertices in topological order: edges spanning >= 2 steps are transitive; in the graph of THOSE,
    # 0 -> 3 has no inner vertex left (0 -> 2 -> 3 needs 2 -> 3, a non-transitive edge): nothing is double transitive
    src, dst = zip(*[(i, j) for i in range(4) for j in range(i + 1, 4)])
    flags, n = OV.transitive_edges(4, src, dst)
    assert [f for f in flags] == [0, 1, 1, 0, 1, 0] and n == 3
    assert OV.transitive_edges(4, src, dst, remove_trans=2)[1] == 0
    # on 5 vertices 0 -> 4 is double transitive through 0 -> 2 -> 4
    src, dst = zip(*[(i, j) for i in range(5) for j in range(i + 1, 5)])
    flags, n = OV.transitive_edges(5, src, dst, remove_trans=2)
    assert n == 1 and flags[list(zip(src, dst)).index((0, 4))] == 1
    # branch reduction: the transitive edge 0 -> 2 (length 50) schedules the out-edges of 0 and the in-edges of 2 that are
    # no longer than it
    flags, _ = OV.transitive_edges(4, [0, 1, 0, 0, 3], [1, 2, 2, 3, 2], ovlen=[40, 90, 50, 60, 50])
    assert flags == [2, 0, 3, 0, 2]


def test_library_parser_matches_oracle(tmp_path):
    from hylight_amd import api
    rng = random.Random(5)
    p = tmp_path / "ov.txt"
    p.write_text("\n".join(_savage_lines(rng, 400) + ["junk", "\t \t", "7\t7\t1\t-\t-\t+\t+\t99\t-\t300\t-\ts\ts"]) + "\n")
    for args in ((150, 0, False), (100, 90, False), (120, 85, True)):
        want = OV.parse_overlaps(str(p), *args)
        got = api.vq_parse_overlaps(str(p), *args)
        assert got[1:] == want[1:] and got[0] == want[0]
    # the path's own SAVAGE files parse completely (single-end rows only)
    for name in sorted(os.listdir(GOLD)):
        if name.endswith(".savage") or "savage" in name:
            f = os.path.join(GOLD, name)
            assert api.vq_parse_overlaps(f, 0, 0) == OV.parse_overlaps(f, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,e", [(1, 30, 200), (2, 200, 3000), (3, 50, 2400), (4, 3000, 40000)])
def test_library_transitive_edges_match_oracle(seed, n, e):
    from hylight_amd import api
    rng = random.Random(seed)
    # overlap-like: vertices on a line, edges to later vertices within a window (+ a few random ones, duplicates allowed)
    src, dst, ln = [], [], []
    for _ in range(e):
        u = rng.randrange(n)
        v = min(n - 1, u + 1 + int(rng.expovariate(0.15))) if rng.random() < 0.9 else rng.randrange(n)
        if u == v:
            continue
        src.append(u); dst.append(v); ln.append(rng.randrange(50, 500))
    for rt in (1, 2, 3):
        want = OV.transitive_edges(n, src, dst, ln if rt == 1 else None, rt)
        got = api.vq_transitive_edges(n, src, dst, ln if rt == 1 else None, rt)
        assert got[1] == want[1] and got[0] == want[0], (seed, rt)
    assert OV.transitive_edges(n, src, dst)[1] > 0


@pytest.mark.gpu
def test_library_transitive_edges_hub_vertex():
    """A vertex with more out-edges than its wave's LDS table holds takes the other kernel."""
    from hylight_amd import api
    n = 3000
    src = [0] * (n - 1) + list(range(1, n - 1))
    dst = list(range(1, n)) + list(range(2, n))
    want = OV.transitive_edges(n, src, dst)
    got = api.vq_transitive_edges(n, src, dst)
    assert got[1] == want[1] == n - 2 and got[0] == want[0]


# ---- quality-aware overlap score (EdgeCalculator.cpp:26-139, round 3) -----------------------------------------------------
def test_oracle_overlap_score_known_answers():
    import math
    # constant quality '=' (Q28: what HyLight.extend_con writes, HyLight.py:289-305), every base matching:
    # p = (1 - e)^2 + e^2 / 3 per position, score = exp(mean log p) = p
    e = math.pow(10, -28 / 10.0)
    p = (1 - e) * (1 - e) + (e * e) / 3.0
    s, mr = OV.overlap_score("ACGTACGTAC", "GTACGTAC", "=" * 10, "=" * 8, 2)
    assert abs(s - p) < 1e-15 and mr == 0.0
    # one substitution among 8 positions
    px = e * (1 - e) / 3.0 + e * (1 - e) / 3.0 + (2 / 9.0) * e * e
    s, mr = OV.overlap_score("ACGTACGTAC", "GTACCTAC", "=" * 10, "=" * 8, 2)
    assert abs(s - math.exp((7 * math.log(p) + math.log(px)) / 8)) < 1e-15 and mr == 0.125
    # that substitution is unacceptable once `mismatch` exceeds its probability; N positions do not count; early returns
    assert OV.overlap_score("ACGTACGTAC", "GTACCTAC", "=" * 10, "=" * 8, 2, mismatch=0.01) == (0.0, 1.0)
    s, mr = OV.overlap_score("ACGTACGTAC", "GTNCGTAC", "=" * 10, "=" * 8, 2)
    assert abs(s - p) < 1e-15 and mr == 0.0
    assert OV.overlap_score("ACGT", "ACGT", "====", "====", 4) == (0.0, 1.0)                 # pos behind read 1
    assert OV.overlap_score("ACGT", "ACGT", "====", "====", 0, min_read_len=5) == (0.0, 1.0)
    assert OV.overlap_score("NNNN", "ACGT", "====", "====", 0) == (0.0, 1.0)                 # no countable position
    # orientation: read 2 reverse-complemented, its qualities reversed
    s, mr, pos3 = OV.single_single_edge("ACGTACGTAC", "IIIIIIIII5", "GTACGTAC", "5IIIIIII", 2, True, False)
    want, _ = OV.overlap_score("ACGTACGTAC", "GTACGTAC", "IIIIIIIII5", "IIIIIII5", 2)
    assert s == want and pos3 == 0


@pytest.mark.gpu
def test_library_overlap_scores_match_oracle(tmp_path):
    from hylight_amd import api
    rng = random.Random(11)
    reads = {}
    fq = tmp_path / "singles.fastq"
    with open(fq, "w") as f:
        for k in range(1, 61):
            n = rng.randint(160, 900)
            seq = "".join(rng.choice("ACGT") for _ in range(n))
            if k % 7 == 0:
                seq = seq[:50] + "N" * 3 + seq[53:]
            if k % 9 == 0:
                seq = seq.lower()                                   # upper-cased on loading
            q = "".join(chr(33 + rng.randint(2, 41)) for _ in range(n)) if k % 3 else "=" * n
            reads[k] = (seq, q)
            f.write(f"@{k} extra\n{seq}\n+\n{q}\n")
    comp = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}
    ovs = []
    for _ in range(400):                                            # overlaps that really overlap, with a few substitutions
        a, b = rng.sample(range(1, 61), 2)
        ori1, ori2 = rng.random() < 0.7, rng.random() < 0.7
        sa = reads[a][0].upper()
        s1 = sa if ori1 else "".join(comp[c] for c in reversed(sa))
        pos = rng.randint(0, len(s1) - 40)
        # rewrite read b so that its oriented form continues read a's from pos (plus noise): keep its own qualities
        L = len(reads[b][0])
        tgt = (s1[pos:] + "".join(rng.choice("ACGT") for _ in range(L)))[:L]
        tgt = "".join(c if rng.random() > 0.01 or c == "N" else rng.choice("ACGT") for c in tgt)
        sb = tgt if ori2 else "".join(comp[c] for c in reversed(tgt))
        ovs.append((a, b, pos, ori1, ori2, sb))
    # every overlap gets its own copy of read b (ids 1000+)
    with open(fq, "a") as f:
        for k, (a, b, pos, ori1, ori2, sb) in enumerate(ovs):
            reads[1000 + k] = (sb, reads[b][1])
            f.write(f"@{1000 + k}\n{sb}\n+\n{reads[b][1]}\n")
    recs = [dict(id1=a, id2=1000 + k, pos1=pos, ori1="+" if o1 else "-", ori2="+" if o2 else "-")
            for k, (a, b, pos, o1, o2, sb) in enumerate(ovs)]
    recs.append(dict(id1=1, id2=2, pos1=len(reads[1][0]) + 5, ori1="+", ori2="+"))          # pos behind read 1
    for mismatch, min_len in ((0.0, 0), (1e-4, 0), (0.0, 400)):
        got = api.vq_overlap_scores(fq, recs, mismatch=mismatch, min_read_len=min_len)
        n_pos = 0
        for r, (s, mr, p3) in zip(recs, got):
            a, b = reads[r["id1"]], reads[r["id2"]]
            ws, wmr, wp3 = OV.single_single_edge(a[0], a[1], b[0], b[1], r["pos1"], r["ori1"] == "+", r["ori2"] == "+", mismatch, min_len)
            assert (s, mr, p3) == (ws, wmr, wp3), (r, s, ws)          # the same doubles, bit for bit
            n_pos += s > 0
        assert n_pos > (150 if mismatch == 0.0 and min_len == 0 else 10)
