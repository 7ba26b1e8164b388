"""CPU oracle for SURVEY 8f rank 3 (STARTED): the front of the SAVAGE / ViralQuasispecies overlap-graph assembler.
TEST INFRASTRUCTURE ONLY (tests/ may import it; the product never does).

PARITY UNPINNED.  tools/HaploConduct/src needs Boost (OverlapGraph.h:17-19, EdgeCalculator.cpp uses boost::trim_if /
split, ViralQuasispecies.cpp boost::program_options) and cannot be built in this image; the reference ships no test
vectors for it.  What follows restates the text of the two functions and nothing checks it against a run of the
reference.

  parse_overlaps       EdgeCalculator.cpp:561-666 (construct_edges, in front of process_overlaps), Overlap.h:37-72,196-203
  transitive_edges     GraphAlgos.cpp:746-795 (findTransEdges, nonemptyIntersect), :938-993 (removeTransitiveEdges up to
                       the deletion, incl. the branch-reduction schedule)
  overlap_score        EdgeCalculator.cpp:26-139 (score, phred_to_prob, overlap_score) and the single-single branch of
                       compute_overlap (:186-222) with Read.h:158-200 (reverse complement, reversed qualities) - round 3
"""
import ctypes


def _atoi(s):
    """C atoi: optional blanks, sign, leading digits; 0 when there is none."""
    s = s.lstrip(" \t\n\v\f\r")
    sign, i = 1, 0
    if s[:1] in "+-":
        sign = -1 if s[0] == "-" else 1
        i = 1
    j = i
    while j < len(s) and s[j].isdigit():
        j += 1
    return sign * int(s[i:j]) if j > i else 0


def _strtoul0(s):
    """strtoul(s, NULL, 0) for the ids of an overlaps file (decimal, 0x.. or 0.. prefixes)."""
    s = s.lstrip(" \t\n\v\f\r")
    neg = s[:1] == "-"
    if s[:1] in "+-":
        s = s[1:]
    base = 10
    if s[:2].lower() == "0x":
        base, s = 16, s[2:]
    elif s[:1] == "0":
        base = 8
    digits = "0123456789abcdef"[:base]
    j = 0
    while j < len(s) and s[j].lower() in digits:
        j += 1
    v = int(s[:j], base) if j else 0
    return ctypes.c_uint64(-v if neg else v).value


def parse_overlaps(path, min_len=150, min_perc=0, relax_pe=False, max_overlaps=100000000):
    """-> (edge candidates in file order, n_nonedge, n_skipped); a candidate is a dict of the 13 columns."""
    kept, nonedge, skipped = [], 0, 0
    data = open(path, "rb").read().decode("latin-1")
    lines = data.split("\n")
    if lines and lines[-1] == "":
        lines.pop()                                  # getline: no empty line behind the last '\n'
    for i, line in enumerate(lines):
        if i >= max_overlaps:
            break
        line = line.strip("\t ")                     # EdgeCalculator.cpp:583
        f = line.split("\t") if line else []         # :589-593
        if len(f) != 13:                             # :597
            skipped += 1
            continue
        dash = f[3] == "-"
        o = dict(id1=_strtoul0(f[0]), id2=_strtoul0(f[1]), pos1=_atoi(f[2]), pos2=0 if dash else _atoi(f[3]),
                 ord=f[4] if len(f[4]) == 1 else f[4].replace(" ", ""),
                 ori1=f[5] if len(f[5]) == 1 else f[5].replace(" ", ""),
                 ori2=f[6] if len(f[6]) == 1 else f[6].replace(" ", ""),
                 perc1=_atoi(f[7]), perc2=0 if dash else _atoi(f[8]), len1=_atoi(f[9]), len2=0 if dash else _atoi(f[10]),
                 type1=f[11] if len(f[11]) == 1 else "".join(c for c in f[11] if c not in "\n\t "),
                 type2=f[12] if len(f[12]) == 1 else "".join(c for c in f[12] if c not in "\n\t "))
        ok = (o["pos1"] >= 0 and o["pos2"] >= 0 and 0 <= o["perc1"] <= 100 and 0 <= o["perc2"] <= 100 and o["len1"] >= 0
              and o["len2"] >= 0 and o["ori1"] in ("+", "-") and o["ori2"] in ("+", "-") and o["type1"] in ("s", "p")
              and o["type2"] in ("s", "p") and o["ord"] in ("1", "2", "-")
              and (("s" in (o["type1"], o["type2"])) == (o["ord"] == "-")))
        if not ok:
            raise ValueError(f"{path}: line {i + 1} is not a valid overlap")       # the reference exits / asserts
        if o["id1"] == o["id2"]:                     # :601
            skipped += 1
            continue
        perc = int(0.5 * (o["perc1"] + o["perc2"])) if o["perc2"] > 0 else o["perc1"]       # Overlap.h:196-203
        ss = o["type1"] == "s" and o["type2"] == "s"
        anyp = "p" in (o["type1"], o["type2"])
        if o["len1"] >= min_len and ss:              # :608
            decided = True
        elif o["len1"] >= 0.5 * min_len and o["len2"] >= 0.5 * min_len and anyp:     # :614
            decided = True
        elif relax_pe and o["len1"] + o["len2"] >= min_len and anyp:                 # :622
            decided = True
        else:
            decided = False
        if not decided:
            nonedge += 1                             # :630 nonedge_overlaps
        elif perc >= min_perc:
            kept.append(o)
        else:
            skipped += 1
    return kept, nonedge, skipped


def _nonempty_intersect(l1, l2):                     # GraphAlgos.cpp:779-795
    i = j = 0
    while i < len(l1) and j < len(l2):
        if l1[i] == l2[j]:
            return True
        if l1[i] < l2[j]:
            i += 1
        else:
            j += 1
    return False


def _find_trans_edges(n, edges):                     # GraphAlgos.cpp:746-776 with removeTrans = false
    """edges: list of (u, v, k); returns the sub-list whose u -> v has a w with u -> w and w -> v."""
    adj_out = [[] for _ in range(n)]
    adj_in = [[] for _ in range(n)]
    for u, v, _ in edges:
        adj_out[u].append(v)
        adj_in[v].append(u)
    for l in adj_out:
        l.sort()
    for l in adj_in:
        l.sort()
    return [(u, v, k) for u, v, k in edges if _nonempty_intersect(adj_out[u], adj_in[v])]


def transitive_edges(n_vertices, src, dst, ovlen=None, remove_trans=1):
    """-> (flags per edge: bit 0 transitive in the last round, bit 1 scheduled by the branch reduction; count of bit 0)."""
    assert remove_trans in (1, 2, 3)
    cur = [(int(u), int(v), k) for k, (u, v) in enumerate(zip(src, dst))]
    for _ in range(remove_trans):                    # :958-969
        cur = _find_trans_edges(n_vertices, cur)
    flags = [0] * len(src)
    for _, _, k in cur:
        flags[k] |= 1
    if remove_trans == 1 and ovlen is not None:      # :970-993
        for u, v, k in cur:
            L = ovlen[k]
            for j, (a, b) in enumerate(zip(src, dst)):
                if (a == u and ovlen[j] <= L) or (b == v and ovlen[j] <= L):
                    flags[j] |= 2
    return flags, len(cur)


# ---- quality-aware overlap score (EdgeCalculator.cpp:26-139) -----------------------------------------------------------
import math

_COMP = {"A": "T", "T": "A", "C": "G", "G": "C", "N": "N"}


def phred_to_prob(phred):
    """EdgeCalculator.cpp:62-66: pow(10, -phred/10.0)."""
    return math.pow(10, -phred / 10.0)


def overlap_score(seq1, seq2, score1, score2, pos, mismatch=0.0, min_read_len=0):
    """EdgeCalculator.cpp:70-139 -> (score, mismatch_rate).  Doubles in the reference's order of operations; log / pow /
    exp are the platform's libm on both sides."""
    mismatch_rate = 1.0
    if pos >= len(seq1):
        return 0.0, mismatch_rate
    if len(seq1) < min_read_len or len(seq2) < min_read_len:
        return 0.0, mismatch_rate
    L = min(len(seq1) - pos, len(seq2))
    total_score, total_len, mismatch_count = 0.0, 0.0, 0
    for i in range(L):
        nt1, nt2 = seq1[i + pos], seq2[i]
        assert nt1 in "ATCGN" and nt2 in "ATCGN"
        p1, p2 = phred_to_prob(ord(score1[i + pos]) - 33), phred_to_prob(ord(score2[i]) - 33)
        if nt1 == "N" or nt2 == "N":                        # score() returns 1: skipped
            continue
        if nt1 == nt2:
            p = (1 - p1) * (1 - p2) + (p1 * p2) / 3.0
        else:
            p = p1 * (1 - p2) / 3.0 + p2 * (1 - p1) / 3.0 + (2 / 9.0) * p1 * p2
            mismatch_count += 1
        if p < mismatch:                                    # score() returns 2: unacceptable mismatch
            return 0.0, mismatch_rate
        total_score += math.log(p)
        total_len += 1
    if total_len == 0:
        return 0.0, mismatch_rate
    import numpy as np
    mismatch_rate = float(np.float32(mismatch_count)) / total_len      # float(mismatch_count)/total_len
    total_score = (1.0 / total_len) * total_score
    return math.exp(total_score), mismatch_rate


def single_single_edge(seq_a, phred_a, seq_b, phred_b, pos1, ori1, ori2, mismatch=0.0, min_read_len=0):
    """compute_overlap, type1 == type2 == "s" (EdgeCalculator.cpp:186-222): reads upper-cased on loading
    (FastqStorage.cpp:124), reverse complement / reversed qualities for a '-' orientation -> (score, mismatch_rate, pos3)."""
    def oriented(seq, phred, fwd):
        seq = seq.upper()
        return (seq, phred) if fwd else ("".join(_COMP[c] for c in reversed(seq)), phred[::-1])
    s1, q1 = oriented(seq_a, phred_a, ori1)
    s2, q2 = oriented(seq_b, phred_b, ori2)
    score, mr = overlap_score(s1, s2, q1, q2, pos1, mismatch, min_read_len)
    return score, mr, len(s1) - pos1 - len(s2)
