"""Helpers of the full-size GPU tests (BASELINE.json configs C2..C5): predicates every final row must satisfy
(script/filter_overlap_slr2.py:77-152) and a streaming file comparison."""
import os


def check_rows(path, len_over, iden, min_rows, pair_once=True):
    """14 columns + trailing TAB, one row per unordered pair, overhang rule, scores as printed, sort -k12 -nr."""
    seen = set()
    prev = None
    n = 0
    for line in open(path):
        c = line.rstrip("\n").split("\t")
        assert len(c) == 15 and c[14] == ""                        # slr2:151
        q, ql, qs, qe, strand, t, tl, ts, te, mc, ln = c[0], int(c[1]), int(c[2]), int(c[3]), c[4], c[5], int(c[6]), \
            int(c[7]), int(c[8]), int(c[9]), int(c[10])
        assert q != t and mc >= len_over                           # slr2:101-105 (column 10 against min_ovlp_len)
        key = (q, t) if q < t else (t, q)
        assert key not in seen or not pair_once                    # slr2:133-136 (per chunk: with both directions reported,
        seen.add(key)                                              #  a pair may come back from another chunk)
        if strand == "-":
            ts, te = tl - te, tl - ts
        assert min(qs, ts) + min(ql - qe, tl - te) <= min(4, max(qe - qs, te - ts) * 0.8)      # slr2:116-131
        assert c[11] == format(0.4 * (mc / ((ql + tl) / 2)) + 0.6 * (mc / ln), ".4f")           # slr2:142
        assert c[13] == format(mc / ln, ".4f") and float(c[12]) >= iden                         # slr2:144,146
        s = float(c[11])
        assert prev is None or s <= prev                           # utils.py:69
        prev = s
        n += 1
    assert n >= min_rows, n
    return n


def same_file(a, b):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(1 << 22), fb.read(1 << 22)
            if x != y:
                return False
            if not x:
                return True
