"""GPU: size-independent properties at the FULL size of BASELINE.json configs[1] (C2: 10 000 reads x 8 kb, 5 strains x
400 kb, --nsplit 100; 1.3e9 anchors, ~2e6 aligned candidate rows per pass) - far beyond what the CPU oracle can check
directly: a pass is deterministic, sharding over 4 ranks and merging reproduces the unsharded file byte for byte, every
final row satisfies the predicates of pass 2 and the rows come out in `sort -k12 -nr` order."""
import os

import pytest

from hylight_amd import api
from hylight_amd import workloads as W

pytestmark = pytest.mark.gpu

LEN_OVER, MC, IDEN = 6000, 2, 0.95     # script/HyLight.py:130


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    d = tmp_path_factory.mktemp("c2full")
    fa = d / "s1.fa"
    W.make_long(W.config("C2"), str(fa))
    out = d / "s1_s1.paf"
    api.split_reads2(fa, fa, 100, d, out, len_over=LEN_OVER, mc=MC, iden=IDEN, long=True)
    return d, fa, out


def test_full_size_pass_is_deterministic_and_shardable(c2):
    d, fa, out = c2
    ref = open(out).read()
    assert ref.count("\n") > 40_000
    again = d / "again.paf"
    api.split_reads2(fa, fa, 100, d, again, len_over=LEN_OVER, mc=MC, iden=IDEN, long=True)
    assert open(again).read() == ref
    parts = []
    for rank in range(4):
        p = d / f"part{rank}.paf"
        api.split_reads2(fa, fa, 100, d, p, len_over=LEN_OVER, mc=MC, iden=IDEN, long=True, rank=rank, world=4)
        assert os.path.getsize(p) > 0
        parts.append(p)
    merged = d / "merged.paf"
    api.merge_scored_paf(parts, merged)
    assert open(merged).read() == ref


def test_full_size_rows_satisfy_pass2_and_order(c2):
    d, fa, out = c2
    seen = set()
    prev = None
    n = 0
    for line in open(out):
        c = line.rstrip("\n").split("\t")
        assert len(c) == 15 and c[14] == ""                        # 14 columns + trailing TAB (slr2:151)
        q, ql, qs, qe, strand, t, tl, ts, te, mc, ln = c[0], int(c[1]), int(c[2]), int(c[3]), c[4], c[5], int(c[6]), \
            int(c[7]), int(c[8]), int(c[9]), int(c[10])
        assert q != t and mc >= LEN_OVER
        key = (q, t) if q < t else (t, q)
        assert key not in seen                                     # one row per unordered pair (slr2:133-136)
        seen.add(key)
        if strand == "-":
            ts, te = tl - te, tl - ts
        assert min(qs, ts) + min(ql - qe, tl - te) <= min(4, max(qe - qs, te - ts) * 0.8)      # slr2:116-131
        assert c[11] == format(0.4 * (mc / ((ql + tl) / 2)) + 0.6 * (mc / ln), ".4f")           # slr2:142
        assert c[13] == format(mc / ln, ".4f") and float(c[12]) >= IDEN
        s = float(c[11])
        assert prev is None or s <= prev                           # sort -k12 -nr (utils.py:69)
        prev = s
        n += 1
    assert n > 40_000


@pytest.mark.parametrize("var", ["HLMI_NARROW_UNPACKED", "HLMI_CHAIN_UNPACKED", "HLMI_ANCHOR_PAIRS", "HLMI_ANCHOR_SPLIT", "HLMI_NO_RANK_WORD", "HLMI_SNP_SORT", "HLMI_NO_SHIFT_CERT", "HLMI_NO_GAP1_CERT", "HLMI_NO_GAP2_CERT", "HLMI_NO_SUFFIX_TRIM", "HLMI_NO_ONE_PIECE_CERT", "HLMI_NO_EXT_CERT", "HLMI_CHAIN_NO_DP16", "HLMI_CHAIN_NO_SMALL"])
def test_full_size_fallback_forms_agree(c2, monkeypatch, var):
    """The forms of the kernels that unusual inputs take (32-bit DP, two-register chain state, key + value anchors,
    two-array index search) give the same file at full size."""
    d, fa, out = c2
    monkeypatch.setenv(var, "1")
    alt = d / f"alt_{var}.paf"
    api.split_reads2(fa, fa, 100, d, alt, len_over=LEN_OVER, mc=MC, iden=IDEN, long=True)
    assert open(alt).read() == open(out).read()
