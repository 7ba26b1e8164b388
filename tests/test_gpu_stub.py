"""GPU: the stub rule (hlmi_ava_opts::stub_oh; proof at oracle/ava_oracle.c:is_stub).  An alignment piece that ends so deep
inside both reads that no end extension (up to max(256, max_gap) rows) can reach a sequence end fails the overhang test of
filter_trans_ovlp_inline_v4.py:52-64 whatever its extensions find.  With stub_oh >= 0 such a piece is reported without its
extensions (it only occupies a line of the filter's 1000-line windows).  Pieces like that come from chains that are cut in
the middle of an overlap: a structural difference between two strains - here deletions of 80-150 bases, more than the 39
diagonals one alignment block may shift by - in reads long enough for the cut to lie more than max_gap bases inside both.

  * rows with the rule on == the oracle's rows with the rule on, bit for bit (stubs materialised on both sides)
  * the stage's final rows do not depend on the rule (HLMI_NO_STUB switches it off): that is its correctness statement
"""
import os

import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from hylight_amd import workloads as W
from oracle import ava as OA

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def divergent(tmp_path_factory):
    """A small C5-like set: C5's strain divergence and read errors on 6 strains x 30 kb."""
    d = tmp_path_factory.mktemp("stub")
    sim = dict(W.CONFIGS["C5"]["sim"], n_strains=6, genome_len=30_000, n_reads=260, mean_len=7_000)
    reads, _ = S.simulate_reads(seed=411, min_len=2_000, max_len=20_000, **sim)
    fa = d / "r.fa"
    S.write_fasta(reads, fa)
    return d, fa


@pytest.fixture(scope="module")
def structural(tmp_path_factory):
    """Two strains of 160 kb that differ by 1 % SNPs and by a deletion of 80-150 bases every ~25 kb; 150 reads of 35-60 kb
    with C3's read errors: a chain across a deletion is cut there, most cuts lie more than max_gap inside both reads."""
    d = tmp_path_factory.mktemp("stub_sv")
    rng = np.random.default_rng(97)
    a = S._BASES[rng.integers(0, 4, size=160_000)]
    b = a.copy()
    pos = rng.choice(len(b), size=len(b) // 100, replace=False)
    b[pos] = S._BASES[(np.searchsorted(S._BASES, b[pos]) + rng.integers(1, 4, size=len(pos))) % 4]
    keep = np.ones(len(b), dtype=bool)
    for at in range(20_000, 150_000, 25_000):
        keep[at:at + int(rng.integers(80, 150))] = False
    strains = [a, b[keep]]
    reads = []
    for i in range(150):
        g = strains[i % 2]
        base, _, s0, rev = S._draw_read(rng, g, int(rng.integers(35_000, 60_000)), 0.003, 0.001, 0.001, 0.5, False)
        reads.append(S.Read(f"sv{i:03d}", base, None, i % 2, s0, s0 + len(base), rev))
    fa = d / "sv.fa"
    S.write_fasta(reads, fa)
    return d, fa


@pytest.mark.parametrize("mode", ["long", "long_divergent", "short_constants"])
def test_rows_with_stubs_match_the_oracle(divergent, structural, mode):
    d, fa = structural if mode == "long" else divergent
    if mode.startswith("long"):
        og, oo = api.ava_opts_long(), OA.opts_long()
    else:                              # the end bonus enters the rule's score bound: blocks >= min_dp_score + end_bonus
        og, oo = api.ava_opts_short(), OA.opts_short()
        for o in (og, oo):
            o.pair_once = 1
    og.stub_oh = oo.stub_oh = 3
    api.ava(fa, fa, d / f"g_{mode}.paf", og)
    st = api.last_stats()
    OA.ava(fa, fa, d / f"o_{mode}.paf", oo)
    pieces, stubs = OA.last_counts()
    got, want = open(d / f"g_{mode}.paf").read(), open(d / f"o_{mode}.paf").read()
    assert got == want and want.count("\n") == pieces
    if mode == "long":
        assert stubs > 0.2 * pieces > 100                       # the rule bites on this input ...
        assert st["align_ext_held"] > 0 and st["align_tasks_long"] > 0
        # and the rows differ from the fully extended ones (the stubs are visible here, by design)
        OA.ava(fa, fa, d / "o_full.paf")
        assert open(d / "o_full.paf").read() != want
    if mode == "short_constants":
        assert stubs > 0.3 * pieces > 300
        assert st["align_ext_held"] > st["align_ext_late"] > 0  # some held-back extensions had to run after all


def test_stage_output_does_not_depend_on_the_rule(structural, monkeypatch):
    d, fa = structural
    stage = dict(len_over=1500, mc=2, iden=0.90)
    on, off = d / "on.paf", d / "off.paf"
    api.split_reads2(fa, fa, 4, d, on, long=True, **stage)
    st_on = api.last_stats()
    monkeypatch.setenv("HLMI_NO_STUB", "1")
    api.split_reads2(fa, fa, 4, d, off, long=True, **stage)
    st_off = api.last_stats()
    assert open(on).read() == open(off).read() and os.path.getsize(on) > 0
    assert st_on["ava_rows"] == st_off["ava_rows"] and st_on["rows_after_v4"] == st_off["rows_after_v4"]
    assert st_on["align_ext_held"] > 0 and st_off.get("align_ext_held", 0) == 0
    assert st_on["align_tasks_long"] < st_off["align_tasks_long"]          # (the held-back extensions are LONG tasks here)
    # the candidates' tasks report scores only: fewer CIGAR ops come out of the overlapper, the same final rows
    assert st_on["align_tasks_score_only"] > 0 and st_on["cigar_ops"] < 0.9 * st_off["cigar_ops"]
    monkeypatch.delenv("HLMI_NO_STUB")
    monkeypatch.setenv("HLMI_STUB_FULL_ROWS", "1")                  # stubs, but with their blocks' CIGARs (the round-3a form)
    full = d / "full_rows.paf"
    api.split_reads2(fa, fa, 4, d, full, long=True, **stage)
    assert open(full).read() == open(on).read() and api.last_stats()["align_tasks_score_only"] == 0
