"""GPU: what ONE of the eight ranks of BASELINE.json configs[3] (C4) computes, through the entry point the multi-GPU
driver uses (hlmi_job_run(rank 0, world 8) = StageRunner.run(share=(0, 8))): the long-read all-vs-all share and the two
short-read calls of the hybrid pipeline (HyLight.py:200: short reads vs polished long contigs, :207: short reads vs the
short reads no contig explained; len_over 70, mc 3, short mode).

C4 itself (1 M long + 10 M short reads on 100 strains x 2 Mb, 5 000x pooled depth) is 3.7e13 anchors for the long call
alone - a quarter of an hour per rank - so the test runs C4@0.1: the same recipe, depth, divergence and --nsplit 1000 with
a tenth of the reads and genome (100 000 long reads, 1 000 000 short reads).  The share is checked like the other
full-size runs: predicates + order, equal to the same chunks computed inside a 2-rank split (rank 0 of 8 = ranks 0 and 8
of 16), deterministic."""
import pytest

from fullsize import check_rows, same_file
from hylight_amd import api
from hylight_amd import simulate as S
from hylight_amd import workloads as W
from hylight_amd.stage import StageRunner

pytestmark = pytest.mark.gpu
SCALE = 0.1


@pytest.fixture(scope="module")
def c4(tmp_path_factory):
    d = tmp_path_factory.mktemp("c4")
    cfg = W.config("C4", SCALE)
    long_fa, short_fa = str(d / "long.fa"), str(d / "short.fa")
    n, _, strains = W.make_long(cfg, long_fa)
    n_short = W.make_short(cfg, strains, short_fa)
    assert n == 100_000 and n_short == 1_000_000
    # "polished long contigs": 40 kb pieces of every strain
    contigs = []
    for k, g in enumerate(strains):
        for a in range(0, len(g) - 10_000, 40_000):
            contigs.append(S.Read(f"longr_con_{len(contigs)}", g[a:a + 40_000].copy(), None, k, a, a + 40_000, False))
    con_fa = str(d / "long_con_polished.fa")
    S.write_fasta(contigs, con_fa)
    return d, cfg, long_fa, short_fa, con_fa, strains


def test_c4_long_share_of_rank0(c4):
    d, cfg, long_fa, short_fa, con_fa, strains = c4
    r = StageRunner(long_fa, long_fa, cfg["nsplit"], long_mode=True)
    try:
        out = str(d / "long_r0.paf")
        rows = r.run(out, share=(0, 8), **cfg["stage"])
        st = api.last_stats()
        assert st["queries"] == 100_000 and 12_000 < st["targets"] < 13_000 and st["anchors"] > 3e10
        assert rows == sum(1 for _ in open(out))
        check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 20)
        halves = []
        for k in (0, 8):                                  # chunks c % 8 == 0  =  c % 16 in {0, 8}
            p = str(d / f"long_r{k}of16.paf")
            r.run(p, share=(k, 16), **cfg["stage"])
            halves.append(p)
        merged = str(d / "long_merged.paf")
        api.merge_scored_paf(halves, merged)
        assert same_file(merged, out)
    finally:
        r.close()


def test_c4_short_vs_contigs_share_of_rank0(c4):
    d, cfg, long_fa, short_fa, con_fa, strains = c4
    st_short = cfg["stage_short"]
    r = StageRunner(short_fa, con_fa, cfg["nsplit"], long_mode=False)      # HyLight.py:200
    try:
        out = str(d / "shortr1_r0.paf")
        rows = r.run(out, share=(0, 8), **st_short)
        st = api.last_stats()
        assert st["queries"] == 1_000_000 and rows > 10_000
        check_rows(out, st_short["len_over"], st_short["iden"], 10_000)
        again = str(d / "shortr1_again.paf")
        r.run(again, share=(0, 8), **st_short)
        assert same_file(again, out)
    finally:
        r.close()


def test_c4_short_vs_remaining_short_share_of_rank0(c4):
    """HyLight.py:207: targets = the short reads pick_up left over, here every fifth pair (200 000 reads)."""
    d, cfg, long_fa, short_fa, con_fa, strains = c4
    remain = str(d / "remain.fa")
    with open(short_fa) as f, open(remain, "w") as o:
        for i, line in enumerate(f):
            if (i // 4) % 5 == 0:                          # 4 lines = one pair in the 2-line FASTA
                o.write(line)
    st_short = cfg["stage_short"]
    r = StageRunner(short_fa, remain, cfg["nsplit"], long_mode=False)
    try:
        out = str(d / "shortr2_r0.paf")
        rows = r.run(out, share=(0, 8), **st_short)
        st = api.last_stats()
        assert st["queries"] == 1_000_000 and 24_000 < st["targets"] < 26_000
        assert rows == sum(1 for _ in open(out)) and rows > 1_000
        # short mode keeps no pair-once rule: predicates per row, order
        check_rows(out, st_short["len_over"], st_short["iden"], 1_000, pair_once=False)
    finally:
        r.close()
