"""Pins oracle/filters.py (the CPU restatement) against golden vectors produced by running
the reference's own scripts (tests/golden/make_goldens.py).  CPU only."""
import json

import pytest

from oracle import filters as F


def test_v4_window_filter_fixture_A(golden):
    assert F.window_filter(golden.lines("fxA_ava.paf"), 4, min_len=30, min_o=3) == golden.lines("fxA_v4.paf")


@pytest.mark.parametrize("kw,gold", [
    (dict(min_len=30, min_o=3), "fxB_v4.paf"),
    (dict(min_len=100, min_iden=0.9, min_o=40), "fxB_v4_len100_oh40.paf"),
])
def test_v4_window_filter_dense_quirks(golden, kw, gold):
    # fixture B: >60 rows per query per window, duplicate pairs inside / across windows, self hits
    out = F.window_filter(golden.lines("fxB_dense.paf"), 4, **kw)
    assert out == golden.lines(gold)
    assert len(out) < len(golden.lines("fxB_dense.paf"))


@pytest.mark.parametrize("kw,gold", [
    (dict(min_len=90, min_iden=0.99, min_o=2, sfo=True), "fxC_v3.sfo"),
    (dict(min_len=90, min_iden=0.9, min_o=30, sfo=True), "fxC_v3_oh30.sfo"),
    (dict(min_len=90, min_iden=0.9, min_o=30, sfo=False), "fxC_v3_oh30.score"),
])
def test_v3_window_filter(golden, kw, gold):
    assert F.window_filter(golden.lines("fxC_contigs.paf"), 3, **kw) == golden.lines(gold)


@pytest.mark.parametrize("tag", ["fxC_v3", "fxC_v3_oh30"])
def test_sfo2overlaps(golden, tag):
    assert F.sfo2overlaps(golden.lines(tag + ".sfo")) == golden.lines(tag + ".savage")


@pytest.mark.parametrize("tag,args", [("len90_oh30", (90, 0.9, 30, 0.8)), ("len90_oh1", (90, 0.98, 1, 0.8))])
def test_filter_ovlp_inline_and_minimap22sfo(golden, tag, args):
    # reference CLIs chained as in polyte.tune_params.py:507-515
    kept = F.ovlp_inline_filter(golden.lines("fxC_contigs.paf"), *args)
    assert kept == golden.lines(f"fxC_inline_{tag}.paf")
    assert F.minimap22sfo(kept, 0, 0) == golden.lines(f"fxC_inline_{tag}.sfo")


def test_minimap22sfo_thresholds(golden):
    assert F.minimap22sfo(golden.lines("fxC_contigs.paf"), 200, 99) == golden.lines("fxC_m22sfo_m200_p99.sfo")


def test_intermediate_sort(golden):
    assert F.sort_intermediate(golden.lines("fxA_v4.paf")) == golden.lines("fxA_v4_sorted.paf")


@pytest.mark.parametrize("mode", ["long", "short"])
def test_snp_pileup_and_support(golden, mode):
    g = json.loads(golden.text(f"fxA_snp_{mode}.json"))
    snp, partners, intervals = F.snp_pileup(golden.lines("fxA_v4_sorted.paf"), long_mode=(mode == "long"))
    assert {f"{r}:{p}": v for (r, p), v in snp.items()} == g["snp"]
    assert sum(len(v) for v in partners.values()) == g["n_map_po"]
    assert sum(len(v) for v in intervals.values()) == g["n_start_po"]
    for mc in (2, 3):
        mut = F.supported_pair_counts(snp, partners, intervals, mc)
        assert {f"{a}:{b}": v for (a, b), v in mut.items()} == g[f"mutation_mc{mc}"]


@pytest.mark.parametrize("mode", ["long", "short"])
def test_numpy_pair_counts_equal_the_definition(golden, mode):
    """pair_counts_np (what worker() uses on full-size chunks) against the reference's own mutation counts and against the
    per-event definition: the golden rows, and simulated all-vs-all rows with real CIGARs (truth_paf of the simulator)."""
    long_mode = mode == "long"
    g = json.loads(golden.text(f"fxA_snp_{mode}.json"))
    rows = golden.lines("fxA_v4_sorted.paf")
    for mc in (2, 3):
        mut = F.pair_counts_np(rows, long_mode, mc)
        assert mut is not None and {f"{a}:{b}": v for (a, b), v in mut.items()} == g[f"mutation_mc{mc}"]
    from hylight_amd import simulate as S
    reads, _ = S.simulate_reads(seed=91, n_strains=3, genome_len=25_000, n_reads=220, mean_len=4_000, min_len=1_500, max_len=9_000,
                                snp_rate=0.02, err_sub=0.01, err_ins=0.004, err_del=0.004, keep_gpos=True)
    sim = F.sort_intermediate(S.truth_paf(reads, pair_once=long_mode))
    assert len(sim) > 2_000
    for mc in (2, 3):
        snp, partners, intervals = F.snp_pileup(sim, long_mode)
        want = dict(F.supported_pair_counts(snp, partners, intervals, mc))
        assert len(want) > 100 and F.pair_counts_np(sim, long_mode, mc) == want
    # rows the numpy form refuses go the definitional way inside worker(): a CIGAR it does not know, an interval without extent
    assert F.pair_counts_np(["a\t100\t0\t50\t+\tb\t100\t0\t50\t50\t50\t0\tcg:Z:25=1B24="], True, 2) is None
    assert F.pair_counts_np(["a\t100\t0\t50\t+\tb\t100\t7\t7\t50\t50\t0\tcg:Z:50="], True, 2) is None
    assert F.worker(golden.lines("fxA_ava.paf"), long_mode, 1000 if long_mode else 70, 2, 0.95, fast=False) == \
        F.worker(golden.lines("fxA_ava.paf"), long_mode, 1000 if long_mode else 70, 2, 0.95, fast=True)


def test_x_digit_sum(golden):
    for k, v in json.loads(golden.text("sum_before_X.json")).items():
        assert F.x_digit_sum(k) == v


@pytest.mark.parametrize("tag,kw", [
    ("long_len1000", dict(long_mode=True, min_ovlp_len=1000, mc=2, iden=0.95)),
    ("long_len3000_iden99", dict(long_mode=True, min_ovlp_len=3000, mc=2, iden=0.99)),
    ("short_len70", dict(long_mode=False, min_ovlp_len=70, mc=3, iden=0.95)),
])
def test_worker_end_to_end(golden, tag, kw):
    # reference: filter_overlap_slr2.main with a minimap2 PATH shim; one chunk = all reads
    assert F.worker(golden.lines("fxA_ava.paf"), **kw) == golden.lines(f"fxA_worker_{tag}.paf")


def test_stage_nsplit4(golden):
    # reference: utils.split_reads2(nsplit=4) with the shim selecting rows by target chunk
    names = [l[1:] for l in golden.lines("fxA_reads.fa") if l.startswith(">")]
    raw = golden.lines("fxA_ava.paf")
    chunks = []
    for lo, hi in F.chunk_ranges(2 * len(names), 4):
        tn = set(names[lo // 2:(hi + 1) // 2])
        chunks.append([l for l in raw if l.split("\t")[5] in tn])
    assert len(chunks) == 4
    assert F.stage(chunks, True, 1000, 2, 0.95) == golden.lines("fxA_stage_nsplit4.paf")
