"""GPU: BASELINE.json configs[2] (C3) at FULL size - 100 000 synthetic ONT reads, mean 10 kb, 20 strains x 1 Mb at
ANI 98.5-99.5 %, --nsplit 200, the constants of script/HyLight.py:130 - through the same entry points bench.py uses.
A pass is ~8e10 anchors and ~1e8 aligned candidate rows, far beyond the CPU oracle, so the checks are the
size-independent ones: a pass is deterministic; the 8 slices bench.py steps through (one rank's share of an 8-rank job
each) merge to exactly the unsharded pass; every final row satisfies pass 2's predicates and the file is in
`sort -k12 -nr` order; forcing the fallback forms of the kernels on one slice changes nothing."""
import os

import pytest

from fullsize import check_rows, same_file
from hylight_amd import api
from hylight_amd import workloads as W
from hylight_amd.stage import StageRunner

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3(tmp_path_factory):
    d = tmp_path_factory.mktemp("c3full")
    cfg = W.config("C3")
    fa = str(d / "s1.fa")
    n, bases, _ = W.make_long(cfg, fa)
    assert n == 100_000 and bases > 9e8
    r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True)
    out = str(d / "s1_s1.paf")
    rows = r.run(out, **cfg["stage"])
    st = api.last_stats()
    yield d, cfg, r, out, rows, st
    r.close()


def test_c3_counts_are_the_configured_workload(c3):
    d, cfg, r, out, rows, st = c3
    assert st["queries"] == st["targets"] == 100_000 and st["chunks_run"] >= 199
    assert st["anchors"] > 5e10 and st["ava_rows"] > 5e7          # ~1000x pooled depth
    assert rows == sum(1 for _ in open(out)) and rows > 10_000


def test_c3_rows_satisfy_pass2_and_order(c3):
    d, cfg, r, out, rows, st = c3
    check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 10_000)


def test_c3_slices_merge_to_the_full_pass_and_pass_is_deterministic(c3):
    """The eight slices are computed by eight independent runs (what eight ranks would do, and what bench.py's steps
    are): different chunk sets, different query batches, different launch shapes than the unsharded pass above - the
    merged bytes must be the same."""
    d, cfg, r, out, rows, st = c3
    parts = []
    for k in range(8):
        p = str(d / f"slice{k}.paf")
        r.run(p, share=(k, 8), **cfg["stage"])
        assert os.path.getsize(p) > 0
        parts.append(p)
    merged = str(d / "merged.paf")
    api.merge_scored_paf(parts, merged)
    assert same_file(merged, out)


@pytest.mark.parametrize("var", ["HLMI_NARROW_UNPACKED", "HLMI_CHAIN_UNPACKED", "HLMI_ANCHOR_PAIRS", "HLMI_NO_RANK_WORD", "HLMI_SNP_SORT", "HLMI_NO_SHIFT_CERT", "HLMI_NO_GAP1_CERT", "HLMI_NO_GAP2_CERT", "HLMI_NO_SUFFIX_TRIM", "HLMI_NO_ONE_PIECE_CERT", "HLMI_NO_EXT_CERT", "HLMI_CHAIN_NO_DP16", "HLMI_CHAIN_NO_SMALL", "HLMI_SEED_NO_GUESS"])
def test_c3_fallback_forms_agree_on_a_slice(c3, monkeypatch, var):
    d, cfg, r, out, rows, st = c3
    monkeypatch.setenv(var, "1")
    alt = str(d / f"alt_{var}.paf")
    r.run(alt, share=(3, 8), **cfg["stage"])
    assert same_file(alt, str(d / "slice3.paf"))


@pytest.mark.parametrize("lanes,cuts", [("1", None), ("4", "25,50,75"), ("2", "")])
def test_c3_lanes_agree_on_a_slice(c3, monkeypatch, lanes, cuts):
    """Query batches in flight (runtime.cpp: lanes): one after the other, two (the default the slices above ran with), three or
    four at a time, the set-aside pieces (LONG tasks) aligned at different points of the pass - the same bytes."""
    d, cfg, r, out, rows, st = c3
    monkeypatch.setenv("HLMI_LANES", lanes)
    if cuts is not None:
        monkeypatch.setenv("HLMI_SET_ASIDE_CUTS", cuts)
    alt = str(d / f"alt_lanes_{lanes}_{cuts}.paf")
    r.run(alt, share=(3, 8), **cfg["stage"])
    s2 = api.last_stats()
    assert (s2.get("ava_lanes", 1) == int(lanes)) and s2["lanes_fit"] == 1
    assert s2["align_pieces_deferred"] > 0            # (there are set-aside pieces to move around)
    assert same_file(alt, str(d / "slice3.paf"))
