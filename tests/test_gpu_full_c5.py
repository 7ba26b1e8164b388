"""GPU: BASELINE.json configs[4] (C5: 50 strains, ANI 95-99 %, 1 % / 0.5 % / 0.5 % read errors, --min_identity 0.90
--min_ovlp_len 1500) with its REAL recipe at the largest size a pass of which stays under a minute on one card:
scale 0.06 = 30 000 reads on 50 x 120 kb genomes, i.e. the configured 2 500x pooled depth and divergence with fewer
reads (the full 500 000 reads are ~280 passes of this size; a pass of 50 000 reads takes 79 s, of 100 000 reads 128 s).
Four fifths of the alignment tasks need a DP here: the stress the config names ("banded-DP LDS occupancy").  Most pieces are
fragments of chains cut at long gaps; the stub rule leaves their end extensions out and has their
tasks report scores only.  Checks: row predicates and order, slices merge to the unsharded
pass (= determinism across different batchings), the fallback DP forms agree on a slice."""
import os

import pytest

from fullsize import check_rows, same_file
from hylight_amd import api
from hylight_amd import workloads as W
from hylight_amd.stage import StageRunner

pytestmark = pytest.mark.gpu
SCALE = 0.06


@pytest.fixture(scope="module")
def c5(tmp_path_factory):
    d = tmp_path_factory.mktemp("c5")
    cfg = W.config("C5", SCALE)
    fa = str(d / "s1.fa")
    n, bases, _ = W.make_long(cfg, fa)
    assert n == 30_000
    r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True)
    out = str(d / "s1_s1.paf")
    rows = r.run(out, **cfg["stage"])
    st = api.last_stats()
    yield d, cfg, r, out, rows, st
    r.close()


def test_c5_is_dp_bound_and_rows_are_valid(c5):
    d, cfg, r, out, rows, st = c5
    assert st["align_tasks_dp"] > 0.6 * st["align_tasks"] and st["anchors"] > 5e9 and st["subruns"] >= 2
    # the gaps of more than 256 bases between chained anchors (divergent strains) are LONG blocks now, not cuts of the chain:
    # every overlap is one row, with its CIGAR (round 3: fragments, most of them stub candidates with score-only tasks)
    assert st["align_tasks_long"] > 1e6 and st["align_tasks_long"] > 0.02 * st["ava_rows"]
    # C5's final output is (nearly) empty - the work is in the candidate rows, and candidate rows/s is what this workload
    # measures.  Why: an error of read R at position p is an X against ALL of R's partners there, so on R's own key v = con
    # and c = con - v = 0: not supported (slr2:383-396 wants c >= mc).  The keys that ARE supported sit on the partners' side:
    # a position of read T collects the X of every read piled up on it that errs there, and at this pooled depth (hundreds of
    # reads per position, 1 % errors) at least mc of them do at nearly every position while con - v stays large.  So every
    # alignment crosses supported keys, its pair count / matchcount exceeds -thre 0.0025 (slr2:90-96) and the pair is dropped:
    # the reference's filter is made for corrected reads.  (Rounds 1-3 kept ~500 rows here: fragments whose partners'
    # fragments never met in the pile-up.)  The filter chain at this depth is compared with the oracle under a sweep of -thre
    # in tests/test_gpu_workloads_oracle.py::test_c5_depth_sample_matches_the_oracle.
    assert st["rows_after_v4"] > 1e5 and st["snp_events"] > 1e8
    check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 0)
    assert rows == sum(1 for _ in open(out))


def test_c5_slices_merge_to_the_full_pass(c5):
    d, cfg, r, out, rows, st = c5
    parts = []
    for k in range(4):
        p = str(d / f"slice{k}.paf")
        r.run(p, share=(k, 4), **cfg["stage"])
        parts.append(p)
    merged = str(d / "merged.paf")
    api.merge_scored_paf(parts, merged)
    assert same_file(merged, out)


@pytest.mark.parametrize("var", ["HLMI_NARROW_UNPACKED", "HLMI_NARROW_LONG_UNPACKED", "HLMI_STUB_FULL_ROWS", "HLMI_CHAIN_UNPACKED", "HLMI_NO_STUB"])
def test_c5_fallback_forms_agree_on_a_slice(c5, monkeypatch, var):
    d, cfg, r, out, rows, st = c5
    monkeypatch.setenv(var, "1")
    alt = str(d / f"alt_{var}.paf")
    r.run(alt, share=(7, 12), **cfg["stage"])
    monkeypatch.delenv(var)
    ref = str(d / "ref_slice.paf")
    if not os.path.exists(ref):
        r.run(ref, share=(7, 12), **cfg["stage"])
    assert same_file(alt, ref)
