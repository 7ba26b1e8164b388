"""GPU: BASELINE.json configs[4] (C5) and configs[3] (C4, long reads) at their FULL size - 500 000 reads on 50 strains x 2 Mb
with --min_identity 0.90 --min_ovlp_len 1500, and 1 000 000 long reads on 100 strains x 2 Mb - each through the entry point
a rank of the multi-GPU job uses (hlmi_job_run).  The complete read set is resident in HBM and sketched (all 5 / 10 Gbases
are queries of every chunk).  Round 4: C5 as one rank's share of an 8-rank job (8 chunks; and as two shares whose merge equals it; the COMPLETE pass
of 60 chunks with HL_FULL_PASS=1),
C4's long reads as the share one rank of an 8-rank job computes (125 of the 1 000 chunks) - row predicates, rows == sum of
the slices, no refused sub-run, the HBM high-water mark of the pass.  The read sets are made by the block-parallel simulator
(hylight_amd/simulate.py:simulate_reads_to_fasta) in seconds."""
import os
import time

import pytest

from fullsize import check_rows, same_file
from hylight_amd import api
from hylight_amd import workloads as W
from hylight_amd.stage import StageRunner

pytestmark = pytest.mark.gpu


def _host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2**30
    except Exception:
        return 0.0


def _free_gb(path):
    st = os.statvfs(path)
    return st.f_bavail * st.f_frsize / 2**30


def _slice_rows(path):
    with open(path) as f:
        return sum(1 for _ in f)


def _c5_share(tmp_path, share, halves, min_cand):
    """`share` = (k, n) of C5's 60 chunks x all 500 000 queries in one hlmi_job_run (the stage cuts it into sub-runs by itself),
    then again as the two shares `halves` that make it up: their rows add up and their merge is the same file."""
    cfg = W.config("C5")
    fa = str(tmp_path / "c5.fa")
    t0 = time.time()
    n, bases, _ = W.make_long(cfg, fa)
    t_sim = time.time() - t0
    assert n == 500_000 and bases > 4.5e9
    t0 = time.time()
    r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True)
    try:
        t_open = time.time() - t0
        assert r.job.num_queries == 500_000 and r.job.num_chunks == 60
        out = str(tmp_path / "pass.paf")
        t0 = time.time()
        r.prepare()
        t_sketch = time.time() - t0
        t0 = time.time()
        rows = r.run(out, share=share, **cfg["stage"])
        t_run = time.time() - t0
        st = api.last_stats()
        n_chunks = len(range(share[0], 60, share[1]))
        print(f"C5 full size: simulate {t_sim:.1f} s, parse + upload {t_open:.1f} s, sketch of 5 Gbases {t_sketch:.2f} s, {n_chunks} of 60 chunks "
              f"(c % {share[1]} == {share[0]}) {t_run:.1f} s -> {rows} overlaps; anchors {st['anchors']:.3g}, candidate rows {st['ava_rows']:.3g}, "
              f"LONG tasks {st.get('align_tasks_long', 0):.3g}, sub-runs {st['subruns']:.0f} (refused {st.get('subruns_refused', 0):.0f}), "
              f"HBM high-water mark {st['hbm_peak_in_use_gb']:.0f} GB", flush=True)
        assert st["queries"] == 500_000 and st["chunks_run"] == n_chunks
        assert st["minimizers_q"] > 1.2e9 and st["ava_rows"] > min_cand and st["align_tasks_long"] > st["ava_rows"]
        assert st.get("subruns_refused", 0) == 0 and 0 < st["hbm_peak_in_use_gb"] < 288
        assert rows == _slice_rows(out) == st["rows_out"]
        check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 0)
        parts = []
        for k, sh in enumerate(halves):
            part = str(tmp_path / f"pass.part{k}")
            r.run(part, share=sh, **cfg["stage"])
            print(f"  share {sh}: {api.last_stats()['t_total_s']:.1f} s", flush=True)
            parts.append(part)
        assert sum(_slice_rows(p) for p in parts) == rows
        api.merge_scored_paf(parts, str(tmp_path / "merged.paf"))
        assert same_file(str(tmp_path / "merged.paf"), out)
    finally:
        r.close()
    os.remove(fa)


def test_c5_full_size_one_rank_share(tmp_path):
    """Eight of C5's 60 chunks, spread over the file (c % 8 == 1: what one rank of an 8-rank job computes), and the same as two
    shares of a 16-rank job."""
    if _free_gb(tmp_path) < 12 or _host_gb() < 24:
        pytest.skip("needs 12 GB of scratch space and 24 GB of host memory")
    _c5_share(tmp_path, (1, 8), [(1, 16), (9, 16)], 1e8)


@pytest.mark.skipif(not os.environ.get("HL_FULL_PASS"), reason="a COMPLETE C5 pass takes ~10 minutes on one card: HL_FULL_PASS=1 "
                    "(run once per round, log under profiles/)")
def test_c5_full_size_complete_pass(tmp_path):
    """All 60 chunks x all 500 000 queries, and again as the two shares of a 2-rank job."""
    if _free_gb(tmp_path) < 12 or _host_gb() < 24:
        pytest.skip("needs 12 GB of scratch space and 24 GB of host memory")
    _c5_share(tmp_path, (0, 1), [(0, 2), (1, 2)], 1e9)


def test_c4_full_size_long_reads_one_rank_share(tmp_path):
    """configs[3]'s long reads at full size, the share ONE RANK of an 8-rank job computes: 125 of the 1 000 chunks (c % 8 == 3)
    against all 1 000 000 queries."""
    if _free_gb(tmp_path) < 24 or _host_gb() < 48:
        pytest.skip("needs 24 GB of scratch space and 48 GB of host memory")
    cfg = W.config("C4")
    fa = str(tmp_path / "c4.fa")
    t0 = time.time()
    n, bases, _ = W.make_long(cfg, fa)
    t_sim = time.time() - t0
    assert n == 1_000_000 and bases > 9e9
    t0 = time.time()
    r = StageRunner(fa, fa, cfg["nsplit"], long_mode=True)
    try:
        t_open = time.time() - t0
        assert r.job.num_queries == 1_000_000 and 990 <= r.job.num_chunks <= 1000
        out = str(tmp_path / "share.paf")
        t0 = time.time()
        r.prepare()                                                # 10 Gbases: sketched in parts of 3 Gbases (csrc/stage.cpp)
        t_sketch = time.time() - t0
        t0 = time.time()
        rows = r.run(out, share=(3, 8), **cfg["stage"])
        t_run = time.time() - t0
        st = api.last_stats()
    finally:
        r.close()
    os.remove(fa)
    print(f"C4 full size (long reads): simulate {t_sim:.1f} s, parse + upload {t_open:.1f} s, sketch of 10 Gbases {t_sketch:.2f} s, one rank's "
          f"share ({st['chunks_run']:.0f} of 1000 chunks) {t_run:.1f} s -> {rows} overlaps; anchors {st['anchors']:.3g}, "
          f"candidate rows {st['ava_rows']:.3g}, sub-runs {st['subruns']:.0f} (refused {st.get('subruns_refused', 0):.0f}), HBM high-water mark "
          f"{st['hbm_peak_in_use_gb']:.0f} GB", flush=True)
    assert st["queries"] == 1_000_000 and 120 <= st["chunks_run"] <= 125 and 120_000 < st["targets"] < 126_000
    assert st["anchors"] > 1e11 and st["minimizers_q"] > 2e9 and st["ava_rows"] > 2e8
    # (a sub-run whose anchors or output turn out above its budget is given up after the counting pass / the first query batch
    #  and retried smaller - it leaves no trace in the counts; the estimates from the sub-runs before it should make that rare)
    assert st.get("subruns_refused", 0) <= 0.1 * st["subruns"] and 0 < st["hbm_peak_in_use_gb"] < 288
    assert rows == _slice_rows(out) == st["rows_out"]
    # (at 5 000x pooled depth the mc = 2 support filter leaves next to nothing of the candidate rows)
    check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 0)


def test_c4_full_size_short_reads_one_rank_share(tmp_path):
    """configs[3]'s short-read side at full size: all 10 000 000 short reads (5 M pairs 2 x 250, 2.5 Gbases) resident and
    sketched, against the "polished long contigs" of HyLight.py:200 (40 kb pieces of the 100 strains x 2 Mb: 5 000
    contigs, 625 --nsplit chunks of 8) - the share one rank of an 8-rank job computes (78 chunks), short mode
    (len_over 70, mc 3)."""
    if _free_gb(tmp_path) < 12 or _host_gb() < 48:
        pytest.skip("needs 12 GB of scratch space and 48 GB of host memory")
    from hylight_amd import simulate as S
    cfg = W.config("C4")
    t0 = time.time()
    _, strains, _, _ = S._population(S.SEED_DEFAULT, cfg["sim"]["n_strains"], cfg["sim"]["genome_len"], 1, cfg["sim"]["mean_len"], 1000,
                                     40000, cfg["sim"]["snp_rate"], cfg["sim"]["strain_indel_rate"])
    short_fa, con_fa = str(tmp_path / "short.fa"), str(tmp_path / "long_con_polished.fa")
    n_short = W.make_short(cfg, strains, short_fa)
    contigs = [S.Read(f"longr_con_{k}_{a}", g[a:a + 40_000].copy(), None, k, a, a + 40_000, False)
               for k, g in enumerate(strains) for a in range(0, len(g) - 10_000, 40_000)]
    S.write_fasta(contigs, con_fa)
    t_sim = time.time() - t0
    assert n_short == 10_000_000 and len(contigs) == 5_000
    st_short = cfg["stage_short"]
    t0 = time.time()
    r = StageRunner(short_fa, con_fa, cfg["nsplit"], long_mode=False)
    try:
        t_open = time.time() - t0
        assert r.job.num_queries == 10_000_000 and r.job.num_chunks == 625
        out = str(tmp_path / "shortr1_r0.paf")
        t0 = time.time()
        r.prepare()
        t_sketch = time.time() - t0
        t0 = time.time()
        rows = r.run(out, share=(0, 8), **st_short)
        t_run = time.time() - t0
        st = api.last_stats()
    finally:
        r.close()
    os.remove(short_fa)
    print(f"C4 full size (short reads vs contigs): simulate {t_sim:.1f} s, parse + upload {t_open:.1f} s, sketch of 2.5 Gbases {t_sketch:.2f} s, "
          f"one rank's share (79 of 625 chunks) {t_run:.1f} s -> {rows} overlaps; anchors {st['anchors']:.3g}, candidate rows {st['ava_rows']:.3g}")
    assert st["queries"] == 10_000_000 and 620 < st["targets"] < 640 and st["anchors"] > 1e9
    assert rows == sum(1 for _ in open(out)) == st["rows_out"] and rows > 100_000
    check_rows(out, st_short["len_over"], st_short["iden"], 100_000)
    assert t_run < 240
