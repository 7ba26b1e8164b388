"""GPU: BASELINE.json configs[4] (C5) and configs[3] (C4, long reads) at their FULL size - 500 000 reads on 50 strains x 2 Mb
with --min_identity 0.90 --min_ovlp_len 1500, and 1 000 000 long reads on 100 strains x 2 Mb - each through the entry point
a rank of the multi-GPU job uses (hlmi_job_run).  The complete read set is resident in HBM and sketched (all 5 / 10 Gbases
are queries of every chunk); of the --nsplit target chunks the test runs as many as fit its time budget: one of C5's 60
(the reference's unit of work, utils.py:54: one worker per chunk; 1/7.5 of one rank's share of an 8-rank job), four of
C4's 1000 (1/31 of a rank's share).  The pair-once rule (strcmp(qname, tname) < 0) makes a chunk's work proportional to
the rank of its targets' names: chunk 40 of C5 (reads r333k..r341k) sits in the middle.  A whole C5 pass is ~60 of these,
i.e. ~2.5 minutes on one card.  The read sets are made by the block-parallel simulator
(hylight_amd/simulate.py:simulate_reads_to_fasta) in seconds."""
import os
import time

import pytest

from fullsize import check_rows
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


def test_c5_full_size_one_chunk(tmp_path):
    if _free_gb(tmp_path) < 12 or _host_gb() < 24:
        pytest.skip("needs 12 GB of scratch space and 24 GB of host memory")
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
        out = str(tmp_path / "chunk40.paf")
        t0 = time.time()
        r.prepare()
        t_sketch = time.time() - t0
        t0 = time.time()
        rows = r.run(out, share=(40, 60), **cfg["stage"])          # chunk c belongs to slice c % 60: exactly chunk 40
        t_run = time.time() - t0
        st = api.last_stats()
    finally:
        r.close()
    os.remove(fa)
    print(f"C5 full size: simulate {t_sim:.1f} s, parse + upload {t_open:.1f} s, sketch of 5 Gbases {t_sketch:.2f} s, one of 60 chunks "
          f"{t_run:.1f} s -> {rows} overlaps; anchors {st['anchors']:.3g}, candidate rows {st['ava_rows']:.3g}, "
          f"held-back end extensions {st['align_ext_held']:.3g} (run after all: {st['align_ext_late']:.3g})")
    assert st["queries"] == 500_000 and st["chunks_run"] == 1 and 8_000 < st["targets"] < 8_700
    assert st["minimizers_q"] > 1.2e9 and st["anchors"] > 2e9 and st["ava_rows"] > 3e7 and st["align_ext_held"] > st["ava_rows"]
    assert rows == sum(1 for _ in open(out)) == st["rows_out"]
    check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 20)
    assert t_run < 120


def test_c4_full_size_long_reads_four_chunks(tmp_path):
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
        rows = r.run(out, share=(5, 250), **cfg["stage"])          # chunks 5, 255, 505, 755
        t_run = time.time() - t0
        st = api.last_stats()
    finally:
        r.close()
    os.remove(fa)
    print(f"C4 full size (long reads): simulate {t_sim:.1f} s, parse + upload {t_open:.1f} s, sketch of 10 Gbases {t_sketch:.2f} s, four of "
          f"1000 chunks {t_run:.1f} s -> {rows} overlaps; anchors {st['anchors']:.3g}, candidate rows {st['ava_rows']:.3g}")
    assert st["queries"] == 1_000_000 and st["chunks_run"] == 4 and 3_900 < st["targets"] < 4_100
    assert st["anchors"] > 5e9 and st["minimizers_q"] > 2e9 and st["ava_rows"] > 1e7
    assert rows == sum(1 for _ in open(out)) == st["rows_out"]
    # (at 5 000x pooled depth the mc = 2 support filter leaves next to nothing of four chunks' 1.9e7 candidate rows)
    check_rows(out, cfg["stage"]["len_over"], cfg["stage"]["iden"], 0)
    assert t_run < 240


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
