"""GPU parity of the short-read mode: the overlapper with the constants of
script/filter_overlap_slr2.py:55 and the short-mode filter chain (prpare_mutation, `fkey in mutation` rule),
against the CPU oracle.  Bit-exact."""
import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from oracle import ava as OA
from oracle import filters as F

pytestmark = pytest.mark.gpu


def _short_reads_and_contigs(seed, n_reads=1500, genome_len=30000):
    """Two strains; 'contigs' = a few long pieces of strain 0, queries = 250 bp reads of both strains."""
    reads, strains = S.simulate_reads(seed=seed, n_strains=2, genome_len=genome_len, n_reads=n_reads, mean_len=250,
                                      min_len=180, max_len=300, err_sub=0.002, err_ins=0.0005, err_del=0.0005,
                                      name_prefix="s")
    g = strains[0]
    cuts = [0, 9000, 17000, genome_len]
    contigs = [S.Read(f"longr_con_{i}", g[a:b].copy(), None, 0, a, b, False) for i, (a, b) in enumerate(zip(cuts, cuts[1:]))]
    return reads, contigs


def test_short_mode_opts_are_the_reference_constants():
    o = api.ava_opts_short()
    assert (o.k, o.w, o.hpc, o.min_chain_score, o.min_cnt, o.match, o.mismatch, o.min_dp_score, o.end_bonus) == \
           (21, 11, 0, 30, 2, 4, 2, 60, 100)          # -k21 -w11 (no -H) -m30 -n2 -A4 -B2 -s60 --end-bonus=100
    assert o.pair_once == 0                           # --sr -DP has no -X: both directions are reported
    oo = OA.opts_short()
    for f, _ in api.AvaOpts._fields_:
        assert getattr(o, f) == getattr(oo, f), f


def test_short_reads_vs_contigs_matches_oracle(tmp_path):
    reads, contigs = _short_reads_and_contigs(91)
    q, t = tmp_path / "short.fa", tmp_path / "con.fa"
    S.write_fasta(reads, q)
    S.write_fasta(contigs, t)
    api.ava(t, q, tmp_path / "g.paf", api.ava_opts_short())
    OA.ava(t, q, tmp_path / "o.paf", OA.opts_short())
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    rows = [l.split("\t") for l in want.splitlines()]
    assert len(rows) > 0.6 * len(reads)               # most reads map (strain 1 reads carry ~1 % SNPs)
    full = sum(1 for c in rows if int(c[2]) == 0 and int(c[3]) == int(c[1]))
    assert full > 0.8 * len(rows)                     # end bonus: alignments reach both read ends
    assert got == want


def test_small_groups_chain_the_same_in_both_kernels(tmp_path, monkeypatch):
    """Short reads on contigs leave groups of a dozen anchors: they go through chain_small_kernel (a lane per group).  With
    HLMI_CHAIN_NO_SMALL every group takes chain_kernel (a wave per group) instead: same rows, and the statistic shows which
    kernel did the work."""
    reads, contigs = _short_reads_and_contigs(93)
    q, t = tmp_path / "short.fa", tmp_path / "con.fa"
    S.write_fasta(reads, q)
    S.write_fasta(contigs, t)
    api.ava(t, q, tmp_path / "a.paf", api.ava_opts_short())
    st = api.last_stats()
    assert st["chain_groups_small"] > 0.5 * st["chain_groups"] and st["anchors_small_groups"] > 0
    monkeypatch.setenv("HLMI_CHAIN_NO_SMALL", "1")
    api.ava(t, q, tmp_path / "b.paf", api.ava_opts_short())
    assert api.last_stats()["chain_groups_small"] == 0
    a = open(tmp_path / "a.paf").read()
    assert a == open(tmp_path / "b.paf").read() and a.count("\n") > 0.6 * len(reads)


def test_small_pieces_assemble_the_same_in_both_kernels(tmp_path, monkeypatch):
    """The pieces of short reads have half a dozen alignment tasks: the piece kernels (task references, LONG flags, row
    assembly) run a lane per piece there.  HLMI_ASM_WAVE keeps the wave-per-piece row assembly: same rows."""
    reads, contigs = _short_reads_and_contigs(95)
    q, t = tmp_path / "short.fa", tmp_path / "con.fa"
    S.write_fasta(reads, q)
    S.write_fasta(contigs, t)
    api.ava(t, q, tmp_path / "a.paf", api.ava_opts_short())
    monkeypatch.setenv("HLMI_ASM_WAVE", "1")
    api.ava(t, q, tmp_path / "b.paf", api.ava_opts_short())
    a = open(tmp_path / "a.paf").read()
    assert a == open(tmp_path / "b.paf").read() and a.count("\n") > 0.6 * len(reads)


def test_short_noisy_reads_vs_contigs_matches_oracle(tmp_path):
    """2 % / 1 % / 1 % read errors: the end bonus now decides between clipped and full-length alignments, extensions
    rarely match exactly (third certificate with the bonus row in play) and go through the 64-diagonal DP."""
    reads, strains = S.simulate_reads(seed=94, n_strains=2, genome_len=20000, n_reads=900, mean_len=250, min_len=180,
                                      max_len=300, err_sub=0.02, err_ins=0.01, err_del=0.01, name_prefix="s")
    g = strains[0]
    contigs = [S.Read(f"longr_con_{i}", g[a:b].copy(), None, 0, a, b, False) for i, (a, b) in enumerate([(0, 9000), (9000, 20000)])]
    q, t = tmp_path / "short.fa", tmp_path / "con.fa"
    S.write_fasta(reads, q)
    S.write_fasta(contigs, t)
    api.ava(t, q, tmp_path / "g.paf", api.ava_opts_short())
    OA.ava(t, q, tmp_path / "o.paf", OA.opts_short())
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 0.5 * len(reads)
    assert got == want


def test_short_vs_short_reports_both_directions(tmp_path):
    reads, _ = _short_reads_and_contigs(92, n_reads=400, genome_len=4000)
    fa = tmp_path / "s.fa"
    S.write_fasta(reads, fa)
    api.ava(fa, fa, tmp_path / "g.paf", api.ava_opts_short())
    OA.ava(fa, fa, tmp_path / "o.paf", OA.opts_short())
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    assert got == want
    pairs = {(c[0], c[5]) for c in (l.split("\t") for l in want.splitlines())}
    assert any((b, a) in pairs for a, b in pairs) and all(a != b for a, b in pairs)


def test_short_stage_matches_oracle_pipeline(tmp_path):
    """split_reads2(short_reads, contigs, long=False) as HyLight.py:200 calls it (len_over 70, mc 3)."""
    reads, contigs = _short_reads_and_contigs(93)
    q, t = tmp_path / "short.fa", tmp_path / "con.fa"
    S.write_fasta(reads, q)
    S.write_fasta(contigs, t)
    out = tmp_path / "shortr1.paf"
    api.split_reads2(q, t, 2, tmp_path, out, threads=4, len_over=70, mc=3, iden=0.95, long=False)
    lines = open(t).read().split("\n")[:-1]
    chunks = []
    for i, (lo, hi) in enumerate(F.chunk_ranges(len(lines), 2)):
        cf = tmp_path / f"c{i}.fa"
        cf.write_text("\n".join(lines[lo:hi]) + "\n")
        OA.ava(cf, q, str(cf) + ".paf", OA.opts_short())
        chunks.append(open(str(cf) + ".paf").read().split("\n")[:-1])
    want = F.stage(chunks, False, 70, 3, 0.95)
    got = open(out).read().split("\n")[:-1]
    assert len(want) > 200
    assert got == want
