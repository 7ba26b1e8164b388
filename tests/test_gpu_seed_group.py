"""GPU: anchors grouped by seed_count_kernel / seed_place_kernel (a workgroup per query piece, table of its partners in LDS,
merge by target from LDS-staged runs) give the rows of the sort path (seed fill + radix sort + head selection) and of the
oracle, bit for bit.

HLMI_SEED_GROUP selects the grouping kernels (measured equal to the sort path on C3: off by default).  Reference call:
filter_overlap_slr2.py:51 (minimap2 -x ava-pb)."""
import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from oracle import ava as OA

pytestmark = pytest.mark.gpu


def _sim(seed, n, **kw):
    args = dict(n_strains=2, genome_len=20000, mean_len=5000, min_len=2000, max_len=9000)
    args.update(kw)
    reads, _ = S.simulate_reads(seed=seed, n_reads=n, **args)
    return reads


def _run(tmp_path, fa_t, fa_q, name, monkeypatch, env):
    monkeypatch.delenv("HLMI_SEED_GROUP", raising=False)
    if env:
        monkeypatch.setenv(env, "1")
    api.ava(fa_t, fa_q, tmp_path / name)
    st = dict(api.last_stats())
    return open(tmp_path / name).read(), st


@pytest.mark.parametrize("seed,n,kw", [
    (31, 40, {}),
    (32, 60, dict(n_strains=3, genome_len=15000, err_sub=0.01, err_ins=0.004, err_del=0.004)),
    (33, 30, dict(n_strains=1, genome_len=8000, mean_len=3000, min_len=500)),
])
def test_grouped_equals_sorted_equals_oracle(tmp_path, monkeypatch, seed, n, kw):
    reads = _sim(seed, n, **kw)
    fa = tmp_path / "r.fa"
    S.write_fasta(reads, fa)
    OA.ava(fa, fa, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 50
    got_g, st_g = _run(tmp_path, fa, fa, "g.paf", monkeypatch, "HLMI_SEED_GROUP")
    got_s, st_s = _run(tmp_path, fa, fa, "s.paf", monkeypatch, None)
    assert got_g == want
    assert got_s == want
    assert st_g["anchors_grouped_in_lds"] == st_g["anchors"] > 0
    assert st_s.get("anchors_grouped_in_lds", 0) == 0
    assert st_g["chain_groups"] == st_s["chain_groups"]          # groups of any size, counted by both forms
    assert st_g["pieces"] == st_s["pieces"] and st_g["fixed_points"] == st_s["fixed_points"]


def test_deep_batch_with_several_pieces_per_query(tmp_path, monkeypatch):
    """60 x depth of 12-20 kb reads: ~200 000 anchors per query = several pieces each (ranges of the query's minimizers of
    ~96 k anchors: seed_scan_kernel puts their tables together); the sort path and the oracle agree."""
    reads = _sim(41, 110, n_strains=2, genome_len=30000, mean_len=16000, min_len=12000, max_len=20000)
    fa = tmp_path / "r.fa"
    S.write_fasta(reads, fa)
    got_d, st_d = _run(tmp_path, fa, fa, "d.paf", monkeypatch, "HLMI_SEED_GROUP")
    got_s, st_s = _run(tmp_path, fa, fa, "s.paf", monkeypatch, None)
    OA.ava(fa, fa, tmp_path / "o.paf")
    assert st_d["anchors_grouped_in_lds"] == st_d["anchors"] > 0
    assert st_d["seed_group_pieces"] >= 110 + 20                 # queries of more than 96 k anchors have two pieces and more
    assert st_s.get("anchors_grouped_in_lds", 0) == 0
    assert got_d == got_s == open(tmp_path / "o.paf").read()
    assert len(got_d.splitlines()) > 2000


def test_more_partners_than_the_table_holds(tmp_path, monkeypatch):
    """One long query against 3 000 short targets cut from it on both strands: more partners than the LDS table of a piece
    takes (640 targets; the query's ~250 000 anchors make three pieces) - the kernel says so and the batch goes through the
    sort path; rows as the oracle gives them."""
    rng = np.random.default_rng(7)
    g = S._BASES[rng.integers(0, 4, size=36000)]
    targets = []
    for i in range(3000):
        L = int(rng.integers(250, 400))
        s = int(rng.integers(0, len(g) - L))
        seq = g[s:s + L].copy()
        if i & 1:
            seq = S.revcomp(seq)
        targets.append(S.Read(f"t{i:05d}", seq, None, 0, s, s + L, bool(i & 1)))
    query = [S.Read("a_query", g.copy(), None, 0, 0, len(g), False)]
    fa_t, fa_q = tmp_path / "t.fa", tmp_path / "q.fa"
    S.write_fasta(targets, fa_t)
    S.write_fasta(query, fa_q)
    got_g, st_g = _run(tmp_path, fa_t, fa_q, "g.paf", monkeypatch, "HLMI_SEED_GROUP")
    got_s, st_s = _run(tmp_path, fa_t, fa_q, "s.paf", monkeypatch, None)
    OA.ava(fa_t, fa_q, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 2400
    assert st_g["chain_groups"] > 2048
    assert st_g["seed_group_gave_up"] == 1 and st_g.get("anchors_grouped_in_lds", 0) == 0
    assert got_g == want
    assert got_s == want


def test_short_mode_keeps_the_sort_path(tmp_path, monkeypatch):
    """Both directions of a pair (--sr -DP, filter_overlap_slr2.py:55): a key's run holds the read's own entries, the
    grouping kernel is not used even when asked for."""
    reads = _sim(35, 70, n_strains=3, genome_len=15000, err_sub=0.01, err_ins=0.005, err_del=0.005)
    fa = tmp_path / "r.fa"
    S.write_fasta(reads, fa)
    monkeypatch.setenv("HLMI_SEED_GROUP", "1")
    api.ava(fa, fa, tmp_path / "g.paf", api.ava_opts_short())
    assert api.last_stats().get("anchors_grouped_in_lds", 0) == 0
    OA.ava(fa, fa, tmp_path / "o.paf", OA.opts_short())
    assert open(tmp_path / "g.paf").read() == open(tmp_path / "o.paf").read()
