"""GPU: the stage's final rows (14 columns, filter_overlap_slr2.py:138-151) formatted by the device kernels of row_text.hip
are byte for byte the rows the host formatter writes - "%.4f" of the three scores, the float(score2) < iden test on the printed
decimals, the sort order that follows from column 12 - in long and in short mode.  The stage formats on the device when a filter
group keeps 65 536 rows or more (HLMI_TEXT_GPU: always, HLMI_TEXT_HOST: never)."""
import pytest

from hylight_amd import api
from hylight_amd import simulate as S

pytestmark = pytest.mark.gpu


def _stage(tmp_path, fa, ref, out, monkeypatch, env, **kw):
    for k in ("HLMI_TEXT_GPU", "HLMI_TEXT_HOST"):
        monkeypatch.delenv(k, raising=False)
    if env:
        monkeypatch.setenv(env, "1")
    api.split_reads2(fa, ref, 4, tmp_path, tmp_path / out, threads=4, **kw)
    return open(tmp_path / out, "rb").read(), dict(api.last_stats())


@pytest.mark.parametrize("iden", [0.95, 0.995])
def test_long_mode_rows_from_the_device_formatter(tmp_path, monkeypatch, iden):
    reads, _ = S.simulate_reads(seed=61, n_reads=90, n_strains=3, genome_len=24000, mean_len=6000, min_len=2500, max_len=12000)
    fa = tmp_path / "s1.fa"
    S.write_fasta(reads, fa)
    kw = dict(len_over=1000, mc=2, iden=iden, long=True)
    g, sg = _stage(tmp_path, fa, fa, "gpu.paf", monkeypatch, "HLMI_TEXT_GPU", **kw)
    h, sh = _stage(tmp_path, fa, fa, "host.paf", monkeypatch, "HLMI_TEXT_HOST", **kw)
    assert g == h
    assert g.count(b"\n") > (100 if iden < 0.99 else 5)
    assert sg["rows_text_on_gpu"] > 0 and sg["rows_text_left_to_host"] == 0 and sh.get("rows_text_on_gpu", 0) == 0
    assert all(line.endswith(b"\t") and line.count(b"\t") == 14 for line in g.split(b"\n")[:-1])


def test_short_mode_rows_from_the_device_formatter(tmp_path, monkeypatch):
    """Short reads against contig pieces (script/HyLight.py:200: -len 70 -mc 3, short mode): tens of thousands of rows."""
    rng_reads, _ = S.simulate_reads(seed=5, n_reads=12, n_strains=2, genome_len=30000, mean_len=9000, min_len=8000, max_len=10000)
    con = tmp_path / "con.fa"
    S.write_fasta(rng_reads, con)
    import numpy as np
    rng = np.random.default_rng(11)
    short = []
    for i in range(6000):
        r = rng_reads[int(rng.integers(0, len(rng_reads)))]
        a = int(rng.integers(0, len(r.seq) - 250))
        seq = r.seq[a:a + 250].copy()
        if i & 1:
            seq = S.revcomp(seq)
        k = rng.integers(0, 250, size=1)
        seq[k] = S._BASES[rng.integers(0, 4, size=1)]
        short.append(S.Read(f"sr{i:06d}", seq, None, 0, a, a + 250, bool(i & 1)))
    sfa = tmp_path / "short.fa"
    S.write_fasta(short, sfa)
    kw = dict(len_over=70, mc=3, iden=0.95, long=False)
    g, sg = _stage(tmp_path, sfa, con, "gpu.paf", monkeypatch, "HLMI_TEXT_GPU", **kw)
    h, _ = _stage(tmp_path, sfa, con, "host.paf", monkeypatch, "HLMI_TEXT_HOST", **kw)
    assert g == h and g.count(b"\n") > 3000
    assert sg["rows_text_on_gpu"] >= g.count(b"\n")
