"""GPU parity: the HIP filter chain (through the C ABI) against the reference-generated golden
vectors and against the CPU oracle on the same inputs.  Bit-exact (text equality)."""
import gzip
import os
import shutil

import pytest

from hylight_amd import api
from oracle import filters as F

pytestmark = pytest.mark.gpu


def _plain(golden, name, tmp_path):
    """Golden file as an uncompressed path."""
    p = golden.path(name)
    if not p.endswith(".gz"):
        return p
    out = tmp_path / name
    with gzip.open(p, "rb") as f, open(out, "wb") as g:
        shutil.copyfileobj(f, g)
    return str(out)


def _lines(path):
    with open(path) as f:
        return f.read().split("\n")[:-1]


@pytest.mark.parametrize("src,kw,gold", [
    ("fxA_ava.paf", dict(min_len=30, min_o=3), "fxA_v4.paf"),
    ("fxB_dense.paf", dict(min_len=30, min_o=3), "fxB_v4.paf"),
    ("fxB_dense.paf", dict(min_len=100, min_iden=0.9, min_o=40), "fxB_v4_len100_oh40.paf"),
])
def test_v4_window_filter(golden, tmp_path, src, kw, gold):
    out = tmp_path / "out.paf"
    api.paf_window_filter(4, _plain(golden, src, tmp_path), out, **kw)
    assert _lines(out) == golden.lines(gold)


@pytest.mark.parametrize("kw,gold", [
    (dict(min_len=90, min_iden=0.99, min_o=2, sfo=True), "fxC_v3.sfo"),
    (dict(min_len=90, min_iden=0.9, min_o=30, sfo=True), "fxC_v3_oh30.sfo"),
    (dict(min_len=90, min_iden=0.9, min_o=30, sfo=False), "fxC_v3_oh30.score"),
])
def test_v3_window_filter(golden, tmp_path, kw, gold):
    out = tmp_path / "out.txt"
    api.paf_window_filter(3, _plain(golden, "fxC_contigs.paf", tmp_path), out, **kw)
    assert _lines(out) == golden.lines(gold)


@pytest.mark.parametrize("tag,kw", [
    ("long_len1000", dict(len_over=1000, mc=2, iden=0.95, long_mode=True)),
    ("long_len3000_iden99", dict(len_over=3000, mc=2, iden=0.99, long_mode=True)),
    ("short_len70", dict(len_over=70, mc=3, iden=0.95, long_mode=False)),
])
def test_filter_chunk_against_reference_worker(golden, tmp_path, tag, kw):
    out = tmp_path / "o4.paf"
    api.filter_chunk(_plain(golden, "fxA_ava.paf", tmp_path), out, **kw)
    assert _lines(out) == golden.lines(f"fxA_worker_{tag}.paf")
    st = api.last_stats()
    assert st["rows_out"] == len(golden.lines(f"fxA_worker_{tag}.paf"))


def test_filter_chunk_against_oracle_on_dense_bidirectional_input(golden, tmp_path):
    # fixture B carries both directions of every pair and duplicate rows; no CIGARs (last field is mapq)
    src = _plain(golden, "fxB_dense.paf", tmp_path)
    out = tmp_path / "o4.paf"
    api.filter_chunk(src, out, len_over=500, mc=2, iden=0.9, long_mode=True)
    want = F.worker(golden.lines("fxB_dense.paf"), True, 500, 2, 0.9)
    assert _lines(out) == want and len(want) > 100


@pytest.mark.parametrize("tag,args", [("len90_oh30", (90, 0.9, 30, 0.8)), ("len90_oh1", (90, 0.98, 1, 0.8))])
def test_filter_ovlp_inline_then_minimap22sfo(golden, tmp_path, tag, args):
    # SURVEY 8f rank 2, chained as polyte.tune_params.py:507-515 does
    paf, sfo = tmp_path / "o.paf", tmp_path / "o.sfo"
    api.filter_ovlp_inline(_plain(golden, "fxC_contigs.paf", tmp_path), paf, *args)
    assert _lines(paf) == golden.lines(f"fxC_inline_{tag}.paf")
    api.minimap22sfo(paf, sfo, 0, 0)
    assert _lines(sfo) == golden.lines(f"fxC_inline_{tag}.sfo")


def test_minimap22sfo_thresholds(golden, tmp_path):
    sfo = tmp_path / "o.sfo"
    api.minimap22sfo(_plain(golden, "fxC_contigs.paf", tmp_path), sfo, 200, 99)
    assert _lines(sfo) == golden.lines("fxC_m22sfo_m200_p99.sfo")


def test_empty_input(tmp_path):
    src = tmp_path / "empty.paf"
    src.write_text("")
    out = tmp_path / "o.paf"
    api.filter_chunk(src, out, len_over=1000, mc=2, iden=0.95)
    assert out.read_text() == ""
    api.paf_window_filter(4, src, out, min_len=30, min_o=3)
    assert out.read_text() == ""


def test_malformed_row_is_an_error(tmp_path):
    src = tmp_path / "bad.paf"
    src.write_text("a\t10\t0\t10\t+\tb\n")
    with pytest.raises(api.HlmiError):
        api.filter_chunk(src, tmp_path / "o.paf", len_over=1000, mc=2, iden=0.95)
