"""the native text passes (filter_non_atcg, gfa2fa, pick_up) against goldens produced by the
reference's own functions (tests/golden/make_goldens_text.py).  Host-only code: runs without a GPU."""
import os

import pytest

from hylight_amd import api

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bytes(p):
    with open(p, "rb") as f:
        return f.read()


@pytest.mark.parametrize("model,src", [("fastq", "fxE_reads.fq"), ("fasta", "fxE_reads.fa")])
def test_filter_non_atcg(tmp_path, model, src):
    out = tmp_path / "s1.fa"
    api.filter_non_atcg(os.path.join(G, src), out, model)
    assert _bytes(out) == _bytes(os.path.join(G, f"fxE_non_atcg_{model}.fa"))


def test_gfa2fa(tmp_path):
    out = tmp_path / "c.fa"
    api.gfa2fa(os.path.join(G, "fxE_graph.gfa"), out)
    assert _bytes(out) == _bytes(os.path.join(G, "fxE_gfa2fa.fa"))


def test_gfa2fa_empty_line_is_an_error(tmp_path):
    bad = tmp_path / "bad.gfa"
    bad.write_text("S\ta\tACGT\n\nS\tb\tAC\n")
    with pytest.raises(api.HlmiError):          # the reference raises IndexError on fields[0]
        api.gfa2fa(bad, tmp_path / "o.fa")


@pytest.mark.parametrize("src,tag,mode", [("fxE_reads.fq", "fq", "fastq"), ("fxE_reads.fa", "fa", "fasta")])
def test_pick_up(tmp_path, src, tag, mode):
    out = tmp_path / "remain.fq"
    out.write_text("stale content that must disappear")
    api.pick_up(os.path.join(G, "fxE_ovlp.paf"), os.path.join(G, src), out, mode)
    assert _bytes(out) == _bytes(os.path.join(G, f"fxE_pick_up.{tag}"))


def test_pick_up_creates_no_file_when_every_read_overlaps(tmp_path):
    fq = tmp_path / "r.fq"
    fq.write_text("@a/1\nAC\n+\nII\n@b\nGT\n+\nII\n")
    paf = tmp_path / "o.paf"
    paf.write_text("a\t2\t0\t2\t+\tb\t2\t0\t2\t2\t2\t0\n")
    out = tmp_path / "remain.fq"
    api.pick_up(paf, fq, out, "fastq")
    assert not out.exists()


def test_missing_input_is_an_error(tmp_path):
    with pytest.raises(api.HlmiError):
        api.filter_non_atcg(tmp_path / "nope.fq", tmp_path / "o.fa", "fastq")
