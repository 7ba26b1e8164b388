"""GPU parity of the overlap-graph stage (hlmi_miniasm: host restatement + HIP transitive reduction) and
of sfo2overlaps against goldens produced by the reference's own miniasm 0.3-r179 / sfo2overlaps.py, and
against oracle/_ref/miniasm (the compiled reference) on further inputs when it is present."""
import gzip
import os
import shutil
import subprocess

import pytest

from hylight_amd import api
from hylight_amd import simulate as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "miniasm")
FLAGS = {"n1c1": dict(n_rounds_arg=1, min_dp=1), "n3c3": dict(n_rounds_arg=3, min_dp=3)}   # HyLight.py:140 / :137


def _plain(golden, name, tmp_path):
    p = golden.path(name)
    if not p.endswith(".gz"):
        return p
    out = tmp_path / name
    with gzip.open(p, "rb") as f, open(out, "wb") as g:
        shutil.copyfileobj(f, g)
    return str(out)


@pytest.mark.parametrize("tag", ["n1c1", "n3c3"])
@pytest.mark.parametrize("fmt", ["ug", "sg", "paf", "bed"])
def test_miniasm_fixture_A(golden, tmp_path, tag, fmt):
    out = tmp_path / "o.txt"
    fa = _plain(golden, "fxA_reads.fa", tmp_path) if fmt == "ug" else None
    api.miniasm(golden.path("fxA_stage_nsplit4.paf"), fa, out, bub_dist=10000, max_ext=1, outfmt=fmt, **FLAGS[tag])
    ext = "gfa" if fmt == "ug" else fmt
    assert open(out).read() == golden.text(f"fxA_miniasm_{tag}.{ext}")


def test_line_starts_found_in_windows(golden, tmp_path, monkeypatch):
    """Files of 4 GiB and more are parsed with 64-bit line offsets found window by window (1 GiB each); HLMI_GRAPH_WINDOW_MB
    forces the same path through a small file: 1 MB windows over a ~20 MB PAF, lines straddling the window borders."""
    rows = open(_plain(golden, "fxD1_messy.paf", tmp_path)).read().split("\n")[:-1]
    big = tmp_path / "big.paf"
    with open(big, "w") as f:
        while f.tell() < 20 << 20:
            f.write("\n".join(rows) + "\n")
    outs = {}
    for mb in (None, "1", "3"):
        if mb:
            monkeypatch.setenv("HLMI_GRAPH_WINDOW_MB", mb)
        for fmt in ("ug", "paf"):
            out = tmp_path / f"o{mb}.{fmt}"
            api.miniasm(big, None, out, bub_dist=10000, max_ext=1, outfmt=fmt, **FLAGS["n1c1"])
            outs[(mb, fmt)] = open(out, "rb").read()
        assert api.last_stats()["graph_parse_windows"] == (1 if mb is None else -(-os.path.getsize(big) // (int(mb) << 20)))
    for fmt in ("ug", "paf"):
        assert outs[(None, fmt)] == outs[("1", fmt)] == outs[("3", fmt)] and len(outs[(None, fmt)]) > 1000


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("tag", ["n1c1", "n3c3"])
def test_miniasm_messy_graphs(golden, tmp_path, seed, tag):
    # fixture D: tips, bubbles, bi-loops, internal sequences, short overlaps, asymmetric arcs
    paf = _plain(golden, f"fxD{seed}_messy.paf", tmp_path)
    fa = _plain(golden, "fxD1_reads.fa", tmp_path) if seed == 1 else None
    for fmt, ext in (("ug", "gfa"), ("sg", "sg")):
        out = tmp_path / f"o.{ext}"
        api.miniasm(paf, fa if fmt == "ug" else None, out, bub_dist=10000, max_ext=1, outfmt=fmt, **FLAGS[tag])
        assert open(out).read() == golden.text(f"fxD{seed}_miniasm_{tag}.{ext}")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/miniasm not built")
@pytest.mark.parametrize("seed,n_reads,genome,fake", [(11, 400, 120_000, 60), (12, 400, 120_000, 60), (14, 250, 40_000, 120)])
def test_miniasm_against_compiled_reference(tmp_path, seed, n_reads, genome, fake):
    reads, paf = S.messy_graph_paf(seed, n_reads=n_reads, genome=genome, fake=fake)
    p = tmp_path / "m.paf"
    p.write_text("\n".join(paf) + "\n")
    fa = tmp_path / "m.fa"
    S.write_fasta(reads, fa)
    for flags, kw in (("-n 1 -e 1 -c 1", FLAGS["n1c1"]), ("-n 3 -e 1 -c 3", FLAGS["n3c3"])):
        want = subprocess.run(f"{REF} -d 10000 {flags} -f {fa} {p}", shell=True, check=True, capture_output=True).stdout.decode()
        out = tmp_path / "o.gfa"
        api.miniasm(p, fa, out, bub_dist=10000, max_ext=1, **kw)
        assert open(out).read() == want and want.count("\nS\t") + want.startswith("S\t") >= 1


def test_miniasm_reads_in_fastq_with_awkward_lines(golden, tmp_path):
    """The unitig sequences come from a scan of the read file that copies only the reads on unitigs; the scan follows
    kseq's record rules: wrapped sequence lines, CRLF, and FASTQ quality lines that begin with '@' or '>' must not be
    taken for headers.  Same reads as fixture A in such a file -> the same GFA."""
    fa = _plain(golden, "fxA_reads.fa", tmp_path)
    recs = []
    name = None
    for line in open(fa):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:]
        else:
            recs.append((name, line))
    fq = tmp_path / "reads.fq"
    with open(fq, "w", newline="") as f:
        for i, (n, seq) in enumerate(recs):
            qual = ("@" if i % 3 == 0 else ">" if i % 3 == 1 else "I") + "I" * (len(seq) - 1)
            if i % 2:                                  # wrapped sequence and quality, CRLF
                w = 61
                sl = "\r\n".join(seq[k:k + w] for k in range(0, len(seq), w))
                ql = "\r\n".join(qual[k:k + w] for k in range(0, len(qual), w))
                f.write(f"@{n} some comment\r\n{sl}\r\n+\r\n{ql}\r\n")
            else:
                f.write(f"@{n}\n{seq}\n+{n}\n{qual}\n")
    out = tmp_path / "o.gfa"
    api.miniasm(golden.path("fxA_stage_nsplit4.paf"), fq, out, bub_dist=10000, max_ext=1, outfmt="ug", **FLAGS["n1c1"])
    assert open(out).read() == golden.text("fxA_miniasm_n1c1.gfa")


def test_miniasm_empty_paf_gives_empty_gfa(tmp_path):
    p = tmp_path / "e.paf"
    p.write_text("")
    out = tmp_path / "o.gfa"
    api.miniasm(p, None, out)
    assert open(out).read() == ""      # HyLight.py:173 tests for exactly this


@pytest.mark.parametrize("tag", ["fxC_v3", "fxC_v3_oh30"])
def test_sfo2overlaps(golden, tmp_path, tag):
    out = tmp_path / "o.savage"
    api.sfo2overlaps(_plain(golden, tag + ".sfo", tmp_path), out, num_singles=90)
    assert open(out).read() == golden.text(tag + ".savage")
