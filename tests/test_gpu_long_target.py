"""GPU: targets longer than 2^24 bases.  The reads -> contigs calls of the run (script/HyLight.py:149,180) put contigs in
the target role of split_reads2; a bacterial-size contig is tens of megabases.  Round 2 refused targets of 2^24 bases
and more; target positions now take up to 29 bits.  The stage output must equal the oracle pipeline's."""
import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from oracle import ava as OA
from oracle import filters as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def contig_and_reads(tmp_path_factory):
    d = tmp_path_factory.mktemp("bigt")
    rng = np.random.default_rng(5)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=20_000_000)]
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    lines = []
    # reads from the far end of the contig (positions above 2^24 = 16.8 M), from its start, and across 2^24 itself
    starts = list(rng.integers(17_000_000, 19_980_000, size=70)) + list(rng.integers(0, 500_000, size=30)) + \
        list(rng.integers((1 << 24) - 9_000, (1 << 24) - 1_000, size=12))
    for i, a in enumerate(starts):
        seq = genome[a:a + int(rng.integers(7_000, 12_000))].copy()
        sub = rng.random(len(seq)) < 0.004
        seq[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(sub.sum()))]
        if i % 2:
            seq = comp[seq[::-1]]
        lines.append(f">r{i:03d}\n{seq.tobytes().decode()}\n")
    reads, contig = d / "s1.fa", d / "contigs1.fa"
    reads.write_text("".join(lines))
    contig.write_text(">utg000001l\n" + genome.tobytes().decode() + "\n")
    return d, reads, contig


def test_reads_against_a_20mb_contig(contig_and_reads, monkeypatch):
    d, reads, contig = contig_and_reads
    stage = dict(len_over=3000, mc=2, iden=0.95)                    # HyLight.py:149: min_ovlp_len, mc 2
    out = d / "ov_long_ref.paf"
    api.split_reads2(reads, contig, 60, d, out, long=True, **stage)
    OA.ava(contig, reads, d / "o.paf")
    raw = open(d / "o.paf").read().split("\n")[:-1]
    assert len(raw) >= 100 and max(int(l.split("\t")[8]) for l in raw) > (1 << 24)
    want = F.stage([raw], True, stage["len_over"], stage["mc"], stage["iden"])
    got = open(out).read().split("\n")[:-1]
    assert got == want and len(want) >= 90
    # the raw rows too (positions, CIGARs), in every anchor form
    for var in (None, "HLMI_ANCHOR_SPLIT", "HLMI_ANCHOR_PAIRS", "HLMI_CHAIN_UNPACKED"):
        if var:
            monkeypatch.setenv(var, "1")
        api.ava(contig, reads, d / "g.paf")
        if var:
            monkeypatch.delenv(var)
        assert open(d / "g.paf").read() == open(d / "o.paf").read(), var
