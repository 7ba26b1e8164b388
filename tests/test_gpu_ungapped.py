"""`-r 0` of HyLight.extend_con's contig-vs-contig call (`minimap2 --sr -X -c ... -r 0`, script/HyLight.py:309): minimap2
derives its DP band from -r (1.5 bw + 1 = 1 diagonal), so the call aligns WITHOUT gaps.  hlmi_ava_opts::bandwidth == 0
therefore means: chains without a diagonal shift, blocks and extensions along the diagonal (align_ungapped_kernel;
oracle/ava_oracle.c: W = 1).  Two contigs whose overlap holds one deleted base must not come out as one gapped row that
passes the 0.99 identity filter into sfoverlaps.out."""
import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from oracle import ava as OA
from oracle import filters as F

pytestmark = pytest.mark.gpu


def _contigs_fastq(path, rng):
    g = S._BASES[rng.integers(0, 4, size=40_000)]
    a = g[0:12_000].copy()                                   # overlap of a and b: g[6000:12000]
    b = np.delete(g[6_000:20_000], 3_000)                    # ... with base g[9000] missing in b
    c = g[15_000:26_000].copy()                              # b and c overlap cleanly (g[15000:20000])
    for seq in (a, b, c):                                    # a few substitutions (polished contigs of two strains)
        pos = rng.choice(len(seq), size=len(seq) // 400, replace=False)
        seq[pos] = S._BASES[(np.searchsorted(S._BASES, seq[pos]) + 1) % 4]
    d = S.revcomp(g[24_000:33_000].copy())                   # c and d overlap on opposite strands
    e = g[30_000:40_000].copy()
    e[2_000] = ord("N")                                      # an ambiguous base inside the d / e overlap
    with open(path, "w") as f:
        for k, seq in enumerate((a, b, c, d, e)):
            f.write(f"@{k + 1}\n{seq.tobytes().decode()}\n+\n{'=' * len(seq)}\n")


def test_bandwidth_zero_aligns_without_gaps(tmp_path):
    fq = tmp_path / "contigs_b.fastq"
    _contigs_fastq(fq, np.random.default_rng(5))
    og, oo = api.ava_opts_short(), OA.opts_short()
    for o in (og, oo):
        o.pair_once, o.bandwidth = 1, 0
    api.ava(fq, fq, tmp_path / "g0.paf", og)
    st = api.last_stats()
    OA.ava(fq, fq, tmp_path / "o0.paf", oo)
    got, want = open(tmp_path / "g0.paf").read(), open(tmp_path / "o0.paf").read()
    assert got == want
    rows = want.split("\n")[:-1]
    assert len(rows) >= 4 and st.get("kernel_launches.align_ungapped", 0) >= 1
    for r in rows:
        cg = r.split("\t")[-1]
        assert "I" not in cg[5:] and "D" not in cg[5:], r[:200]
    # the pair with the deleted base: fragments on two diagonals, none of them a dovetail - nothing reaches sfoverlaps.out
    sfo = F.window_filter(rows, variant=3, min_len=90, min_iden=0.99, min_o=2, sfo=True)
    pairs = {tuple(l.split("\t")[:2]) for l in sfo}
    assert ("1", "2") not in pairs and ("2", "3") in pairs and ("3", "4") in pairs and ("4", "5") in pairs
    # with the short-read calls' own bandwidth (50) the same contigs give ONE gapped row for that pair, and it passes the filter
    og.bandwidth = oo.bandwidth = 50
    api.ava(fq, fq, tmp_path / "g50.paf", og)
    OA.ava(fq, fq, tmp_path / "o50.paf", oo)
    got50, want50 = open(tmp_path / "g50.paf").read(), open(tmp_path / "o50.paf").read()
    assert got50 == want50
    rows50 = want50.split("\n")[:-1]
    ab = [r for r in rows50 if r.startswith("1\t") and r.split("\t")[5] == "2"]
    assert len(ab) == 1 and ("D" in ab[0].split("\t")[-1] or "I" in ab[0].split("\t")[-1])
    sfo50 = F.window_filter(rows50, variant=3, min_len=90, min_iden=0.99, min_o=2, sfo=True)
    assert ("1", "2") in {tuple(l.split("\t")[:2]) for l in sfo50}
