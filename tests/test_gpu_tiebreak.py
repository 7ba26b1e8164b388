"""GPU: the last-resort key of the worker's intermediate sort (`sort -nk7 -k8 -k9 -k5`,
script/filter_overlap_slr2.py:57) on rows the device overlapper emitted itself.

Two rows of ONE read pair with equal tlen / tstart / tend are ordered by GNU sort's whole-line byte comparison; the
"first row per pair" rules (slr2:321-326 for the pile-up, slr2:133-136 for pass 2) then pick the row that comes first
in that TEXT order.  The fixture builds such a pair on purpose: the target read S is contained twice in the query
(lead + S + S with |lead| = 9000), so the overlapper reports q[9000:15000] and q[15000:21000] against t[0:6000] -
numerically 9000 < 15000, but as text "15000" sorts before "9000".  The stage on the GPU (fields only, no text on the
device) must keep the row the reference would keep."""
import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from oracle import ava as OA
from oracle import filters as F

pytestmark = pytest.mark.gpu


def _straddling_fixture(seed, n_small=999, lead=9000, unit=6000):
    """a_query = lead + S + S, c_target = S: two rows with columns 6-9 equal.  The v4 window filter drops the second
    row of a pair inside one 1000-row window (filter_trans_ovlp_inline_v4.py:68-72), so the two rows only both reach
    the sort when a window boundary falls between them: 999 short reads b_0000.. (a 200-base piece of the query's lead
    each, plus a 50-base tail that makes v4 reject the row as an internal match before its per-query counter) put
    exactly 999 rows of a_query ahead of the pair."""
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: bases[rng.integers(0, 4, size=n)]
    u = rnd(3 * n_small + 200)
    s = rnd(unit)
    reads = [S.Read("a_query", np.concatenate([u, rnd(lead - len(u)), s, s]), None, 0, 0, 0, False)]
    for i in range(n_small):
        reads.append(S.Read(f"b_{i:04d}", np.concatenate([u[3 * i:3 * i + 200], rnd(50)]), None, 0, 0, 0, False))
    reads.append(S.Read("c_target", s.copy(), None, 0, 0, 0, False))
    return reads


def _pair_rows(lines):
    return [l.split("\t") for l in lines if l.startswith("a_query\t") and "\tc_target\t" in l]


@pytest.mark.parametrize("seed,lead", [(3, 9000), (4, 9990), (5, 4000)])
def test_equal_target_interval_rows_follow_text_order(tmp_path, seed, lead):
    reads = _straddling_fixture(seed, lead=lead)
    fa = tmp_path / "s1.fa"
    S.write_fasta(reads, fa)
    raw_g, raw_o = tmp_path / "raw.paf", tmp_path / "raw_o.paf"
    api.ava(fa, fa, raw_g)
    OA.ava(fa, fa, raw_o)
    assert open(raw_g).read() == open(raw_o).read()
    raw = open(raw_o).read().split("\n")[:-1]
    at = [i for i, l in enumerate(raw) if l.startswith("a_query\t") and "\tc_target\t" in l]
    assert at == [999, 1000]                                            # the window boundary falls between the two rows
    rows = _pair_rows(raw)
    assert rows[0][5:9] == rows[1][5:9] == ["c_target", "6000", "0", "6000"]
    assert [int(r[2]) for r in rows] == [lead, lead + 6000]             # stream order = numeric order of qstart
    assert len(_pair_rows(F.window_filter(raw, 4, 30, None, 3))) == 2   # both reach the sort
    first_by_text = F.sort_intermediate(["\t".join(r) for r in rows])[0].split("\t")[2]
    assert (first_by_text == str(lead + 6000)) == (str(lead + 6000) < str(lead))
    # whole stage on the GPU: the pair keeps the row the TEXT order puts first
    out = tmp_path / "s1_s1.paf"
    api.split_reads2(fa, fa, 1, tmp_path, out, len_over=3000, mc=2, iden=0.95, long=True)
    want = F.stage([raw], True, 3000, 2, 0.95)
    got = open(out).read().split("\n")[:-1]
    assert got == want
    pair = _pair_rows(got)
    assert len(pair) == 1 and pair[0][2] == first_by_text
