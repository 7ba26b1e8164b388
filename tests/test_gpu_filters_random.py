"""GPU: the filter chain (v4 window filter -> pile-up -> pass 2) on randomly generated PAF rows against the oracle
restatement (itself pinned to the reference's goldens).  Dense pairs, both directions, repeated rows, every CIGAR op,
adjacent X runs, rows that fail each predicate: cases the fixtures made from simulated reads hit only rarely."""
import random

import pytest

from hylight_amd import api
from oracle import filters as F

pytestmark = pytest.mark.gpu


def _rows(seed, n_reads, n_rows, long_mode):
    rnd = random.Random(seed)
    lens = [rnd.randint(300, 2500) for _ in range(n_reads)]
    rows = []
    for _ in range(n_rows):
        q, t = rnd.sample(range(n_reads), 2)
        if rnd.random() < 0.05:
            t = q                                                   # self rows are dropped by the filters
        ops, qspan, tspan, nmatch = [], 0, 0, 0
        for _ in range(rnd.randint(1, 40)):
            op = rnd.choices("=XID", weights=(10, 3, 1, 1))[0]
            if ops and ops[-1][1] == op and rnd.random() < 0.7:
                op = "="                                            # mostly alternate, sometimes repeat an op
            k = rnd.randint(1, 60) if op == "=" else rnd.randint(1, 12)
            ops.append((k, op))
            if op in "=XI":
                qspan += k
            if op in "=XD":
                tspan += k
            if op == "=":
                nmatch += k
        if qspan == 0 or tspan == 0 or qspan > lens[q] or tspan > lens[t]:
            continue
        # overhangs: often dovetail-like (one side at a read end), sometimes internal
        qs = rnd.choice((0, lens[q] - qspan, rnd.randint(0, lens[q] - qspan)))
        ts = rnd.choice((0, lens[t] - tspan, rnd.randint(0, lens[t] - tspan)))
        blen = sum(k for k, _ in ops)
        cg = "cg:Z:" + "".join(f"{k}{o}" for k, o in ops)
        line = f"r{q}\t{lens[q]}\t{qs}\t{qs + qspan}\t{rnd.choice('+-')}\tr{t}\t{lens[t]}\t{ts}\t{ts + tspan}\t{nmatch}\t{blen}\t0\t{cg}"
        rows.append(line)
        if rnd.random() < 0.1:
            rows.append(line)                                       # exact duplicate
    return rows


@pytest.mark.parametrize("seed,long_mode,kw", [
    (1, True, dict(len_over=100, mc=2, iden=0.8)),
    (2, True, dict(len_over=300, mc=3, iden=0.9)),
    (3, False, dict(len_over=60, mc=2, iden=0.8)),
    (4, True, dict(len_over=50, mc=1, iden=0.5)),
])
def test_filter_chunk_on_random_rows(tmp_path, seed, long_mode, kw):
    rows = _rows(seed, n_reads=40, n_rows=2500, long_mode=long_mode)
    src = tmp_path / "in.paf"
    src.write_text("\n".join(rows) + "\n")
    out = tmp_path / "out.paf"
    api.filter_chunk(src, out, long_mode=long_mode, **kw)
    want = F.worker(rows, long_mode, kw["len_over"], kw["mc"], kw["iden"])
    got = open(out).read().split("\n")[:-1]
    assert got == want
    assert len(rows) > 1500


@pytest.mark.parametrize("seed,kw", [(5, dict(min_len=30, min_o=3)), (6, dict(min_len=200, min_iden=0.7, min_o=25))])
def test_v4_window_filter_on_random_rows(tmp_path, seed, kw):
    rows = _rows(seed, n_reads=25, n_rows=3500, long_mode=True)     # > 3 windows of 1000 rows, few reads: 60-cap hits
    src = tmp_path / "in.paf"
    src.write_text("\n".join(rows) + "\n")
    out = tmp_path / "out.paf"
    api.paf_window_filter(4, src, out, **kw)
    want = F.window_filter(rows, 4, kw["min_len"], kw.get("min_iden"), kw["min_o"])
    assert open(out).read().split("\n")[:-1] == want
