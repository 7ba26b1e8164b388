"""Effect sizes of the overlapper's documented deviations from minimap2, measured on a sample of
the C2 workload with the CPU oracle.  Not a parity claim (minimap2 is not in the reference tree): each
test states how many candidate rows of the sample a deviation changes, so the list in DESIGN.md carries numbers.

  * fixed window of 64 predecessors  vs  minimap2's predecessor loop (up to 5000 iterations, --max-chain-skip 25)
  * one-piece vs two-piece gap cost (now implemented: -O4,24 -E2,1)

Sample: 6 target reads (one piece of an --nsplit chunk) x the first 2500 reads of C2 as queries."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    from hylight_amd import simulate as S, workloads as W
    d = tmp_path_factory.mktemp("dev")
    cfg = W.config("C2")
    reads, _ = S.simulate_reads(seed=S.SEED_DEFAULT, min_len=1_000, max_len=40_000, **cfg["sim"])
    q, t = str(d / "q.fa"), str(d / "t.fa")
    S.write_fasta(reads[:2500], q)
    S.write_fasta(reads[9990:9996], t)          # names that sort after the queries' (pair-once rule)
    return d, q, t


def _run(q, t, out, env=None, two_piece=True):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import ava as OA\n"
            "o = OA.opts_long()\n"
            "%s"
            "OA.ava(%r, %r, %r, o)\n") % (ROOT, "" if two_piece else "o.gap_open2 = 0\n", t, q, out)
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, **(env or {})))
    return open(out).read().split("\n")[:-1]


def _compare(a, b):
    from collections import Counter
    same = sum((Counter(a) & Counter(b)).values())
    return dict(rows_a=len(a), rows_b=len(b), identical=same)


def test_fixed_predecessor_window_vs_minimap2_loop(sample):
    d, q, t = sample
    spec = _run(q, t, str(d / "spec.paf"))
    mm2 = _run(q, t, str(d / "mm2.paf"), env={"ORACLE_CHAIN_MM2": "1"})
    r = _compare(spec, mm2)
    print("fixed 64-predecessor window vs minimap2-style loop:", r)
    assert r["rows_a"] > 500
    # the window matters only where a chain has to jump more than 64 anchors: a few rows in a thousand at most
    assert r["identical"] >= 0.99 * max(r["rows_a"], r["rows_b"])


def test_two_piece_gap_cost_vs_one_piece(sample):
    d, q, t = sample
    two = _run(q, t, str(d / "two.paf"))
    one = _run(q, t, str(d / "one.paf"), two_piece=False)
    r = _compare(two, one)
    print("two-piece vs one-piece gap cost:", r)
    # C2 has single-base indels only: the second piece (gaps > 20 bases) cannot show
    assert r["identical"] == r["rows_a"] == r["rows_b"]


def test_against_minimap2_when_one_is_installed(sample):
    """Opportunistic: with a `minimap2` on $PATH, run the exact command of script/filter_overlap_slr2.py:51 on the sample
    and report how the specification's rows relate to it (pair recall, coordinate deltas).  minimap2 is not part of the
    reference tree, so without one this is skipped and a3 stays "parity unpinned"."""
    import shutil
    mm2 = shutil.which("minimap2")
    if not mm2:
        pytest.skip("no minimap2 on PATH: parity of the overlapper stays unpinned")
    d, q, t = sample
    spec = [l.split("\t") for l in _run(q, t, str(d / "spec_mm.paf"))]
    with open(d / "real.paf", "w") as out:
        subprocess.run([mm2, "-N", "40", "-t", "1", "-L", "--eqx", "-cx", "ava-pb", "-Hk19", "-m100", "-g10000",
                        "--max-chain-skip", "25", t, q], stdout=out, stderr=subprocess.DEVNULL, check=True)
    real = [l.split("\t") for l in open(d / "real.paf").read().split("\n")[:-1]]
    key = lambda f: (f[0], f[5], f[4])
    best = {}
    for f in real:                                  # the longest row of a (query, target, strand)
        if f[0] != f[5] and (key(f) not in best or int(f[10]) > int(best[key(f)][10])):
            best[key(f)] = f
    mine = {}
    for f in spec:
        if key(f) not in mine or int(f[10]) > int(mine[key(f)][10]):
            mine[key(f)] = f
    both = set(best) & set(mine)
    deltas = sorted(max(abs(int(best[k][i]) - int(mine[k][i])) for i in (2, 3, 7, 8)) for k in both)
    print("minimap2 rows %d (pairs %d), specification rows %d (pairs %d), common pairs %d, "
          "median / p95 / max coordinate delta of the longest row: %s"
          % (len(real), len(best), len(spec), len(mine), len(both),
             (deltas[len(deltas) // 2], deltas[int(0.95 * (len(deltas) - 1))], deltas[-1]) if deltas else None))
    long_pairs = {k for k, f in best.items() if int(f[10]) >= 6000}
    assert len(long_pairs & set(mine)) >= 0.95 * len(long_pairs)      # recall on the pairs the stage could keep
