"""GPU vs oracle, byte for byte, AT THE BENCHED WORKLOADS (BASELINE.json configs[2], [4], [3]):

  * C3 (100 000 reads, --nsplit 200): one WHOLE --nsplit chunk - 504 target reads against all 100 000 queries, the
    reference's unit of work (script/utils.py:54-65) - raw overlapper rows (hlmi_ava = the minimap2 call of
    filter_overlap_slr2.py:51) and the worker's output for the chunk (hlmi_split_reads2_shard restricted to it vs the
    oracle overlapper + oracle/filters.py): every target sees its real pile-up depth (~1000 x pooled);
  * C5 at its real depth and divergence (50 strains, ANI 95-99 %, 2 % read errors; a tenth of the reads on a tenth
    of the genome length = the same depth): targets sampled across the file against ALL queries;
  * C4s (the short-read calls of the hybrid pipeline, script/HyLight.py:200): contig pieces sampled across the file
    against all short reads of C4 at a tenth.

tests/test_gpu_c2_chunks_oracle.py does the same for C2.  The oracle runs its OpenMP query loop on all host cores."""
import os
import subprocess
import sys

import pytest

from hylight_amd import api
from hylight_amd import workloads as W
from oracle import filters as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_proc(target, query, out, long_mode=True, stub=-1):
    """The oracle overlapper in a process of its own (the GPU works meanwhile)."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom oracle import ava as OA\n"
            "o = OA.opts_long() if %r else OA.opts_short()\no.stub_oh = %d\nOA.ava(%r, %r, %r, o)\n") % (ROOT, long_mode, stub, target, query, out)
    return subprocess.Popen([sys.executable, "-c", code])


def _lines(path):
    with open(path) as f:
        return f.read().split("\n")[:-1]


def _first_difference(got, want):
    for i, (g, w) in enumerate(zip(got, want)):
        if g != w:
            return f"row {i}:\n  gpu    {g[:300]}\n  oracle {w[:300]}"
    return f"row counts {len(got)} (gpu) vs {len(want)} (oracle)"


def test_one_whole_chunk_of_c3_matches_the_oracle(tmp_path):
    cfg = W.config("C3")
    fa = str(tmp_path / "s1.fa")
    W.make_long(cfg, fa)
    with open(fa) as f:
        lines = f.read().split("\n")[:-1]
    ranges = F.chunk_ranges(len(lines), cfg["nsplit"])
    assert len(ranges) == 199
    c = 60                                     # 504 reads; by the pair-once rule it meets ~30 % of its overlapping queries
    lo, hi = ranges[c]
    assert hi - lo == 1008
    cf = str(tmp_path / f"sub{c:05d}")
    with open(cf, "w") as f:
        f.write("\n".join(lines[lo:hi]) + "\n")
    all_lines, ranges2 = (lines, ranges) if os.environ.get("HL_TEST_C3_MORE_CHUNKS") else (None, None)
    del lines
    p = _oracle_proc(cf, fa, cf + ".oracle.paf")
    api.ava(cf, fa, cf + ".gpu.paf")
    stage = cfg["stage"]
    api.split_reads2(fa, fa, cfg["nsplit"], tmp_path, tmp_path / "w.paf", long=True, rank=c, world=len(ranges), **stage)
    assert p.wait(timeout=1500) == 0
    want, got = _lines(cf + ".oracle.paf"), _lines(cf + ".gpu.paf")
    assert len(want) > 100_000
    assert got == want, "whole C3 chunk, raw overlapper rows: " + _first_difference(got, want)
    # The stage's own constants keep a handful of rows at this depth (a position of a read collects the X of every read piled
    # up on it that errs there - at ~1000 x pooled depth at least mc of them do nearly everywhere, with con - v still large:
    # supported keys of slr2:370-405 all along every alignment, and -thre 0.0025 drops nearly every pair): the comparison
    # above says little about the pair counts.  So the same chunk's rows go through the filter chain with -thre swept across the
    # distribution of count / matchcount and -len 1000: the survivors - tens of thousands - depend on every pair's count.
    # more chunks on request (HL_TEST_C3_MORE_CHUNKS="7,140,198": a minute of the box's host cores each), raw rows only
    for c2 in [int(x) for x in os.environ.get("HL_TEST_C3_MORE_CHUNKS", "").split(",") if x.strip()]:
        lo2, hi2 = ranges2[c2]
        cf2 = str(tmp_path / f"sub{c2:05d}")
        with open(cf2, "w") as f:
            f.write("\n".join(all_lines[lo2:hi2]) + "\n")
        p2 = _oracle_proc(cf2, fa, cf2 + ".oracle.paf")
        api.ava(cf2, fa, cf2 + ".gpu.paf")
        assert p2.wait(timeout=1500) == 0
        w2, g2 = _lines(cf2 + ".oracle.paf"), _lines(cf2 + ".gpu.paf")
        print(f"C3 chunk {c2}: {len(w2)} candidate rows compared")
        assert g2 == w2, f"whole C3 chunk {c2}, raw overlapper rows: " + _first_difference(g2, w2)
    thresholds = [0.0025, 0.004, 0.008, 0.02, 1.0]
    sweep = F.worker_sweep(want, True, 1000, stage["mc"], stage["iden"], thresholds)
    w = F.sort_scored(F.worker(want, True, stage["len_over"], stage["mc"], stage["iden"]))
    g = _lines(tmp_path / "w.paf")
    print(f"C3 chunk {c}: {len(want)} candidate rows, {len(w)} final rows")
    assert g == w, "whole C3 chunk, worker output: " + _first_difference(g, w)
    sizes = {}
    for t in thresholds:
        api.filter_chunk(cf + ".oracle.paf", cf + f".f{t}.paf", 1000, stage["mc"], stage["iden"], thre=t)
        gt, wt = _lines(cf + f".f{t}.paf"), sweep[t]
        sizes[t] = len(wt)
        assert gt == wt, f"whole C3 chunk, filter chain at -thre {t}: " + _first_difference(gt, wt)
    print("C3 chunk, rows kept per -thre:", sizes)
    assert sizes[1.0] > 100_000 and sizes[0.02] >= 10_000 and sizes[0.008] >= 10_000
    assert sizes[0.0025] < sizes[0.004] < sizes[0.008] < sizes[0.02] <= sizes[1.0]      # the sweep crosses the distribution


def _sampled_targets(fa, out, n, per_rec=2):
    with open(fa) as f:
        lines = f.read().split("\n")[:-1]
    n_rec = len(lines) // per_rec
    pick = sorted({(k * n_rec) // n + (n_rec // (2 * n)) for k in range(n)})
    with open(out, "w") as f:
        for r in pick:
            f.write("\n".join(lines[per_rec * r:per_rec * (r + 1)]) + "\n")
    return len(pick)


def test_c5_depth_sample_matches_the_oracle(tmp_path):
    cfg = W.config("C5", 0.1)                  # 50 000 reads on 50 strains x 200 kb: C5's depth, divergence and error model
    fa = str(tmp_path / "s1.fa")
    W.make_long(cfg, fa)
    tf = str(tmp_path / "targets.fa")
    assert _sampled_targets(fa, tf, 48) == 48
    procs = [_oracle_proc(tf, fa, tf + f".oracle{stub}.paf", stub=stub) for stub in (-1, 3)]
    o = api.ava_opts_long()
    api.ava(tf, fa, tf + ".gpu-1.paf", o)
    o.stub_oh = 3                              # the form the stage entry points run (stub rule)
    api.ava(tf, fa, tf + ".gpu3.paf", o)
    for p in procs:
        assert p.wait(timeout=1500) == 0
    for stub in (-1, 3):
        want, got = _lines(tf + f".oracle{stub}.paf"), _lines(tf + f".gpu{stub}.paf")
        assert len(want) > 5_000
        assert got == want, f"C5 sample (stub_oh {stub}): " + _first_difference(got, want)
    stage = cfg["stage"]
    rows = _lines(tf + ".oracle-1.paf")        # (every row with its CIGAR: the stub rule's bare rows end in cg:Z:*)
    # C5's reads carry 1 % substitutions.  On the erring read's own key v = con (all partners disagree), so c = con - v = 0 and
    # the key is NOT supported; but a position of a partner collects the X of every read that errs there, at this depth at least
    # mc of them nearly everywhere (slr2:383-396), so -thre 0.0025 drops every pair - the reference's filter is made for
    # corrected reads.
    # The filter chain is therefore compared with -thre swept across the distribution of count / matchcount.
    thresholds = [0.0025, 0.02, 0.04, 1.0]
    sweep = F.worker_sweep(rows, True, stage["len_over"], stage["mc"], stage["iden"], thresholds)
    sizes = {}
    for t in thresholds:
        api.filter_chunk(tf + ".oracle-1.paf", tf + f".f{t}.paf", stage["len_over"], stage["mc"], stage["iden"], thre=t)
        gt = _lines(tf + f".f{t}.paf")
        sizes[t] = len(sweep[t])
        assert gt == sweep[t], f"C5 sample, filter chain at -thre {t}: " + _first_difference(gt, sweep[t])
    print(f"C5 sample: {len(rows)} candidate rows of 48 targets, rows kept per -thre: {sizes}")
    assert sizes[1.0] >= 1_000 and sizes[0.0025] <= sizes[0.02] <= sizes[0.04] <= sizes[1.0]


def test_c4_short_calls_sample_matches_the_oracle(tmp_path):
    from hylight_amd import simulate as S
    cfg = W.config("C4", 0.02)                 # 200 000 short reads, 100 strains x 50 kb: C4's short-read depth
    _, _, strains = W.make_long(dict(cfg, sim=dict(cfg["sim"], n_reads=10)), str(tmp_path / "unused.fa"))
    short_fa, con_fa = str(tmp_path / "short.fa"), str(tmp_path / "contigs.fa")
    W.make_short(cfg, strains, short_fa)
    contigs = [S.Read(f"longr_con_{k}_{a}", g[a:a + 40_000].copy(), None, k, a, a + 40_000, False)
               for k, g in enumerate(strains) for a in range(0, len(g) - 10_000, 40_000)]
    S.write_fasta(contigs, con_fa)
    tf = str(tmp_path / "targets.fa")
    n = _sampled_targets(con_fa, tf, 12)
    procs = [_oracle_proc(tf, short_fa, tf + f".oracle{stub}.paf", long_mode=False, stub=stub) for stub in (-1, 3)]
    o = api.ava_opts_short()
    api.ava(tf, short_fa, tf + ".gpu-1.paf", o)
    o.stub_oh = 3
    api.ava(tf, short_fa, tf + ".gpu3.paf", o)
    for p in procs:
        assert p.wait(timeout=1500) == 0
    for stub in (-1, 3):
        want, got = _lines(tf + f".oracle{stub}.paf"), _lines(tf + f".gpu{stub}.paf")
        assert len(want) > 2_000 * n
        assert got == want, f"C4s sample (stub_oh {stub}): " + _first_difference(got, want)
