"""GPU: the driver counterpart reproduces the reference's output tree for the long-read
path up to contigs1.fa; every file is checked against the oracle / compiled reference on the same input."""
import os
import subprocess

import pytest

from hylight_amd import driver, simulate as S
from oracle import ava as OA
from oracle import filters as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "miniasm")


def test_driver_long_read_path(tmp_path):
    reads, _ = S.simulate_reads(seed=81, n_strains=2, genome_len=40000, n_reads=110, mean_len=9000, min_len=7000,
                                max_len=14000)
    reads[3].seq[10] = ord("n")           # lower case + non-ACGT must be sanitised
    reads[4].seq[20] = ord("R")
    reads[5].name = "r5 extra words"      # header is cut at the first space
    fq = tmp_path / "long.fq"
    S.write_fastq(reads, fq)
    out = tmp_path / "OUT"
    assert driver.main(["-l", str(fq), "-o", str(out), "--corrected", "--nsplit", "3", "-t", "4", "--stop_after",
                        "contigs1"]) == 0
    s1 = (out / "1.split_fastx" / "s1.fa").read_text().split("\n")[:-1]
    assert len(s1) == 2 * len(reads)
    assert s1[6] == ">r3" and s1[7][10] == "N" and s1[9][20] == "N" and s1[10] == ">r5"
    assert all(set(l) <= set("ACGTN") for l in s1[1::2])
    # s1_s1.paf == oracle stage on s1.fa with the constants of HyLight.py:130 (len_over 6000)
    fa = out / "1.split_fastx" / "s1.fa"
    chunks = []
    for i, (lo, hi) in enumerate(F.chunk_ranges(len(s1), 3)):
        cf = tmp_path / f"c{i}.fa"
        cf.write_text("\n".join(s1[lo:hi]) + "\n")
        OA.ava(cf, fa, str(cf) + ".paf")
        chunks.append(open(str(cf) + ".paf").read().split("\n")[:-1])
    want = F.stage(chunks, True, 6000, 2, 0.95)
    got = (out / "2.overlap" / "s1_s1.paf").read_text().split("\n")[:-1]
    assert got == want and len(want) > 50
    gfa = (out / "tmp" / "contigs1.gfa").read_text()
    if os.path.exists(REF):
        ref = subprocess.run(f"{REF} -d 10000 -n 1 -e 1 -c 1 -f {fa} {out}/2.overlap/s1_s1.paf", shell=True, check=True,
                             capture_output=True).stdout.decode()
        assert gfa == ref
    fa_lines = (out / "tmp" / "contigs1.fa").read_text().split("\n")[:-1]
    s_lines = [l.split("\t") for l in gfa.split("\n") if l.startswith("S\t")]
    assert len(s_lines) >= 1 and fa_lines == [x for s in s_lines for x in (">" + s[1], s[2])]


def test_miniasm_launcher_is_argv_compatible(tmp_path, golden):
    out = subprocess.run([os.path.join(ROOT, "tools", "miniasm_mi"), "-d", "10000", "-n", "1", "-e", "1", "-c", "1",
                          golden.path("fxA_stage_nsplit4.paf")], check=True, capture_output=True).stdout.decode()
    want = golden.text("fxA_miniasm_n1c1.gfa").split("\n")
    # without -f the S lines carry '*' instead of the sequence
    got = out.split("\n")
    assert len(got) == len(want)
    for g, w in zip(got, want):
        if w.startswith("S\t"):
            gc, wc = g.split("\t"), w.split("\t")
            assert gc[0:2] == wc[0:2] and gc[2] == "*" and gc[3] == wc[3]
        else:
            assert g == w


def test_driver_reproduces_the_reference_drivers_files(tmp_path, golden):
    """Goldens captured from the reference's own HyLight.py run (tests/golden/make_goldens_driver.py: `--corrected
    --nsplit 3 -t 4`, oracle overlapper behind the minimap2 name): s1.fa, s1_s1.paf, contigs1.gfa, contigs1.fa."""
    import gzip
    fq = tmp_path / "fxF_long.fq"
    with gzip.open(golden.path("fxF_long.fq"), "rb") as f:
        fq.write_bytes(f.read())
    out = tmp_path / "OUT"
    assert driver.main(["-l", str(fq), "-o", str(out), "--corrected", "--nsplit", "3", "-t", "4", "--stop_after", "contigs1"]) == 0
    assert (out / "1.split_fastx" / "s1.fa").read_text() == golden.text("fxF_s1.fa")
    assert (out / "2.overlap" / "s1_s1.paf").read_text() == golden.text("fxF_s1_s1.paf")
    assert (out / "tmp" / "contigs1.gfa").read_text() == golden.text("fxF_contigs1.gfa")
    assert (out / "tmp" / "contigs1.fa").read_text() == golden.text("fxF_contigs1.fa")
    assert golden.text("fxF_s1_s1.paf").count("\n") > 100 and golden.text("fxF_contigs1.fa").count(">") >= 1


def _contigs(seed, n=14, genome=60_000):
    """Overlapping pieces of two strains, a few substitutions apart: what extend_con sees (polished contigs)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    _, strains = S.simulate_reads(seed=seed, n_strains=2, genome_len=genome, n_reads=1, snp_rate=0.004)
    recs = []
    for k in range(n):
        g = strains[k % 2]
        a = int(rng.integers(0, genome - 9000))
        b = a + int(rng.integers(4000, 9000))
        seq = g[a:b].copy()
        if k % 3 == 0:
            seq = S.revcomp(seq)
        recs.append((f"longr_con_{k}", seq.tobytes().decode()))
    recs.append(("tiny", "ACGT" * 30))                    # <= 150 bases: dropped (HyLight.py:297)
    return recs


def test_extend_con_files_match_the_oracle_chain(tmp_path):
    """HyLight.extend_con (HyLight.py:282-318): contigs_b.fastq, then `minimap2 --sr -X ... -r 0` | v3 filter -sfo |
    sfo2overlaps - every file against the oracle overlapper + the (reference-pinned) oracle filters."""
    recs = _contigs(7)
    fa = tmp_path / "all_contigs.fa"
    fa.write_text("".join(f">{n}\n{s}\n" for n, s in recs))
    tmp = tmp_path / "tmp"
    tmp.mkdir()
    n = driver.extend_con(str(fa), str(tmp), str(tmp_path / "final_contigs.fa"))
    assert n == len(recs) - 1
    fq = (tmp / "contigs_b.fastq").read_text().split("\n")[:-1]
    assert fq[0::4] == [f"@{k + 1}" for k in range(n)] and fq[1::4] == [s for _, s in recs[:-1]]
    assert all(q == "=" * len(s) for q, s in zip(fq[3::4], fq[1::4])) and set(fq[2::4]) == {"+"}
    assert (tmp / "stageb" / "fastq" / "singles.fastq").read_text() == (tmp / "contigs_b.fastq").read_text()
    o = OA.opts_short()
    o.pair_once, o.bandwidth = 1, 0
    OA.ava(tmp / "contigs_b.fastq", tmp / "contigs_b.fastq", tmp_path / "o.paf", o)
    raw = open(tmp_path / "o.paf").read()
    assert (tmp / "stageb" / "contigs_ava.paf").read_text() == raw and raw.count("\n") >= 10
    sfo = F.window_filter(raw.split("\n")[:-1], variant=3, min_len=90, min_iden=0.99, min_o=2, sfo=True)
    assert (tmp / "stageb" / "sfoverlaps.out").read_text().split("\n")[:-1] == sfo and len(sfo) >= 5
    assert (tmp / "stageb" / "sfoverlap.out.savage").read_text().split("\n")[:-1] == F.sfo2overlaps(sfo)


def test_driver_at_c1_size_reproduces_the_reference_drivers_files(tmp_path, golden):
    """BASELINE.json configs[0] as SURVEY.md 8d specifies it (the example files are not in the reference tree: C2's recipe
    at a tenth of its size): 1 000 long reads, `--corrected --nsplit 100 -t 8`.  Goldens = the reference's own HyLight.py
    run on the same FASTQ (tests/golden/make_goldens_driver.py fxG); the input is made again by the seeded simulator and
    checked against the recorded SHA-256."""
    import hashlib
    import json
    from hylight_amd import workloads as W
    meta = json.loads(golden.text("fxG_meta.json"))
    fq = tmp_path / "fxG_long.fq"
    S.write_fastq(W.c1_reads(), fq)
    assert hashlib.sha256(fq.read_bytes()).hexdigest() == meta["fastq_sha256"]
    out = tmp_path / "OUT"
    assert driver.main(["-l", str(fq), "-o", str(out), "--corrected", "--nsplit", "100", "-t", "8", "--stop_after", "contigs1"]) == 0
    assert hashlib.sha256((out / "1.split_fastx" / "s1.fa").read_bytes()).hexdigest() == meta["s1_fa_sha256"]
    paf = (out / "2.overlap" / "s1_s1.paf").read_text()
    assert paf == golden.text("fxG_s1_s1.paf") and paf.count("\n") == meta["paf_rows"] > 1000
    assert (out / "tmp" / "contigs1.gfa").read_text() == golden.text("fxG_contigs1.gfa")
    assert (out / "tmp" / "contigs1.fa").read_text() == golden.text("fxG_contigs1.fa")
