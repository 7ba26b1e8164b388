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
