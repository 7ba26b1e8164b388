#!/usr/bin/env python3
"""Golden vectors of the driver boundary, captured by RUNNING the reference's own driver:

    python /root/reference/script/HyLight.py -l fxF_long.fq -s fxF_short.fq --corrected --nsplit 3 -t 4 -o OUT

in the build container with a `minimap2` stand-in on $PATH.  minimap2 is not part of the reference tree;
the stand-in answers `minimap2 ... <chunk> <reads>` with the rows of this repo's CPU oracle overlapper
(oracle/ava_oracle.c) for exactly that chunk and read file, so everything downstream of the overlapper - filter
scripts, GNU sorts, the --nsplit chunking of utils.split_reads2, the shipped tools/miniasm/miniasm, gfa2fa - is the
reference's unmodified code.  The reference run dies later (racon and the short-read tools are absent);
the files of the path up to contigs1.fa exist by then and are what is captured:

    fxF_long.fq.gz       input (FASTQ with lower-case / IUPAC bases and a header with a space)
    fxF_s1.fa.gz         OUT/1.split_fastx/s1.fa        (utils.filter_non_atcg)
    fxF_s1_s1.paf        OUT/2.overlap/s1_s1.paf        (utils.split_reads2, len_over 6000, HyLight.py:130)
    fxF_contigs1.gfa.gz  OUT/tmp/contigs1.gfa           (miniasm -d 10000 -n 1 -e 1 -c 1 -f, HyLight.py:140)
    fxF_contigs1.fa.gz   OUT/tmp/contigs1.fa            (HyLight.gfa2fa)

The same at the size SURVEY.md 8d names for C1 (the example data set is not in the reference tree: C2's recipe at a tenth
of its size, 1 000 long reads, `--corrected --nsplit 100 -t 8`): fxG_*.  Its input is not stored - the seeded simulator
makes it again (hylight_amd.workloads.c1_reads) and fxG_meta.json holds the SHA-256 of the FASTQ the goldens belong to.

Usage: python tests/golden/make_goldens_driver.py [fxF] [fxG]      (LC_ALL=C, as for the other goldens)
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_DRIVER = "/root/reference/script/HyLight.py"
sys.path.insert(0, ROOT)

from hylight_amd import simulate as S  # noqa: E402

SHIM = r'''#!/usr/bin/env python3
# stand-in for the absent minimap2 binary: the oracle overlapper on <chunk> <reads> (the last two arguments)
import os, sys, tempfile
sys.path.insert(0, os.environ["HL_REPO"])
from oracle import ava as OA
target, query = sys.argv[-2], sys.argv[-1]
opts = OA.opts_short() if "--sr" in sys.argv else OA.opts_long()
fd, out = tempfile.mkstemp(suffix=".paf")
os.close(fd)
OA.ava(target, query, out, opts)
with open(out) as f:
    sys.stdout.write(f.read())
os.remove(out)
'''


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def fixture_reads():
    reads, _ = S.simulate_reads(seed=83, n_strains=2, genome_len=30_000, n_reads=70, mean_len=9000, min_len=7000,
                                max_len=13_000)
    reads[3].seq[10] = ord("n")           # lower case + non-ACGT must be sanitised (utils.py:96-102)
    reads[4].seq[20] = ord("R")
    reads[5].name = "r5 extra words"      # header is cut at the first space
    return reads


def main_c1():
    import hashlib
    import json
    from hylight_amd import workloads as W
    tmp = tempfile.mkdtemp(prefix="hl_golden_c1_")
    shim_dir = os.path.join(tmp, "shim")
    os.makedirs(shim_dir)
    with open(os.path.join(shim_dir, "minimap2"), "w") as f:
        f.write(SHIM)
    os.chmod(os.path.join(shim_dir, "minimap2"), 0o755)
    env = dict(os.environ, LC_ALL="C", PATH=shim_dir + ":" + os.environ["PATH"], HL_REPO=ROOT)
    fq = os.path.join(tmp, "fxG_long.fq")
    S.write_fastq(W.c1_reads(), fq)
    short = os.path.join(tmp, "fxG_short.fq")
    with open(short, "w") as f:
        f.write("@s0/1\nACGTACGTAC\n+\nIIIIIIIIII\n")
    out = os.path.join(tmp, "OUT")
    r = subprocess.run([sys.executable, REF_DRIVER, "-l", fq, "-s", short, "--corrected", "--nsplit", "100", "-t", "8", "-o", out],
                       cwd=tmp, env=env, capture_output=True, text=True, timeout=7200)
    print("reference driver exit code:", r.returncode)
    meta = dict(cmd="HyLight.py -l fxG_long.fq -s fxG_short.fq --corrected --nsplit 100 -t 8 -o OUT", fastq_sha256=hashlib.sha256(open(fq, "rb").read()).hexdigest())
    for name, rel in {"fxG_s1_s1.paf.gz": "2.overlap/s1_s1.paf", "fxG_contigs1.gfa.gz": "tmp/contigs1.gfa", "fxG_contigs1.fa.gz": "tmp/contigs1.fa"}.items():
        src = os.path.join(out, rel)
        if not os.path.exists(src) or os.path.getsize(src) == 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(f"the reference run did not produce {rel}")
        gz(src, os.path.join(HERE, name))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")
    meta["provenance"] = {
        "driver_and_filters": "the reference's own code, run unmodified (script/HyLight.py, utils.split_reads2, filter_overlap_slr2.py, tools/miniasm)",
        "overlapper": "NOT minimap2 (absent from the reference tree, SURVEY.md D1): a stand-in on $PATH that answers with this repo's "
                      "oracle/ava_oracle.c for the same chunk and read file - the PAF rows entering the reference's filters are this project's "
                      "specification of the overlapper, and these goldens are regenerated when that specification changes",
        "oracle_sha256": hashlib.sha256(open(os.path.join(ROOT, "oracle", "ava_oracle.c"), "rb").read()).hexdigest(),
        "oracle_commit": subprocess.run(["git", "log", "-1", "--format=%h", "--", "oracle/ava_oracle.c"], cwd=ROOT, capture_output=True, text=True).stdout.strip()}
    meta["s1_fa_sha256"] = hashlib.sha256(open(os.path.join(out, "1.split_fastx/s1.fa"), "rb").read()).hexdigest()
    meta["paf_rows"] = sum(1 for _ in open(os.path.join(out, "2.overlap/s1_s1.paf")))
    with open(os.path.join(HERE, "fxG_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    shutil.rmtree(tmp, ignore_errors=True)


def main():
    tmp = tempfile.mkdtemp(prefix="hl_golden_drv_")
    shim_dir = os.path.join(tmp, "shim")
    os.makedirs(shim_dir)
    with open(os.path.join(shim_dir, "minimap2"), "w") as f:
        f.write(SHIM)
    os.chmod(os.path.join(shim_dir, "minimap2"), 0o755)
    env = dict(os.environ, LC_ALL="C", PATH=shim_dir + ":" + os.environ["PATH"], HL_REPO=ROOT)
    fq = os.path.join(tmp, "fxF_long.fq")
    S.write_fastq(fixture_reads(), fq)
    short = os.path.join(tmp, "fxF_short.fq")
    with open(short, "w") as f:          # never reached on the captured part of the path
        f.write("@s0/1\nACGTACGTAC\n+\nIIIIIIIIII\n")
    out = os.path.join(tmp, "OUT")
    r = subprocess.run([sys.executable, REF_DRIVER, "-l", fq, "-s", short, "--corrected", "--nsplit", "3", "-t", "4", "-o", out],
                       cwd=tmp, env=env, capture_output=True, text=True, timeout=1800)
    print("reference driver exit code:", r.returncode, "(it stops after the captured stages: racon etc. are absent)")
    want = {"fxF_s1.fa.gz": "1.split_fastx/s1.fa", "fxF_s1_s1.paf": "2.overlap/s1_s1.paf", "fxF_contigs1.gfa.gz": "tmp/contigs1.gfa",
            "fxF_contigs1.fa.gz": "tmp/contigs1.fa"}
    for name, rel in want.items():
        src = os.path.join(out, rel)
        if not os.path.exists(src) or os.path.getsize(src) == 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit(f"the reference run did not produce {rel}")
        if name.endswith(".gz"):
            gz(src, os.path.join(HERE, name))
        else:
            shutil.copyfile(src, os.path.join(HERE, name))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")
    gz(fq, os.path.join(HERE, "fxF_long.fq.gz"))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fxF", "fxG"]
    if "fxF" in which:
        main()
    if "fxG" in which:
        main_c1()
