#!/usr/bin/env python3
"""Golden vectors for the text passes of SURVEY 8f rank 4, made by CALLING the reference's own functions
(utils.filter_non_atcg, HyLight.gfa2fa, HyLight.pick_up imported from /root/reference/script).  Build container
only; inputs and outputs are committed, the reference's sources are not.  Usage: python tests/golden/make_goldens_text.py
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = "/root/reference/script"
sys.path.insert(0, SCRIPT)
os.chdir(SCRIPT)
import utils as RU  # noqa: E402
import HyLight as RH  # noqa: E402

FQ = (
    "@r1/1 first read, with a description\n"
    "acgtACGTnNRYKM-.*acgt\n+\nIIIIIIIIIIIIIIIIIIIII\n"
    "@r2\textra\tcolumns kept up to the first space\n"
    "GGGGCCCCTTTTAAAA\n+r2\nIIIIIIIIIIIIIIII\n"
    "@r3/2\r\n"
    "ttttuuuuACGTx\r\n+\r\nIIIIIIIIIIIII\r\n"
    "@r4 \n"
    "\n+\n\n"
    "@@odd/name/with/slashes 1\n"
    "ACGTNacgtn\n+\nIIIIIIIIII\n"
    "@r6/1\n"
    "GATTACA\n+\nIIIIIII"          # no newline at the end of the file
)
FA = (
    ">r1/1 first read\nacgtRYACGT\n"
    ">r2\nGGGGcccc\n"
    "r3 header without the marker\nNNNNacgu\n"
    ">r4/2   three spaces\nACGT"   # no final newline
)
PAF = (
    "r1/1\t20\t0\t20\t+\tr5\t30\t0\t20\t20\t20\t60\n"
    "r9\t10\t0\t10\t-\t@odd/x\t10\t0\t10\t10\t10\t0 trailing\n"
    "  r3/9 13 0 13 + zz/1 10 0 10 10 10 0\n"
)
GFA = (
    "H\tVN:Z:1.0\n"
    "S\tutg000001l\tACGTACGTAC\tLN:i:10\n"
    "a\tutg000001l\t0\tread1:1-10\t+\t10\n"
    "L\tutg000001l\t+\tutg000002l\t-\t5M\n"
    "S utg000002l GGGGG\n"
    "S\tutg3\t*\n"
)


def main():
    tmp = tempfile.mkdtemp(prefix="hl_text_")
    for name, text in (("fxE_reads.fq", FQ), ("fxE_reads.fa", FA), ("fxE_ovlp.paf", PAF), ("fxE_graph.gfa", GFA)):
        with open(os.path.join(HERE, name), "w", newline="") as f:
            f.write(text)
    for model, src in (("fastq", "fxE_reads.fq"), ("fasta", "fxE_reads.fa")):
        out = RU.filter_non_atcg(os.path.join(HERE, src), tmp, model)
        shutil.copy(out, os.path.join(HERE, f"fxE_non_atcg_{model}.fa"))
    RH.gfa2fa(os.path.join(HERE, "fxE_graph.gfa"), os.path.join(HERE, "fxE_gfa2fa.fa"))
    for src, tag in (("fxE_reads.fq", "fq"), ("fxE_reads.fa", "fa")):
        out = RH.pick_up(os.path.join(HERE, "fxE_ovlp.paf"), tmp, os.path.join(HERE, src))
        shutil.copy(out, os.path.join(HERE, f"fxE_pick_up.{tag}"))
        os.remove(out)
    shutil.rmtree(tmp)
    print("done")


if __name__ == "__main__":
    main()
