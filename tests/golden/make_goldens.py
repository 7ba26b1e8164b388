#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's own code.

Runs only in the build container (needs /root/reference); the outputs are committed,
the reference's sources never are.  Recipe = SURVEY.md Appendix C:

  * filter_trans_ovlp_inline_v4.py / _v3.py / sfo2overlaps.py : run as CLIs on fixture text
  * filter_overlap_slr2.py : (a) functions prpare_mutation2 / mutation_re / prpare_mutation
                             imported and called on the sorted fixture PAF -> JSON dicts,
                             (b) main() end-to-end with a `minimap2` PATH shim that cats
                             the fixture rows whose target is in the chunk
  * utils.split_reads2     : whole stage with the same shim
  * tools/miniasm          : compiled by oracle/Makefile into oracle/_ref/miniasm, run with
                             HyLight's flags (HyLight.py:137,140) + -p paf/bed/sg dumps

Everything runs under LC_ALL=C (the container default; GNU sort's last-resort order
depends on it).  Usage:  python tests/golden/make_goldens.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SCRIPT = REF + "/script"
sys.path.insert(0, ROOT)

from hylight_amd import simulate as S  # noqa: E402

ENV = dict(os.environ, LC_ALL="C")


def run(cmd, **kw):
    subprocess.check_call(cmd, shell=True, env=kw.pop("env", ENV), **kw)


def write_lines(path, lines):
    with open(path, "w") as f:
        for l in lines:
            f.write(l + "\n")


SHIM = r'''#!/usr/bin/env python3
# stand-in for the absent minimap2 binary (SURVEY.md App. C): prints the rows of $HL_FAKE_PAF
# whose target (col 6) is a record of the chunk file = second-to-last argument.
import os, sys
chunk = sys.argv[-2]
names = set()
with open(chunk) as f:
    for l in f:
        if l[:1] in ">@":
            names.add(l[1:].split()[0])
with open(os.environ["HL_FAKE_PAF"]) as f:
    for l in f:
        if l.split("\t")[5] in names:
            sys.stdout.write(l)
'''


def main():
    os.makedirs(HERE, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="hl_golden_")
    shim_dir = os.path.join(tmp, "shim")
    os.makedirs(shim_dir)
    with open(os.path.join(shim_dir, "minimap2"), "w") as f:
        f.write(SHIM)
    os.chmod(os.path.join(shim_dir, "minimap2"), 0o755)
    # `python` must resolve (the reference shells out to `python`)
    env = dict(ENV, PATH=shim_dir + ":" + ENV["PATH"])

    # ---------------- fixture A: strain mixture with CIGARs (filters a4-a8, miniasm) ------------
    reads, _ = S.simulate_reads(seed=S.SEED_DEFAULT, n_strains=3, genome_len=24_000, n_reads=120,
                                mean_len=6000, min_len=2500, max_len=12_000, keep_gpos=True)
    S.write_fasta(reads, f"{HERE}/fxA_reads.fa")
    pafA = S.truth_paf(reads, min_cols=40)
    write_lines(f"{HERE}/fxA_ava.paf", pafA)
    print("fixture A:", len(reads), "reads", len(pafA), "rows")

    # ---------------- fixture B: dense, no tags (v4 window / 60-cap quirks) ---------------------
    readsB, _ = S.simulate_reads(seed=7, n_strains=1, genome_len=5_000, n_reads=90,
                                 mean_len=2500, min_len=1500, max_len=4000, keep_gpos=True,
                                 name_prefix="d")
    pafB = S.truth_paf(readsB, min_cols=20, with_tags=False, pair_once=False)
    # quirk rows: self hit, short, low identity, duplicate pair across a window boundary
    extra = ["d0\t3000\t0\t3000\t+\td0\t3000\t0\t3000\t3000\t3000\t0",
             "d1\t3000\t0\t20\t+\td2\t3000\t2980\t3000\t20\t20\t0",
             "d1\t3000\t0\t200\t+\td2\t3000\t2800\t3000\t100\t200\t0",
             "d3\t3000\t500\t2500\t-\td4\t3000\t400\t2400\t1990\t2000\t0"]
    pafB = pafB[:1500] + extra + pafB[1500:]
    pafB += [pafB[10], pafB[999], pafB[1000], pafB[2500]]  # duplicate pairs, same and different windows
    write_lines(f"{HERE}/fxB_dense.paf", pafB)
    print("fixture B:", len(pafB), "rows")

    # ---------------- a4: v4 window filter ------------------------------------------------------
    run(f"python {SCRIPT}/filter_trans_ovlp_inline_v4.py -len 30 -oh 3 < {HERE}/fxA_ava.paf > {HERE}/fxA_v4.paf", env=env)
    run(f"python {SCRIPT}/filter_trans_ovlp_inline_v4.py -len 30 -oh 3 < {HERE}/fxB_dense.paf > {HERE}/fxB_v4.paf", env=env)
    run(f"python {SCRIPT}/filter_trans_ovlp_inline_v4.py -len 100 -iden 0.9 -oh 40 < {HERE}/fxB_dense.paf > {HERE}/fxB_v4_len100_oh40.paf", env=env)

    # ---------------- a17: v3 (-sfo and score mode) ; a18: sfo2overlaps -------------------------
    # contig-like ids are integers (HyLight.py:300 renames contigs to @1..@n)
    id_of = {r.name: str(i + 1) for i, r in enumerate(readsB)}
    pafC = []
    for l in pafB:
        c = l.split("\t")
        c[0], c[5] = id_of[c[0]], id_of[c[5]]
        pafC.append("\t".join(c))
    write_lines(f"{HERE}/fxC_contigs.paf", pafC)
    run(f"python {SCRIPT}/filter_trans_ovlp_inline_v3.py -len 90 -iden 0.99 -oh 2 -sfo < {HERE}/fxC_contigs.paf > {HERE}/fxC_v3.sfo", env=env)
    run(f"python {SCRIPT}/filter_trans_ovlp_inline_v3.py -len 90 -iden 0.9 -oh 30 -sfo < {HERE}/fxC_contigs.paf > {HERE}/fxC_v3_oh30.sfo", env=env)
    run(f"python {SCRIPT}/filter_trans_ovlp_inline_v3.py -len 90 -iden 0.9 -oh 30 < {HERE}/fxC_contigs.paf > {HERE}/fxC_v3_oh30.score", env=env)
    for tag in ("fxC_v3", "fxC_v3_oh30"):
        d = os.path.join(tmp, "sfo_" + tag)
        os.makedirs(d)
        shutil.copy(f"{HERE}/{tag}.sfo", d + "/in.sfo")
        run(f"python {SCRIPT}/sfo2overlaps.py --in {d}/in.sfo --out {d}/out.savage --num_singles {len(readsB)} --num_pairs 0", env=env)
        shutil.copy(d + "/out.savage", f"{HERE}/{tag}.savage")

    # ---------------- intermediate sort (slr2:57) ----------------------------------------------
    run(f"sort -nk7 -k8 -k9 -k5 {HERE}/fxA_v4.paf > {HERE}/fxA_v4_sorted.paf", env=env)

    # ---------------- a5/a6: function-level goldens ---------------------------------------------
    sys.path.insert(0, SCRIPT)
    import filter_overlap_slr2 as slr2  # the reference module itself (never copied)
    from collections import defaultdict
    for mode, fn in (("long", slr2.prpare_mutation2), ("short", slr2.prpare_mutation)):
        with open(f"{HERE}/fxA_v4_sorted.paf") as f:
            snp, map_po, start_po = fn(f)
        sps = defaultdict(list)
        for k in start_po.keys():
            sps[k] = sorted(start_po[k], key=lambda x: (x[0], x[1]))
        out = {"snp": dict(snp), "n_map_po": sum(len(v) for v in map_po.values()),
               "n_start_po": sum(len(v) for v in start_po.values())}
        for mc in (2, 3):
            out[f"mutation_mc{mc}"] = slr2.mutation_re(snp, sps, map_po, mc=mc)
        with open(f"{HERE}/fxA_snp_{mode}.json", "w") as f:
            json.dump(out, f, sort_keys=True, separators=(",", ":"))
        print(mode, "snp keys", len(snp), "mutation pairs mc2", len(out["mutation_mc2"]))
    # sum_before_X known answers
    kat = ["cg:Z:5=1X3=\n", "cg:Z:10=12X4=2X\n", "cg:Z:100=\n", "cg:Z:3=20X1=9X\n"]
    with open(f"{HERE}/sum_before_X.json", "w") as f:
        json.dump({k: slr2.sum_before_X(k) for k in kat}, f)

    # ---------------- a2: worker main() end-to-end with the shim --------------------------------
    for tag, args in (("long_len1000", "-len 1000 -mc 2 -iden 0.95 -long_reads"),
                      ("long_len3000_iden99", "-len 3000 -mc 2 -iden 0.99 -long_reads"),
                      ("short_len70", "-len 70 -mc 3 -iden 0.95")):
        d = os.path.join(tmp, "w_" + tag)
        os.makedirs(d)
        shutil.copy(f"{HERE}/fxA_reads.fa", d + "/chunk0")   # one chunk = all reads
        shutil.copy(f"{HERE}/fxA_reads.fa", d + "/reads.fa")
        run(f"python {SCRIPT}/filter_overlap_slr2.py -r reads.fa -c chunk0 -t 3 {args}", cwd=d,
            env=dict(env, HL_FAKE_PAF=f"{HERE}/fxA_ava.paf"))
        shutil.copy(d + "/chunk0_tmp_overlap4.paf", f"{HERE}/fxA_worker_{tag}.paf")

    # ---------------- a1: utils.split_reads2 whole stage ----------------------------------------
    d = os.path.join(tmp, "stage")
    os.makedirs(d)
    shutil.copy(f"{HERE}/fxA_reads.fa", d + "/s1.fa")
    code = (f"import sys; sys.path.insert(0, {SCRIPT!r}); import utils; "
            f"utils.split_reads2({d!r}+'/s1.fa', {d!r}+'/s1.fa', 4, {d!r}, {d!r}+'/s1_s1.paf', {SCRIPT!r}, "
            f"threads=4, len_over=1000, mc=2, iden=0.95, long=True)")
    run(f"python -c \"{code}\"", cwd=d, env=dict(env, HL_FAKE_PAF=f"{HERE}/fxA_ava.paf"))
    shutil.copy(d + "/s1_s1.paf", f"{HERE}/fxA_stage_nsplit4.paf")

    # ---------------- a9-a16: miniasm -----------------------------------------------------------
    run(f"make -C {ROOT}/oracle ref")
    mini = f"{ROOT}/oracle/_ref/miniasm"
    paf = f"{HERE}/fxA_stage_nsplit4.paf"
    fa = f"{HERE}/fxA_reads.fa"
    for tag, flags in (("n1c1", "-d 10000 -n 1 -e 1 -c 1"), ("n3c3", "-d 10000 -n 3 -e 1 -c 3")):
        run(f"{mini} {flags} -f {fa} {paf} > {HERE}/fxA_miniasm_{tag}.gfa 2>/dev/null")
        for p in ("paf", "bed", "sg"):
            run(f"{mini} {flags} -p {p} {paf} > {HERE}/fxA_miniasm_{tag}.{p} 2>/dev/null")
    # filter_ovlp_inline.py + minimap22sfo.py as polyte.tune_params.py:507-515 chains them
    # (`cut -f 1-12 | filter_ovlp_inline.py <len> <iden> <o> <r>` then `minimap22sfo.py -m 0 -p 0`)
    for tag, a in (("len90_oh30", "90 0.9 30 0.8"), ("len90_oh1", "90 0.98 1 0.8")):
        run(f"python {SCRIPT}/filter_ovlp_inline.py {a} < {HERE}/fxC_contigs.paf > {HERE}/fxC_inline_{tag}.paf", env=env)
        run(f"python {SCRIPT}/minimap22sfo.py --in {HERE}/fxC_inline_{tag}.paf --out {HERE}/fxC_inline_{tag}.sfo -m 0 -p 0", env=env)
    run(f"python {SCRIPT}/minimap22sfo.py --in {HERE}/fxC_contigs.paf --out {HERE}/fxC_m22sfo_m200_p99.sfo -m 200 -p 99", env=env)

    # fixture D: imperfect overlaps -> tips, bubbles, bi-loops, internal cuts, short-overlap removal
    for seed in (1, 2, 3):
        readsD, pafD = S.messy_graph_paf(seed)
        write_lines(f"{HERE}/fxD{seed}_messy.paf", pafD)
        fa_d = f"{tmp}/fxD{seed}.fa"
        S.write_fasta(readsD, fa_d)
        if seed == 1:
            shutil.copy(fa_d, f"{HERE}/fxD1_reads.fa")
        for tag, flags in (("n1c1", "-d 10000 -n 1 -e 1 -c 1"), ("n3c3", "-d 10000 -n 3 -e 1 -c 3")):
            f_opt = f"-f {fa_d}" if seed == 1 else ""
            run(f"{mini} {flags} {f_opt} {HERE}/fxD{seed}_messy.paf > {HERE}/fxD{seed}_miniasm_{tag}.gfa "
                f"2>/dev/null")
            run(f"{mini} {flags} -p sg {HERE}/fxD{seed}_messy.paf > {HERE}/fxD{seed}_miniasm_{tag}.sg 2>/dev/null")
    shutil.rmtree(tmp)
    # compress the bulky text fixtures (tests read them through gzip)
    for fn in sorted(os.listdir(HERE)):
        p = os.path.join(HERE, fn)
        if fn.endswith((".py", ".gz")) or os.path.getsize(p) < 150_000:
            continue
        run(f"gzip -9 -n -f {p}")
    print("done")


if __name__ == "__main__":
    main()
