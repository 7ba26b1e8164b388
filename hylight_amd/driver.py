"""Counterpart of the reference driver for the long-read path.

Keeps script/HyLight.py's CLI flags and defaults (HyLight.py:25-52), stage order and output tree
(OUT/1.split_fastx/s1.fa, OUT/2.overlap/s1_s1.paf, OUT/tmp/contigs1.{gfa,fa}, OUT/tmp/ov_long_ref.paf,
...), and calls libhylight_mi.so at the two boundaries the path owns: B1 = split_reads2 and
B3 = miniasm.  External tools the reference shells out to (bfc, ropebwt2, fmlrc2, racon) are still
external; unlike the reference (whose `execute()` swallows most failures) a missing or
failing tool stops the run with a message.  The two short-read overlap calls of the path (HyLight.py:200,207)
run in the library's short mode.  `extend_con` (HyLight.py:282-326) is wired up to the SAVAGE overlap file: contigs ->
contigs_b.fastq -> self-overlaps (the `minimap2 --sr -X ... -r 0` call as hlmi_ava) -> v3 window filter with -sfo ->
sfo2overlaps; the consumers of that file - short-read clustering, POLYTE and the stage-b contig merge
(pipeline_per_stage.py / ViralQuasispecies) - are outside this implementation (SURVEY.md sections 2 and 8f): without
`--stageb_cmd` the run says so on stderr and, because the contractual output final_contigs.fa was not written, exits
with EXIT_NO_FINAL (3) and everything up to tmp/stageb/sfoverlap.out.savage in place (`--stop_after savage` declares
that partial run and exits 0).  The text passes (filter_non_atcg, gfa2fa, pick_up) are the library's native ones.

`--gpus N` (extension): every split_reads2 call is sharded over N GPUs of the node, chunk i -> rank i % N, the way the
reference fans its chunks out with `xargs -P` (script/utils.py:44-69).  The process starts N - 1 further copies of
itself before it touches the GPU (hylight_amd/launch.py); rank 0 runs the pipeline (text passes, graph builds, external
tools, merges), the other ranks only take part in the stage calls (stage.py:StagePool).  Started under torchrun
(RANK / WORLD_SIZE in the environment) it uses the ranks it was given.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
import time

from . import api, launch
from .stage import StagePool

__version__ = "1.0.1-mi355x"
EXIT_NO_FINAL = 3        # everything this implementation covers ran, but final_contigs.fa (the stage-b merge) was not written


def fq_or_fa(path):
    """toolkits.fq_or_fa (script/toolkits.py:7-18): first character decides."""
    with open(path) as f:
        c = f.read(1)
    if c == "@":
        return "fastq"
    if c == ">":
        return "fasta"
    raise SystemExit(f"{path}: neither FASTQ nor FASTA")


def filter_non_atcg(fq, out_dir, model):
    """utils.filter_non_atcg (script/utils.py:81-114): upper-case, [^ATGCN] -> N, header cut at the first
    space; 4-line FASTQ / 2-line FASTA records.  The pass itself is hlmi_filter_non_atcg."""
    new_dir = os.path.join(out_dir, "1.split_fastx")
    os.makedirs(new_dir, exist_ok=True)
    return api.filter_non_atcg(fq, os.path.join(new_dir, "s1.fa"), model)


def gfa2fa(gfa, fa):
    """HyLight.gfa2fa (script/HyLight.py:328-337): S lines only (hlmi_gfa2fa)."""
    api.gfa2fa(gfa, fa)


def pick_up(ovlap, outdir, fq):
    """HyLight.pick_up (script/HyLight.py:347-378): reads whose name (up to the first '/') appears in
    neither column 1 nor column 6 of the PAF (hlmi_pick_up); the output name follows the reference.  When every
    read overlaps, the reference leaves no file behind - callers get the path either way."""
    out = os.path.join(outdir, "sub" + str(time.time())[-3:] + "_remain.fq")
    return api.pick_up(ovlap, fq, out, fq_or_fa(fq))


def _tool(name):
    p = shutil.which(name)
    if not p:
        raise SystemExit(f"external tool `{name}` (README.md:14-19 of the reference) is not on PATH")
    return p


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, shell=True, cwd=cwd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        print(f"Error executing the command: {cmd}\n{r.stderr}", file=sys.stderr)
        raise SystemExit(1)


def contig_ava_opts():
    """minimap2 -t T --sr -X -c -k 21 -w 11 -s 60 -m 30 -n 2 -r 0 -A 4 -B 2 --end-bonus=100 (HyLight.py:309-310): the
    short-read constants, each pair once and never a contig against itself (-X), chaining bandwidth 0 (-r 0)."""
    o = api.ava_opts_short()
    o.pair_once = 1
    o.bandwidth = 0
    return o


def extend_con(input_con, outdir1, out_file, threads=30, len_c=50000000, stageb_cmd=None):
    """HyLight.extend_con (script/HyLight.py:282-326) up to the SAVAGE overlap file.  Returns the number of contigs
    written to contigs_b.fastq.  `stageb_cmd`: command prefix of the reference's stage-b script
    (`python .../pipeline_per_stage.py`) for installations that have it; without it the merge itself is skipped."""
    conb = os.path.join(outdir1, "contigs_b.fastq")
    if os.path.exists(conb):
        os.remove(conb)
    n = 0
    if os.path.exists(input_con):                              # HyLight.py:289-305: every sequence LINE longer than 150
        with open(input_con) as f, open(conb, "a") as w:
            for line in f:
                line = line.strip()
                if line.startswith(">"):
                    continue
                if len(line) > 150:
                    n += 1
                    w.write(f"@{n}\n{line}\n+\n{'=' * len(line)}\n")
    sb = os.path.join(outdir1, "stageb")
    os.makedirs(os.path.join(sb, "fastq"), exist_ok=True)
    raw, sfo, savage = os.path.join(sb, "contigs_ava.paf"), os.path.join(sb, "sfoverlaps.out"), os.path.join(sb, "sfoverlap.out.savage")
    if n:
        api.ava(conb, conb, raw, contig_ava_opts())
        api.paf_window_filter(3, raw, sfo, min_len=90, min_iden=0.99, min_o=2, sfo=True)      # HyLight.py:310-311
        api.sfo2overlaps(sfo, savage, num_singles=n, num_pairs=0)                              # HyLight.py:315-317
        shutil.copyfile(conb, os.path.join(sb, "fastq", "singles.fastq"))
    else:
        for p in (raw, sfo, savage):
            open(p, "w").close()
    if stageb_cmd and n:                                       # HyLight.py:320-324
        _run(f"{stageb_cmd} --no_error_correction --remove_branches true --stage b --min_overlap_len 300 "
             f"--min_overlap_perc 0 --edge_threshold 1 --overlaps ./sfoverlap.out.savage --fastq ./fastq --max_tip_len 1000 "
             f"--len_c {len_c} --num_threads {threads}", cwd=sb)
        singles = os.path.join(sb, "singles.fastq")
        with open(singles) as f, open(out_file, "w") as o:     # fastq2fasta.py
            for k, line in enumerate(f):
                if k % 4 == 0:
                    o.write(">" + line[1:])
                elif k % 4 == 1:
                    o.write(line)
    return n


def build_parser():
    p = argparse.ArgumentParser(prog="python -m hylight_amd.driver",
                                description="Haplotype-aware de novo assembly of metagenome from hybrid sequencing data "
                                            "(long reads, short reads) - MI355X long-read path",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("-s", "--short_reads", dest="short_reads", type=str, required=False)
    p.add_argument("-l", "--long_reads", dest="long_reads", type=str, required=True)
    p.add_argument("-o", "--outdir", dest="outdir", type=str, default="./")
    p.add_argument("-t", "--threads", dest="threads", type=int, default=20)
    p.add_argument("--corrected", dest="corrected", action="store_true")
    p.add_argument("--low_q", dest="low_quality", action="store_true")
    p.add_argument("--nsplit", dest="nsplit", type=int, default=60)
    p.add_argument("--min_identity", dest="min_identity", type=float, default=0.95)
    p.add_argument("--min_ovlp_len", dest="min_ovlp_len", type=int, default=3000)
    p.add_argument("--size", dest="size", default=15000, type=int)
    p.add_argument("--max_tip_len", dest="max_tip_len", type=int, default=10000)
    p.add_argument("--insert_size", dest="insert_size", default=450, type=int)
    p.add_argument("--average_read_len", dest="average_read_len", default=250, type=int)
    p.add_argument("--version", "-v", action="version", version="%(prog)s version: " + __version__)
    p.add_argument("--stop_after", choices=["overlap", "contigs1", "polish", "savage"], default=None,
                   help="(extension) stop after the named stage (savage: the contig overlap file of extend_con)")
    p.add_argument("--device", type=int, default=0, help="(extension) GPU index of a single-GPU run")
    p.add_argument("--gpus", type=int, default=1,
                   help="(extension) shard every overlap stage over this many GPUs of the node (one process per GPU)")
    p.add_argument("--short_contigs", default=None,
                   help="(extension) contigs of the short-read branch (POLYTE, run elsewhere) to merge with the long-read contigs")
    p.add_argument("--stageb_cmd", default=None,
                   help="(extension) command prefix of the reference's pipeline_per_stage.py, if installed")
    return p


def _init_rank(args, world, local):
    """This rank's GPU, the library on it, the process group of the run (RCCL)."""
    import torch
    n_dev = torch.cuda.device_count()                         # (counting devices does not initialise the GPU)
    dev = args.device if world == 1 else local % max(n_dev, 1)
    torch.cuda.set_device(dev)
    api.init(dev, args.threads)
    launch.init_process_group(dev)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and not launch.launched():
        # one process per GPU, started before this one has made any GPU call; this process only waits for them
        return launch.spawn_ranks(args.gpus, [sys.executable, "-m", "hylight_amd.driver", *argv])
    rank, world, local = launch.rank_env()
    if launch.launched() and args.gpus not in (1, world):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    _init_rank(args, world, local)
    pool = StagePool(rank, world)
    try:
        if rank != 0:
            pool.serve()
            return 0
        return _pipeline(args, pool)
    finally:
        pool.shutdown()
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def _pipeline(args, pool):
    """The run itself (rank 0).  `pool.stage` = utils.split_reads2 on every rank."""
    def stage(fa, ref, nsplit, out_file, len_over, mc, iden, long=True):
        return pool.stage(fa, ref, nsplit, out_file, len_over, mc, iden, long)

    outdir, nsplit, iden = args.outdir, args.nsplit, args.min_identity
    len_over, max_tip = args.min_ovlp_len, args.max_tip_len
    long_reads = os.path.abspath(args.long_reads)
    os.makedirs(outdir, exist_ok=True)
    tmp = os.path.join(outdir, "tmp") + "/"
    os.makedirs(tmp, exist_ok=True)

    if args.corrected:
        infile = long_reads
    else:                                                       # HyLight.py:85-112
        if not args.short_reads:
            raise SystemExit("read correction needs --short_reads (or pass --corrected)")
        short_reads = os.path.abspath(args.short_reads)
        bfc, rope, conv, fml = _tool("bfc"), _tool("ropebwt2"), _tool("fmlrc2-convert"), _tool("fmlrc2")
        cor = os.path.join(tmp, "cor_short_reads.fq")
        _run(f"{bfc} -s 3g -t{args.threads} {short_reads} 2>/dev/null | awk 'NR%4==1{{print $1; next}} {{print}}' > {cor}")
        _run(f"cat {cor} | awk 'NR % 4 == 2' | sort | tr NT TN | {rope} -LR | tr NT TN | {conv} {tmp}/comp_msbwt.npy")
        _run(f"{fml} -t 30 -m 2 comp_msbwt.npy {long_reads} fmlrc1.fasta && {fml} -t 30 -m 2 comp_msbwt.npy fmlrc1.fasta "
             f"fmlrc2.fasta && {fml} -t 30 -m 2 comp_msbwt.npy fmlrc2.fasta fmlrc3.fasta && rm fmlrc1.fasta fmlrc2.fasta", cwd=tmp)
        infile = os.path.join(tmp, "fmlrc3.fasta")

    infile = filter_non_atcg(infile, outdir, fq_or_fa(infile))  # HyLight.py:116-118
    ovl_dir = os.path.join(outdir, "2.overlap")
    shutil.rmtree(ovl_dir, ignore_errors=True)
    os.makedirs(ovl_dir)
    # main all-vs-all: len_over is hard-coded to 6000 here in the reference (HyLight.py:130)
    overlap = stage(infile, infile, nsplit, os.path.join(ovl_dir, "s1_s1.paf"), 6000, 2, iden)
    if args.stop_after == "overlap":
        return 0
    gfa, long_con = tmp + "contigs1.gfa", tmp + "contigs1.fa"
    n_arg, c_arg = (3, 3) if args.low_quality else (1, 1)       # HyLight.py:137,140
    api.miniasm(overlap, infile, gfa, bub_dist=max_tip, n_rounds_arg=n_arg, max_ext=1, min_dp=c_arg)
    gfa2fa(gfa, long_con)
    if args.stop_after == "contigs1":
        return 0

    ov_long_ref = stage(infile, long_con, nsplit, tmp + "ov_long_ref.paf", len_over, 2, iden)   # HyLight.py:149
    racon = _tool("racon")
    p1, p2 = tmp + "polish1.fa", tmp + "polish2.fa"
    _run(f"{racon} --no-trimming -u -t 30 {infile} {ov_long_ref} {long_con} > {p1}", cwd=tmp)
    ti = 0
    while ti < 2 and not args.low_quality:                      # HyLight.py:158-190
        remain = pick_up(ov_long_ref, tmp, infile)
        if not os.path.exists(remain) or os.path.getsize(remain) == 0:    # every read overlapped: nothing left to assemble
            break
        ov_remain = stage(remain, remain, nsplit, tmp + "ov_long_remain.paf", len_over, 2, iden)
        remain_gfa = tmp + "remain.gfa"
        api.miniasm(ov_remain, infile, remain_gfa, bub_dist=max_tip, n_rounds_arg=1, max_ext=1, min_dp=1)
        if os.path.getsize(remain_gfa) == 0:                    # HyLight.py:173
            break
        remain_con = tmp + "remain_con.fa"
        gfa2fa(remain_gfa, remain_con)
        ov2 = stage(infile, remain_con, nsplit, tmp + "ov_long_ref2.paf", len_over, 2, iden)
        _run(f"{racon} --no-trimming -u -t 30 {infile} {ov2} {remain_con} >> {p2}; cat {ov2} >> {ov_long_ref}", cwd=tmp)
        ti += 1
    long_con2 = tmp + "long_con_polished.fa"
    with open(long_con2, "w") as o:                             # HyLight.py:192-201
        num = 0
        for p in (p1, p2):
            if not os.path.exists(p):
                continue
            for line in open(p):
                o.write(line if num % 2 == 1 else f">longr_con_{num // 2}\n")
                num += 1
    if args.stop_after == "polish":
        return 0
    if not args.short_reads:
        sys.stderr.write("hylight-mi: long-read path finished (tmp/long_con_polished.fa); no --short_reads given\n")
        return 0
    short_reads = os.path.abspath(args.short_reads) if args.corrected else os.path.join(tmp, "cor_short_reads.fq")
    # the two short-read calls of the same path (HyLight.py:200,207: len_over 70, mc 3, short mode)
    ov_short = stage(short_reads, long_con2, nsplit, tmp + "shortr1.paf", 70, 3, iden, long=False)
    long_con3 = os.path.join(outdir, "long_con_polished.fa")
    _run(f"{racon} --no-trimming -u -t 30 {short_reads} {ov_short} {long_con2} > {long_con3}", cwd=outdir)
    remain_short = pick_up(ov_short, tmp, short_reads)
    if os.path.exists(remain_short) and os.path.getsize(remain_short):
        stage(short_reads, remain_short, nsplit, tmp + "shortr2.paf", 70, 3, iden, long=False)
    # HyLight.py:211-262 (read clustering, POLYTE per cluster) is the short-read branch: not part of this path.  Its
    # contigs can be handed in; the contig-overlap half of extend_con runs either way (HyLight.py:264-280).
    sys.stderr.write("hylight-mi: short-read clustering and POLYTE (HyLight.py:211-262) are outside this implementation"
                     + (": using --short_contigs\n" if args.short_contigs else "; continuing with the long-read contigs only\n"))
    all_con = os.path.join(outdir, "all_contigs.fa")
    with open(all_con, "w") as o:
        for p in ([args.short_contigs] if args.short_contigs else []) + [long_con3]:
            with open(p) as f:
                shutil.copyfileobj(f, o)
    final = os.path.join(outdir, "final_contigs.fa")
    n_con = extend_con(all_con, tmp, final, threads=30, stageb_cmd=None if args.stop_after == "savage" else args.stageb_cmd)
    if args.stop_after == "savage":
        return 0
    if not os.path.exists(final):
        sys.stderr.write(f"hylight-mi: {n_con} contigs, their overlaps are in tmp/stageb/sfoverlap.out.savage; the stage-b merge "
                         "(pipeline_per_stage.py / ViralQuasispecies, HyLight.py:320-324) is not built here, so "
                         f"final_contigs.fa was not written (pass --stageb_cmd to run the reference's): exit status {EXIT_NO_FINAL}\n")
        return EXIT_NO_FINAL
    return 0


if __name__ == "__main__":
    sys.exit(main())
