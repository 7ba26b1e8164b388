#!/usr/bin/env python3
"""bench.py - long-read all-vs-all overlaps/sec of the hot path on synthetic reads (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the split_reads2-equivalent stage (sketch -> index -> seed -> chain -> align ->
v4 window filter -> SNP pile-up filter -> pass 2 -> score-sorted PAF on disk) over the workload, with
the reads already resident in HBM when the timed region starts (hlmi_job_open uploads them).
Workload at N=1 = BASELINE.json configs[1] (C2): 10 000 synthetic ONT reads, mean 8 kb, 5 strains of
400 kb, --nsplit 100; the stage constants are those of the reference's main all-vs-all call
(script/HyLight.py:130: len_over=6000, mc=2, iden=0.95).  With N>1 the --nsplit target chunks are
sharded over the ranks (chunk i -> rank i % N), every rank sketches 1/N of the reads and the sketches
are all-gathered over RCCL; the read set is fixed whatever N is, so scaling is "strong".

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]
    "C2": dict(seed=20241008, n_strains=5, genome_len=400_000, n_reads=10_000, mean_len=8_000, min_len=1_000,
               max_len=40_000, snp_rate=0.01, err_sub=0.003, err_ins=0.001, err_del=0.001, nsplit=100),
    # small variant for quick checks (not a bench line)
    "C2mini": dict(seed=20241008, n_strains=5, genome_len=40_000, n_reads=1_000, mean_len=8_000, min_len=1_000,
                   max_len=40_000, snp_rate=0.01, err_sub=0.003, err_ins=0.001, err_del=0.001, nsplit=100),
    # scale / robustness probes (not bench lines): 4x the reads of C2 at the same depth; a high-divergence mix in
    # the spirit of BASELINE.json configs[4] (strain indels, 1 % / 0.5 % / 0.5 % read errors)
    "C3s": dict(seed=20241008, n_strains=5, genome_len=1_600_000, n_reads=40_000, mean_len=8_000, min_len=1_000,
                max_len=40_000, snp_rate=0.01, err_sub=0.003, err_ins=0.001, err_del=0.001, nsplit=200),
    "C5s": dict(seed=20241008, n_strains=8, genome_len=200_000, n_reads=5_000, mean_len=10_000, min_len=1_000,
                max_len=40_000, snp_rate=0.02, strain_indel_rate=0.001, err_sub=0.01, err_ins=0.005, err_del=0.005,
                nsplit=100, stage=dict(len_over=1500, mc=2, iden=0.90)),
}
STAGE = dict(len_over=6000, mc=2, iden=0.95)     # script/HyLight.py:130
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def make_workload(name, path):
    from hylight_amd import simulate as S
    w = dict(WORKLOADS[name])
    w.pop("nsplit")
    w.pop("stage", None)
    reads, _ = S.simulate_reads(**w)
    S.write_fasta(reads, path)
    return sum(len(r.seq) for r in reads)


KERNEL_OF_TIMER = {"chain": "hlmi::chain_kernel", "align_narrow": "hlmi::align_narrow_kernel", "align_wide": "hlmi::align_kernel",
                   "align_classify": "hlmi::classify_kernel", "seed_fill": "hlmi::seed_kernel<true>",
                   "seed_count": "hlmi::seed_kernel<false>", "anchor_sort": "rocprim::radix_sort_onesweep_config"}


def pmc_traffic(timer, workload):
    """HBM bytes per launch of the kernel behind `timer`, from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, see tools/summarize_pmc.py).  Those passes were run
    on the C2 workload; any other workload reports null."""
    if workload != "C2":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    name = KERNEL_OF_TIMER.get(timer, "")
    k = next((v for n, v in d.items() if name and n.startswith(name)), None)      # template arguments follow the name
    return k["hbm_bytes_per_launch"] if k else None


def pmc_valu(timer, workload):
    """What the kernel behind `timer` is really bound by (chain and banded DP are VALU work): the share
    of the SIMDs' vector-issue cycles it uses, from the committed SQ counter pass (profiles/*_sq_counters.json,
    tools/summarize_sq.py; SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)).  C2 only."""
    if workload != "C2":
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    name = KERNEL_OF_TIMER.get(timer, "")
    k = next((v for n, v in d.items() if n.startswith(name)), None)
    if not k or not k.get("GRBM_GUI_ACTIVE"):
        return None
    busy = k["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0)
    return dict(valu_issue_frac=round(busy, 4), valu_insts_per_launch=k.get("SQ_INSTS_VALU", 0.0) / max(k.get("launches", 1), 1),
                source=os.path.basename(files[-1]))


def cpu_baseline(fa, nsplit, budget_s=25.0):
    """Oracle (CPU port) timed on a bounded sample of the same workload: as many --nsplit target chunks
    (each vs ALL query reads, exactly like one reference worker) as fit the budget, one process per
    host core like the reference's `xargs -P`."""
    import multiprocessing as mp
    from oracle import filters as F
    lines = open(fa).read().split("\n")[:-1]
    # bounded sample: the first 16 target reads of every --nsplit chunk (a full chunk of ~100 reads against
    # 10 k queries is ~1 core-minute in the scalar oracle)
    ranges = [(lo, min(hi, lo + 32)) for lo, hi in F.chunk_ranges(len(lines), nsplit)]
    cores = min(len(os.sched_getaffinity(0)), 32, len(ranges))
    tmp = tempfile.mkdtemp(prefix="hl_cpu_")
    t0 = time.time()
    done, rows = 0, 0
    with mp.get_context("fork").Pool(cores) as pool:
        nxt = 0
        while nxt < len(ranges) and (done == 0 or (time.time() - t0) * (1 + cores / max(done, 1)) < budget_s):
            batch = [(fa, lines, ranges[i], os.path.join(tmp, f"c{i}")) for i in range(nxt, min(nxt + cores, len(ranges)))]
            rows += sum(pool.map(_cpu_chunk, batch))
            done += len(batch)
            nxt += len(batch)
    dt = time.time() - t0
    return dict(value=rows / dt, unit="overlaps/s", cores=cores, kind="port",
                sample=f"first 16 target reads of {done} of the {len(ranges)} --nsplit chunks x all queries, oracle overlapper "
                       f"+ oracle filters, one process per core, {dt:.1f} s wall")


def _cpu_chunk(args):
    fa, lines, (lo, hi), base = args
    from oracle import ava as OA
    from oracle import filters as F
    with open(base + ".fa", "w") as f:
        f.write("\n".join(lines[lo:hi]) + "\n")
    OA.ava(base + ".fa", fa, base + ".paf")
    raw = open(base + ".paf").read().split("\n")[:-1]
    return len(F.worker(raw, True, STAGE["len_over"], STAGE["mc"], STAGE["iden"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hylight_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU; HL_BENCH_BACKEND=gloo lets several ranks share a card (the 1-GPU rehearsal of the N > 1 flow:
    # RCCL wants one device per rank)
    backend = os.environ.get("HL_BENCH_BACKEND", "nccl")
    dev_id = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_id)
    api.init(dev_id, 0)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    wl = WORKLOADS[args.workload]
    work = os.environ.get("HL_BENCH_DIR") or tempfile.mkdtemp(prefix="hl_bench_")
    fa = os.path.join(work, f"{args.workload}.fa")
    if rank == 0 and not os.path.exists(fa):
        make_workload(args.workload, fa + ".tmp")
        os.replace(fa + ".tmp", fa)
    if world > 1:
        obj = [fa]
        dist.broadcast_object_list(obj, src=0)
        fa = obj[0]
        work = os.path.dirname(fa)          # every rank writes its part next to rank 0's files: rank 0 merges them
        dist.barrier()

    from hylight_amd.stage import StageRunner
    runner = StageRunner(fa, fa, wl["nsplit"], long_mode=True, rank=rank, world=world)
    out_paf = os.path.join(work, "out.paf")         # N > 1: ranks write out.paf.part<rank>, rank 0 merges into out.paf

    def step():
        return runner.run(out_paf, **wl.get("stage", STAGE))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.time()
    rows = 0
    stats = None
    for _ in range(args.steps):
        rows = step()
        stats = api.last_stats()
    fence()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt, float(rows)], dtype=torch.float64, device=red_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, rows = float(tmax[0]), int(t[1])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = rows / (dt / args.steps)

    # ---- roofline of the dominant kernel (HIP events on the library stream, last step) --------------
    kms = {k.split(".", 1)[1]: v for k, v in stats.items() if k.startswith("kernel_ms.")}
    kn = {k.split(".", 1)[1]: v for k, v in stats.items() if k.startswith("kernel_launches.")}
    dom = max(kms, key=kms.get)
    # algorithmic bytes per kernel for the whole step (DESIGN.md "Algorithmic bytes")
    A, P = stats.get("anchors", 0.0), stats.get("pieces", 0.0)
    AB = stats.get("anchor_bytes", 16 * A)             # 8 B per anchor when a batch packs them into one word, else 16
    M = stats.get("sketch_minimizers", 0.0)
    algo = {
        # every anchor read once, the alignment pieces (32 B) and their fixed points (8 B) written once; the DP's
        # own arrays (f, p, best child, chain id, peak, member list: 36 B per anchor) are scratch, not counted
        "chain": AB + 32 * P + 8 * stats.get("fixed_points", 0.0),
        "align_narrow": stats.get("align_bases_narrow", 0.0) + 4 * stats.get("cigar_ops", 0.0) + 32 * stats.get("align_tasks_narrow", 0.0),
        "align_wide": stats.get("align_bases_wide", 0.0) + 32 * stats.get("align_tasks_wide", 0.0),
        "align_classify": stats.get("align_bases_classify", 0.0) + (32 + 24 + 1) * stats.get("align_tasks", 0.0),
        "anchor_sort": 2 * AB,                         # one read + one write per anchor
        "seed_fill": 16 * M / 2 + 8 * A + AB,          # query minimizers, index occurrences (y), anchors out
        "seed_count": 16 * M / 2 + 4 * A,              # query minimizers, rank/frequency word of every occurrence
        "sketch_kmer_window": stats.get("sketch_bases", 0.0) * 1 + 16 * M,
    }
    launches = max(kn.get(dom, 1.0), 1.0)
    avg_ms = kms[dom] / launches
    bytes_per_launch = algo.get(dom, 0.0) / launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    roof = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                traffic=pmc_traffic(dom, args.workload), algorithmic_bytes_per_launch=bytes_per_launch,
                avg_launch_ms=avg_ms, launches_per_step=launches, valu=pmc_valu(dom, args.workload),
                kernel_ms_per_step={k: round(v, 3) for k, v in sorted(kms.items(), key=lambda kv: -kv[1])})

    line = dict(metric="long-read all-vs-all overlaps/sec", value=value, unit="overlaps/s", n_gpus=world,
                steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
                scaling="strong", vs_baseline=None, dtype="u8/int32", data="synthetic",
                config=dict(workload=f"{args.workload}: {wl['n_reads']} synthetic ONT reads, mean {wl['mean_len']} bp, "
                                     f"{wl['n_strains']} strains x {wl['genome_len']} bp, ava, --nsplit {wl['nsplit']}",
                            nsplit=wl["nsplit"], parallelism=f"chunks%{world}", **wl.get("stage", STAGE),
                            overlaps_out=rows, candidate_rows=stats.get("ava_rows"), anchors=A),
                roofline=roof,
                stage_seconds={k: stats[k] for k in ("t_ava_s", "t_filter_s", "t_format_sort_write_s", "t_total_s") if k in stats})
    # second half of the BASELINE metric: overlap-graph build seconds (PAF on disk -> GFA on disk), not part of `value`
    try:
        t_g = time.time()
        api.miniasm(out_paf, fa, os.path.join(work, "contigs1.gfa"), bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1)
        line["graph_build_s"] = round(time.time() - t_g, 4)
        line["graph_unitigs"] = sum(1 for l in open(os.path.join(work, "contigs1.gfa")) if l.startswith("S\t"))
    except Exception as e:                                    # the stage number stays valid without it
        line["graph_build_s"] = None
        line["graph_error"] = str(e)[:200]
    if os.environ.get("HL_BENCH_STATS"):
        sys.stderr.write("STATS " + json.dumps({k: round(v, 4) for k, v in sorted(stats.items())}) + "\n")
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(fa, wl["nsplit"])
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
