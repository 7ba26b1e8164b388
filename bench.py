#!/usr/bin/env python3
"""bench.py - long-read all-vs-all overlaps/sec of the hot path on synthetic reads (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--workload C3|C2|...]
    N > 1: bench.py starts its N rank processes itself (one per GPU, before any GPU call: hylight_amd/launch.py); under an
    external launcher (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...) it uses the ranks
    it was given.

Workload at N=1 = the configuration BASELINE.json's metric is quoted on (north_star: "synthetic 100k-long-read
ava"): C3 = configs[2], 100 000 synthetic ONT reads, mean 10 kb, 20 strains x 1 Mb, --nsplit 200, with the stage
constants of the reference's main all-vs-all call (script/HyLight.py:130: len_over=6000, mc=2, iden=0.95).  A full
pass of C3 is ~8e10 anchors / ~1e8 aligned candidate rows; one STEP is one pass of the split_reads2-equivalent
stage (index -> seed -> chain -> align -> v4 window filter -> SNP pile-up filter -> pass 2 -> score-sorted PAF on
disk) over ONE BATCH of it: 1/8 of the --nsplit target chunks per GPU (chunk c belongs to slice c % 8) against ALL
100 000 query reads - exactly the share one rank of an 8-rank job computes (hlmi_job_run(rank, 8)).  Steps walk
through the slices, so 8 steps at N=1 (one step at N=8) are one complete pass; whenever a new pass begins the
query sketch is recomputed and, with N>1, exchanged again (each rank sketches 1/N of the reads, RCCL all-gather)
inside the timed region: nothing is carried over from one pass to the next.  The reads are resident in HBM when
the timed region starts (hlmi_job_open uploads them; `value_e2e` adds parsing + upload).  Per-GPU work per step is
the same for every N, so the line reports "scaling": "weak"; `value` = overlaps written by all ranks / time.
`--workload C2` (configs[1], 10 000 reads) runs whole passes per step instead (`--slices 1`).

Prints ONE JSON line on rank 0 (fields: README.md / DESIGN.md section 7).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# slices of the --nsplit chunks a full pass is cut into (default 1).  C5: its 60 chunks one by one (slices 60..63 are empty;
# a chunk of the full 500 000-read set is ~half a minute), C4: 1000 chunks in 64 slices, C4s: the two short-read calls of
# the hybrid pipeline (script/HyLight.py:200,207) over C4's short reads, 8 slices each
SLICES = {"C3": 8, "C5": 64, "C4": 64, "C4s": 8}

KERNEL_OF_TIMER = {"chain": "hlmi::chain_kernel", "chain_small": "hlmi::chain_small_kernel", "align_narrow": "hlmi::align_narrow_pk_kernel<128, true>",
                   "align_narrow_small": "hlmi::align_narrow_pk_kernel<64, true>", "align_narrow_long": "hlmi::align_narrow_pk_kernel<256, true>",
                   "align_score_narrow": "hlmi::align_narrow_pk_kernel<128, false>", "align_score_narrow_long": "hlmi::align_narrow_pk_kernel<256, false>",
                   "align_wide": "hlmi::align_kernel<256, true>", "align_wide_short": "hlmi::align_kernel<128, true>",
                   "align_score_wide": "hlmi::align_kernel<256, false>", "align_score_wide_short": "hlmi::align_kernel<128, false>",
                   "align_long": "hlmi::align_long_kernel<true, true>", "align_score_long": "hlmi::align_long_kernel<true, false>",
                   "align_long32": "hlmi::align_long32_kernel<false, true>", "align_score_long32": "hlmi::align_long32_kernel<false, false>",
                   "align_classify": "hlmi::classify_kernel<1>", "assemble_write": "hlmi::assemble_kernel<true>",
                   "assemble_count": "hlmi::assemble_kernel<false>", "seed_fill": "hlmi::seed_kernel<true>",
                   "seed_count": "hlmi::seed_kernel<false>", "anchor_sort": "rocprim::radix_sort_onesweep",
                   "seed_group_count": "hlmi::seed_count_kernel", "seed_group_place": "hlmi::seed_place_kernel", "seed_group_scan": "hlmi::seed_scan_kernel", "seed_group_prep": "hlmi::nz_fill_kernel",
                   "filter_v4": "hlmi::window_filter_kernel", "filter_pileup_heavy": "hlmi::snp_pileup_kernel",
                   "filter_pileup_light": "hlmi::snp_pileup_light_kernel"}


def _latest_profile(suffix, workload):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload.lower()}_{suffix}.json")))
    return files[-1] if files else None


def _lib_sha():
    """sha256 (first 16 hex digits) of the library this run loaded."""
    import hashlib
    so = os.path.join(ROOT, "hylight_amd", "libhylight_mi.so")
    return hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None


def profile_provenance(workload):
    """Where `roofline.traffic` / `roofline.valu` come from: they are NOT measured by this run (PMC counters need their own
    rocprofv3 passes: tools/profile_round.sh) but read from the newest committed passes of the same workload - file names,
    and the hash of the library those passes profiled (written beside them by tools/profile_round.sh) next to the hash of the
    library running now: a mismatch says the ratio is of an older build."""
    out = {}
    for suffix in ("pmc_traffic", "sq_counters"):
        f = _latest_profile(suffix, workload)
        if f:
            sha_file = f.replace(f"_{suffix}.json", "_lib_sha.txt")
            out[suffix] = dict(file=os.path.basename(f), profiled_lib_sha=(open(sha_file).read().strip() if os.path.exists(sha_file) else None))
    out["this_lib_sha"] = _lib_sha()
    return out


def pmc_traffic(timer, workload):
    """HBM bytes per launch of the kernel behind `timer`, from the committed rocprofv3 --pmc passes of the SAME
    workload and step definition (profiles/*_<workload>_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes for gfx950, tools/summarize_pmc.py); null when no such pass is committed."""
    f = _latest_profile("pmc_traffic", workload)
    if not f:
        return None
    d = json.load(open(f))
    name = KERNEL_OF_TIMER.get(timer, "")
    k = next((v for n, v in d.items() if name and n.startswith(name)), None)      # template arguments follow the name
    return k["hbm_bytes_per_launch"] if k else None


def pmc_valu(timer, workload):
    """What the kernel behind `timer` is really bound by (chain and banded DP are VALU work): the share
    of the SIMDs' vector-issue cycles it uses, from the committed SQ counter pass (tools/summarize_sq.py: SQ_INSTS_VALU x 2
    cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), and the same over the 0.893 the calibration kernel of
    tools/valu_calib.hip reaches - profiles/r04_valu_calibration.json)."""
    f = _latest_profile("sq_counters", workload)
    if not f:
        return None
    d = json.load(open(f))
    name = KERNEL_OF_TIMER.get(timer, "")
    k = next((v for n, v in d.items() if name and n.startswith(name)), None)
    insts = (k or {}).get("SQ_INSTS_VALU") or (k or {}).get("SQ_ACTIVE_INST_VALU")
    if not k or not k.get("GRBM_GUI_ACTIVE") or not insts:
        return None
    busy = insts * 2.0 / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0)
    return dict(valu_issue_frac=round(busy, 4), valu_issue_vs_saturation=round(busy / 0.893, 4),
                valu_insts_per_launch=insts / max(k.get("launches", 1), 1), frac_wait_any=k.get("frac_wait_any"),
                source=os.path.basename(f))


def graph_large(work):
    """The metric's second half at a size that says something (the C3 step output is ~3 000 rows): the seeded layout PAF of
    tests/test_gpu_graph_scale.py - 20 000 reads, 1.44 M rows, 2.9 M overlap records - through hlmi_miniasm with HyLight's
    flags (script/HyLight.py:140: -d 10000 -n 1 -e 1 -c 1; PAF on disk -> GFA on disk), and through the reference itself
    (oracle/_ref/miniasm = tools/miniasm compiled by oracle/Makefile, main.c:117-194) on this box's host."""
    from hylight_amd import api
    from hylight_amd import simulate as S
    out = {}
    try:
        _, rows = S.layout_paf(1)
        paf = os.path.join(work, "layout.paf")
        with open(paf, "w") as f:
            f.write("\n".join(rows) + "\n")
        gfa = os.path.join(work, "layout.gfa")
        api.miniasm(paf, None, gfa, bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1)          # (first call: pools, page cache)
        ts = []
        for _ in range(3):
            t = time.time()
            api.miniasm(paf, None, gfa, bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1)
            ts.append(time.time() - t)
        st = api.last_stats()
        out.update(graph_build_large_s=round(min(ts), 4), graph_large_rows=len(rows), graph_large_overlaps=st.get("graph_overlaps"),
                   graph_large_arcs=st.get("graph_arcs"), graph_large_unitigs=sum(1 for l in open(gfa) if l.startswith("S\t")))
        ref = os.path.join(ROOT, "oracle", "_ref", "miniasm")
        if os.path.exists(ref):
            rt = []
            for _ in range(2):
                t = time.time()
                r = subprocess.run([ref, "-d", "10000", "-n", "1", "-e", "1", "-c", "1", paf], capture_output=True)
                rt.append(time.time() - t)
            same = r.returncode == 0 and r.stdout == open(gfa, "rb").read()
            out["cpu_baseline_graph"] = dict(value=round(min(rt), 4), unit="s", cores=1, kind="reference",
                                             sample="the same 1.44 M-row PAF through oracle/_ref/miniasm (tools/miniasm, -d 10000 -n 1 -e 1 -c 1)",
                                             gfa_identical=bool(same))
    except Exception as e:
        out["graph_large_error"] = str(e)[:200]
    return out


# ---- CPU baseline ------------------------------------------------------------------------------------------
def _oracle_ava(target_fa, query_fa, out, long_mode, threads):
    """One oracle overlapper run in a process of its own (OMP_NUM_THREADS = threads; the oracle's query loop is an
    OpenMP loop).  Returns seconds."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom oracle import ava as OA\n"
            "OA.ava(%r, %r, %r, OA.opts_long() if %r else OA.opts_short())\n") % (ROOT, target_fa, query_fa, out, bool(long_mode))
    t = time.time()
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
    return time.time() - t


def cpu_baseline(target_fa, query_fa, nsplit, stage, long_mode=True, budget_s=150.0, gpu_check=True):
    """The reference's CPU path timed beside the GPU number on WHOLE --nsplit chunks - the reference's own unit of work
    (script/utils.py:54-65: one worker process per chunk, its targets against ALL queries).
    (1) minimap2 on $PATH: the exact command line of script/filter_overlap_slr2.py:51 / :55 with -t <cores> -> kind
        "reference";
    (2) otherwise the oracle (this repo's CPU restatement, kind "port"): oracle overlapper with its OpenMP query loop on
        all host cores, then the oracle filter chain (the reference's filter is single-threaded Python per chunk; the
        oracle's pile-up is its numpy restatement) -> rows the chunk contributes to the stage output.
    Procedure: the middle chunk of the target file (under the pair-once rule a chunk's work grows with the name rank
    of its reads: the middle one is the mean).  Its first 8 and first 32 targets against all queries give the fit
    t = t_fixed + n_targets * t_target (t_fixed = reading + sketching the query file, which every per-chunk process of the
    reference repeats); when the fit says the whole chunk fits the budget it is RUN and the measured time is the number,
    otherwise the fit is the number and `sample` says so.  More chunks follow while the budget lasts.
    gpu_check: the GPU overlapper's rows for the same whole chunk must equal the oracle's byte for byte, and the GPU
    worker output for that chunk the oracle filter chain's (parity at the benched workload; outside the timed region)."""
    from oracle import filters as F
    cores = len(os.sched_getaffinity(0))
    tmp = tempfile.mkdtemp(prefix="hl_cpu_")
    mm2 = shutil.which("minimap2")
    t_start = time.time()
    with open(target_fa) as f:
        n_lines = sum(1 for _ in f)
    ranges = F.chunk_ranges(n_lines, nsplit)
    per_rec = 4 if open(target_fa).read(1) == "@" else 2

    def chunk_file(c, n_targets=None):
        lo, hi = ranges[c]
        if n_targets is not None:
            hi = min(hi, lo + per_rec * n_targets)
        path = os.path.join(tmp, f"sub{c:05d}" + (f".first{n_targets}" if n_targets else ""))
        with open(target_fa) as src, open(path, "w") as dst:
            for l, text in enumerate(src):
                if l >= hi:
                    break
                if l >= lo:
                    dst.write(text)
        return path, (hi - lo) // per_rec

    def run(tf, out):
        if mm2:
            cmd = ([mm2, "-N", "40", "-t", str(cores), "-L", "--eqx", "-cx", "ava-pb", "-Hk19", "-m100", "-g10000", "--max-chain-skip", "25"]
                   if long_mode else
                   [mm2, "-t", str(cores), "-c", "--sr", "-DP", "--no-long-join", "-k", "21", "-w", "11", "-s", "60", "-m", "30", "-n", "2",
                    "-A", "4", "-B", "2", "--end-bonus=100"]) + [tf, query_fa]
            t = time.time()
            with open(out, "w") as o:
                subprocess.run(cmd, stdout=o, stderr=subprocess.DEVNULL, check=True)
            return time.time() - t
        return _oracle_ava(tf, query_fa, out, long_mode, cores)

    def n_rows(path):
        with open(path) as f:
            return sum(1 for _ in f)

    mid = len(ranges) // 2
    n_full = (ranges[mid][1] - ranges[mid][0]) // per_rec
    fit = None
    if n_full > 64:
        f8, n8 = chunk_file(mid, 8)
        f32, n32 = chunk_file(mid, 32)
        t8, t32 = run(f8, f8 + ".paf"), run(f32, f32 + ".paf")
        t_target = max((t32 - t8) / (n32 - n8), 1e-9)
        t_fixed = max(t8 - n8 * t_target, 0.0)
        fit = dict(t_fixed_s=round(t_fixed, 2), t_target_s=round(t_target, 4), n_small=n8, n_large=n32,
                   rows_per_target=n_rows(f32 + ".paf") / n32,
                   whole_chunk_s_predicted=round(t_fixed + n_full * t_target, 1))
    res = dict(cores=cores, kind="reference" if mm2 else "port", unit="overlaps/s", chunk_targets=n_full, fit=fit)
    what = ("minimap2 (the command of filter_overlap_slr2.py:%d, -t %d)" % (51 if long_mode else 55, cores) if mm2 else
            f"oracle overlapper (OpenMP over the queries, {cores} threads)") + " + oracle filter chain (one thread, as the reference's per-chunk Python)"
    if fit and fit["whole_chunk_s_predicted"] > budget_s - (time.time() - t_start):
        # one whole chunk does not fit the budget: the fit is the number
        T = t_fixed + n_full * t_target
        res.update(value=None, candidate_rows_per_s=fit["rows_per_target"] * n_full / T,
                   sample=f"whole-chunk rate from the fit t = {fit['t_fixed_s']} s + n_targets x {fit['t_target_s']} s (first {n8} and first "
                          f"{n32} targets of chunk {mid} of {len(ranges)} x all queries; a whole chunk of {n_full} targets would take "
                          f"{T:.0f} s, over the budget of {budget_s:.0f} s), {what}; `value` (final overlaps/s) needs a whole chunk's "
                          "pile-up and is not estimated")
        shutil.rmtree(tmp, ignore_errors=True)
        return res
    order = [mid] + [c for k in range(1, len(ranges)) for c in (mid - k, mid + k) if 0 <= c < len(ranges)]
    done, cand, final, t_ava, t_flt = [], 0, 0, 0.0, 0.0
    checks = {}
    for c in order:
        if done and (time.time() - t_start) * (1 + 1.0 / len(done)) > budget_s:
            break
        cf, _ = chunk_file(c)
        ta = run(cf, cf + ".paf")
        rows = open(cf + ".paf").read().split("\n")[:-1]
        tf = time.time()
        kept = F.sort_scored(F.worker(rows, long_mode, stage["len_over"], stage["mc"], stage["iden"]))
        tf = time.time() - tf
        t_ava += ta; t_flt += tf; cand += len(rows); final += len(kept)
        if gpu_check and not done and not mm2:
            from hylight_amd import api
            api.ava(cf, query_fa, cf + ".gpu.paf", api.ava_opts_long() if long_mode else api.ava_opts_short())
            same = open(cf + ".gpu.paf").read().split("\n")[:-1] == rows
            api.split_reads2(query_fa, target_fa, nsplit, tmp, cf + ".gpu.w.paf", long=long_mode, rank=c, world=len(ranges), **stage)
            same_w = open(cf + ".gpu.w.paf").read().split("\n")[:-1] == kept
            checks = dict(chunk=c, candidate_rows=len(rows), gpu_rows_identical=same, final_rows=len(kept), gpu_worker_identical=same_w)
            # the stage's own constants keep next to nothing at pooled depth: the same rows once more through the filter chain
            # with -thre wide open and -len 1000, where the kept set is large and depends on every step before the rate test
            # (tests/test_gpu_workloads_oracle.py sweeps -thre across the pair-count distribution)
            wide = F.worker(rows, long_mode, 1000 if long_mode else stage["len_over"], stage["mc"], stage["iden"], threshold=1.0)
            api.filter_chunk(cf + ".paf", cf + ".gpu.wide.paf", 1000 if long_mode else stage["len_over"], stage["mc"], stage["iden"], thre=1.0,
                             long_mode=long_mode)
            same_wide = open(cf + ".gpu.wide.paf").read().split("\n")[:-1] == wide
            checks.update(rows_kept_thre_1=len(wide), gpu_filter_identical_thre_1=same_wide)
            same_w = same_w and same_wide
            if not (same and same_w):
                sys.stderr.write(f"PARITY FAILURE at the benched workload: chunk {c}: {checks}\n")
        done.append(c)
    # the reference runs `threads` chunk workers side by side (xargs -P, utils.py:65): the filter's single thread per chunk
    # costs t_flt / cores of the machine
    T = t_ava + t_flt / cores
    res.update(value=final / T, candidate_rows_per_s=cand / t_ava, chunks=done, seconds=dict(overlapper=round(t_ava, 2), filter_one_thread=round(t_flt, 2)),
               parity=checks or None,
               sample=f"{len(done)} WHOLE --nsplit chunk(s) (chunk {done[0]} of {len(ranges)} first: {n_full} targets x all queries), {what}; "
                      f"value = final overlaps of these chunks / (overlapper wall on {cores} cores + filter seconds / {cores}: the reference "
                      "runs one single-threaded filter per chunk, `threads` chunks side by side)")
    shutil.rmtree(tmp, ignore_errors=True)
    return res


# ---- algorithmic bytes of the whole stage from the counts of one step -------------------------------
def stage_bytes(st):
    B_t, B_q = st.get("bases_t", 0.0), st.get("bases_q", 0.0)
    M_t, M_q = st.get("index_entries", 0.0), st.get("minimizers_q", 0.0)
    A, E, P = st.get("anchors", 0.0), st.get("cigar_ops", 0.0), st.get("ava_rows", 0.0)
    L = st.get("align_dp_bases", 0.0)                # sum of (Lq + Lt) over the alignment tasks
    X, P2, Pout = st.get("snp_events", 0.0), st.get("rows_after_v4", 0.0), st.get("rows_out", 0.0)
    ava = (B_t + B_q) + 16 * (M_t + M_q) + 32 * M_t + 16 * M_q + 8 * A + 16 * A + 32 * A + 16 * A + L + 4 * E + 64 * P
    flt = 64 * P * 4 + 4 * E + 16 * X * 3 + 8 * 2 * 2 * P2 + 16 * X + 64 * P + 64 * Pout
    return ava, flt


def main():
    from hylight_amd import workloads as W
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", choices=sorted(W.CONFIGS) + ["C4s"])
    ap.add_argument("--scale", type=float, default=None,
                    help="shrink reads and genomes together (quick checks; not a bench line).  Default 1, C4s: 0.1")
    ap.add_argument("--slices", type=int, default=0, help="slices of the --nsplit chunks per pass (0: workload default)")
    ap.add_argument("--slice0", type=int, default=0, help="slice the first step takes (steps walk on from there; under the pair-once "
                    "rule a chunk's work grows with its index: the middle of the file is the mean)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    from hylight_amd import launch
    if args.gpus > 1 and not launch.launched():
        # one process per GPU, started here before anything has touched the GPU (this process never does); a rank that
        # fails ends the run with its status
        sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__), *sys.argv[1:]]))

    import torch
    import torch.distributed as dist
    from hylight_amd import api

    rank, world, local = launch.rank_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU; HL_BACKEND=gloo (HL_BENCH_BACKEND: older name) lets several ranks share a card (the 1-GPU
    # rehearsal of the N > 1 flow: RCCL wants one device per rank)
    backend = os.environ.get("HL_BACKEND") or os.environ.get("HL_BENCH_BACKEND") or "nccl"
    n_dev = torch.cuda.device_count()                   # (counting does not initialise the GPU)
    if backend == "nccl" and world > max(n_dev, 1):
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, this node shows {n_dev}")
    dev_id = local % max(n_dev, 1)
    torch.cuda.set_device(dev_id)
    api.init(dev_id, 0)
    launch.init_process_group(dev_id, backend)
    rccl_ranks = dist.get_world_size() if world > 1 else 1
    red_dev = "cuda" if backend == "nccl" else "cpu"

    short_calls = args.workload == "C4s"
    if args.scale is None:
        args.scale = 0.1 if short_calls else 1.0
    cfg = W.config("C4" if short_calls else args.workload, args.scale)
    slices = args.slices or SLICES.get(args.workload, 1)
    if slices % world and slices > 1:
        raise SystemExit(f"--slices {slices} must be a multiple of the rank count {world}")
    shares = max(1, slices // world)                 # steps per complete pass
    work = os.environ.get("HL_BENCH_DIR") or tempfile.mkdtemp(prefix="hl_bench_")
    fa = os.path.join(work, f"{cfg['name']}.fa")
    short_fa, con_fa, remain_fa = (os.path.join(work, f"{cfg['name']}.{x}.fa") for x in ("short", "contigs", "remain"))
    if rank == 0 and not os.path.exists(fa):
        _, _, strains = W.make_long(cfg, fa)
        if short_calls:
            from hylight_amd import simulate as S
            W.make_short(cfg, strains, short_fa)
            # "polished long contigs" (HyLight.py:200): 40 kb pieces of every strain; "remaining short reads" (:207): the
            # reads pick_up leaves over - here every fifth pair
            contigs = [S.Read(f"longr_con_{k}_{a}", g[a:a + 40_000].copy(), None, k, a, a + 40_000, False)
                       for k, g in enumerate(strains) for a in range(0, len(g) - 10_000, 40_000)]
            S.write_fasta(contigs, con_fa)
            with open(short_fa) as f, open(remain_fa, "w") as o:
                for i, text in enumerate(f):
                    if (i // 4) % 5 == 0:
                        o.write(text)
    if world > 1:
        obj = [fa]
        dist.broadcast_object_list(obj, src=0)
        fa = obj[0]
        work = os.path.dirname(fa)          # every rank writes its part next to rank 0's files: rank 0 merges them
        short_fa, con_fa, remain_fa = (os.path.join(work, f"{cfg['name']}.{x}.fa") for x in ("short", "contigs", "remain"))
        dist.barrier()

    from hylight_amd.stage import StageRunner
    t_open = time.time()
    if short_calls:
        runners = [(StageRunner(short_fa, con_fa, cfg["nsplit"], long_mode=False, rank=rank, world=world), cfg["stage_short"]),
                   (StageRunner(short_fa, remain_fa, cfg["nsplit"], long_mode=False, rank=rank, world=world), cfg["stage_short"])]
    else:
        runners = [(StageRunner(fa, fa, cfg["nsplit"], long_mode=True, rank=rank, world=world), cfg["stage"])]
    runner = runners[0][0]
    torch.cuda.synchronize()
    t_open = time.time() - t_open           # FASTA parse + name ranks + host-to-device upload of the reads
    out_paf = os.path.join(work, "out.paf")         # N > 1: ranks write out.paf.part<rank>, rank 0 merges into out.paf
    stage = runners[0][1]
    t_prepare = []
    step_stats = {}

    def slice_paf(k):                       # every slice of a pass keeps its own file: together they are the stage output
        return out_paf if shares == 1 else f"{out_paf}.slice{k}"

    def step(i):
        new_pass = i % shares == 0 or i == 0
        i += args.slice0
        if new_pass:                        # a new pass over the read set: sketch + exchange are part of it
            t = time.time()
            for rn, _ in runners:
                rn.prepare(force=True)
            t_prepare.append(time.time() - t)
        n = 0
        step_stats.clear()
        for k, (rn, st) in enumerate(runners):
            n += rn.run(slice_paf(i % shares) + (f".call{k}" if k else ""), share=(i % shares, shares), **st)
            for key, v in api.last_stats().items():          # (C4s: the counts and kernel times of both calls of the step)
                step_stats[key] = step_stats.get(key, 0.0) + v
        return n

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    t_prepare.clear()
    fence()
    t0 = time.time()
    rows = 0
    stats = None
    step_rows, step_s = [], []
    share_load = {}             # slice of the pass -> (anchors, seconds): the 8 shares an 8-rank job hands out, seen one after the other
    for i in range(args.steps):
        ts = time.time()
        r = step(args.warmup + i)
        step_s.append(time.time() - ts)
        step_rows.append(r)
        rows += r
        stats = dict(step_stats)
        share_load[(args.warmup + i) % shares if shares > 1 else 0] = (stats.get("anchors", 0.0), step_s[-1])
    fence()
    dt = time.time() - t0
    # Two query batches in flight (lanes) share the card: a kernel's HIP-event time in the timed steps includes
    # the waves of the other lane's kernels.  One more step of the last slice, untimed and with one lane, gives every kernel's
    # duration by itself - the figure earlier rounds reported and a serial rocprofv3 trace shows.
    stats_one_lane = None
    if stats and stats.get("ava_lanes", 0) > len(runners) and not os.environ.get("HLMI_LANES"):
        os.environ["HLMI_LANES"] = "1"
        try:
            step(args.warmup + args.steps - 1)
            stats_one_lane = dict(step_stats)
        finally:
            del os.environ["HLMI_LANES"]
        fence()
    if world > 1:
        # every rank counts the rows of its own chunks
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        tot = torch.tensor([float(rows)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rows = int(tot[0])
    # the one exchange step of the path, per rank: seconds of sketch + all-gather per pass start, bytes received, rounds
    mine = dict(rank=rank, sketch_exchange_s=[round(x, 4) for x in t_prepare], bytes_received=int(getattr(runner, "exchange_bytes", 0)),
                rounds=int(getattr(runner, "exchange_rounds", 0)),
                # load balance without an 8-GPU node: anchors and seconds of every share of the pass this rank walked through
                shares_seen={str(k): dict(anchors=int(a), step_s=round(t, 4)) for k, (a, t) in sorted(share_load.items())})
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    value = rows / dt

    # ---- roofline of the dominant kernel (HIP events on the library stream, last step) --------------
    kms = {k.split(".", 1)[1]: v for k, v in stats.items() if k.startswith("kernel_ms.")}
    kn = {k.split(".", 1)[1]: v for k, v in stats.items() if k.startswith("kernel_launches.")}
    kms1 = ({k.split(".", 1)[1]: v for k, v in stats_one_lane.items() if k.startswith("kernel_ms.")} if stats_one_lane else None)
    # the dominant kernel by its own duration (with lanes a timer that spans several launches and host gaps - the sort - also
    # counts what ran beside it)
    dom = max(kms1, key=kms1.get) if kms1 else max(kms, key=kms.get)
    # algorithmic bytes per kernel for the whole step (DESIGN.md "Algorithmic bytes")
    A, P = stats.get("anchors", 0.0), stats.get("pieces", 0.0)
    AB = stats.get("anchor_bytes", 16 * A)             # 8 B per anchor when a batch packs them into one word, else 16
    M = stats.get("minimizers_q", 0.0)
    E, NT = stats.get("cigar_ops", 0.0), stats.get("align_tasks", 0.0)
    rows_in, rows_v4 = stats.get("ava_rows", 0.0), stats.get("rows_after_v4", 0.0)
    algo = {
        # every anchor read once, the alignment pieces (32 B) and their fixed points (8 B) written once; the DP's
        # own arrays are scratch, not counted
        # (groups of at most 32 anchors go through chain_small_kernel: the bytes are split by the anchors of the two kinds)
        "chain": (AB + 32 * P + 8 * stats.get("fixed_points", 0.0)) * (1.0 - (stats.get("anchors_small_groups", 0.0) / A if A else 0.0)),
        "chain_small": (AB + 32 * P + 8 * stats.get("fixed_points", 0.0)) * (stats.get("anchors_small_groups", 0.0) / A if A else 0.0),
        "align_classify": stats.get("align_bases_classify", 0.0) + (32 + 24 + 1) * NT,
        "anchor_sort": 2 * AB,                         # one read + one write per anchor
        "seed_fill": 16 * M + 8 * A + AB,              # query minimizers, index occurrences (y), anchors out
        "seed_count": 16 * M + 4 * A,                  # query minimizers, rank/frequency word of every occurrence
        # anchors grouped by (query, target, strand) without a sort (seed_group.hip): records of the minimizers + every 4-byte
        # index entry once per kernel, every anchor out once (a long query's pieces read its entries again: traffic, not algorithm)
        "seed_group_count": 16 * M + 4 * stats.get("anchors_grouped_in_lds", 0.0),
        "seed_group_place": 16 * M + 4 * stats.get("anchors_grouped_in_lds", 0.0) + 8 * stats.get("anchors_grouped_in_lds", 0.0),
        "seed_group_scan": 12 * 1024 * 2 * stats.get("seed_group_pieces", 0.0),      # the pieces' tables, read and written back
        "seed_group_prep": (16 + 8 + 16) * M,          # minimizers + counts and runs in, window records out
        # task results (24 B) and their runs in, merged CIGAR ops + 64-byte rows out
        "assemble_write": 24 * NT + 4 * E + 4 * E + 64 * rows_in,
        "assemble_count": 24 * NT,
        "filter_v4": 64 * rows_in + rows_in,           # rows in, one flag out
        # two walks over the CIGARs of the selected rows (their share of all ops), rows + references
        "filter_pileup_heavy": 8 * E * (rows_v4 / rows_in if rows_in else 0.0) + 72 * rows_v4,
        "filter_pileup_light": 8 * E * (rows_v4 / rows_in if rows_in else 0.0) + 72 * rows_v4,
    }
    for key, v in stats.items():                       # every DP launch: its tasks' bases once + task record (32 B) + result (24 B)
        if key.startswith("align_bases."):
            t = key.split(".", 1)[1]
            algo[t] = v + 56 * stats.get("align_n." + t, 0.0)
    launches = max(kn.get(dom, 1.0), 1.0)
    avg_ms = kms[dom] / launches
    bytes_per_launch = algo.get(dom, 0.0) / launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    b_ava, b_flt = stage_bytes(stats)
    last_s = step_s[-1]
    roof = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                traffic=pmc_traffic(dom, args.workload), algorithmic_bytes_per_launch=bytes_per_launch,
                avg_launch_ms=avg_ms, launches_per_step=launches, valu=pmc_valu(dom, args.workload),
                profile_source=profile_provenance(args.workload),
                # whole stage: SURVEY.md 8d's bytes_ava + bytes_filter evaluated on the counts of the last step, over
                # that step's wall time
                stage=dict(bytes_ava=b_ava, bytes_filter=b_flt, step_s=last_s,
                           achieved=(b_ava + b_flt) / last_s / 1e9, frac=(b_ava + b_flt) / last_s / 1e9 / HBM_PEAK_GBS),
                kernel_ms_per_step={k: round(v, 3) for k, v in sorted(kms.items(), key=lambda kv: -kv[1])})
    roof["lanes"] = int(stats.get("ava_lanes", len(runners)) / len(runners))       # (the counts of a step are summed over its calls)
    if kms1:
        l1 = max(stats_one_lane.get("kernel_launches." + dom, 1.0), 1.0)
        a1 = (algo.get(dom, 0.0) / l1) / (kms1[dom] / l1 * 1e-3) / 1e9 if kms1.get(dom, 0) > 0 else 0.0
        roof["one_lane"] = dict(note="one untimed step of the last slice with HLMI_LANES=1: the kernel alone on the card; `achieved` above is over "
                                     "the timed steps, where the other lane's kernels run beside it",
                                avg_launch_ms=kms1[dom] / l1, achieved=a1, frac=a1 / HBM_PEAK_GBS,
                                kernel_ms_per_step={k: round(v, 3) for k, v in sorted(kms1.items(), key=lambda kv: -kv[1])})

    counts = {k: stats.get(k) for k in ("queries", "targets", "chunks_run", "bases_q", "bases_t", "minimizers_q", "index_entries",
                                        "anchors", "anchor_bytes", "anchors_grouped_in_lds", "seed_group_gave_up", "seed_group_pieces", "seed_group_query_table_full", "seed_group_piece_table_full", "chain_groups", "pieces", "fixed_points", "align_tasks",
                                        "align_tasks_dp", "align_tasks_fast", "align_dp_bases", "cigar_ops", "ava_rows",
                                        "rows_after_v4", "snp_events", "pairs", "rows_out")}
    pass_steps = shares
    pass_s = (sum(step_s[-pass_steps:]) / min(pass_steps, len(step_s))) * pass_steps
    pass_rows = rows / args.steps * pass_steps          # all ranks, per complete pass
    line = dict(metric="long-read all-vs-all overlaps/sec", value=value, unit="overlaps/s", n_gpus=world,
                rccl_ranks=rccl_ranks, backend=(backend if world > 1 else None),
                launcher=(os.environ.get("HL_LAUNCHER", "external") if world > 1 else None),
                steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
                scaling="weak" if slices > 1 else "strong", vs_baseline=None, dtype="u8/int32", data="synthetic",
                config=dict(workload=W.describe(cfg) + (" - the two short-read calls of script/HyLight.py:200,207 (short mode: reads vs 40 kb "
                                                            "contig pieces, reads vs every fifth pair)" if short_calls else ""),
                            nsplit=cfg["nsplit"],
                            step=(f"slice {slices} of the --nsplit chunks per pass; one step = {world} slice(s) "
                                  f"(chunk c: slice c % {slices}) x all queries; {shares} steps = one full pass"
                                  if slices > 1 else "one full pass"),
                            parallelism=f"chunks%{slices if slices > 1 else world}", **stage, overlaps_out=rows,
                            candidate_rows_per_s=(stats.get("ava_rows", 0.0) * world) / step_s[-1],
                            last_step=counts),
                roofline=roof,
                # the drop-in call's view: FASTA parsing + upload of the reads + sketch/exchange + one full pass
                value_e2e=pass_rows / (t_open + pass_s) if pass_s > 0 else None,
                step_ms=[round(1e3 * x, 1) for x in step_s],
                e2e=dict(open_parse_upload_s=round(t_open, 3), pass_s=round(pass_s, 3), pass_overlaps=int(pass_rows),
                         sketch_exchange_s=[round(x, 4) for x in t_prepare]),
                exchange=dict(world=(dist.get_world_size() if world > 1 else 1), backend=(backend if world > 1 else None), per_rank=per_rank),
                stage_seconds={k: stats[k] for k in ("t_ava_s", "t_filter_s", "t_rows_to_text_s", "t_rows_download_s", "t_final_sort_s", "t_format_sort_write_s", "t_total_s") if k in stats})
    # second half of the BASELINE metric: overlap-graph build seconds (PAF on disk -> GFA on disk), not part of `value`
    if not args.no_graph and not short_calls:
        try:
            if shares > 1:                  # the pass's output = the merge of its slices (utils.py:69)
                have = [slice_paf(k) for k in range(shares) if os.path.exists(slice_paf(k))]
                api.merge_scored_paf(have, out_paf)
                line["graph_input"] = f"{len(have)} of {shares} slices of the pass"
            t_g = time.time()
            api.miniasm(out_paf, fa, os.path.join(work, "contigs1.gfa"), bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1)
            line["graph_build_s"] = round(time.time() - t_g, 4)
            line["graph_rows_in"] = sum(1 for _ in open(out_paf))
            line["graph_unitigs"] = sum(1 for l in open(os.path.join(work, "contigs1.gfa")) if l.startswith("S\t"))
        except Exception as e:                                    # the stage number stays valid without it
            line["graph_build_s"] = None
            line["graph_error"] = str(e)[:200]
    if not args.no_graph and not short_calls and rank == 0:
        line.update(graph_large(work))
    if os.environ.get("HL_BENCH_STATS"):
        sys.stderr.write("STATS " + json.dumps({k: round(v, 4) for k, v in sorted(stats.items())}) + "\n")
    if not args.no_cpu_baseline and world == 1:
        budget = float(os.environ.get("HL_CPU_BUDGET_S", "150"))
        if short_calls:      # the first of the two calls: short reads (queries) against the contig pieces (chunked targets)
            line["cpu_baseline"] = cpu_baseline(con_fa, short_fa, cfg["nsplit"], cfg["stage_short"], long_mode=False, budget_s=budget)
        else:
            line["cpu_baseline"] = cpu_baseline(fa, fa, cfg["nsplit"], cfg["stage"], long_mode=True, budget_s=budget)
    print(json.dumps(line), flush=True)
    for rn, _ in runners:
        rn.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
