"""Host side of one split_reads2 stage (script/utils.py:41-71) on 1..N GPUs of a node.

One process per GPU.  PyTorch is plumbing here: it owns the device buffers that cross the C ABI
and provides torch.distributed (backend "nccl" = RCCL over xGMI) for the ONE exchange step of the
path: every rank sketches 1/N of the reads, the 16-byte minimizers are all-gathered ONCE per read
set (`prepare()`), then each rank overlaps + filters its share of the --nsplit target chunks
(chunk i -> rank i % N) against ALL reads with no further communication.  Rank 0 merges the
per-rank score-sorted PAFs (`sort -k12 -nr` of utils.py:69).

The exchange is streamed: the complete sketch is allocated once at its final size and filled in
rounds of at most SLAB entries per rank (one all_gather_into_tensor of equal slabs per round, then
one device-to-device copy per rank into place), so the transient memory is N slabs whatever the
sketch size (C4: 40 GB of minimizers).
"""
from __future__ import annotations

import os

from . import api

SLAB = 32 << 20          # minimizers per rank and round (16 B each: 512 MiB per rank slot)


class StageRunner:
    def __init__(self, reads_fa, ref_fa, nsplit, long_mode=True, rank=0, world=1, group=None, slab=SLAB,
                 force_exchange=False):
        """`force_exchange` takes the all-gather path with one rank as well (the RCCL calls of the N > 1 flow on a single
        card: tests/test_gpu_multirank.py).  The job is always api.Job, its buffers live on api.DEVICE."""
        self.rank, self.world, self.group, self.device = rank, world, group, api.DEVICE
        self.force_exchange = force_exchange
        self.slab = max(1, int(slab))
        self.job = api.Job(reads_fa, ref_fa, nsplit, long_mode)
        self._keep = None
        self.exchange_rounds = 0
        self.exchange_bytes = 0          # bytes this rank received in the last exchange (minimizers + per-read counts)

    def close(self):
        self.job.close()
        self._keep = None

    # -- sketch exchange -------------------------------------------------------------------------
    def prepare(self, force=False):
        """Sketch this rank's slice of the reads, exchange, install the complete sketch.  Once per read set:
        later passes (`run`) reuse it unless `force`."""
        if self._keep is not None and not force:
            return
        job, world, rank, dev = self.job, self.world, self.rank, self.device
        if world == 1 and hasattr(job, "sketch_own") and not self.force_exchange:
            # one rank: nothing to exchange - the job sketches into buffers of its own, sized exactly (a caller buffer needs
            # room for the bound, one 16-byte entry per base: 80 GB for C5's 5 Gbases)
            job.sketch_own()
            self._keep = ()
            return
        import torch
        nq = job.num_queries
        lo, hi = rank * nq // world, (rank + 1) * nq // world
        cap = max(job.sketch_bound(lo, hi), 1)
        mz = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        cnt = torch.zeros(max(hi - lo, 1), dtype=torch.int32, device=dev)
        if dev == "cuda":
            torch.cuda.current_stream().synchronize()      # the library works on its own stream: torch's fills first
        n = job.sketch(lo, hi, mz.data_ptr(), cap, cnt.data_ptr()) if hi > lo else 0
        if world == 1 and not self.force_exchange:
            all_mz, all_cnt, total = mz[:max(n, 1)], cnt, n
        else:
            import torch.distributed as dist
            # RCCL moves device buffers directly; a gloo group (tests: two ranks sharing one GPU) stages through the host
            staged = dev == "cuda" and dist.get_backend(self.group) == "gloo"
            xdev = "cpu" if staged else dev
            sizes = torch.tensor([n, hi - lo], dtype=torch.int64, device=xdev)
            gathered = torch.empty(world * 2, dtype=torch.int64, device=xdev)     # flat: gloo and RCCL both take it
            dist.all_gather_into_tensor(gathered, sizes, group=self.group)
            g = gathered.view(world, 2).cpu().tolist()
            n_of, q_of = [int(x[0]) for x in g], [int(x[1]) for x in g]
            total = sum(n_of)
            all_mz = torch.empty((max(total, 1), 2), dtype=torch.int64, device=dev)
            if total == 0:
                all_mz.zero_()
            base = [sum(n_of[:r]) for r in range(world)]
            slab = min(self.slab, max(max(n_of), 1))
            send = torch.zeros((slab, 2), dtype=torch.int64, device=xdev)
            recv = torch.empty(world * slab * 2, dtype=torch.int64, device=xdev)
            self.exchange_rounds = 0
            for off in range(0, max(n_of), slab):
                mine = max(0, min(slab, n - off))
                if mine:
                    send[:mine].copy_(mz[off:off + mine])
                # all-gather of equally sized slabs (RCCL: ring over xGMI, (N-1)/N of the round per link)
                dist.all_gather_into_tensor(recv, send.view(-1), group=self.group)
                rv = recv.view(world, slab, 2)
                for r in range(world):
                    k = max(0, min(slab, n_of[r] - off))
                    if k:
                        all_mz[base[r] + off:base[r] + off + k].copy_(rv[r, :k])
                self.exchange_rounds += 1
            del send, recv, mz
            max_q = max(max(q_of), 1)
            csend = torch.zeros(max_q, dtype=torch.int32, device=xdev)
            csend[:hi - lo].copy_(cnt[:hi - lo])
            crecv = torch.empty(world * max_q, dtype=torch.int32, device=xdev)
            dist.all_gather_into_tensor(crecv, csend, group=self.group)
            all_cnt = torch.zeros(max(nq, 1), dtype=torch.int32, device=dev)
            qbase = 0
            for r in range(world):
                if q_of[r]:
                    all_cnt[qbase:qbase + q_of[r]].copy_(crecv[r * max_q:r * max_q + q_of[r]])
                qbase += q_of[r]
            self.exchange_bytes = 16 * (total - n) + 4 * (nq - (hi - lo))
        if dev == "cuda":
            torch.cuda.synchronize()
        self._keep = (all_mz, all_cnt)          # the library reads these buffers during run()
        job.set_query_sketch(all_mz.data_ptr(), total, all_cnt.data_ptr())

    # -- one pass ----------------------------------------------------------------------------------
    def run(self, out_paf, len_over, mc, iden, merge=True, share=None):
        """One pass over this rank's chunks.  `share` = (i, n) restricts the pass to the i-th of n slices of
        the --nsplit chunks (chunk c belongs to slice c % n; inside a slice the ranks take turns): bench.py's
        unit of work on workloads whose full pass takes tens of seconds.  Returns the number of overlaps this
        rank wrote (rank 0 with merge: of the merged file)."""
        self.prepare()
        si, sn = share if share is not None else (0, 1)
        part = out_paf if self.world == 1 else f"{out_paf}.part{self.rank}"
        self.job.run(si * self.world + self.rank, sn * self.world, len_over, mc, iden, part)
        rows = int(self.job.rows_out()) if hasattr(self.job, "rows_out") else int(api.last_stats().get("rows_out", 0))
        if self.world > 1 and merge:
            import torch.distributed as dist
            dist.barrier(group=self.group)
            if self.rank == 0:
                parts = [f"{out_paf}.part{r}" for r in range(self.world)]
                api.merge_scored_paf(parts, out_paf)
                for p in parts:
                    os.remove(p)
        return rows


class StagePool:
    """Every split_reads2 call of a run on the N ranks of a node (the reference: `xargs -i -P threads` per call,
    script/utils.py:65).  Rank 0 drives the pipeline and calls `stage()`; the other ranks sit in `serve()` and take part
    in every stage call they are told about (the call's arguments are broadcast): each rank opens the same files, sketches
    its slice of the reads, runs its chunks (chunk i -> rank i % N), rank 0 merges.  With one rank this is a plain call."""

    def __init__(self, rank=0, world=1, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.calls = 0
        self.broken = False          # a stage call failed on this rank: the other ranks are inside collectives, not in serve()

    def _tell(self, msg):
        import torch.distributed as dist
        obj = [msg]
        dist.broadcast_object_list(obj, src=0, group=self.group)
        return obj[0]

    def _run(self, fa, ref, nsplit, out_file, len_over, mc, iden, long):
        r = StageRunner(fa, ref, nsplit, long_mode=long, rank=self.rank, world=self.world, group=self.group)
        try:
            n = r.run(out_file, len_over, mc, iden)
        except BaseException:
            self.broken = True
            raise
        finally:
            r.close()
        self.calls += 1
        return n

    def stage(self, fa, ref, nsplit, out_file, len_over, mc, iden, long=True):
        """split_reads2 on all ranks; called on rank 0 only.  Returns out_file."""
        if self.rank != 0:
            raise RuntimeError("StagePool.stage is called on rank 0; the other ranks run serve()")
        args = (os.fspath(fa), os.fspath(ref), int(nsplit), os.fspath(out_file), int(len_over), int(mc), float(iden), bool(long))
        if self.world > 1:
            self._tell(("stage",) + args)
        self._run(*args)
        return out_file

    def serve(self):
        """Ranks 1..N-1: take part in the stage calls rank 0 announces until it says "exit".  Returns the number of
        calls served."""
        while True:
            msg = self._tell(None)
            if msg[0] == "exit":
                return self.calls
            self._run(*msg[1:])

    def shutdown(self):
        """Rank 0, at the end of the run: releases the other ranks.  After a failed stage call nothing is sent (the
        other ranks are not listening): this process then exits non-zero and the launcher ends them."""
        if self.world > 1 and self.rank == 0 and not self.broken:
            self._tell(("exit",))


def split_reads2(fa, ref, nsplit, out_dir, out_file, bin=None, threads=30, len_over=3000, mc=2, iden=0.95,
                 long=False):
    """Drop-in for utils.split_reads2 (script/utils.py:41): same arguments, same return value."""
    r = StageRunner(fa, ref, nsplit, long_mode=long)
    try:
        r.run(out_file, len_over, mc, iden)
    finally:
        r.close()
    return out_file
