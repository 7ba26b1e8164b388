"""Host side of one split_reads2 stage (script/utils.py:41-71) on 1..N GPUs of a node.

One process per GPU.  PyTorch is plumbing here: it owns the device buffers that cross the C ABI
and provides torch.distributed (backend "nccl" = RCCL over xGMI) for the ONE exchange step of the
path: every rank sketches 1/N of the reads, the 16-byte minimizers are all-gathered, then each
rank overlaps + filters its share of the --nsplit target chunks (chunk i -> rank i % N) against
ALL reads with no further communication.  Rank 0 merges the per-rank score-sorted PAFs
(`sort -k12 -nr` of utils.py:69).
"""
from __future__ import annotations

import os

from . import api


class StageRunner:
    def __init__(self, reads_fa, ref_fa, nsplit, long_mode=True, rank=0, world=1, group=None, job=None,
                 device="cuda"):
        """`job` / `device` exist for the CPU (gloo) tests of the exchange logic: the product always uses
        api.Job on "cuda"."""
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.job = job if job is not None else api.Job(reads_fa, ref_fa, nsplit, long_mode)
        self._keep = None

    def close(self):
        self.job.close()

    # -- sketch exchange -------------------------------------------------------------------------
    def _install_sketch(self):
        import torch
        job, world, rank, dev = self.job, self.world, self.rank, self.device
        nq = job.num_queries
        lo, hi = rank * nq // world, (rank + 1) * nq // world
        cap = max(job.sketch_bound(lo, hi), 1)
        mz = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        cnt = torch.zeros(max(hi - lo, 1), dtype=torch.int32, device=dev)
        n = job.sketch(lo, hi, mz.data_ptr(), cap, cnt.data_ptr()) if hi > lo else 0
        if world == 1:
            all_mz, all_cnt, total = mz[:max(n, 1)], cnt, n
        else:
            import torch.distributed as dist
            # RCCL moves device buffers directly; a gloo group (tests: two ranks sharing one GPU) stages through the host
            xdev = "cpu" if dev == "cuda" and dist.get_backend(self.group) == "gloo" else dev
            sizes = torch.tensor([n, hi - lo], dtype=torch.int64, device=xdev)
            gathered = torch.empty(world * 2, dtype=torch.int64, device=xdev)     # flat: gloo and RCCL both take it
            dist.all_gather_into_tensor(gathered, sizes, group=self.group)
            g = gathered.view(world, 2).cpu().tolist()
            max_n, max_q = max(max(x[0] for x in g), 1), max(max(x[1] for x in g), 1)
            # all-gather of equally sized slabs (RCCL: ring over xGMI, (N-1)/N of the sketch per link)
            send = torch.zeros((max_n, 2), dtype=torch.int64, device=xdev)
            send[:n] = mz[:n].to(xdev)
            recv = torch.empty(world * max_n * 2, dtype=torch.int64, device=xdev)
            dist.all_gather_into_tensor(recv, send.view(-1), group=self.group)
            recv = recv.view(world * max_n, 2)
            csend = torch.zeros(max_q, dtype=torch.int32, device=xdev)
            csend[:hi - lo] = cnt[:hi - lo].to(xdev)
            crecv = torch.empty(world * max_q, dtype=torch.int32, device=xdev)
            dist.all_gather_into_tensor(crecv, csend, group=self.group)
            all_mz = torch.cat([recv[r * max_n:r * max_n + g[r][0]] for r in range(world)]).contiguous().to(dev)
            all_cnt = torch.cat([crecv[r * max_q:r * max_q + g[r][1]] for r in range(world)]).contiguous().to(dev)
            total = int(sum(x[0] for x in g))
            if all_mz.shape[0] == 0:
                all_mz = torch.zeros((1, 2), dtype=torch.int64, device=dev)
        if dev == "cuda":
            torch.cuda.synchronize()
        self._keep = (all_mz, all_cnt)          # the library reads these buffers during run()
        job.set_query_sketch(all_mz.data_ptr(), total, all_cnt.data_ptr())

    # -- one pass ----------------------------------------------------------------------------------
    def run(self, out_paf, len_over, mc, iden, merge=True):
        """Returns the number of overlaps this rank wrote (rank 0 with merge: of the merged file)."""
        self._install_sketch()
        part = out_paf if self.world == 1 else f"{out_paf}.part{self.rank}"
        self.job.run(self.rank, self.world, len_over, mc, iden, part)
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


def split_reads2(fa, ref, nsplit, out_dir, out_file, bin=None, threads=30, len_over=3000, mc=2, iden=0.95,
                 long=False):
    """Drop-in for utils.split_reads2 (script/utils.py:41): same arguments, same return value."""
    r = StageRunner(fa, ref, nsplit, long_mode=long)
    try:
        r.run(out_file, len_over, mc, iden)
    finally:
        r.close()
    return out_file
