"""CPU, world_size 2 over gloo: the N>1 logic of hylight_amd/stage.py (sketch slices -> padded all-gather ->
reassembly, chunk i -> rank i % N, merge of the per-rank score-sorted PAFs).  The GPU job is replaced by
an oracle-backed stand-in with the same interface, so what is under test is the exchange + sharding code,
which is identical for gloo and RCCL."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleJob:
    """hylight_amd.api.Job look-alike on host memory, computing with the oracle."""

    def __init__(self, fa, nsplit):
        from oracle import filters as F
        self.fa = fa
        self.lines = open(fa).read().split("\n")[:-1]
        self.names = [l[1:] for l in self.lines[0::2]]
        self.seqs = [l.encode() for l in self.lines[1::2]]
        self.ranges = F.chunk_ranges(len(self.lines), nsplit)
        self.installed = None
        self._rows = 0

    num_queries = property(lambda s: len(s.seqs))
    num_chunks = property(lambda s: len(s.ranges))

    def sketch_bound(self, lo, hi):
        return sum(len(s) for s in self.seqs[lo:hi])

    def sketch(self, lo, hi, mz_ptr, cap, cnt_ptr):
        from oracle import ava as OA
        parts = [OA.sketch(self.seqs[i], rid=i) for i in range(lo, hi)]
        allm = np.concatenate(parts) if parts else np.zeros((0, 2), np.uint64)
        assert len(allm) <= cap
        C.memmove(mz_ptr, allm.ctypes.data, allm.nbytes)
        cnt = np.array([len(p) for p in parts], dtype=np.int32)
        C.memmove(cnt_ptr, cnt.ctypes.data, cnt.nbytes)
        return len(allm)

    def set_query_sketch(self, mz_ptr, n, cnt_ptr):
        mz = np.ctypeslib.as_array(C.cast(mz_ptr, C.POINTER(C.c_uint64)), shape=(max(n, 1), 2))[:n].copy()
        cnt = np.ctypeslib.as_array(C.cast(cnt_ptr, C.POINTER(C.c_int32)), shape=(len(self.seqs),)).copy()
        self.installed = (mz, cnt)

    def run(self, rank, world, len_over, mc, iden, out):
        from oracle import ava as OA
        from oracle import filters as F
        # the installed sketch must be the complete, read-major sketch whoever computed which slice
        mz, cnt = self.installed
        want = [OA.sketch(s, rid=i) for i, s in enumerate(self.seqs)]
        assert cnt.tolist() == [len(w) for w in want]
        assert (mz == np.concatenate(want)).all()
        rows = []
        for c, (lo, hi) in enumerate(self.ranges):
            if c % world != rank:
                continue
            cf = f"{out}.chunk{c}.fa"
            with open(cf, "w") as f:
                f.write("\n".join(self.lines[lo:hi]) + "\n")
            OA.ava(cf, self.fa, cf + ".paf")
            rows += F.worker(open(cf + ".paf").read().split("\n")[:-1], True, len_over, mc, iden)
        rows = F.sort_scored(rows)
        with open(out, "w") as f:
            f.write("".join(r + "\n" for r in rows))
        self._rows = len(rows)

    def rows_out(self):
        return self._rows

    def close(self):
        pass


def patch_product_with_oracle_job():
    """The product has no test back door (api.Job on api.DEVICE, always): the CPU tests of the exchange + sharding code
    replace those two names in THEIR process."""
    from hylight_amd import api
    api.Job = lambda reads_fa, ref_fa, nsplit, long_mode=True: OracleJob(reads_fa, nsplit)
    api.DEVICE = "cpu"


def _worker(rank, world, port, fa, out, nsplit, slab):
    import torch.distributed as dist
    from hylight_amd.stage import StageRunner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    patch_product_with_oracle_job()
    r = StageRunner(fa, fa, nsplit, rank=rank, world=world, slab=slab)
    n = r.run(out, len_over=1000, mc=2, iden=0.95)
    with open(f"{out}.rounds{rank}", "w") as f:
        f.write(str(r.exchange_rounds))
    with open(f"{out}.n{rank}", "w") as f:
        f.write(str(n))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nsplit,slab", [(2, 5, 1 << 20), (2, 5, 700), (3, 7, 257)])
def test_ranks_reproduce_the_single_process_stage(tmp_path, world, nsplit, slab):
    """slab = minimizers per rank and exchange round: the small values force the streamed exchange through many rounds
    with a ragged last one (the C4 sketch is 40 GB: SURVEY.md 8e asks for query batches)."""
    from hylight_amd import simulate as S
    from oracle import ava as OA
    from oracle import filters as F
    reads, _ = S.simulate_reads(seed=71, n_strains=2, genome_len=12000, n_reads=31, mean_len=4000, min_len=2000,
                                max_len=7000)
    fa = str(tmp_path / "s1.fa")
    S.write_fasta(reads, fa)
    out = str(tmp_path / "merged.paf")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, fa, out, nsplit, slab), nprocs=world, join=True)
    # single-process oracle stage
    lines = open(fa).read().split("\n")[:-1]
    chunks = []
    for i, (lo, hi) in enumerate(F.chunk_ranges(len(lines), nsplit)):
        cf = tmp_path / f"ref{i}.fa"
        cf.write_text("\n".join(lines[lo:hi]) + "\n")
        OA.ava(cf, fa, str(cf) + ".paf")
        chunks.append(open(str(cf) + ".paf").read().split("\n")[:-1])
    want = F.stage(chunks, True, 1000, 2, 0.95)
    got = open(out).read().split("\n")[:-1]
    assert got == want and len(want) > 10
    per_rank = [int(open(f"{out}.n{r}").read()) for r in range(world)]
    assert sum(per_rank) == len(want) and all(n > 0 for n in per_rank)
    assert not any(os.path.exists(f"{out}.part{r}") for r in range(world))
    rounds = [int(open(f"{out}.rounds{r}").read()) for r in range(world)]
    assert len(set(rounds)) == 1 and (rounds[0] > 3 if slab < 1000 else rounds[0] == 1)
