"""GPU: the real N > 1 path end to end on ONE card - two ranks (processes) share the GPU, gloo carries the sketch
exchange through the host, everything else is the product code: every rank sketches its half of the reads with the
HIP kernels, the halves are all-gathered and installed, each rank overlaps + filters its chunks, rank 0 merges.
The merged file must equal the single-rank result byte for byte."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

STAGE = dict(len_over=3000, mc=2, iden=0.95)


def _worker(rank, world, port, fa, out):
    import torch
    import torch.distributed as dist
    from hylight_amd import api
    from hylight_amd.stage import StageRunner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    api.init(0, 0)
    r = StageRunner(fa, fa, 12, long_mode=True, rank=rank, world=world)
    r.run(out, **STAGE)
    r.run(out, **STAGE)                      # a second pass reuses the job (bench.py does)
    r.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_rank_result(tmp_path):
    from hylight_amd import api
    from hylight_amd import simulate as S
    reads, _ = S.simulate_reads(seed=77, n_strains=3, genome_len=30_000, n_reads=240, mean_len=6000, min_len=1500,
                                max_len=20_000)
    fa = tmp_path / "r.fa"
    S.write_fasta(reads, fa)
    one = tmp_path / "one.paf"
    api.split_reads2(fa, fa, 12, tmp_path, one, long=True, **STAGE)
    assert os.path.getsize(one) > 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = tmp_path / "two.paf"
    mp.spawn(_worker, args=(2, port, str(fa), str(two)), nprocs=2, join=True)
    assert open(two).read() == open(one).read()
