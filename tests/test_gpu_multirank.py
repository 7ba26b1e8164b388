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


def test_driver_with_two_self_started_ranks_matches_the_single_gpu_run(tmp_path):
    """`python -m hylight_amd.driver --gpus 2`: the process starts its two ranks itself (hylight_amd/launch.py), rank 0
    runs the pipeline, rank 1 serves the stage calls; both share this box's one card, so the exchange goes over gloo
    (HL_BACKEND).  The output tree up to contigs1.fa must equal the --gpus 1 run byte for byte."""
    import subprocess
    from hylight_amd import simulate as S
    reads, _ = S.simulate_reads(seed=81, n_strains=2, genome_len=40000, n_reads=110, mean_len=9000, min_len=7000,
                                max_len=14000)
    fq = tmp_path / "long.fq"
    S.write_fastq(reads, fq)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    outs = {}
    for n in (1, 2):
        out = tmp_path / f"OUT{n}"
        r = subprocess.run([sys.executable, "-m", "hylight_amd.driver", "-l", str(fq), "-o", str(out), "--corrected", "--nsplit", "3",
                            "-t", "4", "--stop_after", "contigs1", "--gpus", str(n)], env=dict(env, HL_BACKEND="gloo"),
                           capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[n] = out
    for rel in ("1.split_fastx/s1.fa", "2.overlap/s1_s1.paf", "tmp/contigs1.gfa", "tmp/contigs1.fa"):
        a, b = (outs[1] / rel).read_bytes(), (outs[2] / rel).read_bytes()
        assert a == b and len(a) > 0, rel
    assert (outs[2] / "2.overlap" / "s1_s1.paf").read_text().count("\n") > 50


def test_bench_starts_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: two ranks on this one card over gloo, a tenth of C2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C2",
                        "--scale", "0.1", "--no-cpu-baseline"], env=dict(env, HL_BACKEND="gloo", HL_BENCH_DIR=str(tmp_path)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["launcher"] == "self" and line["backend"] == "gloo"
    assert line["value"] > 0 and line["config"]["overlaps_out"] > 0


def _rccl_rank(rank, world, port, fa, out):
    import torch
    import torch.distributed as dist
    from hylight_amd import api, launch
    from hylight_amd.stage import StageRunner
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    api.init(0, 0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=rank, world_size=world)
    assert dist.get_backend() == "nccl"
    # the collectives bench.py and the driver issue besides the sketch exchange
    obj = [fa]
    dist.broadcast_object_list(obj, src=0)
    dist.barrier()
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == 1.5
    r = StageRunner(obj[0], obj[0], 12, long_mode=True, rank=rank, world=world, slab=5000, force_exchange=True)
    r.run(out, **STAGE)
    assert r.exchange_rounds > 3                  # the streamed all-gather went through several ragged rounds
    r.close()
    dist.barrier()
    dist.destroy_process_group()


def test_the_rccl_calls_of_the_exchange_on_one_rank(tmp_path):
    """The N > 1 tests above carry the exchange over gloo (two ranks on one card; RCCL wants a device per rank).  This
    one runs the same code path - counts all-gather, slab rounds of all_gather_into_tensor on device buffers, counts
    again, install - over the real backend ("nccl" = RCCL) with a communicator of ONE rank, plus the other collectives
    of bench.py / the driver: dtype, shape, device and stream handling are what an 8-GPU run will meet."""
    from hylight_amd import api
    from hylight_amd import simulate as S
    reads, _ = S.simulate_reads(seed=78, n_strains=3, genome_len=30_000, n_reads=200, mean_len=6000, min_len=1500,
                                max_len=20_000)
    fa = tmp_path / "r.fa"
    S.write_fasta(reads, fa)
    one = tmp_path / "one.paf"
    api.split_reads2(fa, fa, 12, tmp_path, one, long=True, **STAGE)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "rccl.paf"
    mp.spawn(_rccl_rank, args=(1, port, str(fa), str(out)), nprocs=1, join=True)
    assert open(out).read() == open(one).read() and os.path.getsize(one) > 0
