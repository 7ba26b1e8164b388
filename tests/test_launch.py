"""CPU: the product's own process launcher (hylight_amd/launch.py) and the N > 1 control flow of the driver counterpart.
The reference fans every stage out by itself (`xargs -i -P threads`, script/utils.py:65); here `bench.py --gpus N` and
`python -m hylight_amd.driver --gpus N` start their rank processes themselves."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hylight_amd import launch  # noqa: E402


def test_spawn_ranks_sets_the_rank_environment(tmp_path):
    code = ("import os, json; r = os.environ['RANK']\n"
            f"json.dump({{k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', "
            f"'HL_LAUNCHER', 'HSA_ENABLE_IPC_MODE_LEGACY')}}, open(r'{tmp_path}/env' + r + '.json', 'w'))\n")
    assert launch.spawn_ranks(3, [sys.executable, "-c", code]) == 0
    envs = [json.load(open(tmp_path / f"env{r}.json")) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and envs[0]["HL_LAUNCHER"] == "self"
    # runtime switches are inherited from the caller's environment, never invented by the launcher
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") for e in envs)


def test_a_failing_rank_ends_the_run_with_its_status():
    """Rank 1 fails at once, rank 0 would wait for a minute (as it would inside a collective): the launcher ends it and
    reports rank 1's status."""
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)\n"
    t0 = time.time()
    assert launch.spawn_ranks(2, [sys.executable, "-c", code]) == 7
    assert time.time() - t0 < 20


def test_bench_starts_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must get as far as its rank processes (here they stop at
    hlmi_init: no GPU in this container) and hand their failure on as a non-zero exit - it must not refuse to start."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "C2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "[launch] rank" in r.stderr and "WORLD_SIZE=1" not in r.stderr


def _driver_rank(rank, world, port, argv, done_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from hylight_amd import driver
    from hylight_amd import launch as L
    from test_multirank_gloo import patch_product_with_oracle_job
    patch_product_with_oracle_job()
    driver._init_rank = lambda args, world, local: L.init_process_group(backend="gloo")      # no GPU here: gloo ranks on the CPU
    rc = driver.main(argv)
    with open(os.path.join(done_dir, f"rc{rank}"), "w") as f:
        f.write(str(rc))


def test_driver_shards_its_stage_over_two_ranks(tmp_path):
    """driver.main on two ranks over gloo, the GPU job replaced by the oracle-backed stand-in: rank 0 runs the pipeline
    (read sanitising, the stage call, the merge), rank 1 serves the stage call.  OUT/2.overlap/s1_s1.paf must equal the
    single-process oracle stage with the constants of HyLight.py:130."""
    from hylight_amd import simulate as S
    from oracle import ava as OA
    from oracle import filters as F
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    reads, _ = S.simulate_reads(seed=83, n_strains=2, genome_len=30000, n_reads=40, mean_len=9000, min_len=7000,
                                max_len=13000)
    fq = tmp_path / "long.fq"
    S.write_fastq(reads, fq)
    out = tmp_path / "OUT"
    argv = ["-l", str(fq), "-o", str(out), "--corrected", "--nsplit", "3", "-t", "4", "--stop_after", "overlap"]
    port = launch.free_port()
    mp.spawn(_driver_rank, args=(2, port, argv, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"rc{r}").read() for r in range(2)] == ["0", "0"]
    s1 = (out / "1.split_fastx" / "s1.fa").read_text().split("\n")[:-1]
    chunks = []
    for i, (lo, hi) in enumerate(F.chunk_ranges(len(s1), 3)):
        cf = tmp_path / f"c{i}.fa"
        cf.write_text("\n".join(s1[lo:hi]) + "\n")
        OA.ava(cf, out / "1.split_fastx" / "s1.fa", str(cf) + ".paf")
        chunks.append(open(str(cf) + ".paf").read().split("\n")[:-1])
    want = F.stage(chunks, True, 6000, 2, 0.95)
    got = (out / "2.overlap" / "s1_s1.paf").read_text().split("\n")[:-1]
    assert got == want and len(want) > 10
    assert not [p for p in (out / "2.overlap").iterdir() if p.name.endswith((".part0", ".part1"))]      # merged and removed
