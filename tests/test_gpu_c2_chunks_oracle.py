"""GPU vs oracle, row for row, on COMPLETE --nsplit chunks of the full C2 workload (BASELINE.json configs[1]:
10 000 reads, 5 strains x 400 kb, --nsplit 100 -> 104 target reads per chunk, each against all 10 000 queries): the
overlapper's raw rows (hlmi_ava on the chunk file = the minimap2 call of filter_overlap_slr2.py:51) and the worker's
output for the chunk (hlmi_split_reads2_shard restricted to that chunk vs oracle overlapper + oracle filters).  One
chunk is about a core-minute in the oracle; the chunks run in processes of their own (started, not forked: the oracle's
OpenMP runtime does not survive a fork of a process that has used it) while the GPU works."""
import os
import subprocess
import sys

import pytest

from hylight_amd import api
from hylight_amd import workloads as W
from oracle import filters as F

pytestmark = pytest.mark.gpu

CHUNKS = (0, 19, 37, 58, 77, 96)          # first ... last (the last one is shorter)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_chunk(cf, fa, out, threads):
    code = "import sys; sys.path.insert(0, %r)\nfrom oracle import ava as OA\nOA.ava(%r, %r, %r)\n" % (ROOT, cf, fa, out)
    return subprocess.Popen([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS=str(threads)))


def test_whole_chunks_of_c2_match_the_oracle(tmp_path):
    cfg = W.config("C2")
    fa = str(tmp_path / "s1.fa")
    W.make_long(cfg, fa)
    lines = open(fa).read().split("\n")[:-1]
    ranges = F.chunk_ranges(len(lines), cfg["nsplit"])
    assert len(ranges) == 97
    jobs = []
    for c in CHUNKS:
        lo, hi = ranges[c]
        cf = str(tmp_path / f"sub{c:05d}")
        with open(cf, "w") as f:
            f.write("\n".join(lines[lo:hi]) + "\n")
        jobs.append((cf, fa, cf + ".oracle.paf"))
    threads = max(1, len(os.sched_getaffinity(0)) // len(jobs))
    procs = [_oracle_chunk(*j, threads) for j in jobs]
    # the GPU works while the oracle processes run
    for cf, _, _ in jobs:
        api.ava(cf, fa, cf + ".gpu.paf")
    stage = cfg["stage"]
    for c in CHUNKS:
        # chunk c alone = rank c of a `number of chunks`-rank job
        api.split_reads2(fa, fa, cfg["nsplit"], tmp_path, tmp_path / f"w{c}.paf", long=True, rank=c, world=len(ranges),
                         **stage)
    for p in procs:
        assert p.wait(timeout=900) == 0
    for (cf, _, opaf), c in zip(jobs, CHUNKS):
        want = open(opaf).read()
        got = open(cf + ".gpu.paf").read()
        assert want.count("\n") > 3_000
        assert got == want, f"chunk {c}: raw overlapper rows differ"
        w = F.sort_scored(F.worker(want.split("\n")[:-1], True, stage["len_over"], stage["mc"], stage["iden"]))
        g = open(tmp_path / f"w{c}.paf").read().split("\n")[:-1]
        assert len(w) > 30 and g == w, f"chunk {c}: worker output differs"
