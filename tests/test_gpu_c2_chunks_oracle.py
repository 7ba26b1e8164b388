"""GPU vs oracle, row for row, on COMPLETE --nsplit chunks of the full C2 workload (BASELINE.json configs[1]:
10 000 reads, 5 strains x 400 kb, --nsplit 100 -> 104 target reads per chunk, each against all 10 000 queries): the
overlapper's raw rows (hlmi_ava on the chunk file = the minimap2 call of filter_overlap_slr2.py:51) and the worker's
output for the chunk (hlmi_split_reads2_shard restricted to that chunk vs oracle overlapper + oracle filters).  One
chunk is about a core-minute in the scalar oracle; the chunks run in parallel processes."""
import multiprocessing as mp
import os

import pytest

from hylight_amd import api
from hylight_amd import workloads as W
from oracle import filters as F

pytestmark = pytest.mark.gpu

CHUNKS = (0, 19, 37, 58, 77, 96)          # first ... last (the last one is shorter)


def _oracle_chunk(args):
    cf, fa, out = args
    from oracle import ava as OA
    OA.ava(cf, fa, out)
    return out


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
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        res = pool.map_async(_oracle_chunk, jobs)
        # the GPU works while the oracle processes run
        for cf, _, _ in jobs:
            api.ava(cf, fa, cf + ".gpu.paf")
        stage = cfg["stage"]
        for c in CHUNKS:
            # chunk c alone = rank c of a `number of chunks`-rank job
            api.split_reads2(fa, fa, cfg["nsplit"], tmp_path, tmp_path / f"w{c}.paf", long=True, rank=c, world=len(ranges),
                             **stage)
        res.get(timeout=900)
    for (cf, _, opaf), c in zip(jobs, CHUNKS):
        want = open(opaf).read()
        got = open(cf + ".gpu.paf").read()
        assert want.count("\n") > 3_000
        assert got == want, f"chunk {c}: raw overlapper rows differ"
        w = F.sort_scored(F.worker(want.split("\n")[:-1], True, stage["len_over"], stage["mc"], stage["iden"]))
        g = open(tmp_path / f"w{c}.paf").read().split("\n")[:-1]
        assert len(w) > 50 and g == w, f"chunk {c}: worker output differs"
