"""GPU: the transitive reduction (row a14, tools/miniasm/asg.c:148-193) on a DENSE overlap graph.  Equal-length reads
stacked two bases apart: no read contains another, so containment removal (hit.c:225-256) thins nothing and every read end
keeps an arc to each of the ~1 500 reads that overlap it by 2 kb or more - more out-arcs than the per-wave LDS table of
reduce_kernel holds, so the vertices go through the global-memory table (`graph_big_vertices` > 0).  All four output
formats of the compiled reference must come out byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from hylight_amd import api

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "miniasm")


def dense_layout_paf(seed, n_reads=2200, length=5000, step=2, min_ovl=2000, jitter=2):
    """Rows of every pair of reads [i*step, i*step + length) sharing >= min_ovl bases; both strands, ends jittered."""
    rng = np.random.default_rng(seed)
    start = np.arange(n_reads) * step
    rev = rng.random(n_reads) < 0.5
    reach = (length - min_ovl) // step
    a = np.repeat(np.arange(n_reads), reach)
    b = a + np.tile(np.arange(1, reach + 1), n_reads)
    ok = b < n_reads
    a, b = a[ok], b[ok]
    s = start[b] + rng.integers(0, jitter + 1, size=len(a))
    e = start[a] + length - rng.integers(0, jitter + 1, size=len(a))

    def on_read(r, s, e):
        return np.where(rev[r], start[r] + length - e, s - start[r]), np.where(rev[r], start[r] + length - s, e - start[r])
    qs, qe = on_read(a, s, e)
    ts, te = on_read(b, s, e)
    strand = np.where(rev[a] == rev[b], "+", "-")
    ml = (e - s) - rng.integers(0, 30, size=len(a))
    rows = [f"D{x}\t{length}\t{q0}\t{q1}\t{z}\tD{y}\t{length}\t{t0}\t{t1}\t{m}\t{bl}\t255"
            for x, y, q0, q1, z, t0, t1, m, bl in zip(a.tolist(), b.tolist(), qs.tolist(), qe.tolist(), strand.tolist(),
                                                      ts.tolist(), te.tolist(), ml.tolist(), (e - s).tolist())]
    order = rng.permutation(len(rows))
    return [rows[i] for i in order.tolist()]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/miniasm not built")
def test_dense_graph_matches_the_compiled_reference(tmp_path):
    rows = dense_layout_paf(3)
    assert len(rows) > 2_000_000
    paf = tmp_path / "dense.paf"
    paf.write_text("\n".join(rows) + "\n")
    for fmt in ("sg", "ug", "bed", "paf"):
        want = subprocess.run(f"{REF} -d 10000 -n 1 -e 1 -c 1 -p {fmt} {paf}", shell=True, check=True, capture_output=True).stdout
        out = tmp_path / f"o.{fmt}"
        api.miniasm(paf, None, out, bub_dist=10000, n_rounds_arg=1, max_ext=1, min_dp=1, outfmt=fmt)
        if fmt == "sg":                       # (`-p paf` / `-p bed` stop in front of the graph)
            st = api.last_stats()
        assert open(out, "rb").read() == want, fmt
        assert len(want) > 100
    print("dense graph:", {k: st[k] for k in ("graph_arcs", "graph_arcs_reduced", "graph_big_vertices", "graph_big_table_slots",
                                              "kernel_ms.graph_reduce", "kernel_ms.graph_reduce_big", "kernel_ms.graph_device") if k in st})
    assert st["graph_big_vertices"] > 1000 and st["graph_arcs"] > 2_000_000
