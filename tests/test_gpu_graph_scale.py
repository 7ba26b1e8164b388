"""GPU: the overlap-graph stage (rows a9-a14 on the device, a15/a16 on the host) against the compiled reference
(oracle/_ref/miniasm = tools/miniasm built by oracle/Makefile) on PAFs of 1.4 million rows - 2.9 million overlap
records, 20 000 reads, thousands of equal sort keys, duplicate and unpaired arcs, bubbles, bi-loops.  Every output
format the reference offers (`-p ug|sg|paf|bed`, main.c:146-150) must come out byte for byte, under both flag sets
HyLight uses (script/HyLight.py:137,140)."""
import os
import subprocess

import pytest

from hylight_amd import api
from hylight_amd import simulate as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "miniasm")
FLAGS = [("-n 1 -e 1 -c 1", dict(n_rounds_arg=1, min_dp=1)), ("-n 3 -e 1 -c 3", dict(n_rounds_arg=3, min_dp=3))]


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/miniasm not built")
@pytest.mark.parametrize("seed", [1, 2])
def test_million_row_paf_matches_the_compiled_reference(tmp_path, seed):
    _, rows = S.layout_paf(seed)
    assert len(rows) > 1_000_000
    paf = tmp_path / "big.paf"
    paf.write_text("\n".join(rows) + "\n")
    for flags, kw in FLAGS:
        for fmt in ("ug", "sg", "bed", "paf"):
            want = subprocess.run(f"{REF} -d 10000 {flags} -p {fmt} {paf}", shell=True, check=True, capture_output=True).stdout
            out = tmp_path / f"o.{fmt}"
            api.miniasm(paf, None, out, bub_dist=10000, max_ext=1, outfmt=fmt, **kw)
            assert open(out, "rb").read() == want, (flags, fmt)
            assert len(want) > 1000
    st = api.last_stats()
    assert st["graph_rows"] == len(rows) and st["graph_overlaps"] > 2_000_000
