"""CPU: the host-side final order (`sort -k12 -nr`, script/utils.py:54,69) against GNU sort itself.

hlmi_merge_scored_paf is the one product entry point that needs no GPU: it reads the per-rank parts, orders all
lines by column 12 (numeric, descending; ties by reversed whole-line byte order, the C locale's last-resort rule)
and writes them back.  The stage uses the same sort on views into its text buffers."""
import os
import random
import subprocess

import pytest

from hylight_amd import api


def _line(rng, i, scores):
    cols = [f"r{rng.randrange(60)}", "1000", "0", "900", "+-"[i % 2], f"r{rng.randrange(60)}", "1200", "5", "905", "880",
            "901", rng.choice(scores), "0.99", "0.97", ""]
    return "\t".join(cols)


@pytest.mark.parametrize("seed,scores", [
    (1, None),                                                      # "%.4f" scores as the stage writes them
    (2, ["0.9876", "0.98760", "1.0000", "0.5", "12.25", "0.0001", "0", "-0.5", "00.5", ".75", "1e3", "abc", ""]),
])
def test_merge_matches_gnu_sort(tmp_path, seed, scores):
    rng = random.Random(seed)
    if scores is None:
        scores = ["%.4f" % (rng.randrange(9000, 10001) / 10000.0) for _ in range(40)]      # many ties
    parts, everything = [], []
    for p in range(3):
        lines = [_line(rng, i, scores) for i in range(500)]
        everything += lines
        fn = tmp_path / f"part{p}.paf"
        fn.write_text("\n".join(lines) + "\n")
        parts.append(fn)
    (tmp_path / "empty.paf").write_text("")
    parts.append(tmp_path / "empty.paf")
    api.merge_scored_paf(parts, tmp_path / "out.paf")
    (tmp_path / "all.paf").write_text("\n".join(everything) + "\n")
    want = subprocess.run(["sort", "-k12", "-nr", str(tmp_path / "all.paf")], env=dict(os.environ, LC_ALL="C"),
                          capture_output=True, text=True, check=True).stdout
    assert open(tmp_path / "out.paf").read() == want
