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


def test_merge_with_huge_tie_runs_matches_gnu_sort(tmp_path):
    """Three score values over 300 000 lines: every tie run is longer than a thread's share of the rows, so the runs are
    sorted piecewise by all host threads and merged (paf_io.cpp:sort_scored_lines), and the output goes through the
    mapped-file writer (write_lines).  Same bytes as GNU sort."""
    rng = random.Random(7)
    lines = [_line(rng, i, ["0.9990", "1.0000", "0.9876"]) + str(rng.randrange(10 ** 6)) for i in range(300000)]
    (tmp_path / "in.paf").write_text("\n".join(lines) + "\n")
    api.merge_scored_paf([tmp_path / "in.paf"], tmp_path / "out.paf")
    want = subprocess.run(["sort", "-k12", "-nr", str(tmp_path / "in.paf")], env=dict(os.environ, LC_ALL="C"),
                          capture_output=True, check=True).stdout
    assert open(tmp_path / "out.paf", "rb").read() == want


def test_write_errors_and_special_files(tmp_path):
    """A target that is not a regular file is written directly (no temporary + rename); a full device is an error that
    names the file, not a truncated output."""
    (tmp_path / "in.paf").write_text("a\t1\t0\t1\t+\tb\t1\t0\t1\t1\t1\t0.5000\t1.0000\t1.0000\t\n")
    fifo = tmp_path / "fifo"
    os.mkfifo(fifo)
    reader = subprocess.Popen(["cat", str(fifo)], stdout=subprocess.PIPE)
    api.merge_scored_paf([tmp_path / "in.paf"], fifo)
    assert reader.communicate(timeout=30)[0] == open(tmp_path / "in.paf", "rb").read()
    with pytest.raises(Exception, match="/dev/full"):
        api.merge_scored_paf([tmp_path / "in.paf"], "/dev/full")
    out = tmp_path / "out.paf"
    api.merge_scored_paf([tmp_path / "in.paf"], out)
    assert out.read_bytes() == open(tmp_path / "in.paf", "rb").read()
    assert sorted(os.listdir(tmp_path)) == ["fifo", "in.paf", "out.paf"]          # no temporary left behind


def test_fixed4_formatter_writes_what_printf_writes(tmp_path):
    """The integer "%.4f" of the final rows' three score columns (paf_io.cpp:format_fixed4) against printf on ~24 M values:
    the dyadic grid (every exact tie of the fourth decimal), ratios of small integers as the rows have them, random bit
    patterns of every exponent, signs and specials (tests/capi/fixed4_check.cpp)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api.load()                                                  # (builds the library when it is missing)
    exe = tmp_path / "fixed4_check"
    libdir = os.path.join(root, "hylight_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(root, "tests", "capi", "fixed4_check.cpp"), "-o", str(exe),
                    "-L", libdir, "-lhylight_mi", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr
