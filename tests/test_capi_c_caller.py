"""A plain-C caller of the C ABI (tests/capi/job_two_ranks.c): no Python, no PyTorch between the host program and
libhylight_mi.so - the binding a cgo / JNI / C host would use.  CPU: the program compiles and
links against include/hylight_mi.h with gcc.  GPU: it drives hlmi_job_open -> hlmi_job_sketch -> (device copies standing in
for the all-gather) -> hlmi_job_set_query_sketch -> hlmi_job_run(rank, 2) for two ranks on one card and finds the merged
output byte-identical to the single-rank entry point."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "capi", "job_two_ranks.c")


def _compile(out):
    lib_dir = os.path.join(ROOT, "hylight_amd")
    cmd = ["gcc", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), SRC, "-o", str(out),
           "-L" + lib_dir, "-lhylight_mi", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return str(out)


def test_c_caller_compiles_against_the_header(tmp_path):
    from hylight_amd import api
    api.load()                                     # (the library exists)
    exe = _compile(tmp_path / "job_two_ranks")
    assert os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_c_caller_two_ranks_equal_one_rank(tmp_path):
    from hylight_amd import simulate as S
    reads, _ = S.simulate_reads(seed=77, n_strains=2, genome_len=30_000, n_reads=140, mean_len=5_000, min_len=2_000, max_len=9_000)
    fa = tmp_path / "s1.fa"
    S.write_fasta(reads, fa)
    exe = _compile(tmp_path / "job_two_ranks")
    r = subprocess.run([exe, str(fa), "6", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK world=2") and int(r.stdout.split("rows=")[1]) > 50, r.stdout
