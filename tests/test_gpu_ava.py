"""GPU parity of the overlapper: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from hylight_amd import api
from hylight_amd import simulate as S
from oracle import ava as OA
from oracle import filters as F

pytestmark = pytest.mark.gpu


def _write(tmp_path, name, reads):
    p = tmp_path / name
    S.write_fasta(reads, p)
    return p


def _sim(seed, n, **kw):
    args = dict(n_strains=2, genome_len=20000, mean_len=5000, min_len=2000, max_len=9000)
    args.update(kw)
    reads, _ = S.simulate_reads(seed=seed, n_reads=n, **args)
    return reads


def test_sketch_matches_oracle(tmp_path):
    import torch
    reads = _sim(21, 40)
    # edge cases: ambiguous bases, a long homopolymer, a read shorter than k+w, lower case
    reads[3].seq[100:103] = ord("N")
    reads[4].seq[500:900] = ord("A")
    reads[5].seq = reads[5].seq[:15].copy()
    reads[6].seq = np.frombuffer(reads[6].seq.tobytes().lower(), dtype=np.uint8).copy()
    fa = _write(tmp_path, "r.fa", reads)
    job = api.Job(fa, fa, 4, True)
    n = job.num_queries
    assert n == len(reads)
    for lo, hi in ((0, n), (7, 23)):
        cap = job.sketch_bound(lo, hi)
        mz = torch.zeros((cap, 2), dtype=torch.int64, device="cuda")
        cnt = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
        got_n = job.sketch(lo, hi, mz.data_ptr(), cap, cnt.data_ptr())
        got = mz[:got_n].cpu().numpy().view(np.uint64)
        want = [OA.sketch(reads[i].seq.tobytes(), rid=i) for i in range(lo, hi)]
        assert cnt.cpu().numpy().tolist() == [len(w) for w in want]
        want = np.concatenate([w for w in want if len(w)]) if any(len(w) for w in want) else np.zeros((0, 2), np.uint64)
        assert got.shape == want.shape and (got == want).all()
    job.close()


@pytest.mark.parametrize("seed,n,kw", [
    (31, 40, {}),
    (32, 60, dict(n_strains=3, genome_len=15000, err_sub=0.01, err_ins=0.004, err_del=0.004)),
    (33, 30, dict(n_strains=1, genome_len=8000, mean_len=3000, min_len=500)),
])
def test_ava_matches_oracle(tmp_path, seed, n, kw):
    reads = _sim(seed, n, **kw)
    fa = _write(tmp_path, "r.fa", reads)
    api.ava(fa, fa, tmp_path / "g.paf")
    OA.ava(fa, fa, tmp_path / "o.paf")
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 50
    assert got == want


def test_ava_key_value_anchor_form(tmp_path, monkeypatch):
    """Anchors normally travel as one packed 64-bit word; inputs whose id / position widths do not fit take a
    key + value form (same order, same chains).  HLMI_ANCHOR_PAIRS forces it."""
    reads = _sim(32, 60, n_strains=3, genome_len=15000, err_sub=0.01, err_ins=0.004, err_del=0.004)
    fa = _write(tmp_path, "r.fa", reads)
    api.ava(fa, fa, tmp_path / "packed.paf")
    monkeypatch.setenv("HLMI_ANCHOR_PAIRS", "1")
    api.ava(fa, fa, tmp_path / "pairs.paf")
    assert open(tmp_path / "pairs.paf").read() == open(tmp_path / "packed.paf").read()
    assert api.last_stats()["anchor_bytes"] == 16 * api.last_stats()["anchors"]
    # in between: everything but the (target, strand) bits in the word, those in a 2- or 4-byte key of their own
    monkeypatch.delenv("HLMI_ANCHOR_PAIRS")
    for width in ("2", "4"):
        monkeypatch.setenv("HLMI_ANCHOR_SPLIT", width)
        api.ava(fa, fa, tmp_path / "split.paf")
        assert open(tmp_path / "split.paf").read() == open(tmp_path / "packed.paf").read()
        assert api.last_stats()["anchor_bytes"] == (8 + int(width)) * api.last_stats()["anchors"]


def test_ava_index_search_forms(tmp_path, monkeypatch):
    """Seed counting searches one word per index entry (hash ‖ rank word); HLMI_NO_RANK_WORD forces the two-array
    comparison that long k-mers with many reads take.  Both pairing rules: every pair once (long mode) and every
    partner but the read itself (short mode: a key's run is counted as suffix minus the read's own entries)."""
    reads = _sim(35, 70, n_strains=3, genome_len=15000, err_sub=0.01, err_ins=0.005, err_del=0.005)
    fa = _write(tmp_path, "r.fa", reads)
    OA.ava(fa, fa, tmp_path / "o1.paf")
    OA.ava(fa, fa, tmp_path / "o2.paf", OA.opts_short())
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("HLMI_NO_RANK_WORD", env)
        api.ava(fa, fa, tmp_path / "g1.paf")
        api.ava(fa, fa, tmp_path / "g2.paf", api.ava_opts_short())
        assert open(tmp_path / "g1.paf").read() == open(tmp_path / "o1.paf").read()
        assert open(tmp_path / "g2.paf").read() == open(tmp_path / "o2.paf").read()
    assert len(open(tmp_path / "o2.paf").read().splitlines()) > 20


def test_ava_chain_state_forms(tmp_path, monkeypatch):
    """The chaining DP keeps score and predecessor in one packed word per lane when scores stay below 2^22 (every
    real read set); HLMI_CHAIN_UNPACKED forces the two-register form longer sequences take.  Same chains, and both
    equal to the oracle."""
    reads = _sim(33, 80, n_strains=3, genome_len=20000, err_sub=0.02, err_ins=0.01, err_del=0.01)
    fa = _write(tmp_path, "r.fa", reads)
    api.ava(fa, fa, tmp_path / "packed.paf")
    monkeypatch.setenv("HLMI_CHAIN_UNPACKED", "1")
    api.ava(fa, fa, tmp_path / "unpacked.paf")
    OA.ava(fa, fa, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert open(tmp_path / "packed.paf").read() == want
    assert open(tmp_path / "unpacked.paf").read() == want


def test_ava_narrow_band_kernel_forms(tmp_path, monkeypatch):
    """Near-diagonal blocks run eight per wave with two tasks sharing each lane (16-bit halves of one register, score
    from the walked runs, or H of the last row when the quad holds an ambiguous base); HLMI_NARROW_UNPACKED forces the
    four-per-wave 32-bit form that unusual scoring constants take.  Reads with errors and a few N so that both forms see
    DP tasks with and without ambiguous bases."""
    reads = _sim(36, 60, n_strains=2, genome_len=15000, err_sub=0.02, err_ins=0.01, err_del=0.01)
    for k in (3, 11, 17, 29, 41):
        reads[k].seq[700:703] = ord("N")
        reads[k].seq[len(reads[k].seq) // 2] = ord("N")
    fa = _write(tmp_path, "r.fa", reads)
    OA.ava(fa, fa, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 50
    api.ava(fa, fa, tmp_path / "pk.paf")
    assert open(tmp_path / "pk.paf").read() == want
    assert api.last_stats()["align_tasks_narrow"] > 1000
    monkeypatch.setenv("HLMI_NARROW_UNPACKED", "1")
    api.ava(fa, fa, tmp_path / "un.paf")
    assert open(tmp_path / "un.paf").read() == want


@pytest.mark.parametrize("seed,errs", [(71, (0.06, 0.03, 0.03)), (72, (0.10, 0.02, 0.02)), (73, (0.02, 0.06, 0.06))])
def test_ava_noisy_reads(tmp_path, seed, errs):
    """Raw-read error rates: most blocks need a DP, many of them in the 64-diagonal band or in the long near-diagonal
    instance, with tens of runs per task (both packed and 32-bit kernels, the second traceback walk, the run pool)."""
    sub, ins, dele = errs
    reads = _sim(seed, 36, n_strains=2, genome_len=12000, mean_len=4000, min_len=1500, max_len=8000,
                 err_sub=sub, err_ins=ins, err_del=dele)
    fa = _write(tmp_path, "r.fa", reads)
    api.ava(fa, fa, tmp_path / "g.paf")
    OA.ava(fa, fa, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 10          # (whole overlaps: long gaps between anchors no longer cut the chains)
    assert open(tmp_path / "g.paf").read() == want
    st = api.last_stats()
    assert st["align_tasks_dp"] > 0.3 * st["align_tasks"]


@pytest.mark.parametrize("changes", [
    dict(mismatch=16, gap_open=20, gap_ext=2, gap_open2=0),           # scores too large for the 16-bit kernel: the 32-bit form by itself
    dict(match=1, mismatch=1, gap_open=1, gap_ext=1, gap_open2=0),     # many ties: the traceback's tie rules decide every block
    dict(gap_open2=0),                                    # one-piece gap cost
    dict(gap_open=4, gap_ext=3, gap_open2=21, gap_ext2=2),              # second piece cheaper from 18 bases on
    dict(k=15, w=8, bandwidth=300, max_gap=2000, min_chain_score=40),     # other seeds, byte-table gap costs
    dict(bandwidth=3000),                                 # gap-cost table too small: costs computed in the chain DP
])
def test_ava_other_constants(tmp_path, changes):
    """Constants other than the two presets go through the same kernels (or their fallback forms)."""
    reads = _sim(37, 40, n_strains=2, genome_len=12000, err_sub=0.02, err_ins=0.01, err_del=0.01)
    fa = _write(tmp_path, "r.fa", reads)
    go, oo = api.ava_opts_long(), OA.opts_long()
    for k, v in changes.items():
        setattr(go, k, v)
        setattr(oo, k, v)
    api.ava(fa, fa, tmp_path / "g.paf", go)
    OA.ava(fa, fa, tmp_path / "o.paf", oo)
    want = open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 20
    assert open(tmp_path / "g.paf").read() == want


def test_ava_traceback_run_buffer_overflow(tmp_path, monkeypatch):
    """The traceback keeps a task's runs in LDS and walks once; a task with more runs than the buffer holds is
    walked a second time straight into the pool.  HLMI_RUN_BUF_CAP=3 sends nearly every DP task down that path."""
    reads = _sim(34, 50, n_strains=2, genome_len=15000, err_sub=0.03, err_ins=0.02, err_del=0.02)
    fa = _write(tmp_path, "r.fa", reads)
    api.ava(fa, fa, tmp_path / "a.paf")
    monkeypatch.setenv("HLMI_RUN_BUF_CAP", "3")
    api.ava(fa, fa, tmp_path / "b.paf")
    OA.ava(fa, fa, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 50
    assert open(tmp_path / "a.paf").read() == want
    assert open(tmp_path / "b.paf").read() == want


@pytest.mark.parametrize("cap", ["0", "40"])
def test_ava_row_assembly_without_the_lds_stage(tmp_path, monkeypatch, cap):
    """The assembly puts the CIGAR slots of a 64-task step together in LDS when they fit the buffer and stores them one
    by one when they do not: with the buffer cut to nothing / to 40 slots every step / most steps take the direct path."""
    reads = _sim(36, 40, err_sub=0.02, err_ins=0.005, err_del=0.005)
    fa = _write(tmp_path, "r.fa", reads)
    monkeypatch.setenv("HLMI_ASM_STAGE_CAP", cap)
    api.ava(fa, fa, tmp_path / "g.paf")
    monkeypatch.delenv("HLMI_ASM_STAGE_CAP")
    OA.ava(fa, fa, tmp_path / "o.paf")
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    assert len(want.splitlines()) > 50
    assert got == want


def test_ava_target_subset_and_ambiguous_bases(tmp_path):
    reads = _sim(41, 36)
    reads[2].seq[1000:1004] = ord("N")
    reads[9].seq[10:12] = ord("n")
    q = _write(tmp_path, "q.fa", reads)
    t = _write(tmp_path, "t.fa", reads[10:22])
    api.ava(t, q, tmp_path / "g.paf")
    OA.ava(t, q, tmp_path / "o.paf")
    assert open(tmp_path / "g.paf").read() == open(tmp_path / "o.paf").read()


def test_ava_empty_and_tiny_inputs(tmp_path):
    reads = _sim(51, 3)
    fa = _write(tmp_path, "r.fa", reads)
    empty = tmp_path / "e.fa"
    empty.write_text("")
    api.ava(empty, fa, tmp_path / "g.paf")
    assert open(tmp_path / "g.paf").read() == ""
    api.ava(fa, empty, tmp_path / "g.paf")
    assert open(tmp_path / "g.paf").read() == ""
    tiny = tmp_path / "tiny.fa"
    tiny.write_text(">a\nACGT\n>b\nACGTACGTTTGA\n")
    api.ava(tiny, tiny, tmp_path / "g.paf")
    assert open(tmp_path / "g.paf").read() == ""


def test_split_reads2_matches_oracle_pipeline(tmp_path):
    """Whole stage (a1-a8): GPU overlapper + GPU filters vs oracle overlapper + oracle filters."""
    reads = _sim(61, 90, n_strains=3, genome_len=24000, mean_len=6000, min_len=2500, max_len=12000)
    fa = _write(tmp_path, "s1.fa", reads)
    out = tmp_path / "s1_s1.paf"
    api.split_reads2(fa, fa, 4, tmp_path, out, threads=4, len_over=1000, mc=2, iden=0.95, long=True)
    # oracle: same chunking, oracle overlapper per chunk, oracle filters
    lines = open(fa).read().split("\n")[:-1]
    chunks = []
    for i, (lo, hi) in enumerate(F.chunk_ranges(len(lines), 4)):
        cf = tmp_path / f"chunk{i}.fa"
        cf.write_text("\n".join(lines[lo:hi]) + "\n")
        OA.ava(cf, fa, tmp_path / f"chunk{i}.paf")
        chunks.append(open(tmp_path / f"chunk{i}.paf").read().split("\n")[:-1])
    want = F.stage(chunks, True, 1000, 2, 0.95)
    got = open(out).read().split("\n")[:-1]
    assert len(want) > 100
    assert got == want
    st = api.last_stats()
    assert st["rows_out"] == len(want)


def test_ava_extremely_repetitive_chunk(tmp_path):
    """1030 copies of one 160-base read in a chunk: the occurrence count at the (1 - 2e-4) quantile of the chunk's
    distinct minimizers lies beyond the 1024-bin histogram the index keeps per chunk; the cut-off then comes from the
    exact quantile (minimap2 would only clamp its mid_occ and carry on) and the run completes like the oracle's."""
    rng = np.random.default_rng(3)
    unit = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=160)]
    reads = [S.Read(f"rep{i:04d}", unit.copy(), None, 0, 0, 0, False) for i in range(1030)]
    more, _ = S.simulate_reads(seed=4, n_strains=1, genome_len=20000, n_reads=20, mean_len=4000, min_len=2000, max_len=6000,
                               name_prefix="u")
    fa = tmp_path / "rep.fa"
    S.write_fasta(reads + more, fa)
    api.ava(fa, fa, tmp_path / "g.paf")
    assert api.last_stats().get("index_exact_quantiles", 0) >= 1
    OA.ava(fa, fa, tmp_path / "o.paf")
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    assert want.count("\n") > 500_000
    assert got == want


def test_ava_long_gaps_take_the_second_gap_piece(tmp_path):
    """minimap2's gap cost has two pieces (-O4,24 -E2,1 in the ava-pb preset HyLight uses): gaps of more than 20 bases are
    cheaper under the second one.  Strain B differs from strain A by 24-36-base insertions and deletions, so alignments
    between reads of different strains contain such gaps inside 64-diagonal blocks: the GPU rows must equal the
    oracle's with the second piece on and off, and the long gaps must be there.  (The PAF carries no alignment score:
    the second piece shows in the rows only where it tips a choice - a long gap against a run of substitutions, a piece
    at the minimum DP score - so on most inputs, this one included, the two settings print the same rows.)"""
    rng = np.random.default_rng(61)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    a = bases[rng.integers(0, 4, size=24_000)]
    parts, pos = [], 0
    for k, cut in enumerate(range(1500, 23_000, 1800)):
        parts.append(a[pos:cut])
        ln = int(rng.integers(24, 37))
        if k % 2:
            parts.append(bases[rng.integers(0, 4, size=ln)])      # insertion in B
            pos = cut
        else:
            pos = cut + ln                                       # deletion in B
    parts.append(a[pos:])
    b = np.concatenate(parts)
    reads = []
    for i in range(30):
        g = a if i % 2 == 0 else b
        s = int(rng.integers(0, len(g) - 7000))
        e = s + int(rng.integers(4000, 7000))
        seq = g[s:e].copy()
        err = rng.random(len(seq)) < 0.004
        seq[err] = bases[rng.integers(0, 4, size=int(err.sum()))]
        if i % 3 == 0:
            seq = S.revcomp(seq)
        reads.append(S.Read(f"g{i:02d}", np.ascontiguousarray(seq), None, i % 2, s, e, i % 3 == 0))
    fa = _write(tmp_path, "gaps.fa", reads)
    api.ava(fa, fa, tmp_path / "g.paf")
    OA.ava(fa, fa, tmp_path / "o.paf")
    want = open(tmp_path / "o.paf").read()
    assert open(tmp_path / "g.paf").read() == want
    import re
    long_gaps = [int(n) for n in re.findall(r"(\d+)[ID]", want) if int(n) > 20]
    assert len(long_gaps) >= 20, len(long_gaps)
    one = OA.opts_long()
    one.gap_open2 = 0
    OA.ava(fa, fa, tmp_path / "o1.paf", one)
    g1 = api.ava_opts_long()
    g1.gap_open2 = 0
    api.ava(fa, fa, tmp_path / "g1.paf", g1)
    assert open(tmp_path / "g1.paf").read() == open(tmp_path / "o1.paf").read()


def test_ava_alignment_in_spans_of_pieces(tmp_path, monkeypatch):
    """A batch with more alignment tasks than the 32-bit run pool can index is aligned in spans of consecutive pieces
    (divergent read sets at full batch size); HLMI_ALIGN_SPAN_TASKS forces spans of a few thousand tasks here."""
    reads = _sim(36, 80, n_strains=3, genome_len=15000, err_sub=0.01, err_ins=0.004, err_del=0.004)
    fa = _write(tmp_path, "r.fa", reads)
    api.ava(fa, fa, tmp_path / "one.paf")
    monkeypatch.setenv("HLMI_ALIGN_SPAN_TASKS", "3000")
    api.ava(fa, fa, tmp_path / "spans.paf")
    assert api.last_stats().get("align_spans", 0) >= 2
    assert open(tmp_path / "spans.paf").read() == open(tmp_path / "one.paf").read()


def test_ava_cigar_runs_merging_across_many_tasks(tmp_path):
    """Rows whose alignment tasks are almost all one exact-match run: the row's CIGAR collapses into a few ops, each
    the sum of hundreds of tasks' runs, across several 64-task steps of the assembly (copies of one 24 kb read: exact, with
    one substitution, one inserted base, one deleted base, a reverse-complement copy, and prefix / suffix pieces)."""
    rng = np.random.default_rng(77)
    base = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=24000)].copy()
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    def mk(name, seq):
        return S.Read(name, np.ascontiguousarray(seq), None, 0, 0, 0, False)
    sub = base.copy(); sub[11000] = ord("A") if sub[11000] != ord("A") else ord("C")
    ins = np.concatenate([base[:7000], np.frombuffer(b"G", dtype=np.uint8), base[7000:]])
    dele = np.concatenate([base[:16000], base[16001:]])
    two = base.copy(); two[300] = ord("T") if two[300] != ord("T") else ord("G"); two[23000] = ord("C") if two[23000] != ord("C") else ord("A")
    reads = [mk("dup0", base), mk("dup1", base.copy()), mk("sub", sub), mk("ins", ins), mk("del", dele), mk("two", two),
             mk("rc", comp[base[::-1]]), mk("head", base[:15000].copy()), mk("tail", base[9000:].copy())]
    fa = tmp_path / "dups.fa"
    S.write_fasta(reads, fa)
    api.ava(fa, fa, tmp_path / "g.paf")
    OA.ava(fa, fa, tmp_path / "o.paf")
    got, want = open(tmp_path / "g.paf").read(), open(tmp_path / "o.paf").read()
    rows = want.splitlines()
    assert len(rows) >= 20
    # the exact copies: one op for the whole read
    assert any("cg:Z:24000=" in r for r in rows)
    assert got == want
