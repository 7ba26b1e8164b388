"""CPU: pins the overlapper oracle (oracle/ava_oracle.c).  minimap2 itself is absent from the
reference tree (parity unpinned), so the anchors here are the published
properties of the algorithm and the simulator's ground truth."""
import numpy as np
import pytest

from hylight_amd import simulate as S
from oracle import ava as OA


def test_hash_is_a_bijection_on_small_masks():
    # Li 2016 section 2.2: the integer hash is invertible on [0, 4^k)
    for k in (3, 5, 7):
        mask = (1 << 2 * k) - 1
        vals = {OA.hash64(x, mask) for x in range(mask + 1)}
        assert len(vals) == mask + 1 and max(vals) <= mask


# Literal known answers of the 64-bit invertible mix (Li 2016 section 2.2; the same function minimap2 applies to every
# k-mer), computed once with an independent pure-Python restatement of the published seven
# steps and written down here: a change of any shift, constant or mask position in oracle/ava_oracle.c or - through the
# GPU-vs-oracle sketch tests - in csrc/sketch.hip moves them.
HASH_KAT = {
    19: [(0x0, 0x1df06f29bc), (0x1, 0x29b794f8ce), (0x2, 0x3f6f2a0674), (0x123456789, 0xa635aa6a1),
         (0x2aaaaaaaaa, 0x1a2ccac738), (0x3fffffffff, 0x1c5d2677be)],                    # -k19: long mode (slr2:51)
    21: [(0x0, 0x1df06f29bc0), (0x1, 0x69b794f8ce), (0x2, 0x33f6f2a0674), (0x123456789, 0x1ac74bc9de6),
         (0x2aaaaaaaaa, 0x392647df628), (0x3ffffffffff, 0xddf0b551bf)],                  # -k21: short mode (slr2:55)
    15: [(0x0, 0x3ff06f15), (0x1, 0x3794f8e6), (0x2, 0x2f3f0620), (0x23456789, 0x28de583e),
         (0x2aaaaaaa, 0x304a3cb6), (0x3fffffff, 0x864d0ee)],
}


def test_hash_known_answers():
    for k, table in HASH_KAT.items():
        mask = (1 << 2 * k) - 1
        for x, want in table:
            assert OA.hash64(x, mask) == want, (k, hex(x))


def _hash64_inverse(y, bits):
    """The mix undone step by step: odd multipliers have inverses modulo 2^bits, x ^= x >> s is undone by iterating."""
    mod = 1 << bits

    def unxorshift(v, s):
        x = v
        for _ in range(bits // s + 1):
            x = v ^ (x >> s)
        return x
    y = y * pow(1 + (1 << 31), -1, mod) % mod
    y = unxorshift(y, 28)
    y = y * pow(21, -1, mod) % mod
    y = unxorshift(y, 14)
    y = y * pow(265, -1, mod) % mod
    y = unxorshift(y, 24)
    return (y + 1) * pow((1 << 21) - 1, -1, mod) % mod          # ~x + (x << 21) = x (2^21 - 1) - 1


def test_hash_round_trips_through_its_inverse():
    rng = np.random.default_rng(17)
    for k in (15, 19, 21, 27):
        bits = 2 * k
        for x in [0, 1, (1 << bits) - 1] + [int(v) for v in rng.integers(0, 1 << bits, size=200, dtype=np.uint64)]:
            assert _hash64_inverse(OA.hash64(x, (1 << bits) - 1), bits) == x


def _mz(seq, **kw):
    m = OA.sketch(seq, **kw)
    return [(int(x) >> 8, int(x) & 0xff, (int(y) & 0xffffffff) >> 1, int(y) & 1) for x, y in m]


def test_hpc_sketch_ignores_homopolymer_length():
    rng = np.random.default_rng(5)
    core = "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
    # remove existing runs, then stretch every 7th base into a run of 3
    comp = "".join(c for i, c in enumerate(core) if i == 0 or c != core[i - 1])
    stretched = "".join(c * 3 if i % 7 == 0 else c for i, c in enumerate(comp))
    a, b = _mz(comp.encode()), _mz(stretched.encode())
    assert [x[0] for x in a] == [x[0] for x in b] and len(a) > 20       # same hashes, same order
    assert any(sa != sb for (_, sa, _, _), (_, sb, _, _) in zip(a, b))  # but different spans
    # positions are ends of runs in original coordinates
    for h, span, pos, strand in b:
        assert pos + 1 == len(stretched) or stretched[pos] != stretched[pos + 1]


def test_sketch_is_strand_symmetric():
    rng = np.random.default_rng(9)
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, 600)).encode()
    rc = S.revcomp(np.frombuffer(s, dtype=np.uint8)).tobytes()
    a, b = _mz(s), _mz(rc)
    assert sorted(x[0] for x in a) == sorted(x[0] for x in b)


def test_ambiguous_bases_reset_kmers_and_short_reads_give_nothing():
    assert _mz(b"ACGTACGTAC") == []
    rng = np.random.default_rng(2)
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, 300))
    with_n = s[:150] + "N" + s[150:]
    a = _mz(with_n.encode())
    for h, span, pos, strand in a:       # no k-mer may span the N at position 150
        assert pos < 150 or pos - span + 1 > 150
    assert len(a) > 10


def test_long_homopolymer_span_overflow():
    rng = np.random.default_rng(3)
    s = "".join("ACGT"[i] for i in rng.integers(0, 4, 200))
    t = s[:100] + "A" * 300 + s[100:]
    for h, span, pos, strand in _mz(t.encode()):
        assert span < 256


@pytest.fixture(scope="module")
def small_set(tmp_path_factory):
    d = tmp_path_factory.mktemp("ava")
    reads, _ = S.simulate_reads(seed=11, n_strains=2, genome_len=20000, n_reads=50, mean_len=5000, min_len=2500,
                                max_len=9000, keep_gpos=True)
    fa = d / "reads.fa"
    S.write_fasta(reads, fa)
    OA.ava(fa, fa, d / "o.paf")
    return reads, [l.rstrip("\n").split("\t") for l in open(d / "o.paf")]


def test_ava_recovers_simulated_overlaps(small_set):
    reads, rows = small_set
    truth = {}
    for l in S.truth_paf(reads, min_cols=300):
        c = l.split("\t")
        truth[(c[0], c[5])] = c
    found = {(c[0], c[5]): c for c in rows}
    assert set(truth) <= set(found)                       # recall 1.0 for overlaps >= 300 columns
    close = 0
    for k, t in truth.items():
        c = found[k]
        assert c[4] == t[4]
        if max(abs(int(c[i]) - int(t[i])) for i in (2, 3, 7, 8)) <= 8:
            close += 1
    assert close >= 0.98 * len(truth)


def test_ava_rows_are_consistent_paf(small_set):
    reads, rows = small_set
    by_name = {r.name: r for r in reads}
    import re
    for c in rows:
        assert c[0] < c[5]                                 # pair once: strcmp(qname, tname) < 0
        ops = re.findall(r"(\d+)([=XID])", c[-1][5:])
        q = sum(int(n) for n, o in ops if o in "=XI")
        t = sum(int(n) for n, o in ops if o in "=XD")
        assert q == int(c[3]) - int(c[2]) and t == int(c[8]) - int(c[7])
        assert sum(int(n) for n, o in ops if o == "=") == int(c[9])
        assert sum(int(n) for n, o in ops) == int(c[10])
        # '=' columns really are equal bases
        qs = by_name[c[0]].seq
        if c[4] == "-":
            qs = S.revcomp(qs)
            qpos = len(qs) - int(c[3])
        else:
            qpos = int(c[2])
        ts, tpos = by_name[c[5]].seq, int(c[7])
        for n, o in ops:
            n = int(n)
            if o == "=":
                assert (qs[qpos:qpos + n] == ts[tpos:tpos + n]).all()
            if o == "X":
                assert (qs[qpos:qpos + n] != ts[tpos:tpos + n]).all()
            if o in "=XI":
                qpos += n
            if o in "=XD":
                tpos += n
