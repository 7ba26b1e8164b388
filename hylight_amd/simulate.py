"""Seeded synthetic strain-mixture read simulator (SURVEY.md §8d input recipe).

Not part of the reference: the reference ships no test data, so
the synthetic workloads of BASELINE.json (C2..C5) and the small fixtures under
tests/golden/ are generated here.  Deterministic for a given seed.

Model: one uniform-random ACGT ancestor, `n_strains` strains = ancestor + SNPs
(+ optional short indels), log-uniform abundances over one decade, reads with
Gamma(k=4) lengths, per-base substitution / insertion / deletion errors and 50 %
reverse-strand reads.  For every read the simulator keeps the genome coordinate
of each base, so `truth_paf()` can emit the exact pairwise alignment (PAF with a
`cg:Z:` =/X/I/D CIGAR, the format minimap2 `-c --eqx` writes).
"""
from __future__ import annotations

import numpy as np

_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)

SEED_DEFAULT = 20241008


class Read:
    __slots__ = ("name", "seq", "gpos", "strain", "start", "end", "rev")

    def __init__(self, name, seq, gpos, strain, start, end, rev):
        self.name = name      # str
        self.seq = seq        # np.uint8 array, read orientation
        self.gpos = gpos      # np.int64 per read base (read orientation): strain coord or -1 (inserted)
        self.strain = strain
        self.start = start    # strain interval covered [start, end)
        self.end = end
        self.rev = rev        # True if read is the reverse complement of the strain


def make_strains(rng, n_strains, genome_len, snp_rate, indel_rate=0.0):
    """Return list of strain sequences (uint8 arrays).  Strain 0 is the ancestor.  `snp_rate` is one rate or a
    (lo, hi) pair: every strain then draws its own divergence uniformly from that range ("ANI
    98.5-99.5 % (per-strain uniform)")."""
    anc = _BASES[rng.integers(0, 4, size=genome_len)]
    strains = [anc]
    for _ in range(1, n_strains):
        s = anc.copy()
        rate = float(rng.uniform(snp_rate[0], snp_rate[1])) if isinstance(snp_rate, (tuple, list)) else snp_rate
        n_snp = rng.binomial(genome_len, rate)
        pos = rng.choice(genome_len, size=n_snp, replace=False)
        # substitute with a different base
        shift = rng.integers(1, 4, size=n_snp)
        code = np.searchsorted(_BASES, s[pos])
        s[pos] = _BASES[(code + shift) % 4]
        if indel_rate > 0:
            n_id = rng.binomial(genome_len, indel_rate)
            ipos = np.sort(rng.choice(genome_len, size=n_id, replace=False))
            is_del = rng.random(n_id) < 0.5
            s = np.insert(np.delete(s, ipos[is_del]),
                          ipos[~is_del] - np.searchsorted(ipos[is_del], ipos[~is_del]),
                          _BASES[rng.integers(0, 4, size=int((~is_del).sum()))])
        strains.append(s)
    return strains


def _draw_read(rng, g, L, err_sub, err_ins, err_del, rev_frac, keep_gpos):
    """One read of length L (before errors) from strain sequence g: (bases, gpos or None, start, reverse?)."""
    s = int(rng.integers(0, len(g) - L + 1))
    frag = g[s:s + L]
    # errors
    u = rng.random(L)
    is_del = u < err_del
    is_sub = (u >= err_del) & (u < err_del + err_sub)
    frag2 = frag.copy()
    nsub = int(is_sub.sum())
    if nsub:
        code = np.searchsorted(_BASES, frag2[is_sub])
        frag2[is_sub] = _BASES[(code + rng.integers(1, 4, size=nsub)) % 4]
    keep = ~is_del
    # never delete the first/last base (keeps start/end exact)
    keep[0] = keep[-1] = True
    base = frag2[keep]
    gp = (np.arange(s, s + L, dtype=np.int64))[keep] if keep_gpos else None
    n_ins = rng.binomial(len(base), err_ins)
    if n_ins:
        ipos = np.sort(rng.integers(1, len(base), size=n_ins))  # insert before index ipos (never at 0)
        ibase = _BASES[rng.integers(0, 4, size=n_ins)]
        base = np.insert(base, ipos, ibase)
        if keep_gpos:
            gp = np.insert(gp, ipos, -1)
    rev = bool(rng.random() < rev_frac)
    if rev:
        base = _COMP[base[::-1]]
        if keep_gpos:
            gp = gp[::-1].copy()
    return np.ascontiguousarray(base), (np.ascontiguousarray(gp) if keep_gpos else None), s, rev


def _population(seed, n_strains, genome_len, n_reads, mean_len, min_len, max_len, snp_rate, strain_indel_rate):
    rng = np.random.default_rng(seed)
    strains = make_strains(rng, n_strains, genome_len, snp_rate, strain_indel_rate)
    ab = 10.0 ** rng.uniform(0.0, 1.0, size=n_strains)
    ab = ab / ab.sum()
    strain_of = rng.choice(n_strains, size=n_reads, p=ab)
    lens = np.clip(rng.gamma(4.0, mean_len / 4.0, size=n_reads).astype(np.int64), min_len, max_len)
    return rng, strains, strain_of, lens


def simulate_reads(seed=SEED_DEFAULT, n_strains=5, genome_len=400_000, n_reads=10_000,
                   mean_len=8_000, min_len=1_000, max_len=40_000,
                   snp_rate=0.01, strain_indel_rate=0.0,
                   err_sub=0.003, err_ins=0.001, err_del=0.001,
                   rev_frac=0.5, name_prefix="r", keep_gpos=False):
    """Generate reads.  Returns (reads, strains)."""
    rng, strains, strain_of, lens = _population(seed, n_strains, genome_len, n_reads, mean_len, min_len, max_len, snp_rate,
                                                strain_indel_rate)
    reads = []
    for i in range(n_reads):
        st = int(strain_of[i])
        g = strains[st]
        L = int(min(lens[i], len(g)))
        base, gp, s, rev = _draw_read(rng, g, L, err_sub, err_ins, err_del, rev_frac, True)
        reads.append(Read(f"{name_prefix}{i}", base, gp if keep_gpos else None, st, s, s + L, rev))
    return reads, strains


_BLOCK_CTX = None          # (strains, strain_of, lens, recipe): inherited by the forked block workers


def _fasta_block(job):
    b, lo, hi, seed, part = job
    strains, strain_of, lens, (err_sub, err_ins, err_del, rev_frac, name_prefix) = _BLOCK_CTX
    rng = np.random.default_rng([seed, b + 1])
    bases = 0
    with open(part, "wb") as f:
        for i in range(lo, hi):
            g = strains[int(strain_of[i])]
            L = int(min(lens[i], len(g)))
            base, _, _, _ = _draw_read(rng, g, L, err_sub, err_ins, err_del, rev_frac, False)
            f.write(b">" + f"{name_prefix}{i}".encode() + b"\n" + base.tobytes() + b"\n")
            bases += len(base)
    return bases


def simulate_reads_to_fasta(path, seed=SEED_DEFAULT, n_strains=5, genome_len=400_000, n_reads=10_000,
                            mean_len=8_000, min_len=1_000, max_len=40_000, snp_rate=0.01, strain_indel_rate=0.0,
                            err_sub=0.003, err_ins=0.001, err_del=0.001, rev_frac=0.5, name_prefix="r", block=25_000,
                            workers=None):
    """The recipe of simulate_reads for read sets of hundreds of thousands of reads (the full C4 / C5), written straight
    to a 2-line FASTA by a pool of processes: strains, abundances, strain and length of every read come from the same
    seeded stream as in simulate_reads; the per-read draws (position, errors, orientation) of block b of `block` reads
    come from the stream seeded [seed, b + 1].  Deterministic for a given (seed, block).  Returns (n_reads, bases, strains)."""
    import multiprocessing as mp
    import os
    import shutil
    global _BLOCK_CTX
    _, strains, strain_of, lens = _population(seed, n_strains, genome_len, n_reads, mean_len, min_len, max_len, snp_rate,
                                              strain_indel_rate)
    _BLOCK_CTX = (strains, strain_of, lens, (err_sub, err_ins, err_del, rev_frac, name_prefix))
    jobs = [(b, lo, min(lo + block, n_reads), seed, f"{path}.part{b}") for b, lo in enumerate(range(0, n_reads, block))]
    workers = workers or max(1, min(len(jobs), len(os.sched_getaffinity(0)), 16))
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            bases = sum(pool.map(_fasta_block, jobs, chunksize=1))
    finally:
        _BLOCK_CTX = None
    with open(path, "wb") as out:
        for _, _, _, _, part in jobs:
            with open(part, "rb") as f:
                shutil.copyfileobj(f, out, 16 << 20)
            os.remove(part)
    return n_reads, int(bases), strains


def write_fasta(reads, path):
    with open(path, "wb") as f:
        for r in reads:
            f.write(b">" + r.name.encode() + b"\n")
            f.write(r.seq.tobytes() + b"\n")


def write_fastq(reads, path, qual=b"I"):
    with open(path, "wb") as f:
        for r in reads:
            f.write(b"@" + r.name.encode() + b"\n")
            f.write(r.seq.tobytes() + b"\n+\n" + qual * len(r.seq) + b"\n")


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def _pair_alignment(q: Read, t: Read):
    """Exact alignment of query read q against target read t (same strain coordinates).

    Returns None when the reads share no genome column, else a PAF row tuple
    (qs, qe, strand, ts, te, nmatch, blen, cigar_string)."""
    if q.end <= t.start or t.end <= q.start:
        return None
    strand_rev = q.rev != t.rev
    qseq = revcomp(q.seq) if strand_rev else q.seq
    qgp = q.gpos[::-1] if strand_rev else q.gpos
    tseq, tgp = t.seq, t.gpos
    # both lists now traverse the genome in the same direction: ascending if t is forward
    asc = not t.rev
    i = j = 0
    nq, nt = len(qseq), len(tseq)
    cols = []  # (op, qi, tj)

    def ahead(a, b):  # genome coordinate a comes before b in traversal order
        return a < b if asc else a > b

    while i < nq and j < nt:
        a, b = qgp[i], tgp[j]
        if a < 0:
            cols.append(("I", i, j)); i += 1
        elif b < 0:
            cols.append(("D", i, j)); j += 1
        elif a == b:
            cols.append(("=" if qseq[i] == tseq[j] else "X", i, j)); i += 1; j += 1
        elif ahead(a, b):
            cols.append(("I", i, j)); i += 1
        else:
            cols.append(("D", i, j)); j += 1
    # trim to first/last '=' column
    first = next((k for k, c in enumerate(cols) if c[0] == "="), None)
    if first is None:
        return None
    last = len(cols) - 1 - next(k for k, c in enumerate(reversed(cols)) if c[0] == "=")
    cols = cols[first:last + 1]
    qs_, ts_ = cols[0][1], cols[0][2]
    qe_, te_ = cols[-1][1] + 1, cols[-1][2] + 1
    # run-length encode
    ops = []
    for c in cols:
        if ops and ops[-1][0] == c[0]:
            ops[-1][1] += 1
        else:
            ops.append([c[0], 1])
    nmatch = sum(n for o, n in ops if o == "=")
    blen = sum(n for o, n in ops)
    cigar = "".join(f"{n}{o}" for o, n in ops)
    if strand_rev:
        qs, qe = nq - qe_, nq - qs_
    else:
        qs, qe = qs_, qe_
    return qs, qe, "-" if strand_rev else "+", ts_, te_, nmatch, blen, cigar


def truth_paf(reads, min_cols=50, same_strain_only=False, pair_once=True, with_tags=True):
    """All pairwise true alignments as PAF lines (query-major, file order).

    Only valid for strains without indels relative to the ancestor (shared
    coordinate system).  `pair_once` mimics minimap2's dual-skip: a pair is
    reported only with strcmp(qname, tname) < 0."""
    lines = []
    order = sorted(range(len(reads)), key=lambda k: reads[k].start)
    starts = np.array([reads[k].start for k in order])
    for qi, q in enumerate(reads):
        cands = []
        hi = np.searchsorted(starts, q.end, side="left")
        for oi in range(hi):
            ti = order[oi]
            t = reads[ti]
            if ti == qi or t.end <= q.start:
                continue
            if same_strain_only and t.strain != q.strain:
                continue
            if pair_once and not (q.name < t.name):
                continue
            cands.append(ti)
        for ti in sorted(cands):
            t = reads[ti]
            al = _pair_alignment(q, t)
            if al is None:
                continue
            qs, qe, strand, ts, te, nmatch, blen, cigar = al
            if blen < min_cols:
                continue
            row = [q.name, str(len(q.seq)), str(qs), str(qe), strand, t.name, str(len(t.seq)),
                   str(ts), str(te), str(nmatch), str(blen), "0"]
            if with_tags:
                nm = blen - nmatch
                row += [f"NM:i:{nm}", "tp:A:S", f"cg:Z:{cigar}"]
            lines.append("\t".join(row))
    return lines


def messy_graph_paf(seed, n_reads=260, genome=70_000, drop=0.12, trim=0.25, fake=25):
    """Reads + a deliberately imperfect tag-less PAF (dropped rows, one-sided trimmed overlaps, false
    suffix/prefix overlaps, shuffled order): exercises miniasm's tip / bubble / bi-loop / short-overlap
    cleaning, which clean simulated overlaps never trigger."""
    import random
    rnd = random.Random(seed)
    reads, _ = simulate_reads(seed=seed, n_strains=1, genome_len=genome, n_reads=n_reads, mean_len=5000,
                              min_len=2500, max_len=11000, keep_gpos=True, name_prefix="m")
    out = []
    for l in truth_paf(reads, min_cols=500, with_tags=False):
        if rnd.random() < drop:
            continue
        c = l.split("\t")
        if rnd.random() < trim:
            d = rnd.randint(50, 900)
            qs, qe, ts, te = int(c[2]), int(c[3]), int(c[7]), int(c[8])
            if qe - qs > d + 2100 and te - ts > d + 2100:
                if rnd.random() < 0.5:
                    if c[4] == "+":
                        qs += d; ts += d
                    else:
                        qe -= d; ts += d
                else:
                    if c[4] == "+":
                        qe -= d; te -= d
                    else:
                        qs += d; te -= d
                c[2], c[3], c[7], c[8] = map(str, (qs, qe, ts, te))
                c[9] = str(int(c[9]) - d)
                c[10] = str(int(c[10]) - d)
        out.append("\t".join(c))
    names = [r.name for r in reads]
    length = {r.name: len(r.seq) for r in reads}
    for _ in range(fake):
        a, b = rnd.sample(names, 2)
        if not a < b:
            a, b = b, a
        ov = rnd.randint(2100, 3500)
        if length[a] < ov + 10 or length[b] < ov + 10:
            continue
        out.append("\t".join([a, str(length[a]), str(length[a] - ov), str(length[a]), "+", b, str(length[b]), "0",
                              str(ov), str(ov - 20), str(ov), "0"]))
    rnd.shuffle(out)
    return reads, out


def simulate_short_pairs(seed, strains, n_pairs, read_len=250, insert_mean=450.0, insert_sd=27.0, err_sub=0.001,
                         name_prefix="p"):
    """Paired 2 x `read_len` reads, insert size N(mean, sd), substitution errors only ("10 M paired
    2x250 bp (insert N(450,27)), short 0.1 % sub").  Interleaved: read 2i is the forward mate `<name>/1`, read 2i+1
    the reverse-complemented mate `<name>/2` of the same fragment; fragments are drawn from the strains with the same
    log-uniform abundances as the long reads.  Vectorised: the C4 share draws millions of pairs."""
    rng = np.random.default_rng(seed)
    ns = len(strains)
    ab = 10.0 ** rng.uniform(0.0, 1.0, size=ns)
    ab /= ab.sum()
    strain_of = rng.choice(ns, size=n_pairs, p=ab)
    ins = np.maximum(np.rint(rng.normal(insert_mean, insert_sd, size=n_pairs)).astype(np.int64), read_len)
    glen = np.array([len(g) for g in strains], dtype=np.int64)
    start = (rng.random(n_pairs) * (glen[strain_of] - ins + 1)).astype(np.int64)
    flip = rng.random(n_pairs) < 0.5                     # which strand the fragment is read from
    reads = []
    idx = np.arange(read_len, dtype=np.int64)
    for st in range(ns):
        sel = np.nonzero(strain_of == st)[0]
        if not len(sel):
            continue
        g = strains[st]
        left = g[start[sel][:, None] + idx[None, :]]                                     # fragment's first bases
        right = _COMP[g[(start[sel] + ins[sel])[:, None] - 1 - idx[None, :]]]             # revcomp of its last bases
        for block in (left, right):
            e = rng.random(block.shape) < err_sub
            n_e = int(e.sum())
            if n_e:
                code = np.searchsorted(_BASES, block[e])
                block[e] = _BASES[(code + rng.integers(1, 4, size=n_e)) % 4]
        for k, i in enumerate(sel):
            a, b = (right[k], left[k]) if flip[i] else (left[k], right[k])
            s0, s1 = int(start[i]), int(start[i] + ins[i])
            reads.append((int(i), Read(f"{name_prefix}{i}/1", a, None, st, s0, s1, bool(flip[i])),
                          Read(f"{name_prefix}{i}/2", b, None, st, s0, s1, not flip[i])))
    reads.sort(key=lambda x: x[0])
    out = []
    for _, r1, r2 in reads:
        out.append(r1)
        out.append(r2)
    return out


_SHORT_CTX = None          # (strains, strain_of, ins, start, flip, recipe): inherited by the forked block workers


def _short_block(job):
    b, lo, hi, seed, part = job
    strains, strain_of, ins, start, flip, (read_len, err_sub, name_prefix) = _SHORT_CTX
    rng = np.random.default_rng([seed, b + 1])
    idx = np.arange(read_len, dtype=np.int64)
    n = hi - lo
    r1 = np.empty((n, read_len), dtype=np.uint8)
    r2 = np.empty((n, read_len), dtype=np.uint8)
    so = strain_of[lo:hi]
    for st in np.unique(so):
        sel = np.nonzero(so == st)[0]
        g = strains[int(st)]
        a = start[lo:hi][sel]
        left = g[a[:, None] + idx[None, :]]
        right = _COMP[g[(a + ins[lo:hi][sel])[:, None] - 1 - idx[None, :]]]
        for block in (left, right):
            e = rng.random(block.shape) < err_sub
            n_e = int(e.sum())
            if n_e:
                code = np.searchsorted(_BASES, block[e])
                block[e] = _BASES[(code + rng.integers(1, 4, size=n_e)) % 4]
        f = flip[lo:hi][sel]
        r1[sel] = np.where(f[:, None], right, left)
        r2[sel] = np.where(f[:, None], left, right)
    with open(part, "wb") as out:
        for k in range(n):
            i = lo + k
            out.write(b">%s%d/1\n" % (name_prefix, i) + r1[k].tobytes() + b"\n>%s%d/2\n" % (name_prefix, i) + r2[k].tobytes() + b"\n")
    return 2 * n


def simulate_short_pairs_to_fasta(path, seed, strains, n_pairs, read_len=250, insert_mean=450.0, insert_sd=27.0, err_sub=0.001,
                                  name_prefix="p", block=250_000, workers=None):
    """The recipe of simulate_short_pairs for millions of pairs (C4: 5 M pairs), written straight to an interleaved 2-line
    FASTA by a pool of processes: strain, insert size, position and strand of every fragment come from the same seeded
    stream as in simulate_short_pairs, the substitution errors of block b of `block` pairs from the stream seeded
    [seed, b + 1].  Returns the number of reads written."""
    import multiprocessing as mp
    import os
    import shutil
    global _SHORT_CTX
    rng = np.random.default_rng(seed)
    ns = len(strains)
    ab = 10.0 ** rng.uniform(0.0, 1.0, size=ns)
    ab /= ab.sum()
    strain_of = rng.choice(ns, size=n_pairs, p=ab)
    ins = np.maximum(np.rint(rng.normal(insert_mean, insert_sd, size=n_pairs)).astype(np.int64), read_len)
    glen = np.array([len(g) for g in strains], dtype=np.int64)
    start = (rng.random(n_pairs) * (glen[strain_of] - ins + 1)).astype(np.int64)
    flip = rng.random(n_pairs) < 0.5
    _SHORT_CTX = (strains, strain_of, ins, start, flip, (read_len, err_sub, name_prefix.encode()))
    jobs = [(b, lo, min(lo + block, n_pairs), seed, f"{path}.part{b}") for b, lo in enumerate(range(0, n_pairs, block))]
    workers = workers or max(1, min(len(jobs), len(os.sched_getaffinity(0)), 16))
    try:
        with mp.get_context("fork").Pool(workers) as pool:
            n = sum(pool.map(_short_block, jobs, chunksize=1))
    finally:
        _SHORT_CTX = None
    with open(path, "wb") as out:
        for _, _, _, _, part in jobs:
            with open(part, "rb") as f:
                shutil.copyfileobj(f, out, 16 << 20)
            os.remove(part)
    return n


def layout_paf(seed, n_reads=20_000, genome=2_000_000, mean_len=10_000, min_ovl=2_500, jitter=30, dup_frac=0.01,
               fake_frac=0.002, drop=0.05, name_prefix="L"):
    """A large tag-less PAF straight from a read layout (no alignment is computed): every pair of reads whose genome
    intervals share >= min_ovl bases gives one row with the overlap's coordinates on both reads (both strands occur),
    ends jittered by a few bases, a fraction of the rows dropped, some pairs reported twice with shifted coordinates
    and a few false suffix-prefix overlaps between unrelated reads.  Millions of rows in seconds: the scale test of
    the overlap-graph stage (tests/test_gpu_graph_scale.py).  Returns (read_lengths, lines)."""
    rng = np.random.default_rng(seed)
    length = np.clip(rng.gamma(4.0, mean_len / 4.0, size=n_reads).astype(np.int64), 3_000, 40_000)
    start = np.sort(rng.integers(0, genome - 3_000, size=n_reads))
    length = np.minimum(length, genome - start)
    end = start + length
    rev = rng.random(n_reads) < 0.5
    # partners of read i: the reads j > i (by start) that begin at least min_ovl before i ends
    last = np.searchsorted(start, end - min_ovl, side="right")
    cnt = np.maximum(last - (np.arange(n_reads) + 1), 0)
    a = np.repeat(np.arange(n_reads), cnt)
    b = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt) + a + 1
    keep = rng.random(len(a)) >= drop
    a, b = a[keep], b[keep]
    dup = rng.random(len(a)) < dup_frac
    a, b = np.concatenate([a, a[dup]]), np.concatenate([b, b[dup]])
    s = np.maximum(start[a], start[b]) + rng.integers(0, jitter + 1, size=len(a))
    e = np.minimum(end[a], end[b]) - rng.integers(0, jitter + 1, size=len(a))
    ok = e - s >= 2_000
    a, b, s, e = a[ok], b[ok], s[ok], e[ok]

    def on_read(r, s, e):
        return np.where(rev[r], end[r] - e, s - start[r]), np.where(rev[r], end[r] - s, e - start[r])
    qs, qe = on_read(a, s, e)
    ts, te = on_read(b, s, e)
    strand = np.where(rev[a] == rev[b], "+", "-")
    ml = (e - s) - rng.integers(0, 40, size=len(a))
    rows = [f"{name_prefix}{x}\t{length[x]}\t{q0}\t{q1}\t{z}\t{name_prefix}{y}\t{length[y]}\t{t0}\t{t1}\t{m}\t{bl}\t255"
            for x, y, q0, q1, z, t0, t1, m, bl in zip(a.tolist(), b.tolist(), qs.tolist(), qe.tolist(), strand.tolist(),
                                                      ts.tolist(), te.tolist(), ml.tolist(), (e - s).tolist())]
    n_fake = int(len(rows) * fake_frac)
    fa, fb = rng.integers(0, n_reads, size=n_fake), rng.integers(0, n_reads, size=n_fake)
    for x, y in zip(fa.tolist(), fb.tolist()):
        if x == y:
            continue
        ov = int(min(length[x], length[y], 2_100 + int(rng.integers(0, 1_500))))
        rows.append(f"{name_prefix}{x}\t{length[x]}\t{length[x] - ov}\t{length[x]}\t+\t{name_prefix}{y}\t{length[y]}\t0\t{ov}\t"
                    f"{ov - 20}\t{ov}\t255")
    order = rng.permutation(len(rows))
    return length, [rows[i] for i in order.tolist()]
