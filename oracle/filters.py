"""ORACLE (test infrastructure, not product code): CPU restatement of the PAF filter stages.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (hylight_amd/) never does.  Each function restates - in its own
data-structure terms, not line by line - what the cited reference code computes, and is
pinned against golden vectors produced by running the reference itself
(tests/golden/make_goldens.py -> tests/test_oracle_filters.py).

Reference rows:
  a4  script/filter_trans_ovlp_inline_v4.py:31-85     window_filter(variant=4)
  a17 script/filter_trans_ovlp_inline_v3.py:39-102    window_filter(variant=3)
  a2  script/filter_overlap_slr2.py:57                sort_intermediate
  a5  script/filter_overlap_slr2.py:289-367 (long), :229-287 (short)   snp_pileup
  a6  script/filter_overlap_slr2.py:69-71,370-405     supported_pair_counts
  a7  script/filter_overlap_slr2.py:77-161            pass2
  a8  script/utils.py:54,69                           sort_scored
  a18 script/sfo2overlaps.py:18-200                   sfo2overlaps
"""
from __future__ import annotations

import re
from collections import defaultdict

WINDOW = 1000      # filter_trans_ovlp_inline_v4.py:33
QUERY_CAP = 60     # filter_trans_ovlp_inline_v4.py:82

_NUM = re.compile(rb"[ \t]*(-?)([0-9]*)(?:\.([0-9]*))?")
_CG = re.compile(r"(\d+)([=XIDMNSHP])")


# --------------------------------------------------------------------------------------------
# GNU sort emulation (LC_ALL=C): numeric keys "field k .. end of line", last-resort bytewise
# --------------------------------------------------------------------------------------------
def _gnu_numeric(b: bytes):
    """Value GNU `sort -n` sees at the start of `b` as an exact (sign, int, frac) comparable."""
    m = _NUM.match(b)
    sign, ip, fp = m.group(1), m.group(2), m.group(3) or b""
    ip = ip.lstrip(b"0")
    fp = fp.rstrip(b"0")
    mag = (len(ip), ip, fp)           # compare integer part by length then digits, then fraction
    if not ip and not fp:
        return (0, (0, b"", b""))       # zero (also "-", "+", text)
    return (-1, _neg(mag)) if sign else (1, mag)


class _neg:
    """Order-reversing wrapper so that negative magnitudes sort descending."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v

    def __lt__(self, o):
        return self.v > o.v

    def __eq__(self, o):
        return self.v == o.v


def _field_tail(line: bytes, k: int) -> bytes:
    """Bytes from the start of (1-based) tab-separated field k to the end of the line."""
    pos = 0
    for _ in range(k - 1):
        nxt = line.find(b"\t", pos)
        if nxt < 0:
            return b""
        pos = nxt + 1
    return line[pos:]


def gnu_sort(lines, numeric_fields, reverse=False):
    """`sort -n -k<f1> -k<f2> ... [-r]` over text lines (str, no trailing newline)."""
    bl = [l.encode() for l in lines]
    keyed = sorted(bl, key=lambda b: tuple(_gnu_numeric(_field_tail(b, k)) for k in numeric_fields) + (b,),
                   reverse=reverse)
    return [b.decode() for b in keyed]


def sort_intermediate(lines):
    """`sort -nk7 -k8 -k9 -k5` (filter_overlap_slr2.py:57): the global -n applies to every key."""
    return gnu_sort(lines, (7, 8, 9, 5))


def sort_scored(lines):
    """`sort -k12 -nr` (utils.py:54,69): numeric descending on the score column."""
    return gnu_sort(lines, (12,), reverse=True)


# --------------------------------------------------------------------------------------------
# shared row helpers
# --------------------------------------------------------------------------------------------
def _cols(line):
    c = line.split("\t")
    return c[0], int(c[1]), int(c[2]), int(c[3]), c[4], c[5], int(c[6]), int(c[7]), int(c[8]), int(c[9]), int(c[10])


def _is_internal(ql, qs, qe, strand, tl, ts, te, min_o):
    """Overhang test of minimap Alg. 5 as both filters apply it (v4:52-66, slr2:116-131)."""
    if strand == "-":
        ts, te = tl - te, tl - ts
    overhang = min(qs, ts) + min(ql - qe, tl - te)
    maplen = max(qe - qs, te - ts)
    return overhang > min(min_o, maplen * 0.8)


def pair_key(a, b):
    return (a, b) if a <= b else (b, a)


def to_sfo(q, ql, qs, qe, strand, t, tl, ts, te, mc, ln):
    """PAF row -> SFO row (filter_trans_ovlp_inline_v3.py:83-102)."""
    if strand == "+":
        ori, oha, ohb = "N", qs - ts, tl - ts - (ql - qs)
    else:
        ori, oha, ohb = "I", qs - (tl - te), te - (ql - qs)
    ola = min(ql - oha, tl) if oha >= 0 else min(tl + oha, ql)
    return "\t".join(map(str, (q, t, ori, oha, ohb, ola, ola, ln - mc)))


# --------------------------------------------------------------------------------------------
# a4 / a17 : streaming window filter
# --------------------------------------------------------------------------------------------
def window_filter(lines, variant=4, min_len=60, min_iden=None, min_o=0, sfo=False):
    """Windows of 1000 input rows, state reset per window.

    variant 4: predicates, overhang test, first row per unordered pair, then at most 60
               printed rows per query (the per-query counter advances for every row that
               reaches it, printed or not).
    variant 3: pair de-duplication happens BEFORE the overhang test; no per-query cap;
               output is SFO (sfo=True) or "q t score" lines.
    """
    if min_iden is None:
        min_iden = 0.6 if variant == 4 else 0.8
    out = []
    for w0 in range(0, len(lines), WINDOW):
        seen, per_query = set(), defaultdict(int)
        for line in lines[w0:w0 + WINDOW]:
            q, ql, qs, qe, strand, t, tl, ts, te, mc, ln = _cols(line)
            if ln < min_len or mc / ln < min_iden or q == t:
                continue
            pk = pair_key(q, t)
            if variant == 3:
                if pk in seen:
                    continue
                seen.add(pk)
                if _is_internal(ql, qs, qe, strand, tl, ts, te, min_o):
                    continue
                if sfo:
                    out.append(to_sfo(q, ql, qs, qe, strand, t, tl, ts, te, mc, ln))
                else:
                    mlen = (ql + tl) / 2
                    out.append("\t".join([q, t, str(0.1 * (ln / mlen) + 0.9 * (mc / ln))]))
                continue
            if _is_internal(ql, qs, qe, strand, tl, ts, te, min_o):
                continue
            if pk in seen:
                continue
            seen.add(pk)
            per_query[q] += 1
            if per_query[q] > QUERY_CAP:
                continue
            out.append(line)
    return out


# --------------------------------------------------------------------------------------------
# a5 : SNP pile-up from CIGAR X runs
# --------------------------------------------------------------------------------------------
def cigar_ops(last_field):
    """(len, op) list of a `cg:Z:` field; [] for anything else (incl. '*')."""
    s = last_field.strip()
    if not s.startswith("cg:Z:"):
        return []
    return [(int(n), o) for n, o in _CG.findall(s[5:])]


def snp_pileup(sorted_lines, long_mode=True):
    """Returns (snp, partners, intervals):
       snp[(read,pos)]      = number of X runs ending at that key
       partners[(read,pos)] = list of the other read of each such run
       intervals[read]      = [(start,end), ...] aligned intervals recorded for the read
    long_mode (prpare_mutation2): first row per unordered pair only; both reads get keys and
    intervals; query key is in forward-query coordinates (+1 on the minus strand).
    short mode (prpare_mutation): every non-self row; target side only; I ops ignored."""
    snp, partners, intervals = defaultdict(int), defaultdict(list), defaultdict(list)
    used = set()
    for line in sorted_lines:
        c = line.split("\t")
        if len(c) < 6:
            continue
        q, t = c[0], c[5]
        # slr2:318 strips the field before the "*" test, slr2:253 (short mode) does not, so in
        # short mode a "*\n" row still records its interval
        if q == t or (c[-1].strip() if long_mode else c[-1]) == "*":
            continue
        if long_mode:
            pk = pair_key(q, t)
            if pk in used:
                continue
            used.add(pk)
        ql, qs, qe, ts, te = int(c[1]), int(c[2]), int(c[3]), int(c[7]), int(c[8])
        minus = c[4] != "+"
        intervals[t].append((ts, te))
        if long_mode:
            intervals[q].append((qs, qe))
        qpos = (ql - qe) if minus else qs
        tpos = ts
        for n, op in cigar_ops(c[-1]):
            if op == "=":
                qpos += n; tpos += n
            elif op == "I":
                qpos += n
            elif op == "D":
                tpos += n
            elif op == "X":
                qpos += n; tpos += n
                kt = (t, tpos)
                snp[kt] += 1; partners[kt].append(q)
                if long_mode:
                    kq = (q, ql - qpos + 1) if minus else (q, qpos)
                    snp[kq] += 1; partners[kq].append(t)
    return snp, partners, intervals


# --------------------------------------------------------------------------------------------
# a6 : coverage-supported SNPs -> per-pair disagreement counts
# --------------------------------------------------------------------------------------------
def supported_pair_counts(snp, partners, intervals, mc):
    """mutation[pair] = number of supported SNP keys at which the pair disagrees.
    A key with v >= mc supporters is kept when at least mc further reads span it strictly
    (#intervals with start < pos < end, minus v)."""
    from bisect import bisect_left, bisect_right
    mutation = defaultdict(int)
    # every recorded interval has start < end (aligned spans), so the intervals with end <= pos are among those with
    # start < pos and  #(start < pos < end) = #(start < pos) - #(end <= pos): two searches in sorted lists instead of a
    # scan of the read's intervals per key (a target of a full-size chunk has ~10^3 intervals and ~10^4 keys)
    starts, ends, odd = {}, {}, set()
    for read, iv in intervals.items():
        if any(s >= e for s, e in iv):
            odd.add(read)                 # (hand-made rows only: counted by the definition below)
        starts[read] = sorted(s for s, _ in iv)
        ends[read] = sorted(e for _, e in iv)
    for (read, pos), v in snp.items():
        if v < mc:
            continue
        if read in odd:
            spanning = sum(1 for s, e in intervals.get(read, ()) if s < pos < e)
        else:
            spanning = bisect_left(starts.get(read, ()), pos) - bisect_right(ends.get(read, ()), pos)
        if spanning - v < mc:
            continue
        for other in partners[(read, pos)]:
            mutation[pair_key(read, other)] += 1
    return mutation


# --------------------------------------------------------------------------------------------
# a7 : pass 2 - predicates, scores, 14-column rows
# --------------------------------------------------------------------------------------------
def x_digit_sum(last_field):
    """Sum of the single digit preceding every 'X' (slr2:156-161): a 12X run counts 2."""
    return sum(int(last_field[i - 1]) for i, ch in enumerate(last_field) if ch == "X")


def pass2(sorted_lines, mutation, long_mode, min_ovlp_len, iden, threshold=0.0025, min_o=4):
    out, emitted = [], set()
    for line in sorted_lines:
        c = line.split("\t")
        q, ql, qs, qe, strand, t, tl, ts, te, mc, ln = _cols(line)
        pk = pair_key(q, t)
        if pk in mutation and (not long_mode or mutation[pk] / mc > threshold):
            continue
        if q == t or mc < min_ovlp_len:
            continue
        if _is_internal(ql, qs, qe, strand, tl, ts, te, min_o):
            continue
        if pk in emitted:
            continue
        emitted.add(pk)
        mis = x_digit_sum(c[-1]) / mc
        score = format(0.4 * (mc / ((ql + tl) / 2)) + 0.6 * (mc / ln), ".4f")
        score2 = format(1 - mis, ".4f")
        score3 = format(mc / ln, ".4f")
        if float(score2) < iden:
            continue
        out.append("\t".join([q, str(ql), str(qs), str(qe), strand, t, str(tl), str(ts), str(te),
                              str(mc), str(ln), score, score2, score3, ""]))
    return out


def pair_counts_np(sorted_lines, long_mode, mc):
    """snp_pileup + supported_pair_counts in one go with numpy - the SAME quantities (those two functions are the
    definition and stay pinned to the reference's goldens; tests/test_oracle_filters.py holds this one to them on the golden
    inputs and on simulated chunks).  A full-size chunk is ~4e5 rows and ~6e7 X events: minutes per chunk in the per-event
    Python loops, seconds here.  Returns None when a CIGAR is not plain `<digits><=XIDMNSHP>...` text or an interval has no
    extent (the caller then takes the definitional path)."""
    import numpy as np
    ids, names = {}, []

    def rid(name):
        i = ids.get(name)
        if i is None:
            i = ids[name] = len(names)
            names.append(name)
        return i

    used = set()
    rq, rt, rql, rqs, rqe, rts, rte, rminus, cigs = [], [], [], [], [], [], [], [], []
    for line in sorted_lines:                                   # row selection: as snp_pileup
        c = line.split("\t")
        if len(c) < 6:
            continue
        q, t = c[0], c[5]
        if q == t or (c[-1].strip() if long_mode else c[-1]) == "*":
            continue
        if long_mode:
            pk = pair_key(q, t)
            if pk in used:
                continue
            used.add(pk)
        last = c[-1].strip()
        rq.append(rid(q)); rt.append(rid(t)); rql.append(int(c[1])); rqs.append(int(c[2])); rqe.append(int(c[3]))
        rts.append(int(c[7])); rte.append(int(c[8])); rminus.append(c[4] != "+")
        cigs.append(last[5:].encode() if last.startswith("cg:Z:") else b"")
    if not rq:
        return {}
    rq, rt, rql, rqs, rqe, rts, rte = (np.asarray(x, dtype=np.int64) for x in (rq, rt, rql, rqs, rqe, rts, rte))
    rminus = np.asarray(rminus, dtype=bool)
    if (rts >= rte).any() or (long_mode and (rqs >= rqe).any()):
        return None
    big = np.frombuffer(b"".join(cigs), dtype=np.uint8)
    row_off = np.zeros(len(cigs) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in cigs], out=row_off[1:])
    if len(big) and big.min() < 48:
        return None
    L = np.flatnonzero(big > 57)                                 # one op letter per CIGAR op (digits are 48..57)
    if not len(L):
        return {}
    op = big[L]
    lut = np.zeros(256, dtype=bool)
    lut[np.frombuffer(b"=XIDMNSHP", dtype=np.uint8)] = True
    if not lut[op].all():
        return None
    ndig = np.empty(len(L), dtype=np.int64)
    ndig[0] = L[0]
    ndig[1:] = L[1:] - L[:-1] - 1
    if ndig.min() < 1 or ndig.max() > 9:
        return None
    nonempty = row_off[1:] > row_off[:-1]
    if (big[row_off[1:][nonempty] - 1] <= 57).any():             # a CIGAR ends with its last op letter
        return None
    n = big[L - 1].astype(np.int32) - 48
    for p in range(2, int(ndig.max()) + 1):
        m = np.flatnonzero(ndig >= p)
        n[m] += (big[L[m] - p].astype(np.int32) - 48) * np.int32(10 ** (p - 1))
    first = np.searchsorted(L, row_off[:-1], side="left")        # index of each row's first op (len(L): none)
    # positions inside a row: cumulated op lengths minus their value at the row's first op; 32-bit sums that wrap
    # around give the right DIFFERENCES (a row spans far less than 2^31 bases)
    isx = op == ord("X")
    adv = (op == ord("=")) | isx
    dq = np.where(adv | (op == ord("I")), n, np.int32(0))
    dt = np.where(adv | (op == ord("D")), n, np.int32(0))
    del adv
    cq, ct = np.cumsum(dq, dtype=np.int32), np.cumsum(dt, dtype=np.int32)
    del dq, dt
    xi = np.flatnonzero(isx)
    xr = np.searchsorted(first, xi, side="right") - 1            # row of every X op (rows without ops share `first` with the next)
    has_ops = first < np.concatenate((first[1:], [len(L)]))
    if not has_ops.all():                                        # map through the rows that have ops
        rows_with = np.flatnonzero(has_ops)
        xr = rows_with[np.searchsorted(first[rows_with], xi, side="right") - 1]
    f = first[xr]
    cq0 = np.where(f > 0, cq[np.maximum(f, 1) - 1], np.int32(0))
    ct0 = np.where(f > 0, ct[np.maximum(f, 1) - 1], np.int32(0))
    q0 = np.where(rminus, rql - rqe, rqs)
    qpos = q0[xr] + (cq[xi] - cq0).astype(np.int64)
    tpos = rts[xr] + (ct[xi] - ct0).astype(np.int64)
    del cq, ct
    ev_read, ev_pos, ev_other = rt[xr], tpos, rq[xr]
    iv_read, iv_s, iv_e = rt, rts, rte
    if long_mode:
        qkey = np.where(rminus[xr], rql[xr] - qpos + 1, qpos)
        ev_read = np.concatenate((ev_read, rq[xr])); ev_pos = np.concatenate((ev_pos, qkey)); ev_other = np.concatenate((ev_other, rt[xr]))
        iv_read = np.concatenate((iv_read, rq)); iv_s = np.concatenate((iv_s, rqs)); iv_e = np.concatenate((iv_e, rqe))
    if not len(ev_read):
        return {}
    if ev_pos.min() < 0 or ev_pos.max() >= 1 << 31 or iv_s.min() < 0 or iv_e.max() >= 1 << 31:
        return None
    key = ev_read << 32 | ev_pos
    uk, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    sk = np.sort(iv_read << 32 | iv_s)
    ek = np.sort(iv_read << 32 | iv_e)
    ur = uk >> 32 << 32
    spanning = (np.searchsorted(sk, uk, side="left") - np.searchsorted(sk, ur, side="left")) \
        - (np.searchsorted(ek, uk, side="right") - np.searchsorted(ek, ur, side="left"))
    good = (cnt >= mc) & (spanning - cnt >= mc)
    ge = good[inv]
    a, b = ev_read[ge], ev_other[ge]
    pu, pc = np.unique(np.minimum(a, b) << 32 | np.maximum(a, b), return_counts=True)
    return {pair_key(names[int(k) >> 32], names[int(k) & 0xffffffff]): int(v) for k, v in zip(pu, pc)}


def worker(raw_lines, long_mode, min_ovlp_len, mc, iden, fast=True, threshold=0.0025, min_o=4):
    """One chunk: filter_overlap_slr2.main after the overlapper (slr2:51-152).  fast=False: the pile-up by the
    definitional per-event functions (snp_pileup, supported_pair_counts) instead of their numpy restatement.
    threshold / min_o: the script's -thre and -oh (slr2:24,26; the stage never overrides their defaults)."""
    return worker_sweep(raw_lines, long_mode, min_ovlp_len, mc, iden, [threshold], fast, min_o)[threshold]


def worker_sweep(raw_lines, long_mode, min_ovlp_len, mc, iden, thresholds, fast=True, min_o=4):
    """worker() for several values of -thre at once: window filter, intermediate sort and pile-up are the same for all of
    them (the pair counts of slr2:370-405 enter pass 2 only through `count / matchcount > thre`, slr2:90-96)."""
    kept = window_filter(raw_lines, variant=4, min_len=30, min_o=3)
    srt = sort_intermediate(kept)
    mutation = pair_counts_np(srt, long_mode, mc) if fast else None
    if mutation is None:
        snp, partners, intervals = snp_pileup(srt, long_mode)
        mutation = supported_pair_counts(snp, partners, intervals, mc)
    return {t: pass2(srt, mutation, long_mode, min_ovlp_len, iden, threshold=t, min_o=min_o) for t in thresholds}


def chunk_ranges(n_lines, nsplit):
    """Line ranges produced by `split -l int(nu/(8*nsplit)+1)*8` (utils.py:44-47)."""
    per = int(n_lines / (8 * nsplit) + 1) * 8
    return [(s, min(s + per, n_lines)) for s in range(0, n_lines, per)]


def stage(chunk_raw_lines, long_mode, min_ovlp_len, mc, iden):
    """split_reads2 after chunking: per-chunk worker + per-chunk sort + merged sort (utils.py:54-69)."""
    allrows = []
    for raw in chunk_raw_lines:
        allrows += sort_scored(worker(raw, long_mode, min_ovlp_len, mc, iden))
    return sort_scored(allrows)


# --------------------------------------------------------------------------------------------
# the short-read cluster path's filter and converter
# --------------------------------------------------------------------------------------------
def ovlp_inline_filter(lines, min_ovlp_len, min_identity, o=1000, r=0.8):
    """script/filter_ovlp_inline.py:12-106.  Per window of 1000 rows: drop short / divergent / internal rows
    (rm_intermatch), then drop self hits and keep, per unordered pair, the LONGEST overlap (column 11; the
    earlier row wins ties) - printed at the position where the pair first appeared (rm_dupovlp)."""
    out = []
    for w0 in range(0, len(lines), WINDOW):
        best, order = {}, []
        for line in lines[w0:w0 + WINDOW]:
            q, ql, qs, qe, strand, t, tl, ts, te, mc, ln = _cols(line)
            if ln < min_ovlp_len or mc / ln < min_identity:
                continue
            if strand == "-":
                ts, te = tl - te, tl - ts
            if min(qs, ts) + min(ql - qe, tl - te) > min(o, max(qe - qs, te - ts) * r):
                continue
            if q == t:
                continue
            pk = pair_key(q, t)
            if pk not in best:
                best[pk] = (ln, line)
                order.append(pk)
            elif ln > best[pk][0]:
                best[pk] = (ln, line)
        out += [best[pk][1] for pk in order]
    return out


def minimap22sfo(lines, min_overlap_len=0, min_pident=0.0):
    """script/minimap22sfo.py:28-75: PAF -> SFO with the ids put in (string) order."""
    out = []
    for line in lines:
        q, ql, qs, qe, strand, t, tl, ts, te, mc, ln = _cols(line)
        if ln < min_overlap_len or mc / float(ln) < min_pident / 100.0:
            continue
        if strand == "+":
            ori, oha, ohb = "N", qs - ts, tl - ts - (ql - qs)
        else:
            ori, oha, ohb = "I", qs - (tl - te), te - (ql - qs)
        ola = min(ql - oha, tl) if oha >= 0 else min(tl + oha, ql)
        a, b = q, t
        if a > b:
            a, b = b, a
            oha, ohb = (-oha, -ohb) if ori == "N" else (ohb, oha)
        out.append("\t".join(map(str, (a, b, ori, oha, ohb, ola, ola, ln - mc))))
    return out


# --------------------------------------------------------------------------------------------
# a18 : SFO -> SAVAGE overlaps (single-end reads only: HyLight passes --num_pairs 0)
# --------------------------------------------------------------------------------------------
def sfo2overlaps(sfo_lines):
    rows = []
    for line in sfo_lines:
        a, b, ori, oha, ohb, ola, olb, k = line.split()
        ia, ib = int(a), int(b)
        if ia > ib:  # canonical id order (sfo2overlaps.py:41-47,112-122)
            if ori == "I":
                a, b, oha, ohb, ola, olb = b, a, ohb, oha, olb, ola
            else:
                a, b, oha, ohb, ola, olb = b, a, str(-int(oha)), str(-int(ohb)), olb, ola
            ia, ib = ib, ia
            body = "\t".join([a, b, ori, oha, ohb, ola, olb, k])
        else:
            body = line.rstrip("\n")
        rows.append(f"{ia}\t{ib}\t{body}")
    # sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq   (sfo2overlaps.py:53)
    bl = sorted((r.encode() for r in rows),
                key=lambda b: tuple(_gnu_numeric(b.split(b"\t")[i]) for i in range(4)) + (b,))
    uniq = [b.decode() for i, b in enumerate(bl) if i == 0 or b != bl[i - 1]]
    out = []
    for r in uniq:
        c = r.split()
        ida, idb = c[0], c[1]
        if int(ida) == int(idb):
            continue
        oha, ohb, ola, olb = int(c[5]), int(c[6]), int(c[7]), int(c[8])
        ori = "+" if c[4] == "N" else "-"
        ovlen = min(ola, olb)
        if oha >= 0:
            lena = ola + oha + (0 if ohb >= 0 else -ohb)
            lenb = olb + ohb if ohb >= 0 else olb
            id1, id2, pos1, ori1, ori2 = ida, idb, str(oha), "+", ori
        else:
            lena = ola if ohb >= 0 else ola - ohb
            lenb = -oha + olb + (ohb if ohb >= 0 else 0)
            id1, id2, pos1, ori1, ori2 = idb, ida, str(-oha), ori, "+"
        perc = min(round(100 * ovlen / min(lena, lenb)), 100)   # Python round = half-to-even
        out.append("\t".join([id1, id2, pos1, "-", "-", ori1, ori2, "{:.0f}".format(perc), "-",
                              str(ovlen), "-", "s", "s"]))
    return [l for i, l in enumerate(out) if i == 0 or l != out[i - 1]]   # final `uniq`
