/* ORACLE (test infrastructure, NOT product code): sequential CPU restatement of the overlapper
 * specification (DESIGN.md "Overlapper spec").
 *
 * PARITY UNPINNED against minimap2: the reference calls an external, un-vendored, un-pinned
 * `minimap2` (lh3/minimap2; bioconda "latest", 2.28 series at the reference date) at
 * script/filter_overlap_slr2.py:51 with `-x ava-pb -Hk19 -m100 -g10000 --max-chain-skip 25 -c
 * --eqx`; no source, binary or golden PAF exists under /root/reference.  This file restates the
 * PUBLISHED algorithm (Li 2016 Alg. 1-2, Li 2018 section 2.1) with the integer / fixed-window
 * choices listed in DESIGN.md, and is the bit-exact oracle for the HIP kernels - not for
 * minimap2.  Known answers pinned in tests/test_oracle_ava.py: the invertible hash, HPC edge
 * cases, simulator truth (pair, strand, coordinates).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int k, w, hpc, min_chain_score, max_gap, bandwidth, min_cnt, min_mid_occ;
    double mid_occ_frac;
    int match, mismatch, gap_open, gap_ext, ambi;
    int min_dp_score, end_bonus, pair_once;
    int gap_open2, gap_ext2;   /* second piece of the gap cost: a gap of L bases costs min(open + ext L, open2 + ext2 L); open2 <= 0: one piece */
    int stub_oh;               /* >= 0: the rows go to a consumer that drops internal matches with this overhang bound
                                * (filter_trans_ovlp_inline_v4.py:64, -oh 3): pieces that are certain to be reported and certain
                                * to be dropped there are written without their end extensions; < 0: off */
    int zdrop;                 /* > 0: an end extension stops after a row i (checked at every 32nd row) whose best cell lies more than
                                * this below the best cell so far - minimap2's -z (400 for the ava-pb preset); 0: no z-drop */
} ava_opts_t;   /* same layout as hlmi_ava_opts (include/hylight_mi.h) */

/* ---- fixed constants of the spec -------------------------------------------- */
#define CHAIN_PRED   64      /* predecessors examined per anchor                            */
#define BLOCK_MIN    32      /* min distance between alignment fixed points                 */
#define BLOCK_MAX    256     /* rows / cols up to which a block takes the band rule below; longer blocks: LONG blocks */
#define BAND_W       64      /* diagonals per block                                         */
#define BAND_PAD     12      /* padding around [min(0,delta), max(0,delta)]                 */
#define NARROW_W     16      /* blocks with |delta| <= NARROW_DELTA use a 16-diagonal band  */
#define NARROW_PAD   5
#define NARROW_DELTA 5
#define EXT_MAX      256     /* rows of an end extension: max(EXT_MAX, max_gap) (ext_rows())   */
#define ZDROP_STEP   32      /* the z-drop test runs after rows 32, 64, ...                 */
#define HALF_W       32      /* diagonals of a LONG extension and of a LONG block with |delta| <= HALF_DELTA */
#define HALF_DELTA   7
#define MAX_MID_OCC  1000000
#define NEG_INF      (-(1 << 29))

/* Blocks and extensions.  minimap2 -c -g10000 fills the gap between two chained anchors whatever its
 * length and extends chain ends until a z-drop or max_gap bases; so does the specification:
 *   - a block may have any number of rows / columns (a chain link spans at most max_gap); only a diagonal shift above
 *     BAND_W - 2 BAND_PAD - 1 = 39 between two fixed points still splits a chain into two pieces (the band has 64 diagonals);
 *     blocks with more than BLOCK_MAX rows or columns (LONG blocks) take a band centred on the two corners' diagonals: 32
 *     diagonals when the shift is at most 7 (pad >= 12), else 64;
 *   - an end extension runs over up to max(EXT_MAX, max_gap) rows and stops at a z-drop (ava_opts_t::zdrop); one that can run
 *     more than EXT_MAX rows (LONG extension: min(rows, columns + 31) > EXT_MAX) takes a band of 32 diagonals (-15 .. +16)
 *     instead of 64 (-31 .. +32).
 * MEASUREMENT switches (tests/test_deviation_effects.py only; the specification is the defaults):
 *   ORACLE_BLOCK_MAX   rows / cols above which a block splits its chain (none; 256 = the specification up to round 3)
 *   ORACLE_SHIFT_MAX   diagonal shift of one block (39); wider blocks get a band of shift + 2 pad + 1
 *   ORACLE_EXT_MAX     rows of an end extension (max(256, max_gap); 256 = the specification up to round 3)
 *   ORACLE_EXT_BAND    diagonals of an end extension (64)
 */
static int g_block_max = 1 << 30, g_shift_max = BAND_W - 2 * BAND_PAD - 1, g_ext_max = 0, g_ext_band = BAND_W;
static int g_chain_mm2;
static int g_ungapped;       /* bandwidth == 0 (minimap2 -r 0, script/HyLight.py:309): the DP band is the diagonal alone */
static long g_n_pieces, g_n_stubs;      /* pieces reported / of them as stubs (oracle_last_counts; atomic adds) */
static void read_switches(void) {
    const char *e;
    g_block_max = (e = getenv("ORACLE_BLOCK_MAX")) ? atoi(e) : 1 << 30;
    g_shift_max = (e = getenv("ORACLE_SHIFT_MAX")) ? atoi(e) : BAND_W - 2 * BAND_PAD - 1;
    g_ext_max = (e = getenv("ORACLE_EXT_MAX")) ? atoi(e) : 0;
    g_ext_band = (e = getenv("ORACLE_EXT_BAND")) ? atoi(e) : BAND_W;
    g_chain_mm2 = getenv("ORACLE_CHAIN_MM2") ? 1 : 0;
}
/* rows an end extension may run over */
static int ext_rows(const ava_opts_t *o) { return g_ext_max > 0 ? g_ext_max : (o->max_gap > EXT_MAX ? o->max_gap : EXT_MAX); }

/* ---- sequences ---------------------------------------------------------------------------- */
typedef struct {
    int n;
    char **name;
    uint8_t **code;   /* 0..3 ACGT, 4 other */
    int *len;
    int *rank;        /* strcmp rank of the name among all names of both sets */
} seqset_t;

static int nt4(int c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

static void seqset_free(seqset_t *s) {
    for (int i = 0; i < s->n; ++i) { free(s->name[i]); free(s->code[i]); }
    free(s->name); free(s->code); free(s->len); free(s->rank);
}

/* FASTA/FASTQ, multi-line tolerant; name = header up to first blank */
static int seqset_read(const char *path, seqset_t *s) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *d = (char *)malloc(sz + 1);
    if (fread(d, 1, sz, f) != (size_t)sz) { fclose(f); free(d); return -1; }
    d[sz] = 0;
    fclose(f);
    memset(s, 0, sizeof *s);
    int cap = 0;
    long p = 0;
    while (p < sz) {
        long e = p;
        while (e < sz && d[e] != '\n') ++e;
        if (d[p] == '>' || d[p] == '@') {
            int fq = d[p] == '@';
            if (s->n == cap) {
                cap = cap ? cap * 2 : 1024;
                s->name = (char **)realloc(s->name, cap * sizeof(char *));
                s->code = (uint8_t **)realloc(s->code, cap * sizeof(uint8_t *));
                s->len = (int *)realloc(s->len, cap * sizeof(int));
            }
            long q = p + 1;
            while (q < e && !isspace((unsigned char)d[q])) ++q;
            s->name[s->n] = strndup(d + p + 1, q - p - 1);
            /* sequence lines */
            long b = e + 1, tot = 0, capb = 0;
            uint8_t *buf = 0;
            while (b < sz && d[b] != '>' && d[b] != '@' && d[b] != '+') {
                long e2 = b;
                while (e2 < sz && d[e2] != '\n') ++e2;
                long l = e2 - b;
                if (l && d[e2 - 1] == '\r') --l;
                if (tot + l > capb) { capb = (tot + l) * 2 + 16; buf = (uint8_t *)realloc(buf, capb); }
                for (long i = 0; i < l; ++i) buf[tot + i] = (uint8_t)nt4(d[b + i]);
                tot += l;
                b = e2 + 1;
            }
            if (fq && b < sz && d[b] == '+') {
                while (b < sz && d[b] != '\n') ++b;
                ++b;
                long ql = 0;
                while (b < sz && ql < tot) {
                    long e2 = b;
                    while (e2 < sz && d[e2] != '\n') ++e2;
                    ql += e2 - b;
                    b = e2 + 1;
                }
            }
            s->code[s->n] = buf ? buf : (uint8_t *)malloc(1);
            s->len[s->n] = (int)tot;
            ++s->n;
            p = b;
        } else p = e + 1;
    }
    free(d);
    return 0;
}

/* ---- S1: sketch ----------------------------------------------------------------------------- */
/* Thomas Wang's invertible integer hash as minimap2 uses it (Li 2016, section 2.2) */
uint64_t oracle_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

typedef struct { uint64_t x, y; } mz_t;   /* x = hash<<8 | span ; y = rid<<32 | pos<<1 | strand */

/* Minimizers of one read, position order.  Returns the count (<= cap written). */
int64_t oracle_sketch_codes(const uint8_t *c, int len, uint32_t rid, int k, int w, int hpc, mz_t *out, int64_t cap) {
    if (len <= 0) return 0;
    const uint64_t mask = (1ULL << 2 * k) - 1, shift = 2 * (k - 1);
    uint64_t *sx = (uint64_t *)malloc((size_t)len * 8);    /* slot key (x) or UINT64_MAX */
    uint32_t *sp = (uint32_t *)malloc((size_t)len * 4);    /* slot position<<1 | strand  */
    int32_t *sl = (int32_t *)malloc((size_t)len * 4);      /* symbols since the last ambiguous base */
    int *runq = (int *)calloc(k, sizeof(int));
    int ns = 0, l = 0, qn = 0, qh = 0, span = 0;
    uint64_t fwd = 0, rev = 0;
    for (int i = 0; i < len; ++i) {
        int b = c[i];
        uint64_t x = UINT64_MAX;
        uint32_t pz = 0;
        if (b < 4) {
            int run = 1;
            if (hpc) {
                while (i + run < len && c[i + run] == b) ++run;
                i += run - 1;
            }
            if (hpc) {
                if (qn == k) { span -= runq[qh]; runq[qh] = run; qh = (qh + 1) % k; }
                else { runq[(qh + qn) % k] = run; ++qn; }
                span += run;
            } else span = l + 1 < k ? l + 1 : k;
            fwd = (fwd << 2 | (uint64_t)b) & mask;
            rev = (rev >> 2) | (3ULL ^ (uint64_t)b) << shift;
            ++l;
            if (l >= k && span < 256 && fwd != rev) {
                int z = fwd < rev ? 0 : 1;
                x = oracle_hash64(z ? rev : fwd, mask) << 8 | (uint64_t)span;
                pz = (uint32_t)i << 1 | (uint32_t)z;
            }
        } else {
            l = 0; qn = 0; qh = 0; span = 0;
        }
        sx[ns] = x; sp[ns] = pz; sl[ns] = l; ++ns;
    }
    int64_t n = 0;
    for (int j = 0; j < ns; ++j) {
        if (sx[j] == UINT64_MAX) continue;
        int sel = 0;
        for (int s = j; s < j + w && s < ns && !sel; ++s) {      /* windows [s-w+1, s] containing j */
            if (sl[s] < w + k - 1 || s - w + 1 < 0) continue;      /* window not eligible */
            uint64_t mn = UINT64_MAX;
            for (int t = s - w + 1; t <= s; ++t) if (sx[t] < mn) mn = sx[t];
            if (mn == sx[j]) sel = 1;
        }
        if (sel) {
            if (n < cap) { out[n].x = sx[j]; out[n].y = (uint64_t)rid << 32 | sp[j]; }
            ++n;
        }
    }
    free(sx); free(sp); free(sl); free(runq);
    return n;
}

int64_t oracle_sketch(const char *seq, int len, uint32_t rid, int k, int w, int hpc, uint64_t *out_xy, int64_t cap) {
    uint8_t *c = (uint8_t *)malloc(len > 0 ? len : 1);
    for (int i = 0; i < len; ++i) c[i] = (uint8_t)nt4(seq[i]);
    int64_t n = oracle_sketch_codes(c, len, rid, k, w, hpc, (mz_t *)out_xy, cap);
    free(c);
    return n;
}

/* ---- S2: index of the target chunk ---------------------------------------------------------- */
typedef struct { uint64_t key; uint64_t y; } ient_t;   /* key = hash (x>>8) */

static int cmp_ient(const void *a, const void *b) {
    const ient_t *x = (const ient_t *)a, *y = (const ient_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->y < y->y ? -1 : x->y > y->y ? 1 : 0;
}
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

typedef struct {
    ient_t *e;
    int64_t n;
    int mid_occ;
} index_t;

static void index_build(const seqset_t *T, const ava_opts_t *o, index_t *ix) {
    int64_t cap = 0;
    for (int i = 0; i < T->n; ++i) cap += T->len[i];
    mz_t *tmp = (mz_t *)malloc((size_t)(cap + 1) * sizeof(mz_t));
    ix->e = (ient_t *)malloc((size_t)(cap + 1) * sizeof(ient_t));
    ix->n = 0;
    for (int i = 0; i < T->n; ++i) {
        int64_t m = oracle_sketch_codes(T->code[i], T->len[i], (uint32_t)i, o->k, o->w, o->hpc, tmp, cap);
        for (int64_t j = 0; j < m; ++j) { ix->e[ix->n].key = tmp[j].x >> 8; ix->e[ix->n].y = tmp[j].y; ++ix->n; }
    }
    free(tmp);
    qsort(ix->e, ix->n, sizeof(ient_t), cmp_ient);
    /* occurrence threshold: count at the (1-f) quantile of distinct minimizers, +1, clamped */
    uint32_t *cnt = (uint32_t *)malloc((size_t)(ix->n + 1) * 4);
    int64_t nd = 0;
    for (int64_t i = 0; i < ix->n;) {
        int64_t j = i;
        while (j < ix->n && ix->e[j].key == ix->e[i].key) ++j;
        cnt[nd++] = (uint32_t)(j - i);
        i = j;
    }
    ix->mid_occ = o->min_mid_occ;
    if (nd && o->mid_occ_frac > 0) {
        qsort(cnt, nd, 4, cmp_u32);
        int64_t q = (int64_t)(uint32_t)((1.0 - o->mid_occ_frac) * (double)nd);
        if (q >= nd) q = nd - 1;
        int t = (int)cnt[q] + 1;
        if (t > ix->mid_occ) ix->mid_occ = t;
        if (ix->mid_occ > MAX_MID_OCC) ix->mid_occ = MAX_MID_OCC;
    }
    free(cnt);
}

static int64_t index_find(const index_t *ix, uint64_t key, int64_t *cnt) {
    int64_t lo = 0, hi = ix->n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (ix->e[m].key < key) lo = m + 1; else hi = m; }
    int64_t j = lo;
    while (j < ix->n && ix->e[j].key == key) ++j;
    *cnt = j - lo;
    return lo;
}

/* ---- S3: anchors ------------------------------------------------------------------------------ */
typedef struct { uint32_t t, strand, tpos, qpos, qspan, gen; } anchor_t;   /* gen = generation order */

/* Order of the anchors of one query: by target, strand, then by position on the QUERY in aligned orientation.
 * Anchors are generated query minimizer by query minimizer (ascending forward position; the occurrences of one
 * minimizer ascending by target position), so this is the generation order on the forward strand and its reverse
 * on the reverse strand - a grouping, not a sort by coordinate. */
static int cmp_anchor(const void *a, const void *b) {
    const anchor_t *x = (const anchor_t *)a, *y = (const anchor_t *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    if (x->strand != y->strand) return x->strand < y->strand ? -1 : 1;
    if (x->gen != y->gen) return (x->gen < y->gen) != (x->strand != 0) ? -1 : 1;
    return 0;
}

static int ilog2_32(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }

/* ---- S5: alignment ---------------------------------------------------------------------------- */
typedef struct { uint32_t *op; int n, m; } cigar_t;   /* op = len<<4 | code ; '='7 'X'8 'I'1 'D'2 */
static void cig_push(cigar_t *c, int code, int len) {
    if (len <= 0) return;
    if (c->n && (int)(c->op[c->n - 1] & 15) == code) { c->op[c->n - 1] += (uint32_t)len << 4; return; }
    if (c->n == c->m) { c->m = c->m ? c->m * 2 : 64; c->op = (uint32_t *)realloc(c->op, c->m * 4); }
    c->op[c->n++] = (uint32_t)len << 4 | (uint32_t)code;
}

static inline int sub_score(const ava_opts_t *o, int a, int b) {
    if (a > 3 || b > 3) return -o->ambi;
    return a == b ? o->match : -o->mismatch;
}

/* Banded affine DP over rows i=0..m (query), diagonals d=j-i in [dlo, dlo+BAND_W).
 * q[i], t[j] are accessed through stride (+1 forward, -1 for left extensions).
 * mode 0: global, must end at (m,n); returns score, ops appended in forward order.
 * mode 1: extension, best cell (max H; ties: smaller i+j, then smaller i); *bi,*bj returned.
 * Cells: H = max(M, E, F) with priority M, E, F on ties; E (gap in query, consumes target, 'D'),
 * F (gap in target, consumes query, 'I'); open preferred over extend on ties. */
static __thread int g_last_rank;
static __thread int g_one_piece;      /* set by the callers of band_dp for a task in the 32-diagonal band of the LONG tasks */
/* Two-piece gap cost (minimap2 -O4,24 -E2,1: filter_overlap_slr2.py:51 leaves the ava-pb preset's values in place): every
 * gap state exists once per piece, E = max(E1, E2), F = max(F1, F2) with the first piece on ties.  Traceback byte:
 * bits 0-1 source of H (0 M, 1 E, 2 F), bit 2 / 3 E1 / F1 extended, bit 4 / 5 E2 / F2 extended, bit 6 / 7 E / F is the
 * second piece. */
static int band_dp(const ava_opts_t *o, const uint8_t *q, int qstride, int m, const uint8_t *t, int tstride, int n,
                   int dlo, int W, int mode, int end_row, int *bi, int *bj, uint32_t *rev_ops, int *n_rev) {
    /* z-drop (extensions): after row i = 32, 64, ... the best cell of that row is held against the best cell so far; more
     * than zdrop below it (or no cell of the row inside the rectangle any more) ends the extension: rows > i do not exist */
    const int zdrop = mode == 1 ? o->zdrop : 0;
    /* the 32-diagonal band of the LONG tasks knows the first piece alone: a gap long enough for the second one (>= 20 bases
     * with -O4,24 -E2,1) does not fit a band of +-15 around its diagonal except right across it */
    const int two = o->gap_open2 > 0 && !g_one_piece;
    const int go[2] = {o->gap_open, two ? o->gap_open2 : o->gap_open}, ge[2] = {o->gap_ext, two ? o->gap_ext2 : o->gap_ext};
    int rows = m + 1;
    /* the matrices live in one buffer per thread that only grows: a malloc per call is an mmap + munmap for every block of
     * more than a few rows, and with many OpenMP threads those serialise in the kernel */
    static __thread int32_t *arena;
    static __thread size_t arena_cells;
    const size_t cells = (size_t)rows * W;
    if (cells > arena_cells) {
        free(arena);
        arena_cells = cells + cells / 2 + 4096;
        arena = (int32_t *)malloc(arena_cells * 21 + 64);
    }
    int32_t *H = arena, *E[2], *F[2];
    E[0] = H + cells; E[1] = E[0] + cells; F[0] = E[1] + cells; F[1] = F[0] + cells;
    uint8_t *tb = (uint8_t *)(F[1] + cells);
    int best = NEG_INF, best_i = 0, best_j = 0, best_h = NEG_INF;   /* best: score + end bonus (ranking), best_h: score */
    for (int i = 0; i <= m; ++i) {
        int row_max = NEG_INF;
        for (int dd = 0; dd < W; ++dd) {
            int j = i + dlo + dd, idx = i * W + dd;
            int h = NEG_INF, e[2] = {NEG_INF, NEG_INF}, f[2] = {NEG_INF, NEG_INF};
            uint8_t b = 0;
            if (j >= 0 && j <= n) {
                if (i == 0 && j == 0) h = 0;
                else {
                    int mm = NEG_INF;
                    if (i > 0 && j > 0) {   /* diagonal: same dd in row i-1 */
                        int hp = H[(i - 1) * W + dd];
                        if (hp > NEG_INF) mm = hp + sub_score(o, q[(i - 1) * qstride], t[(j - 1) * tstride]);
                    }
                    for (int p = 0; p < 1 + two; ++p) {
                        if (j > 0 && dd > 0) {  /* horizontal: (i, j-1) = dd-1 same row */
                            int hl = H[idx - 1], el = E[p][idx - 1];
                            int open = hl > NEG_INF ? hl - go[p] - ge[p] : NEG_INF, ext = el > NEG_INF ? el - ge[p] : NEG_INF;
                            if (open >= ext) { e[p] = open; } else { e[p] = ext; b |= p ? 16 : 4; }
                        }
                        if (i > 0 && dd + 1 < W) {  /* vertical: (i-1, j) = dd+1 in row i-1 */
                            int hu = H[(i - 1) * W + dd + 1], fu = F[p][(i - 1) * W + dd + 1];
                            int open = hu > NEG_INF ? hu - go[p] - ge[p] : NEG_INF, ext = fu > NEG_INF ? fu - ge[p] : NEG_INF;
                            if (open >= ext) { f[p] = open; } else { f[p] = ext; b |= p ? 32 : 8; }
                        }
                    }
                    int eb = e[0], fb = f[0];
                    if (two && e[1] > e[0]) { eb = e[1]; b |= 64; }
                    if (two && f[1] > f[0]) { fb = f[1]; b |= 128; }
                    if (mm >= eb && mm >= fb) { h = mm; b |= 0; }
                    else if (eb >= fb) { h = eb; b |= 1; }
                    else { h = fb; b |= 2; }
                    if (h < NEG_INF) h = NEG_INF;
                }
                if (h > row_max) row_max = h;
                if (mode == 1 && h > NEG_INF) {
                    int hb = h + (i == end_row ? o->end_bonus : 0);   /* reaching the query end earns the bonus */
                    if (hb > best || (hb == best && (i + j < best_i + best_j || (i + j == best_i + best_j && i < best_i)))) {
                        best = hb; best_i = i; best_j = j; best_h = h;
                    }
                }
            }
            H[idx] = h; tb[idx] = b;
            for (int p = 0; p < 2; ++p) { E[p][idx] = e[p]; F[p][idx] = f[p]; }
        }
        if (zdrop > 0 && i > 0 && i % ZDROP_STEP == 0 && (row_max == NEG_INF || best - row_max > zdrop)) break;
    }
    int ei, ej, score;
    if (mode == 0) { ei = m; ej = n; score = H[m * W + (n - m - dlo)]; }
    else { ei = best_i; ej = best_j; score = best_h; }
    /* traceback */
    int i = ei, j = ej, state = 0, piece = 0, nr = 0;
    while (i > 0 || j > 0) {
        int idx = i * W + (j - i - dlo);
        uint8_t b = tb[idx];
        if (state == 0) {
            int src = b & 3;
            if (src == 0) {
                int eq = q[(i - 1) * qstride] == t[(j - 1) * tstride];
                rev_ops[nr++] = eq ? 7 : 8;
                --i; --j;
            } else { state = src; piece = src == 1 ? (b >> 6) & 1 : (b >> 7) & 1; }   /* 1: E, 2: F */
        } else if (state == 1) {
            rev_ops[nr++] = 2;   /* D */
            if (!(b & (piece ? 16 : 4))) state = 0;
            --j;
        } else {
            rev_ops[nr++] = 1;   /* I */
            if (!(b & (piece ? 32 : 8))) state = 0;
            --i;
        }
    }
    *n_rev = nr;
    g_last_rank = mode == 1 ? best : score;   /* extension: score + end bonus of the chosen cell */
    if (bi) *bi = ei;
    if (bj) *bj = ej;
    return score;
}

typedef struct {
    int qs, qe, ts, te;   /* aligned-orientation query coords, target coords */
    int score;
    cigar_t cg;
} piece_t;

/* global block between fixed points (q0,t0) -> (q1,t1); returns 0 if it violates the block limits */
static int block_ok(int q0, int t0, int q1, int t1) {
    int m = q1 - q0, n = t1 - t0, delta = n - m;
    if (m < 0 || n < 0 || m > g_block_max || n > g_block_max) return 0;
    if ((delta < 0 ? -delta : delta) > (g_ungapped ? 0 : g_shift_max)) return 0;
    return 1;
}

static void align_block(const ava_opts_t *o, const uint8_t *q, const uint8_t *t, int q0, int t0, int q1, int t1,
                        piece_t *p) {
    int m = q1 - q0, n = t1 - t0, delta = n - m, nr;
    uint32_t *scratch = (uint32_t *)malloc((size_t)(m + n + 2) * 4);
    /* band rule: near-diagonal blocks get the narrow band, the rest the wide one */
    const int ad = delta < 0 ? -delta : delta;
    const int lng = m > BLOCK_MAX || n > BLOCK_MAX;              /* LONG block: 64 diagonals centred on the corners' diagonals */
    int narrow = !lng && ad <= NARROW_DELTA;
    int W = narrow ? NARROW_W : (lng && ad <= HALF_DELTA ? HALF_W : BAND_W);
    if (!narrow && !(lng && ad <= HALF_DELTA) && ad + 2 * BAND_PAD + 1 > W) W = ad + 2 * BAND_PAD + 1;   /* measurement only */
    int dlo = (delta < 0 ? delta : 0) - (narrow ? NARROW_PAD : lng ? (W - 1 - ad) / 2 : BAND_PAD);
    if (g_ungapped) { W = 1; dlo = 0; }                          /* (delta == 0: block_ok) the diagonal is the band */
    g_one_piece = lng && ad <= HALF_DELTA && !g_ungapped;
    p->score += band_dp(o, q + q0, 1, m, t + t0, 1, n, dlo, W, 0, -1, 0, 0, scratch, &nr);
    g_one_piece = 0;
    for (int x = nr - 1; x >= 0; --x) cig_push(&p->cg, (int)scratch[x], 1);
    free(scratch);
}

/* one chain (anchors ascending) -> alignment pieces -> PAF rows */
/* bare: the row of a stub candidate (see is_stub_candidate) - its content never reaches anybody, so it carries none: columns
 * 10 and 11 are 0 and the CIGAR is "*" */
static void emit_piece(FILE *out, const ava_opts_t *o, const seqset_t *Q, int qi, const seqset_t *T, int ti, int strand, piece_t *p, int bare) {
    if (p->cg.n && p->score >= o->min_dp_score) {
        long nm = 0, bl = 0;
        for (int x = 0; x < p->cg.n && !bare; ++x) {
            long l = p->cg.op[x] >> 4;
            bl += l;
            if ((p->cg.op[x] & 15) == 7) nm += l;
        }
        int ql = Q->len[qi];
        int qs = strand ? ql - p->qe : p->qs, qe = strand ? ql - p->qs : p->qe;
        fprintf(out, "%s\t%d\t%d\t%d\t%c\t%s\t%d\t%d\t%d\t%ld\t%ld\t0\tNM:i:%ld\ttp:A:S\tcg:Z:", Q->name[qi], ql, qs, qe,
                strand ? '-' : '+', T->name[ti], T->len[ti], p->ts, p->te, nm, bl, bl - nm);
        for (int x = 0; x < p->cg.n && !bare; ++x) {
            static const char opc[16] = {'?', 'I', 'D', '?', '?', '?', '?', '=', 'X', '?', '?', '?', '?', '?', '?', '?'};
            fprintf(out, "%u%c", p->cg.op[x] >> 4, opc[p->cg.op[x] & 15]);
        }
        if (bare) fputc('*', out);
        fputc('\n', out);
    }
    p->cg.n = 0;
    p->score = 0;
}

/* diagonals of an end extension over m rows and n columns: a LONG extension - one whose 64-diagonal band (-31 .. +32) stays
 * inside the target for more than EXT_MAX rows - takes 32 diagonals (-15 .. +16) */
static int ext_band(int m, int n) {
    if (g_ext_band != BAND_W) return g_ext_band;                 /* (measurement switch) */
    const int rows = m < n + (BAND_W / 2 - 1) ? m : n + (BAND_W / 2 - 1);
    return rows > EXT_MAX ? HALF_W : BAND_W;
}

static void extend_left(const ava_opts_t *o, const uint8_t *q, const uint8_t *t, piece_t *p) {
    const int xm = ext_rows(o);
    int m = p->qs < xm ? p->qs : xm, n = p->ts < xm + g_ext_band ? p->ts : xm + g_ext_band, bi, bj, nr;
    if (m <= 0 || n <= 0) return;
    uint32_t *scratch = (uint32_t *)malloc((size_t)(m + n + 2) * 4);
    const int xw = ext_band(m, n);
    g_one_piece = xw == HALF_W && g_ext_band == BAND_W && !g_ungapped;
    int sc = band_dp(o, q + p->qs - 1, -1, m, t + p->ts - 1, -1, n, g_ungapped ? 0 : -(xw / 2 - 1), g_ungapped ? 1 : xw, 1,
                     p->qs <= xm ? p->qs : -1, &bi, &bj, scratch, &nr);
    g_one_piece = 0;
    if (g_last_rank <= 0 || nr == 0) { free(scratch); return; }   /* nothing gained (the end bonus counts here, not in the score) */
    /* rev_ops run from the far end towards the fixed point on reversed sequences = forward order */
    cigar_t pre = {0, 0, 0};
    for (int x = 0; x < nr; ++x) cig_push(&pre, (int)scratch[x], 1);
    for (int x = 0; x < p->cg.n; ++x) cig_push(&pre, (int)(p->cg.op[x] & 15), (int)(p->cg.op[x] >> 4));
    free(p->cg.op);
    free(scratch);
    p->cg = pre;
    p->qs -= bi; p->ts -= bj; p->score += sc;
}

static void extend_right(const ava_opts_t *o, const uint8_t *q, int ql, const uint8_t *t, int tl, piece_t *p) {
    const int xm = ext_rows(o);
    int m = ql - p->qe < xm ? ql - p->qe : xm;
    int n = tl - p->te < xm + g_ext_band ? tl - p->te : xm + g_ext_band, bi, bj, nr;
    if (m <= 0 || n <= 0) return;
    uint32_t *scratch = (uint32_t *)malloc((size_t)(m + n + 2) * 4);
    const int xw = ext_band(m, n);
    g_one_piece = xw == HALF_W && g_ext_band == BAND_W && !g_ungapped;
    int sc = band_dp(o, q + p->qe, 1, m, t + p->te, 1, n, g_ungapped ? 0 : -(xw / 2 - 1), g_ungapped ? 1 : xw, 1,
                     ql - p->qe <= xm ? ql - p->qe : -1, &bi, &bj, scratch, &nr);
    g_one_piece = 0;
    if (g_last_rank > 0 && nr != 0) {          /* (else nothing gained: the end bonus counts here, not in the score) */
        for (int x = nr - 1; x >= 0; --x) cig_push(&p->cg, (int)scratch[x], 1);
        p->qe += bi; p->te += bj; p->score += sc;
    }
    free(scratch);
}

/* Stub rule (ava_opts_t::stub_oh = h >= 0).  An end extension moves a piece end by at most X = ext_rows() query and
 * X + BAND_W target bases.  A piece whose left end has qs > X + h and ts > X + BAND_W + h (or the same on
 * its right end, measured from the sequence ends) therefore keeps an overhang > h whatever the extensions find, and the
 * consumer's test `overhang > min(h, 0.8 maplen)` drops the row before it touches any state.  The row still counts as a
 * line of the consumer's 1000-line windows, so it has to be written exactly when the full specification writes it:
 * score >= min_dp_score.  Extensions only add a positive score, except that one reaching the query end may be taken
 * with a score down to 1 - end_bonus; such an end is not an interior end, so blocks >= min_dp_score + end_bonus
 * decides "reported" before any extension has run.  Such a piece is written without its end extensions.
 * A piece that meets the geometric half of the rule - a stub CANDIDATE - fails the consumer's test whether or not it is
 * extended, so nobody ever reads its row's content: a candidate's row (extended or not) is written bare (columns 10, 11 = 0,
 * CIGAR "*"; the consumer's first test drops a row of length 0 without a division).  The product therefore computes only the
 * SCORES of a candidate's blocks and extensions. */
static int is_stub_candidate(const ava_opts_t *o, const piece_t *p, int ql, int tl) {
    if (o->stub_oh < 0 || g_ext_band != BAND_W) return 0;
    const int h = o->stub_oh, X = ext_rows(o);
    return (p->qs > X + h && p->ts > X + BAND_W + h) || (ql - p->qe > X + h && tl - p->te > X + BAND_W + h);
}
static int is_stub(const ava_opts_t *o, const piece_t *p, int ql, int tl) {
    const int bonus = o->end_bonus > 0 ? o->end_bonus : 0;
    if (!p->cg.n || p->score < o->min_dp_score + bonus) return 0;
    return is_stub_candidate(o, p, ql, tl);
}

static void close_piece(FILE *out, const ava_opts_t *o, const seqset_t *Q, int qi, const uint8_t *qa, const seqset_t *T, int ti,
                        int strand, piece_t *p) {
    const int ql = Q->len[qi], tl = T->len[ti];
    const int cand = is_stub_candidate(o, p, ql, tl);      /* (on the unextended piece, as the stub test itself) */
    const int stub = is_stub(o, p, ql, tl);
    if (!stub) {
        extend_left(o, qa, T->code[ti], p);
        extend_right(o, qa, ql, T->code[ti], tl, p);
    }
    if (p->cg.n && p->score >= o->min_dp_score) {
        __atomic_fetch_add(&g_n_pieces, 1, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_n_stubs, stub, __ATOMIC_RELAXED);
    }
    emit_piece(out, o, Q, qi, T, ti, strand, p, cand);
}

static void align_chain(FILE *out, const ava_opts_t *o, const seqset_t *Q, int qi, const uint8_t *qa /* aligned orientation */,
                        const seqset_t *T, int ti, int strand, const anchor_t *a, const int *chain, int m) {
    const uint8_t *t = T->code[ti];
    piece_t p;
    memset(&p, 0, sizeof p);
    int open = 0, cq = 0, ct = 0;   /* current fixed point */
    for (int x = 0; x < m; ++x) {
        const anchor_t *an = &a[chain[x]];
        int qe = (int)an->qpos + 1, te = (int)an->tpos + 1;
        if (!open) {    /* start a piece at the start of this anchor */
            int sp = (int)an->qspan;
            int q0 = qe - sp, t0 = te - sp;
            if (q0 < 0 || t0 < 0) { int sh = q0 < t0 ? -q0 : -t0; q0 += sh; t0 += sh; }
            if (q0 < 0) q0 = 0;
            if (t0 < 0) t0 = 0;
            if (!block_ok(q0, t0, qe, te)) continue;
            p.qs = q0; p.ts = t0; p.score = 0; p.cg.n = 0;
            align_block(o, qa, t, q0, t0, qe, te, &p);
            cq = qe; ct = te; open = 1;
            continue;
        }
        if (!((qe - cq >= BLOCK_MIN && te - ct >= BLOCK_MIN) || x == m - 1)) continue;
        if (qe <= cq || te <= ct) continue;
        if (block_ok(cq, ct, qe, te)) {
            align_block(o, qa, t, cq, ct, qe, te, &p);
            cq = qe; ct = te;
        } else {        /* split: close the piece here, reopen at this anchor */
            p.qe = cq; p.te = ct;
            close_piece(out, o, Q, qi, qa, T, ti, strand, &p);
            open = 0;
            --x;        /* revisit this anchor as the start of a new piece */
        }
    }
    if (open) {
        p.qe = cq; p.te = ct;
        close_piece(out, o, Q, qi, qa, T, ti, strand, &p);
    }
    free(p.cg.op);
}

/* ---- S4: chaining of one (target, strand) group ------------------------------------------------ */
/* DP: Li 2018 eq. (1)-(2) over the previous CHAIN_PRED anchors, integer gap cost
 *     gamma(dd) = dd*k/100 + (floor(log2 dd) >> 1).
 * Chain extraction (parallel-friendly restatement of "best-scoring chains first"): the parent links
 * p[] form a forest; every anchor hands its trunk to its best child (larger f, then smaller index);
 * a chain starts at every anchor that is not its parent's best child, follows best-child links and
 * is cut at its highest-scoring anchor (first one on ties).  score = f[peak] - f[parent of start]. */
static void chain_group(FILE *out, const ava_opts_t *o, const seqset_t *Q, int qi, const uint8_t *qa, const seqset_t *T,
                        const anchor_t *a, int n) {
    int32_t *f = (int32_t *)malloc(n * 4), *p = (int32_t *)malloc(n * 4), *bc = (int32_t *)malloc(n * 4);
    /* ORACLE_CHAIN_MM2=1 (tests/test_deviation_effects.py only): the predecessor loop of minimap2's chaining as published
     * (Li 2018 and the --max-chain-skip / max-iteration heuristics HyLight's command line sets: skip 25, 5000
     * iterations) instead of the specification's fixed window of CHAIN_PRED - to MEASURE what the fixed window changes. */
    const int mm2_mode = g_chain_mm2;
    int32_t *tmark = mm2_mode ? (int32_t *)calloc((size_t)n, 4) : 0;
    if (tmark) for (int i = 0; i < n; ++i) tmark[i] = -1;
    for (int i = 0; i < n; ++i) {
        int32_t best = (int32_t)a[i].qspan, bp = -1;
        int n_skip = 0;
        const int window = mm2_mode ? 5000 : CHAIN_PRED;
        for (int j = i - 1; j >= 0 && j >= i - window; --j) {
            int32_t dr = (int32_t)a[i].tpos - (int32_t)a[j].tpos, dq = (int32_t)a[i].qpos - (int32_t)a[j].qpos;
            if (dq > o->max_gap) break;                       /* query positions only grow going back */
            int32_t cand = NEG_INF;
            if (!(dr <= 0 || dr > o->max_gap || dq == 0)) {
                int32_t dd = dr > dq ? dr - dq : dq - dr;
                if (dd <= o->bandwidth) {
                    int32_t dg = dr < dq ? dr : dq;
                    int32_t sc = dg < (int32_t)a[i].qspan ? dg : (int32_t)a[i].qspan;
                    int32_t pen = dd ? (dd * o->k) / 100 + (ilog2_32((uint32_t)dd) >> 1) : 0;
                    cand = f[j] + sc - pen;
                }
            }
            if (cand > NEG_INF && cand > best) { best = cand; bp = j; if (mm2_mode && n_skip > 0) --n_skip; }
            else if (mm2_mode && cand > NEG_INF && tmark[j] == i) { if (++n_skip > 25) break; }
            if (mm2_mode && p[j] >= 0) tmark[p[j]] = i;
        }
        f[i] = best; p[i] = bp;
    }
    free(tmark);
    for (int i = 0; i < n; ++i) bc[i] = -1;
    for (int i = 0; i < n; ++i)            /* ascending i: a later child wins only with a strictly larger f */
        if (p[i] >= 0 && (bc[p[i]] < 0 || f[i] > f[bc[p[i]]])) bc[p[i]] = i;
    int *path = (int *)malloc(n * sizeof(int));
    for (int s = 0; s < n; ++s) {
        if (p[s] >= 0 && bc[p[s]] == s) continue;      /* continues its parent's chain */
        int m = 0, best_len = 1, cur = s;
        int32_t best_f = f[s];
        path[m++] = s;
        while (bc[cur] >= 0) {
            cur = bc[cur];
            path[m++] = cur;
            if (f[cur] > best_f) { best_f = f[cur]; best_len = m; }
        }
        int32_t sc = best_f - (p[s] >= 0 ? f[p[s]] : 0);
        if (sc < o->min_chain_score || best_len < o->min_cnt) continue;
        align_chain(out, o, Q, qi, qa, T, (int)a[0].t, (int)a[0].strand, a, path, best_len);
    }
    free(f); free(p); free(bc); free(path);
}

/* ---- driver: one target chunk vs all queries ---------------------------------------------------- */
static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

static void assign_ranks(seqset_t *T, seqset_t *Q) {
    int n = T->n + Q->n;
    char **all = (char **)malloc((size_t)(n ? n : 1) * sizeof(char *));
    for (int i = 0; i < T->n; ++i) all[i] = T->name[i];
    for (int i = 0; i < Q->n; ++i) all[T->n + i] = Q->name[i];
    qsort(all, n, sizeof(char *), cmp_str);
    T->rank = (int *)malloc((size_t)(T->n ? T->n : 1) * sizeof(int));
    Q->rank = (int *)malloc((size_t)(Q->n ? Q->n : 1) * sizeof(int));
    for (int s = 0; s < 2; ++s) {
        seqset_t *S = s ? Q : T;
        for (int i = 0; i < S->n; ++i) {   /* rank = index of the first equal name */
            int lo = 0, hi = n;
            while (lo < hi) { int m = (lo + hi) >> 1; if (strcmp(all[m], S->name[i]) < 0) lo = m + 1; else hi = m; }
            S->rank[i] = lo;
        }
    }
    free(all);
}

void oracle_last_counts(long *pieces, long *stubs) { *pieces = g_n_pieces; *stubs = g_n_stubs; }

int oracle_ava(const char *target_fa, const char *query_fa, const ava_opts_t *o, const char *out_paf) {
    seqset_t Ts, Qs, *T = &Ts, *Q = &Qs;
    if (seqset_read(target_fa, T) != 0) return -2;
    if (seqset_read(query_fa, Q) != 0) { seqset_free(T); return -2; }
    if (!(o->k & 1) || o->k > 28 || o->w < 1 || o->w > 64) { seqset_free(T); seqset_free(Q); return -1; }
    assign_ranks(T, Q);
    read_switches();
    g_ungapped = o->bandwidth == 0;
    g_n_pieces = g_n_stubs = 0;
    FILE *out = fopen(out_paf, "w");
    if (!out) { seqset_free(T); seqset_free(Q); return -2; }
    index_t ix;
    index_build(T, o, &ix);
    int64_t qcap = 0;
    for (int i = 0; i < Q->n; ++i) if (Q->len[i] > qcap) qcap = Q->len[i];
    /* Queries are independent of each other: one query per loop trip, its rows into a buffer of its own, the buffers
     * written in query order - the output does not depend on the number of threads (OMP_NUM_THREADS; bench.py's CPU leg
     * uses all host cores, the tests whatever the box has). */
    char **qbuf = (char **)calloc((size_t)(Q->n ? Q->n : 1), sizeof(char *));
    size_t *qbuf_n = (size_t *)calloc((size_t)(Q->n ? Q->n : 1), sizeof(size_t));
#pragma omp parallel
    {
    mz_t *qm = (mz_t *)malloc((size_t)(qcap + 1) * sizeof(mz_t));
    uint8_t *qrc = (uint8_t *)malloc((size_t)qcap + 1);
    anchor_t *an = 0;
    int64_t an_cap = 0;
#pragma omp for schedule(dynamic, 16)
    for (int qi = 0; qi < Q->n; ++qi) {
        int ql = Q->len[qi];
        int64_t nm = oracle_sketch_codes(Q->code[qi], ql, (uint32_t)qi, o->k, o->w, o->hpc, qm, qcap);
        int64_t na = 0;
        for (int64_t x = 0; x < nm; ++x) {
            int64_t cnt, s = index_find(&ix, qm[x].x >> 8, &cnt);
            if (cnt == 0 || cnt > ix.mid_occ) continue;
            uint32_t qspan = (uint32_t)(qm[x].x & 0xff), qpos = (uint32_t)qm[x].y >> 1, qz = (uint32_t)qm[x].y & 1;
            for (int64_t e = s; e < s + cnt; ++e) {
                uint32_t t = (uint32_t)(ix.e[e].y >> 32), tpos = (uint32_t)ix.e[e].y >> 1, tz = (uint32_t)ix.e[e].y & 1;
                if (o->pair_once ? Q->rank[qi] >= T->rank[t] : Q->rank[qi] == T->rank[t]) continue;   /* pair once: strcmp(q,t) < 0 only; self always skipped */
                if (na == an_cap) { an_cap = an_cap ? an_cap * 2 : 4096; an = (anchor_t *)realloc(an, an_cap * sizeof(anchor_t)); }
                anchor_t *a = &an[na++];
                a->t = t; a->strand = qz ^ tz; a->tpos = tpos; a->qspan = qspan; a->gen = (uint32_t)(na - 1);
                a->qpos = a->strand ? (uint32_t)(ql - (int)(qpos + 1 - qspan) - 1) : qpos;
            }
        }
        if (!na) continue;
        qsort(an, na, sizeof(anchor_t), cmp_anchor);
        for (int i = 0; i < ql; ++i) { uint8_t c = Q->code[qi][ql - 1 - i]; qrc[i] = c < 4 ? 3 - c : 4; }
        FILE *qout = open_memstream(&qbuf[qi], &qbuf_n[qi]);
        for (int64_t b = 0; b < na;) {
            int64_t e = b;
            while (e < na && an[e].t == an[b].t && an[e].strand == an[b].strand) ++e;
            chain_group(qout, o, Q, qi, an[b].strand ? qrc : Q->code[qi], T, an + b, (int)(e - b));
            b = e;
        }
        fclose(qout);
    }
    free(an); free(qm); free(qrc);
    }
    for (int qi = 0; qi < Q->n; ++qi) {
        if (qbuf_n[qi]) fwrite(qbuf[qi], 1, qbuf_n[qi], out);
        free(qbuf[qi]);
    }
    free(qbuf); free(qbuf_n);
    free(ix.e);
    fclose(out);
    seqset_free(T); seqset_free(Q);
    return 0;
}
