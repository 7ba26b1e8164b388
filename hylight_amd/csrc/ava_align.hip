// ava_align.hip - S5 of the overlapper spec: base-level alignment of the chain pieces.
//
// Every piece is a list of fixed points; consecutive fixed points bound an independent global
// alignment block (|diagonal shift| <= 39, any length) and the two piece ends get a local
// extension (up to max(256, max_gap) rows, z-drop).  Blocks of more than 256 rows or columns and
// extensions of more than 256 rows are LONG tasks (align_long_kernel: the band walks the task in
// tiles of 256 rows, its traceback planes live in a scratch area per wave).  Passes:
//   make_tasks   one self-contained 32-byte record per block / extension
//   classify     lane per task: square blocks with <= kmax substitutions are finished on the spot (proof at the
//                kernel), the rest is split into near-diagonal (16-diagonal band) and wide (64 diagonals) lists
//   align_narrow four tasks per wave (one per DPP row), align (wide) one task per wave.  Lanes = diagonals, rows one
//                by one; affine gaps (Gotoh): F from lane d+1 of the previous row, E from a max-plus prefix scan
//                across the lanes (E(d) = max_{d'<d} Ht(d') - open - ext*(d-d')); the five traceback bits of a cell
//                stay with its lane as bit planes; the traceback jumps along diagonals with count-trailing-ones
//   assemble     wave per piece: merged CIGAR + PafRec
// The recurrences, tie rules and the best-cell rule of the extensions are those of
// oracle/ava_oracle.c:band_dp.  VALU bound by nature: reads ~1 B per
// DP row per sequence from HBM.
#include <algorithm>
#include <memory>
#include <type_traits>

#include "ava_internal.h"
#include "dev_prims.h"
#include "wave_ops.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
constexpr int WAVES = WG / 64;
constexpr int SEQ_T_MAX = EXT_MAX + BAND_W;     // 320
constexpr int NR_SHORT = 128;                   // rows of an align_narrow_kernel<NR_SHORT> task; longer near-diagonal
                                                // blocks (3-4 % of them) run in the <BLOCK_MAX> instance with twice the LDS
constexpr int WIDE_SHORT = 128;                 // rows of an align_kernel<WIDE_SHORT> task (most extensions are short): half the
                                                // plane LDS of the <EXT_MAX> instance, twice the resident waves
inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }

struct Task {               // 32 B, self-contained: the kernels reach the bases without touching Piece / offset tables
    uint32_t piece;
    uint16_t kind;          // 0 block, 1 left extension, 2 right extension; bit 2: query on the reverse strand; bit 3: TASK_ONE
    uint16_t narrow;        // block: 1 = 16-diagonal band; extensions: (row that reaches the query end + 1) << 1, 0 = out of reach
    uint64_t qa, ta;        // offsets in qcodes / tcodes of window element 0 (see load_window4 for the directions)
    int16_t m, n;           // rows (query), cols (target)
    int16_t dlo;            // first diagonal of the band
    int16_t trim;           // near-diagonal blocks: exactly matching bases cut off the end of both sequences (m, n are without
                            // them); the traceback starts with a run of that many '='
};
constexpr uint16_t TASK_REV = 4;
constexpr uint16_t TASK_ONE = 8;             // extension: no gap long enough for the second piece of the gap cost can be on or tie the best path
struct TaskOut {            // 24 B
    int32_t score;
    int32_t bi, bj;         // extension: rows / cols consumed
    uint32_t runs_off, n_runs;
    uint32_t pad;           // bits 0-3 / 4-7: CIGAR code of the first / last run (the assembly decides the merges of neighbouring
                            // tasks from these without touching the runs); bit 31: extension that reached the query end
};
__device__ __forceinline__ uint32_t end_codes(uint32_t first_run, uint32_t last_run) { return (first_run & 15u) | (last_run & 15u) << 4; }

__global__ void piece_task_count_kernel(const Piece *pieces, size_t n, uint32_t *cnt) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = pieces[i].n_fp + 1;      // (n_fp - 1) blocks + 2 extensions
}

// Where the task records come from.  Per task 8 bytes: its piece, its kind (0 block, 1 left, 2 right extension) and the
// index of its first fixed point; per piece 32 bytes with everything else the record needs.  The classifier reaches a
// task's bases in three dependent reads (reference, then piece geometry and fixed points side by side, then the bases)
// instead of five, and fetches the reference a round ahead.
struct TaskRef { uint32_t pk, fp; };       // piece | kind << 30; block k: fixed point k - 1, left extension: 0, right: the last
struct PieceGeom {                         // 32 B
    uint64_t qo, to;                       // first base of the query / target in the code arrays
    uint32_t ql, tl;
    uint32_t strand;
    uint32_t stub_cand;                    // 1: one end lies so far inside both sequences that no end extension can bring the
                                           // piece's overhang down to the consumer's bound (stub rule, see align_span)
};
struct TaskGeom {
    const TaskRef *ref;
    const PieceGeom *pg;
    const FixPt *fps;
};
// Window element x of a task: query = qcodes[qa + x] (complemented and read downwards, qcodes[qa - x], when exactly one
// of "left extension" and "reverse strand" holds; complemented whenever the strand is reverse); target =
// tcodes[ta + x], downwards for a left extension.
// ext_max = ext_rows(opts): rows an end extension may run over
__device__ __forceinline__ Task build_task(const PieceGeom &p, uint32_t piece, uint32_t kind, FixPt f0, FixPt f1, int ext_max) {
    const int ql = (int)p.ql, tl = (int)p.tl;
    const uint64_t qo = p.qo, to = p.to;
    const uint16_t rev = p.strand ? TASK_REV : 0;
    // aligned query position pos -> offset in qcodes
    auto qaddr = [&](int pos) { return p.strand ? qo + (uint64_t)(ql - 1 - pos) : qo + (uint64_t)pos; };
    if (kind == 1) {                    // left extension: elements run downwards from the fixed point
        const int qs = (int)f0.q, ts = (int)f0.t;
        // (columns: the band ends BAND_W / 2 diagonals right of the main one - more than m + BAND_W target bases are never
        //  looked at, and the 64-diagonal kernels stage no more than SEQ_T_MAX of them)
        const int m = qs < ext_max ? qs : ext_max, n2 = ts < m + BAND_W ? ts : m + BAND_W;
        return Task{piece, (uint16_t)(1u | rev), (uint16_t)(qs <= ext_max ? (qs + 1) << 1 : 0), qaddr(qs - 1),
                    to + (uint64_t)(ts - 1), (int16_t)m, (int16_t)n2, (int16_t)(-(BAND_W / 2 - 1)), 0};
    }
    if (kind == 2) {                    // right extension
        const int qe = (int)f0.q, te = (int)f0.t;
        const int m = ql - qe < ext_max ? ql - qe : ext_max, n2 = tl - te < m + BAND_W ? tl - te : m + BAND_W;
        return Task{piece, (uint16_t)(2u | rev), (uint16_t)(ql - qe <= ext_max ? (ql - qe + 1) << 1 : 0), qaddr(qe),
                    to + (uint64_t)te, (int16_t)m, (int16_t)n2, (int16_t)(-(BAND_W / 2 - 1)), 0};
    }
    // block between two fixed points
    const int q0 = (int)f0.q, t0 = (int)f0.t, m = (int)f1.q - q0, n2 = (int)f1.t - t0;
    const int delta = n2 - m, ad = delta < 0 ? -delta : delta;
    // band rule: near-diagonal blocks use the 16-diagonal band; a LONG block (more than BLOCK_MAX rows or
    // columns) takes 64 diagonals centred on the diagonals of its two corners
    const bool lng = m > BLOCK_MAX || n2 > BLOCK_MAX;
    const bool narrow = !lng && ad <= NARROW_DELTA;
    const int lw = ad <= HALF_DELTA ? HALF_W : BAND_W;           // band of a LONG block
    return Task{piece, rev, (uint16_t)(narrow ? 1 : 0), qaddr(q0), to + (uint64_t)t0, (int16_t)m, (int16_t)n2,
                (int16_t)((delta < 0 ? delta : 0) - (narrow ? NARROW_PAD : lng ? (lw - 1 - ad) / 2 : BAND_PAD)), 0};
}
// references of the tasks of every piece and the piece's geometry (one wave per piece)
__global__ __launch_bounds__(WG) void task_ref_kernel(const Piece *pieces, const uint32_t *task_off, size_t n, const uint32_t *qlen,
                                                       const uint32_t *tlen, const uint64_t *qoff, const uint64_t *toff,
                                                       const FixPt *fps, int stub_oh, int ext_max, TaskRef *task_ref, PieceGeom *pg) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < n; i += n_waves) {
        const Piece p = pieces[i];
        const uint32_t n_tasks = p.n_fp + 1;
        TaskRef *out = task_ref + task_off[i];
        for (uint32_t k = (uint32_t)lane; k < n_tasks; k += 64) {
            const uint32_t kind = k == 0 ? 1u : (k == p.n_fp ? 2u : 0u);
            out[k] = TaskRef{(uint32_t)i | kind << 30, p.fp_off + (k == 0 ? 0u : k - 1)};
        }
        if (lane == 0) {
            const uint32_t ql = qlen[p.q], tl = tlen[p.t];
            uint32_t cand = 0;
            if (stub_oh >= 0) {          // (oracle/ava_oracle.c:is_stub - the geometric half of the rule)
                const FixPt a = fps[p.fp_off], b = fps[p.fp_off + p.n_fp - 1];
                const uint32_t dq = (uint32_t)(ext_max + stub_oh), dt = (uint32_t)(ext_max + BAND_W + stub_oh);
                cand = ((a.q > dq && a.t > dt) || (ql - b.q > dq && tl - b.t > dt)) ? 1u : 0u;
            }
            pg[i] = PieceGeom{qoff[p.q], toff[p.t], ql, tl, p.strand, cand};
        }
    }
}

// the same with a lane per piece (pieces of a handful of fixed points: short reads)
__global__ __launch_bounds__(WG) void task_ref_lane_kernel(const Piece *pieces, const uint32_t *task_off, size_t n, const uint32_t *qlen,
                                                            const uint32_t *tlen, const uint64_t *qoff, const uint64_t *toff,
                                                            const FixPt *fps, int stub_oh, int ext_max, TaskRef *task_ref, PieceGeom *pg) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Piece p = pieces[i];
    const uint32_t n_tasks = p.n_fp + 1;
    TaskRef *out = task_ref + task_off[i];
    for (uint32_t k = 0; k < n_tasks; ++k) {
        const uint32_t kind = k == 0 ? 1u : (k == p.n_fp ? 2u : 0u);
        out[k] = TaskRef{(uint32_t)i | kind << 30, p.fp_off + (k == 0 ? 0u : k - 1)};
    }
    const uint32_t ql = qlen[p.q], tl = tlen[p.t];
    uint32_t cand = 0;
    if (stub_oh >= 0) {
        const FixPt a = fps[p.fp_off], b = fps[p.fp_off + p.n_fp - 1];
        const uint32_t dq = (uint32_t)(ext_max + stub_oh), dt = (uint32_t)(ext_max + BAND_W + stub_oh);
        cand = ((a.q > dq && a.t > dt) || (ql - b.q > dq && tl - b.t > dt)) ? 1u : 0u;
    }
    pg[i] = PieceGeom{qoff[p.q], toff[p.t], ql, tl, p.strand, cand};
}

struct AlignArgs {
    Task *tasks;            // records of the tasks that need a DP (written by the classifier, read by the DP kernels)
    TaskGeom geom;
    size_t n_tasks;
    const uint32_t *list;   // task ids this launch works on (n_list of them)
    size_t n_list;
    const uint8_t *qcodes, *tcodes;
    long long q_total, t_total;   // bytes in qcodes / tcodes (4-base loads stay inside)
    int match, mismatch, go, ge, ambi;
    int go2, ge2;           // second piece of the gap cost (go2 <= 0: one piece); only the 64-diagonal kernel can meet gaps long enough for it
    int end_bonus;          // ranks extension cells that reach the query end (0 in long mode)
    int ext_max;            // rows an end extension may run over (ext_rows)
    int ext_all_long;       // 1: every extension runs in align_long_kernel (the z-drop could change an extension of <= EXT_MAX rows)
    int ungapped;           // 1: bandwidth 0 (minimap2 -r 0: the DP band is the diagonal alone) - blocks are square, tasks the
                            // certificates do not settle are compared along the diagonal (align_ungapped_kernel)
    int kmax;               // fast path: max substitutions for which the diagonal is provably the unique optimum
    int kgap1;              // 1: blocks with |n - m| = 1 and one substitution finish in the classifier (fifth certificate)
    int kgap2;              // 1: ... and with two substitutions, when no path with a two-base gap one way and a one-base gap the
                            // other way avoids them all (seventh certificate)
    int one_ok;             // 1: extensions may be certified for the one-piece rows (HLMI_NO_ONE_PIECE_CERT: test hook)
    int kext_plain;         // 1: an extension with ONE substitution and no bonus row finishes in the classifier (sixth certificate)
    int kext_bonus;         // extensions whose end cell lies in the bonus row finish there with up to this many substitutions
    int trim_ok;            // 1: near-diagonal DP tasks lose their exactly matching suffix (HLMI_NO_SUFFIX_TRIM: test hook)
    int kshift;             // 1: a square block with kmax + 1 substitutions also finishes here when the one-base shift
                            // that could avoid them all does not match (fourth certificate, see classify_kernel)
    TaskOut *out;
    uint32_t *runs;
    uint32_t cap_runs;
    uint32_t *counters;     // [0] runs cursor, [1] overflow
    uint32_t run_buf_cap;   // runs the first traceback walk may keep in LDS (test hook HLMI_RUN_BUF_CAP lowers it)
    uint8_t *cls_bare;      // stub rule: DP class (1..4, as cls) of the tasks of stub CANDIDATE pieces - nobody reads their rows'
                            // content, so these run the score-only kernels (no traceback planes, no walk, no runs); null: the
                            // candidates' tasks are classified like the others
    uint8_t *defer_flag;    // classification: tasks left to the second pass (flag per task, then their list and its length)
    const uint32_t *defer_list, *defer_count;
};

// exclusive prefix sum over the wave; total = sum of all lanes
__device__ __forceinline__ uint32_t wave_excl_sum_u32(uint32_t v, int lane, uint32_t &total) {
    (void)lane;
    const uint32_t x = wave_prefix_sum_incl_dpp(v);          // six DPP adds (the shuffle form was six trips through the LDS crossbar)
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}

// CIGAR-run pool: a wave reserves RUN_CHUNK entries with ONE global atomic and hands them out locally
// (a single contended counter word saturates near 90 M atomics/s - more than 100 M tasks per step would
// serialise on it).  Only lane 0 allocates.
constexpr uint32_t RUN_CHUNK = 4096;
constexpr uint32_t RUN_CHUNK_SMALL = 256;       // per 16-lane group leader (narrow kernels)
__device__ __forceinline__ uint32_t pool_take(const AlignArgs &a, uint32_t n, uint32_t &chunk_off, uint32_t &chunk_left,
                                              bool &ok, uint32_t chunk = RUN_CHUNK) {
    if (n > chunk_left) {
        const uint32_t want = n > chunk ? n : chunk;
        chunk_off = atomicAdd(&a.counters[0], want);
        chunk_left = want;
        if ((unsigned long long)chunk_off + want > a.cap_runs) { a.counters[1] = 1; chunk_left = 0; ok = false; return 0; }
    }
    const uint32_t off = chunk_off;
    chunk_off += n;
    chunk_left -= n;
    return off;
}

// 4 codes at base[idx .. idx+3] (little endian); bytes outside [0, total) read as 4
__device__ __forceinline__ uint32_t load_codes4(const uint8_t *base, long long idx, long long total) {
    if (idx >= 0 && idx + 4 <= total) {
        uint32_t v;
        __builtin_memcpy(&v, base + idx, 4);
        return v;
    }
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) v |= (uint32_t)(idx + k >= 0 && idx + k < total ? base[idx + k] : 4) << (8 * k);
    return v;
}
// complement of 4 packed codes (3 - c for ACGT, 4 stays 4)
__device__ __forceinline__ uint32_t comp_codes4(uint32_t x) {
    const uint32_t n = x & 0x04040404u;
    return (x ^ 0x03030303u) & ~((n >> 1) | (n >> 2));
}
// window elements x .. x+3 (byte 0 = element x) of a sequence whose element 0 sits at offset a0
__device__ __forceinline__ uint32_t load_window4(const uint8_t *codes, long long total, long long a0, bool down, bool comp, int x) {
    uint32_t v = down ? __builtin_bswap32(load_codes4(codes, a0 - x - 3, total)) : load_codes4(codes, a0 + x, total);
    if (comp) v = comp_codes4(v);
    return v;
}

// the same without the bounds test (the code arrays carry DevReads::PAD bytes of code 4 on both sides; the windows of
// the DP kernels stay within a few bases of their reads)
__device__ __forceinline__ uint32_t load_window4p(const uint8_t *codes, long long a0, bool down, bool comp, int x) {
    uint32_t v;
    __builtin_memcpy(&v, codes + (down ? a0 - x - 3 : a0 + x), 4);
    if (down) v = __builtin_bswap32(v);
    if (comp) v = comp_codes4(v);
    return v;
}

// ---- pass 1: classification + diagonal fast path, one lane per task --------------------------------
// cls[task] = 0 done here (empty task or fast path), 1 / 3 DP in the 16-diagonal band (<= / > NR_SHORT rows),
// 4 / 2 DP in the 64-diagonal band (<= / > WIDE_SHORT rows).
// Fast path: square block, no ambiguous base, at most kmax substitutions.  With delta = 0 any gapped path has
// >= 1 insertion and >= 1 deletion and <= m-1 diagonal moves, i.e. scores <= match*(m-1) - 2*(open+ext); the
// all-diagonal path with k mismatches scores match*(m-k) - mismatch*k, which is strictly larger while
// k*(match+mismatch) < match + 2*(open+ext).  It is then the unique optimum, so the DP + traceback would return
// exactly these runs.

// stats[]: 0 bases (Lq + Lt) of all tasks, 1 of the square blocks compared here, 2 of the narrow DP tasks, 3 of the
// wide DP tasks, 4 tasks finished on the diagonal fast path, 5 DP tasks, 6 DP rows
enum { ST_BASES = 0, ST_BASES_SQUARE, ST_BASES_NARROW, ST_BASES_WIDE, ST_FAST, ST_DP, ST_DP_ROWS, ST_NARROW_SMALL, ST_WIDE_ONE, ST_STUB_EXT, ST_EXT_CERT, ST_EXT_DP_ROW, ST_EXT_DP_K, ST_LONG, ST_BASES_LONG, N_ALIGN_STATS };
constexpr int NR_SMALL = 64;                    // narrow tasks with fewer rows than this run in the instance with half the plane LDS
constexpr uint8_t CLS_LONG = 5;                 // class of the LONG tasks (align_long_kernel)
constexpr uint8_t CLS_LONG32 = 7;               // LONG tasks in the 32-diagonal band, two to a wave (align_long32_kernel)
constexpr uint8_t CLS_UNGAPPED = 6;             // bandwidth 0: every task a certificate does not settle (align_ungapped_kernel)

// 8 window elements x .. x+7 (byte 0 = element x); same conventions as load_window4
__device__ __forceinline__ uint64_t load_codes8(const uint8_t *base, long long idx, long long total) {
    if (idx >= 0 && idx + 8 <= total) {
        uint64_t v;
        __builtin_memcpy(&v, base + idx, 8);
        return v;
    }
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v |= (uint64_t)(idx + k >= 0 && idx + k < total ? base[idx + k] : 4) << (8 * k);
    return v;
}
__device__ __forceinline__ uint64_t load_window8(const uint8_t *codes, long long total, long long a0, bool down, bool comp, int x) {
    uint64_t v = down ? __builtin_bswap64(load_codes8(codes, a0 - x - 7, total)) : load_codes8(codes, a0 + x, total);
    if (comp) {
        const uint64_t n = v & 0x0404040404040404ull;
        v = (v ^ 0x0303030303030303ull) & ~((n >> 1) | (n >> 2));
    }
    return v;
}

// the same without the bounds test: the code arrays carry DevReads::PAD bytes of code 4 on both sides, and the
// classifier's loads stay within a few bases of its blocks
__device__ __forceinline__ uint64_t load_window8p(const uint8_t *codes, long long a0, bool down, bool comp, int x) {
    uint64_t v;
    __builtin_memcpy(&v, codes + (down ? a0 - x - 7 : a0 + x), 8);
    if (down) v = __builtin_bswap64(v);
    if (comp) {
        const uint64_t n = v & 0x0404040404040404ull;
        v = (v ^ 0x0303030303030303ull) & ~((n >> 1) | (n >> 2));
    }
    return v;
}

// One lane per task: a square block is compared 8 bases at a time and given up at the third mismatch (kmax <= 3
// would still pass with 3), so the typical task costs 8-9 iterations of two 8-byte loads; a block with m != n gets
// its common prefix / suffix measured the same way (second certificate below); the runs of the tasks that finish
// here are allocated with one pool request per wave.
// Two passes.  PASS 1 goes over all tasks and settles the square blocks; the blocks with m != n and the end
// extensions (a tenth of the tasks, scattered one or two to a wave, each with two or three times the work of a square
// block) are only flagged.  PASS 2 runs the second and third certificate over the list of the flagged tasks, every lane
// busy with the same kind of work (in one pass those few lanes cost a third of the kernel).
template <int PASS>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(PASS == 1 ? 4 : 5))) void classify_kernel(AlignArgs a, uint8_t *cls, unsigned long long *stats) {
    __shared__ unsigned long long s_stat[WAVES][N_ALIGN_STATS];
    constexpr int CMP_PIECES = PASS == 1 ? 64 * (BLOCK_MAX / 8) : 1;     // 8-base pieces of a wave's 64 square blocks
    __shared__ uint8_t s_mm[WAVES][CMP_PIECES], s_up[WAVES][CMP_PIECES], s_dn[WAVES][CMP_PIECES];
    __shared__ uint64_t s_qa[WAVES][PASS == 1 ? 64 : 1], s_ta[WAVES][PASS == 1 ? 64 : 1];
    __shared__ uint32_t s_geo[WAVES][PASS == 1 ? 64 : 1], s_amb[WAVES][2];
    const int lane = threadIdx.x & 63;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t n_thr = (size_t)gridDim.x * blockDim.x;
    uint32_t chunk_off = 0, chunk_left = 0;
    uint32_t st[N_ALIGN_STATS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // < 2^32 per thread by far
    const size_t n_units = PASS == 1 ? a.n_tasks : (size_t)*a.defer_count;
    const size_t rounds = (n_units + n_thr - 1) / n_thr;              // uniform trip count: the allocation is per wave
    // the reference of a task (PASS 2: its index in the list) is fetched a round ahead
    TaskRef ref_n{0, 0};
    uint32_t ti_n = 0;
    if (PASS == 1) { if (tid < n_units) ref_n = a.geom.ref[tid]; }
    else if (tid < n_units) ti_n = a.defer_list[tid];
    for (size_t r = 0; r < rounds; ++r) {
        const size_t u = r * n_thr + tid;
        bool live = u < n_units;
        const size_t ti = PASS == 1 || !live ? u : (size_t)ti_n;
        TaskRef ref = ref_n;
        if (PASS == 2 && live) ref = a.geom.ref[ti];
        {
            const size_t un = u + n_thr;
            if (PASS == 1) { if (un < n_units) ref_n = a.geom.ref[un]; }
            else if (un < n_units) ti_n = a.defer_list[un];
        }
        Task tk{};
        bool held = false;           // end extension of a stub candidate: decided after the piece's blocks are scored
        bool bare = false;           // task of a stub candidate: only its score is wanted
        if (live) {
            const uint32_t pc = ref.pk & 0x3fffffffu, kind = ref.pk >> 30;
            const PieceGeom pg = a.geom.pg[pc];
            held = PASS == 1 && kind != 0 && pg.stub_cand != 0;
            bare = a.cls_bare != nullptr && pg.stub_cand != 0;
            const FixPt f0 = a.geom.fps[ref.fp];
            FixPt f1{0, 0};
            if (kind == 0) f1 = a.geom.fps[ref.fp + 1];
            tk = build_task(pg, pc, kind, f0, f1, a.ext_max);
        }
        const int m = tk.m, n = tk.n;
        uint8_t c = 2;
        bool try_fast = false;
        if (live) {
            if (m <= 0 || n <= 0 || held) c = 0;
            else if ((tk.kind & 3) == 0) {
                if (m > BLOCK_MAX || n > BLOCK_MAX)                    // LONG block: no certificate is tried (no shared minimizer over
                    c = (n > m ? n - m : m - n) <= HALF_DELTA ? CLS_LONG32 : CLS_LONG;   // more than 256 bases: the sequences differ there)
                else { c = tk.narrow ? (m <= NR_SHORT ? 1 : 3) : 2; try_fast = (m == n); }
            } else if ((m < n - tk.dlo ? m : n - tk.dlo) > EXT_MAX) { c = CLS_LONG32; tk.dlo = (int16_t)(-(HALF_W / 2 - 1)); }   // 32 diagonals
            else if (a.ext_all_long) c = CLS_LONG;
            // (an extension's rows end where its band leaves the target: min(m, n - dlo) - a dovetail's extension into the
            //  few bases the shorter side has left is a short task whatever the other side's length; LONG after the
            //  certificates of the second pass)
            if (c == 2 && (m < n - tk.dlo ? m : n - tk.dlo) <= WIDE_SHORT) c = 4;      // rows the 64-diagonal kernel really runs
        }
        if (PASS == 1 && live) {                                      // second / third certificate: the other pass
            const bool defer = c != 0 && a.kmax >= 0 && ((tk.kind & 3) != 0 || (m != n && c != CLS_LONG && c != CLS_LONG32));
            a.defer_flag[ti] = defer ? 1 : 0;
            if (defer) { live = false; try_fast = false; }
        }
        int k = 0, mpos[3] = {0, 0, 0}, last_x = -1;
        bool ambig = false, shift_ok = false;
        if constexpr (PASS == 1) {
            // The square blocks of the wave's 64 tasks are compared 8 bases per lane and step with the LANES SPREAD OVER
            // THE 8-BASE PIECES of all of them (a lane per task would run every lane as long as the longest block of
            // the wave, and its loads would touch 64 different lines): piece c of the wave belongs to the task whose first
            // piece is the last one marked at or before c (a running prefix maximum over the marks), neighbouring lanes read
            // neighbouring bytes, and each piece leaves one byte of mismatch bits in LDS for its task to read back.
            const int wv = threadIdx.x >> 6;
            // (the owner marks and the mismatch bytes share an array: a piece's mark is read before its byte is written)
            uint8_t *own = s_mm[wv], *mm = s_mm[wv], *mup = s_up[wv], *mdn = s_dn[wv];
            const uint32_t nch = try_fast ? (uint32_t)(m + 7) >> 3 : 0u;
            const uint32_t incl = wave_prefix_sum_incl_dpp(nch), tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t st0 = incl - nch;
            for (uint32_t c = (uint32_t)lane; c < tot; c += 64) own[c] = 0;
            if (lane < 2) s_amb[wv][lane] = 0;
            s_qa[wv][lane] = tk.qa; s_ta[wv][lane] = tk.ta;
            s_geo[wv][lane] = st0 | (uint32_t)m << 16 | ((tk.kind & TASK_REV) ? 0x80000000u : 0u);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (nch) own[st0] = (uint8_t)lane;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            int carry = 0;
            constexpr int CU = 4;                                   // pieces per lane and trip: their loads are issued together
            for (uint32_t c0 = 0; c0 < tot; c0 += 64 * CU) {
                int ow[CU], xs[CU], left[CU];
                bool rv[CU];
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    const uint32_t c = c0 + 64u * u + (uint32_t)lane;
                    int o = c < tot ? (int)own[c] : 0;
                    o = wave_prefix_max_incl_dpp(o);
                    o = o > carry ? o : carry;
                    carry = __builtin_amdgcn_readlane(o, 63);
                    const uint32_t geo = s_geo[wv][o];
                    ow[u] = o;
                    xs[u] = (int)(c - (geo & 0xffffu)) * 8;
                    left[u] = c < tot ? (int)(geo >> 16 & 0x7fffu) - xs[u] : 0;      // bases of the piece inside its block (>= 1)
                    rv[u] = (geo >> 31) != 0;
                }
                uint64_t q8[CU], t8[CU];
                uint32_t q_end = 0, t_end = 0;         // lane 63: the base behind the trip's last piece (fourth certificate)
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    q8[u] = t8[u] = 0;
                    if (left[u] > 0) {
                        q8[u] = load_window8p(a.qcodes, (long long)s_qa[wv][ow[u]], rv[u], rv[u], xs[u]);
                        t8[u] = load_window8p(a.tcodes, (long long)s_ta[wv][ow[u]], false, false, xs[u]);
                        if (u == CU - 1 && lane == 63 && a.kshift) {
                            q_end = (uint32_t)load_window8p(a.qcodes, (long long)s_qa[wv][ow[u]], rv[u], rv[u], xs[u] + 8) & 0xffu;
                            t_end = (uint32_t)load_window8p(a.tcodes, (long long)s_ta[wv][ow[u]], false, false, xs[u] + 8) & 0xffu;
                        }
                    }
                }
                int nq[CU], nt[CU];
                if (a.kshift) {
#pragma unroll
                    for (int u = 0; u < CU; ++u) {
                        const int q_after = u + 1 < CU ? __builtin_amdgcn_readlane((int)((uint32_t)q8[u + 1 < CU ? u + 1 : u] & 0xffu), 0) : (int)q_end;
                        const int t_after = u + 1 < CU ? __builtin_amdgcn_readlane((int)((uint32_t)t8[u + 1 < CU ? u + 1 : u] & 0xffu), 0) : (int)t_end;
                        nq[u] = wave_shl1((int)((uint32_t)q8[u] & 0xffu), q_after);
                        nt[u] = wave_shl1((int)((uint32_t)t8[u] & 0xffu), t_after);
                    }
                }
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    if (left[u] <= 0) continue;
                    const uint64_t keep = left[u] >= 8 ? ~0ull : (1ull << (8 * left[u])) - 1ull;
                    uint64_t d = (q8[u] ^ t8[u]) & keep;
                    if ((q8[u] | t8[u]) & keep & 0x0404040404040404ull) atomicOr(&s_amb[wv][ow[u] >> 5], 1u << (ow[u] & 31));
                    d = (d | d >> 1 | d >> 2) & 0x0101010101010101ull;      // codes are 0..4: three bits
                    // byte y set -> bit y: the four flags of a half gather in its top byte
                    auto squeeze = [](uint64_t f) {
                        return (uint8_t)((((uint32_t)f * 0x10204080u) >> 28) | (((uint32_t)(f >> 32) * 0x10204080u) >> 28) << 4);
                    };
                    mm[c0 + 64u * u + (uint32_t)lane] = squeeze(d);
                    if (a.kshift) {       // fourth certificate: bit i = q[i + 1] != t[i] / q[i] != t[i + 1]
                        // the windows one base further: the base behind a piece is the first of the next piece (the next lane;
                        // where that is another block's, the position lies outside [p1, pk) and is never looked at)
                        const uint64_t q9 = q8[u] >> 8 | (uint64_t)(uint32_t)nq[u] << 56, t9 = t8[u] >> 8 | (uint64_t)(uint32_t)nt[u] << 56;
                        uint64_t du = q9 ^ t8[u], dd = q8[u] ^ t9;
                        du = (du | du >> 1 | du >> 2) & 0x0101010101010101ull;
                        dd = (dd | dd >> 1 | dd >> 2) & 0x0101010101010101ull;
                        mup[c0 + 64u * u + (uint32_t)lane] = squeeze(du);
                        mdn[c0 + 64u * u + (uint32_t)lane] = squeeze(dd);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (try_fast) {
                ambig = (s_amb[wv][lane >> 5] >> (lane & 31)) & 1u;
                for (uint32_t j = 0; j < nch && k <= a.kmax + a.kshift; ++j) {
                    uint32_t bits = mm[st0 + j];
                    while (bits) {
                        const int y = __ffs((int)bits) - 1;
                        bits &= bits - 1;
                        if (k < 3) mpos[k] = (int)(8 * j) + y;
                        ++k;
                    }
                }
            }
            // Fourth certificate, square blocks with exactly kmax + 1 substitutions at p1 < .. < pk.  A gapped path scores at
            // most match (m - 1 - x) - mismatch x - 2 (open + ext) with x mismatching diagonal moves, and by the choice of
            // kmax it beats the diagonal only with x = 0, one inserted and one deleted base (the host checks that longer or
            // further gaps cannot make up their cost): it follows the main diagonal without a mismatch up to p1 at the
            // latest, runs one diagonal higher or lower past pk with every pair matching, and comes back.  The least
            // constrained such path leaves at p1 and returns after pk: if neither (q[i + 1], t[i]) nor (q[i], t[i + 1])
            // match for all i in [p1, pk), no gapped path ties or beats the diagonal - it is the unique optimum.
            if (try_fast && !ambig && a.kshift && k == a.kmax + 1) {
                const int p1 = mpos[0], pk = mpos[k - 1];
                uint32_t any_up = 0, any_dn = 0;                     // a mismatch of the shifted segment inside [p1, pk)
                for (int j = p1 >> 3; j <= (pk - 1) >> 3; ++j) {
                    uint32_t keepb = 0xffu;
                    if (j == p1 >> 3) keepb &= 0xffu << (p1 & 7);
                    if (j == (pk - 1) >> 3) keepb &= 0xffu >> (7 - ((pk - 1) & 7));
                    any_up |= mup[st0 + (uint32_t)j] & keepb;
                    any_dn |= mdn[st0 + (uint32_t)j] & keepb;
                }
                shift_ok = any_up != 0 && any_dn != 0;
            }
            if (try_fast && k > 0 && !ambig) {                  // last substitution of the diagonal (suffix trim below)
                for (int j = (int)nch - 1; j >= 0; --j) {
                    const uint32_t bits = mm[st0 + (uint32_t)j];
                    if (bits) { last_x = 8 * j + 31 - __clz((int)bits); break; }
                }
            }
            __builtin_amdgcn_wave_barrier();                   // the next round reuses the arrays
        }
        bool fast = try_fast && !ambig && k <= a.kmax;
        if (shift_ok) fast = true;
        uint32_t runs[7];
        uint32_t nr = 0;
        int fast_score = a.match * (m - k) - a.mismatch * k;
        // Second certificate, blocks with m != n: if a prefix on the start diagonal and a suffix on the end diagonal
        // match exactly and together cover the shorter sequence, "prefix, ONE gap of |n - m|, suffix" scores
        // match * min(m,n) - open - ext * |n - m|, which no alignment of an m x n block can beat (fewer columns or
        // more gap cost).  Every optimal alignment is then of this form; the traceback prefers the diagonal on ties,
        // i.e. it leaves the end diagonal as late as it can: the gap sits at the leftmost admissible place.
        // Third certificate, the end extensions: if the two sequences agree base for base over L = min(m, n) elements
        // (no ambiguous base), the cell (L, L) scores match * L and every other cell (i, j) at most match * min(i, j):
        // it is the unique best cell and the alignment is L matches.  With an end bonus in play (short mode) a cell of
        // the bonus row could outrank it, so the certificate then needs (L, L) to lie in that row itself.
        uint32_t ext_flag = 0;
        int ext_take = -1;                   // sixth certificate: rows = columns of the cell the extension stops in
        if (PASS == 2 && live && c != 0 && (tk.kind & 3) != 0 && a.kmax >= 0) {
            const bool rev = (tk.kind & TASK_REV) != 0, left = (tk.kind & 3) == 1;
            const int L = m < n ? m : n;
            const int end_row = (int)(tk.narrow >> 1) - 1;
            // the row that earns the end bonus exists in this task's band: the band holds the cells with j - i >= -(BAND_W / 2 - 1),
            // so a query end more than that many rows below the last target base is out of reach (a short read running off the end
            // of its target: the usual right end of a read-to-read overlap) and the task is an extension without a bonus
            const bool bonus_row = a.end_bonus != 0 && end_row >= 0 && end_row <= n + (BAND_W / 2 - 1);
            bool same = !bonus_row || L == end_row;
            // One-piece certificate for the extensions the DP will get.  The second piece of the gap cost prices a gap at or
            // below the first from g* = ceil((open2 - open) / (ext - ext2)) bases on; a path with such a gap ends in a cell
            // (i, j) with at most match * min(i, j) - (open2 + ext2 g*) <= match * L - cost*, L = min(m, n).  The main diagonal alone reaches
            // match * L - (match + mismatch) * k at (L, L) with k substitutions: if that is MORE than the bound, no path with
            // a long gap is the best cell's path or ties with it anywhere on it (a tie would make a long-gap path of the best
            // score), every cell on the traceback has the same arrivals under both costs, and the DP with the first piece
            // alone returns the same cell and path (k <= allow below; no end bonus in play, no ambiguous base).
            int allow = -1, k_ext = 0;
            bool amb_ext = false;
            if (a.go2 > 0 && a.ge > a.ge2 && a.end_bonus == 0 && a.one_ok) {
                int g = (a.go2 - a.go + (a.ge - a.ge2) - 1) / (a.ge - a.ge2);
                g = g < 1 ? 1 : g;
                allow = (a.go2 + a.ge2 * g - 1) / (a.match + a.mismatch);      // (match + mismatch) * k < cost*
            }
            // (the first two substitutions' places: sixth certificate below)
            const int k_want = bonus_row ? (L == end_row ? a.kext_bonus : 0) : a.kext_plain;
            int x1 = -1, x2 = -1;
            auto more = [&]() { return same || (!amb_ext && (k_ext <= allow || k_ext <= k_want)); };
            auto take8 = [&](int x, uint64_t q8, uint64_t t8) {
                const int rest = L - x;
                const uint64_t keep = rest >= 8 ? ~0ull : (1ull << (8 * rest)) - 1ull;
                uint64_t d = (q8 ^ t8) & keep;
                amb_ext |= ((q8 | t8) & keep & 0x0404040404040404ull) != 0;
                d = (d | d >> 1 | d >> 2) & 0x0101010101010101ull;
                if (d && x2 < 0) {
                    const int y1 = (__ffsll((long long)d) - 1) >> 3;
                    if (x1 < 0) { x1 = x + y1; const uint64_t d2 = d & (d - 1); if (d2) x2 = x + ((__ffsll((long long)d2) - 1) >> 3); }
                    else x2 = x + y1;
                }
                k_ext += __popcll(d);
                same = same && d == 0 && !amb_ext;
            };
            for (int x = 0; x < L && more(); x += 16) {            // two steps' loads in flight together
                const uint64_t q8 = load_window8p(a.qcodes, (long long)tk.qa, left != rev, rev, x);
                const uint64_t t8 = load_window8p(a.tcodes, (long long)tk.ta, left, false, x);
                uint64_t q8b = 0, t8b = 0;
                if (x + 8 < L) {
                    q8b = load_window8p(a.qcodes, (long long)tk.qa, left != rev, rev, x + 8);
                    t8b = load_window8p(a.tcodes, (long long)tk.ta, left, false, x + 8);
                }
                take8(x, q8, t8);
                if (x + 8 < L && more()) take8(x + 8, q8b, t8b);
            }
            if (!same && !amb_ext && k_ext <= allow) { tk.kind |= TASK_ONE; ++st[ST_WIDE_ONE]; }
            if (same) {
                runs[nr++] = (uint32_t)L << 4 | OP_EQ;
                fast_score = a.match * L;
                // the row that reaches the query end earns the bonus in the caller's keep / drop decision
                ext_flag = L == end_row ? 0x80000000u : 0u;
                fast = true;
                c = 0;
            } else if (!amb_ext && k_ext >= 1 && k_ext <= k_want) {
                // Sixth certificate: an extension whose first L = min(m, n) pairs differ in k places and hold no ambiguous base.
                // U = match + mismatch, G = the cheapest one-base gap (open + ext of the cheaper piece); the host admits the
                // certificate only when the inequalities used here hold for the scoring constants.
                // * Any cell (i, j) reached with a gap scores at most match min(i, j) - G; cell (i, i) of the main diagonal
                //   scores D(i) = match i - U (substitutions before i).  A path into a diagonal cell that leaves the diagonal
                //   needs an insertion AND a deletion: at most match (i - 1) - 2 G < D(i) for k U < 2 G + match, so the
                //   diagonal is the unique best path into its own cells.
                // * No bonus row (long mode, or the query end out of reach), k = 1 at x: the candidates are (x, x) with match x
                //   - every pair before the substitution matches - and (L, L) with match (L - 1) - mismatch; gapped cells stay
                //   at or below match L - G <= D(L) (G >= U), and where they tie they hold more bases (i + j > 2 L), which the
                //   best-cell rule ranks behind; the same bound keeps them at or below (x, x) when that is the larger one.  Ties
                //   between the two go to (x, x): fewer bases.
                // * The end cell (L, L) lies in the bonus row (L = end_row, short mode), k <= kext_bonus: it ranks D(L) + bonus;
                //   the other cells of that row need a gap - at most match L - G + bonus, below it for k U < G - and every cell
                //   of another row ranks at most match (L - 1) < D(L) + bonus for k U < bonus + match.
                int take = L;                                     // rows (= columns) of the chosen cell
                if (!bonus_row && a.match * x1 >= a.match * (L - 1) - a.mismatch) take = x1;
                const int subs = take == L ? k_ext : 0;
                int prev = 0;
                if (take > 0) {
                    if (subs >= 1) {
                        if (x1 > 0) runs[nr++] = (uint32_t)x1 << 4 | OP_EQ;
                        if (subs == 2 && x2 == x1 + 1) { runs[nr++] = 2u << 4 | OP_X; prev = x2 + 1; }
                        else {
                            runs[nr++] = 1u << 4 | OP_X; prev = x1 + 1;
                            if (subs == 2) {
                                runs[nr++] = (uint32_t)(x2 - prev) << 4 | OP_EQ;
                                runs[nr++] = 1u << 4 | OP_X; prev = x2 + 1;
                            }
                        }
                    }
                    if (take > prev) runs[nr++] = (uint32_t)(take - prev) << 4 | OP_EQ;
                }
                // element 0 of a left extension is the base next to the fixed point: the runs above are in element order,
                // the row wants them in sequence order
                if (left) for (uint32_t i = 0; i < nr / 2; ++i) { const uint32_t v = runs[i]; runs[i] = runs[nr - 1 - i]; runs[nr - 1 - i] = v; }
                fast_score = a.match * (take - subs) - a.mismatch * subs;
                ext_flag = bonus_row && take == end_row ? 0x80000000u : 0u;
                ext_take = take;
                fast = true;
                c = 0;
                ++st[ST_EXT_CERT];
            } else if (bonus_row && L != end_row) ++st[ST_EXT_DP_ROW];       // (why an extension goes to the DP: statistics)
            else ++st[ST_EXT_DP_K];
        } else if (PASS == 2 && live && c != 0 && (tk.kind & 3) == 0 && m != n && a.kmax >= 0) {
            const bool rev = (tk.kind & TASK_REV) != 0;
            const int mn = m < n ? m : n, gap = n > m ? n - m : m - n;
            const bool del = n > m;                                       // the gap consumes target bases
            // Both certificates come from two scans: from the block's start along the start diagonal, from its end along
            // the end diagonal, each up to its first substitution (second certificate) or, for a one-base gap, its
            // second (fifth certificate).  a1 < a2: first substitutions of the start diagonal (mn: none), b1 > b2: last
            // ones of the end diagonal (-1: none).  "prefix [0,p) + gap + suffix [p,mn)" has
            // s(p) = #(a < p) + #(b >= p) substitutions.
            //   second certificate, s* = 0 (b1 < a1): no alignment of an m x n block beats match * min(m,n) - open -
            //     ext * |n - m| (fewer columns or more gap cost); the traceback prefers the diagonal on ties, i.e. it
            //     leaves the end diagonal as late as it can: the gap sits at the leftmost admissible place b1 + 1.
            //   fifth certificate, |n - m| = 1 and s* = 1: any alignment that is not of this form needs a further
            //     inserted + deleted base - at least open + 2 ext more gap cost and one diagonal move fewer, which
            //     avoiding ONE substitution cannot pay for (checked on the host) - so the optimal alignments are the
            //     placements with s(p) = 1: p in (b2, a1] with the substitution at b1, or p in (b1, a2] with it at a1;
            //     the first range lies left of the second and the traceback takes the leftmost p, as above.
            const int want = a.kgap1 && gap == 1 ? (a.kgap2 ? 3 : 2) : 1;
            int found_a = 0, found_b = 0, a1 = mn, a2 = mn, a3 = mn, b1 = -1, b2 = -1, b3 = -1;
            bool amb = false;
            // both scans in one loop: the start diagonal forwards, the end diagonal backwards, their four loads of a trip in
            // flight together (one after the other the kernel waited 80 % of its time on a chain of dependent trips)
            const int sq = m > n ? m - n : 0, st_ = n > m ? n - m : 0;   // shift of the end diagonal in q / t
            for (int x = 0; x < mn && (found_a < want || found_b < want); x += 8) {
                const bool da = found_a < want, db = found_b < want;
                const int e0 = mn - 8 - x;                                // end diagonal: elements mn-8-x .. mn-1-x
                uint64_t q8 = 0, t8 = 0, q8e = 0, t8e = 0;
                if (da) { q8 = load_window8p(a.qcodes, (long long)tk.qa, rev, rev, x); t8 = load_window8p(a.tcodes, (long long)tk.ta, false, false, x); }
                if (db) { q8e = load_window8p(a.qcodes, (long long)tk.qa, rev, rev, e0 + sq); t8e = load_window8p(a.tcodes, (long long)tk.ta, false, false, e0 + st_); }
                const int left = mn - x;
                if (da) {
                    const uint64_t keep = left >= 8 ? ~0ull : (1ull << (8 * left)) - 1ull;
                    amb |= ((q8 | t8) & keep & 0x0404040404040404ull) != 0;
                    uint64_t d = (q8 ^ t8) & keep;
                    d = (d | d >> 1 | d >> 2) & 0x0101010101010101ull;
                    while (d && found_a < want) {
                        const int y = (__ffsll((long long)d) - 1) >> 3;
                        d &= d - 1;
                        if (found_a == 0) a1 = x + y; else if (found_a == 1) a2 = x + y; else a3 = x + y;
                        ++found_a;
                    }
                }
                if (db) {
                    const uint64_t keep = left >= 8 ? ~0ull : ~0ull << (8 * (8 - left));
                    amb |= ((q8e | t8e) & keep & 0x0404040404040404ull) != 0;
                    uint64_t d = (q8e ^ t8e) & keep;
                    d = (d | d >> 1 | d >> 2) & 0x0101010101010101ull;
                    while (d && found_b < want) {
                        const int z = 7 - (__clzll((long long)d) >> 3);      // highest set byte
                        d &= ~(0xffull << (8 * z));
                        if (found_b == 0) b1 = e0 + z; else if (found_b == 1) b2 = e0 + z; else b3 = e0 + z;
                        ++found_b;
                    }
                }
            }
            int p_star = -1, xpos = -1, xpos2 = -1;
            if (!amb) {
                if (b1 < a1) p_star = b1 + 1;                                             // s* = 0
                else if (want >= 2 && b2 < a1) { p_star = b2 + 1; xpos = b1; }            // s* = 1, substitution in the suffix
                else if (want >= 2 && b1 < a2) { p_star = b1 + 1; xpos = a1; }   // ... in the prefix
            }
            // Seventh certificate, |n - m| = 1 and s* = 2 (a1 < b1 here: with b1 <= a1 a placement had s <= 1).  U = match +
            // mismatch, c(L) = cost of a gap of L bases.  Against "prefix, one gap base, suffix" with two substitutions - score
            // match mn - 2 U - c(1) - a path with k further inserted and k further deleted bases has k diagonal moves fewer and
            // more gap cost: with k = 1 and the two deleted (inserted) bases in ONE gap it pays match + c(2) and scores HIGHER
            // if it meets no mismatch at all (2 U > match + c(2) with the usual constants), every other arrangement - a
            // mismatch on the way, three separate gaps, k >= 2 - stays below (the host checks U < match + c(2),
            // 2 U < match + 2 c(1), 2 U < 2 match + c(2) + c(3) - c(1)).  A path without a mismatch follows the start diagonal up
            // to row a1 at the latest and the end diagonal from row b1 + 1 at the earliest; in between it runs on the diagonal
            // two to the far side of the end diagonal (long gap first: rows [a1, b1) at least) or on the one before the start
            // diagonal (short gap first: rows [a1 + 1, b1] at least).  A mismatch of the block with itself shifted that way in
            // each of the two ranges rules both out: the optimal alignments are then the placements with s(p) = 2 - p in
            // (b3, min(a1, b2)] with the substitutions at b2, b1, or in (max(a1, b2), min(a2, b1)] with a1, b1, or in
            // (max(a2, b1), a3] with a1, a2; the ranges lie left to right in this order and the traceback takes the leftmost p.
            if (p_star < 0 && !amb && want == 3) {
                int ps = -1, x1 = -1, x2 = -1;
                const int mab = a1 < b2 ? a1 : b2, Mab = a1 > b2 ? a1 : b2, m2 = a2 < b1 ? a2 : b1, M2 = a2 > b1 ? a2 : b1;
                if (b2 >= 0 && b3 < mab) { ps = b3 + 1; x1 = b2; x2 = b1; }
                else if (a1 < mn && b1 >= 0 && Mab < m2) { ps = Mab + 1; x1 = a1; x2 = b1; }
                else if (a2 < mn && M2 < a3) { ps = M2 + 1; x1 = a1; x2 = a2; }
                if (ps >= 0 && a1 < b1) {
                    // S = the shorter sequence (rows of the argument above), L = the longer one
                    auto load_s = [&](int x) { return del ? load_window8p(a.qcodes, (long long)tk.qa, rev, rev, x) : load_window8p(a.tcodes, (long long)tk.ta, false, false, x); };
                    auto load_l = [&](int x) { return del ? load_window8p(a.tcodes, (long long)tk.ta, false, false, x) : load_window8p(a.qcodes, (long long)tk.qa, rev, rev, x); };
                    bool far_ne = false, near_ne = false, ambx = false;
                    for (int x = a1; x < b1 && !far_ne; x += 8) {                 // S[i] vs L[i + 2], i in [a1, b1)
                        const uint64_t s8 = load_s(x), l8 = load_l(x + 2);
                        const int left = b1 - x;
                        const uint64_t keep = left >= 8 ? ~0ull : (1ull << (8 * left)) - 1ull;
                        ambx |= ((s8 | l8) & keep & 0x0404040404040404ull) != 0;
                        far_ne = ((s8 ^ l8) & keep) != 0;
                    }
                    for (int x = a1 + 1; x <= b1 && !near_ne; x += 8) {           // S[i] vs L[i - 1], i in [a1 + 1, b1]
                        const uint64_t s8 = load_s(x), l8 = load_l(x - 1);
                        const int left = b1 + 1 - x;
                        const uint64_t keep = left >= 8 ? ~0ull : (1ull << (8 * left)) - 1ull;
                        ambx |= ((s8 | l8) & keep & 0x0404040404040404ull) != 0;
                        near_ne = ((s8 ^ l8) & keep) != 0;
                    }
                    if (far_ne && near_ne && !ambx) { p_star = ps; xpos = x1; xpos2 = x2; }
                }
            }
            if (p_star < 0 && !amb && b1 >= 0) last_x = b1;     // (suffix trim below; positions count along the shorter sequence)
            if (p_star >= 0) {
                auto seg = [&](int from, int to) {            // [from, to) of one diagonal, with the substitutions that lie inside
                    int prev = from;                          // (xpos < xpos2 when both are set; neighbours share a run)
                    for (int k = 0; k < 2; ++k) {
                        const int xp = k ? xpos2 : xpos;
                        if (xp < from || xp >= to) continue;
                        if (xp > prev) runs[nr++] = (uint32_t)(xp - prev) << 4 | OP_EQ;
                        if (nr && xp == prev && prev > from && (runs[nr - 1] & 15u) == OP_X) runs[nr - 1] += 1u << 4;
                        else runs[nr++] = 1u << 4 | OP_X;
                        prev = xp + 1;
                    }
                    if (to > prev) runs[nr++] = (uint32_t)(to - prev) << 4 | OP_EQ;
                };
                seg(0, p_star);
                runs[nr++] = (uint32_t)gap << 4 | (del ? OP_D : OP_I);
                seg(p_star, mn);
                const int subs = (xpos >= 0 ? 1 : 0) + (xpos2 >= 0 ? 1 : 0);
                int gap_cost = a.go + a.ge * gap;
                if (a.go2 && a.go2 + a.ge2 * gap < gap_cost) gap_cost = a.go2 + a.ge2 * gap;       // the cheaper piece
                fast_score = a.match * (mn - subs) - a.mismatch * subs - gap_cost;
                fast = true;
                c = 0;
            }
        } else if (fast) {
            c = 0;
            int prev = 0, xs = -1, xe = -1;
            for (int i = 0; i < k; ++i) {
                const int x = mpos[i];
                if (x == xe) { xe = x + 1; continue; }
                if (xs >= 0) { runs[nr++] = (uint32_t)(xe - xs) << 4 | OP_X; prev = xe; }
                if (x > prev) runs[nr++] = (uint32_t)(x - prev) << 4 | OP_EQ;
                xs = x; xe = x + 1;
            }
            if (xs >= 0) { runs[nr++] = (uint32_t)(xe - xs) << 4 | OP_X; prev = xe; }
            if (m > prev) runs[nr++] = (uint32_t)(m - prev) << 4 | OP_EQ;
        }
        if (bare) nr = 0;                                     // (the score is all a candidate's task reports)
        uint32_t wave_total;
        const uint32_t mine = wave_excl_sum_u32(nr, lane, wave_total);
        uint32_t base = 0;
        bool ok = true;
        if (wave_total) {
            if (lane == 0) base = pool_take(a, wave_total, chunk_off, chunk_left, ok);
            base = (uint32_t)__shfl((int)base, 0, 64);
            ok = __shfl((int)ok, 0, 64) != 0;
        }
        if (fast) {
            if (ok) for (uint32_t q = 0; q < nr; ++q) a.runs[base + mine + q] = runs[q];
            const bool ext = (tk.kind & 3) != 0;              // extensions report the cell they stop in
            const int L = ext_take >= 0 ? ext_take : (m < n ? m : n);
            a.out[ti] = TaskOut{fast_score, ext ? L : m, ext ? L : n, base + mine, ok ? nr : 0, (nr ? end_codes(runs[0], runs[nr - 1]) : 0u) | ext_flag};
        } else if (live && c == 0) {
            a.out[ti] = TaskOut{0, 0, 0, 0, 0, 0};
        }
        // Suffix trim of the near-diagonal DP tasks.  Behind the last substitution of the end diagonal the two sequences
        // agree base for base (no ambiguous code): T bases.  In the (banded) recurrence a matching pair makes the diagonal
        // move at least as good as any gap into the same cell - H(i-1,j-1) + match >= H(i,j-k) - open - k ext for every k
        // (take the path into (i,j-k): if it ends with a diagonal move, the same gap one row higher plus this match scores
        // at least as much; if it ends in a gap, joining or shortening the gaps does) and likewise for the other gap
        // direction, all within the same diagonals - and the traceback prefers the diagonal on ties: from (m, n) it walks
        // the T cells of the suffix diagonally whatever lies before.  The cells of the first m - T rows do not depend on
        // the suffix, so the DP of the block without it followed by T matches is the DP of the block, bit for bit
        // (needs match > 0 and gap costs >= 0, as the certificates; one row is always left).
        int m_dp = m;
        if (live && !fast && (c == 1 || c == 3) && a.kmax >= 0 && last_x >= 0 && a.trim_ok && !a.ungapped) {
            const int mn = m < n ? m : n;
            int T = mn - 1 - last_x;
            T = T < mn - 1 ? T : mn - 1;
            if (T > 0) {
                tk.m = (int16_t)(m - T); tk.n = (int16_t)(n - T); tk.trim = (int16_t)T;
                m_dp = m - T;
                c = m_dp <= NR_SHORT ? 1 : 3;
            }
        }
        if (live) {
            if (a.ungapped && c != 0) c = CLS_UNGAPPED;       // (the certificates above hold for the diagonal band as well)
            cls[ti] = bare ? 0 : c;
            if (a.cls_bare) a.cls_bare[ti] = bare ? c : 0;
            if (c != 0) a.tasks[ti] = tk;                      // a DP kernel will want the record
            if (held) ++st[ST_STUB_EXT];
            else if (m > 0 && n > 0) {
                const uint32_t bases = (uint32_t)(m + n);
                st[ST_BASES] += bases;
                if ((tk.kind & 3) == 0 && m == n) st[ST_BASES_SQUARE] += bases;
                if (c == 0) ++st[ST_FAST];
                else {
                    ++st[ST_DP];
                    if (c == CLS_LONG || c == CLS_LONG32) { ++st[ST_LONG]; st[ST_BASES_LONG] += bases; st[ST_DP_ROWS] += (uint32_t)(m < n - tk.dlo ? m : n - tk.dlo); }
                    else if (c == 2 || c == 4) { st[ST_BASES_WIDE] += bases; st[ST_DP_ROWS] += (uint32_t)(m < n - tk.dlo ? m : n - tk.dlo); }
                    else { st[ST_BASES_NARROW] += bases; st[ST_DP_ROWS] += (uint32_t)m_dp; if (c == 1 && m_dp < NR_SMALL && !bare) ++st[ST_NARROW_SMALL]; }
                }
            }
        }
    }
    // counters: wave sum -> block sum -> one atomic per block and counter
#pragma unroll
    for (int k = 0; k < N_ALIGN_STATS; ++k) {
        unsigned long long v = st[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_stat[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < N_ALIGN_STATS) {
        unsigned long long v = 0;
        for (int w = 0; w < WAVES; ++w) v += s_stat[w][threadIdx.x];
        if (v) atomicAdd(&stats[threadIdx.x], v);
    }
}

// sort key of a DP task list: rows / 4, so the four tasks of a wave of align_narrow_kernel run about equally long
// (the sort is stable: inside a bucket the tasks keep their order, neighbours in the reads stay neighbours)
__global__ void task_rows_key_kernel(const Task *tasks, const uint32_t *list, size_t n, uint32_t *key) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) key[i] = (uint32_t)tasks[list[i]].m >> 2;
}


// sort key of the LONG task list: rows the task really runs, descending
__global__ void task_rows_desc_key_kernel(const Task *tasks, const uint32_t *list, size_t n, uint32_t *key) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Task &t = tasks[list[i]];
    const int rows = (int)t.m < (int)t.n - (int)t.dlo ? (int)t.m : (int)t.n - (int)t.dlo;
    key[i] = 0xffffu - (uint32_t)(rows < 0 ? 0 : rows);
}

// ---- pass 2a: DP of near-diagonal blocks, FOUR tasks per wave (one per row of 16 lanes) --------------------------
// Same recurrences and tie rules as align_kernel with W = 16; every cross-lane step is a DPP row operation, so
// the four groups of a wave never interact.  What differs is the bookkeeping:
//   * no validity masks: the band starts from H(0,j) and -inf elsewhere, cells left of column 0 stay -inf by
//     themselves, cells right of column n and rows below m compute garbage that nothing valid depends on
//     (every DP dependence goes to an equal or larger j) and that the traceback never visits;
//   * the traceback bits stay with the lane (= diagonal) that produced them: five bit planes, one bit per
//     row, shifted into 32-bit accumulators by v_addc (carry-in = the compare) and flushed to LDS every 32 rows
//     (row i = bit 31 - (i-1)%32 of word (i-1)/32);
//       DIAG  mm == h              the cell took the diagonal move (ties: M > E > F)
//       EGEF  e >= f               otherwise E, else F
//       EEXT  e + open > h         E of the cell to the RIGHT extends this cell's E (its flagE)
//       FEXT  f - ext > h - o - e  F of the cell BELOW extends this cell's F (its flagF)
//       NE    q != t               '=' or 'X' of a diagonal move
//   * a diagonal of the traceback is then a run of ones in ONE word: count-trailing-ones finds its length, the
//     NE word its mismatches; the walk is per event, not per base, and is done twice (count, then write) so the
//     runs go straight into the pool in forward order.
constexpr int DP_BIAS = 1 << 24;             // added to every DP score: 0 then acts as -inf (block scores stay within +-4096)
constexpr int NT_PAD = 16;                   // st index of target offset 0
constexpr int NT_EXTRA = NT_PAD + NARROW_DELTA + NARROW_W + 7;     // st length = rows + NT_EXTRA
enum { PL_DIAG = 0, PL_EGEF, PL_EEXT, PL_FEXT, PL_NE, N_PLANES };

// acc = 2 * acc + bit: one v_addc_co_u32 with the compare mask as carry-in
__device__ __forceinline__ uint32_t shift_in(uint32_t acc, bool bit) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long m = __ballot(bit);
    unsigned long long co;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(co) : "v"(acc), "s"(m));
#endif
    return acc;
}

struct NarrowWalk {
    const uint32_t *pl;                     // planes of this wave: word (plane, chunk, lane) at (plane * chunks + chunk) * 64 + lane
    int chunks;
    int lb;                                 // first lane of the group
    int m, n, dlo;                          // the walk starts at cell (m, n)
    int wmask;                              // band width - 1
    bool keep_order;                        // left extensions: the reversed sequences make end -> start the forward order
    int trim;                               // bases matched behind (m, n): the path ends with that many '='
    __device__ __forceinline__ uint32_t word(int plane, int c, int lane) const { return pl[(plane * chunks + c) * 64 + lane]; }
};
struct WalkScore {                          // score of the walked path from its runs (no ambiguous base on it)
    int match, mismatch, go, ge;
    int total;
};
// Traceback of one task by one lane.  Emits the runs in reverse (end -> start) and returns their number.  The
// first `cap` of them are kept in `buf` (LDS, emission order): a task with no more than that is walked once and its
// runs copied out by the whole group; with `out` (second walk of a longer task) they are written to
// out[total-1 .. 0].  Every iteration consumes at least one row or column, so m + n + 1 bounds the trip count; the
// cap only guards a corrupted plane.
// `ends` receives end_codes(first run, last run) of the forward order.
__device__ uint32_t narrow_walk(const NarrowWalk &w, uint32_t *out, uint32_t total, uint32_t *buf = nullptr, uint32_t cap = 0,
                                WalkScore *ws = nullptr, uint32_t *ends = nullptr) {
    int i = w.m, j = w.n, state = 0;
    uint32_t cur_op = OP_EQ, cur_len = (uint32_t)w.trim, n_runs = 0, e_first = 0, e_last = 0;
    auto put = [&]() {
        if (!n_runs) e_first = cur_op;
        e_last = cur_op;
        if (out) out[w.keep_order ? n_runs : total - 1 - n_runs] = cur_len << 4 | cur_op;
        else if (n_runs < cap) buf[n_runs] = cur_len << 4 | cur_op;
        ++n_runs;
        // a merged gap run is one gap of the DP (with open > 0 a gap never re-opens next to itself; with open = 0 it
        // costs the same either way)
        if (ws) ws->total += cur_op == OP_EQ ? ws->match * (int)cur_len
                           : (cur_op == OP_X ? -ws->mismatch * (int)cur_len : -(ws->go + ws->ge * (int)cur_len));
    };
    // One segment per iteration - a stretch of '=' or of 'X' on a diagonal (as far as one plane word reaches), or one
    // gap base - through ONE body without state-specific branches: the few lanes of a wave that walk at the same time
    // are in different states, and every branch they disagree on is executed for each of them in turn.
    for (int it = 0; (i > 0 || j > 0) && it < 4 * (EXT_MAX + SEQ_T_MAX); ++it) {
        uint32_t op, len;
        if (i == 0) {                                          // row 0: H(0,j) is a gap from the corner
            op = OP_D; len = (uint32_t)j; j = 0;
        } else {
            const int d = w.lb + ((j - i - w.dlo) & w.wmask);
            const int c = (i - 1) >> 5, sh = 31 - ((i - 1) & 31);          // row i = bit sh of word c; row i-1 = bit sh+1
            // the five words an iteration can need, requested together
            const uint32_t dg = w.word(PL_DIAG, c, d) >> sh;   // bit 0 = row i, bit 1 = row i-1, ... zeros above the word
            const uint32_t ne = w.word(PL_NE, c, d) >> sh;
            const uint32_t eg = w.word(PL_EGEF, c, d) >> sh;
            const uint32_t ee = w.word(PL_EEXT, c, (d - 1) & 63) >> sh;
            const int i2 = i > 1 ? i - 2 : 0;
            const uint32_t fe = w.word(PL_FEXT, i2 >> 5, (d + 1) & 63) >> (31 - (i2 & 31));
            // a cell reached in state 0 that did not take the diagonal opens a gap: E if e >= f, else F
            const int st = state != 0 ? state : ((dg & 1u) ? 0 : ((eg & 1u) ? 1 : 2));
            if (st == 0) {
                const uint32_t inv = ~dg;
                const int r = inv ? __ffs((int)inv) - 1 : 32;                 // diagonal moves in a row (this word)
                const bool isx = (ne & 1u) != 0;
                const uint32_t flip = isx ? ~ne : ne;                         // first row whose '=' / 'X' kind differs
                int l = flip ? __ffs((int)flip) - 1 : 32;
                l = l < r ? l : r;
                op = isx ? OP_X : OP_EQ; len = (uint32_t)l;
                i -= l; j -= l;
                state = 0;
            } else if (st == 1) {
                op = OP_D; len = 1;
                state = (ee & 1u) ? 1 : 0;                     // E of this cell extends the E of the cell to the left
                --j;
            } else {
                op = OP_I; len = 1;
                state = (i > 1 && (fe & 1u)) ? 2 : 0;          // F of this cell extends the F of the cell above
                --i;
            }
        }
        if (op == cur_op) cur_len += len;
        else { if (cur_len) put(); cur_op = op; cur_len = len; }
    }
    if (cur_len) put();
    if (ends) *ends = w.keep_order ? end_codes(e_first, e_last) : end_codes(e_last, e_first);
    return n_runs;
}
constexpr int RUN_BUF_NARROW = 48, RUN_BUF_WIDE = 128;   // runs kept in LDS by the first walk (per task)

template <bool AMBI, int NR_CHUNKS>
__device__ __forceinline__ void narrow_rows(const AlignArgs &a, int rows, int m, int dlo, int l, const uint8_t *sq,
                                            const uint8_t *st, uint32_t (*pl)[NR_CHUNKS][64], int lane, int &Hend) {
    const int go = a.go, ge = a.ge, goe = go + ge, gel = ge * l, goel = go + ge * l;
    const int j0 = dlo + l;
    // scores carry DP_BIAS and "-inf" is 0 (plus or minus a few hundred): lanes a DPP shift has no source for
    // are then simply zero-filled (bound_ctrl), no identity register to rebuild every row
    int H = j0 == 0 ? DP_BIAS : (j0 > 0 ? DP_BIAS - (go + ge * j0) : 0);      // row 0
    int G = H - goe;                                                 // max(H - open - ext, F - ext) with F = -inf
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    const uint8_t *tp = st + NT_PAD + dlo + l - 1;                    // target base of row i: tp[i]
    int v_match = a.match, v_mis = -a.mismatch;
    asm volatile("" : "+v"(v_match), "+v"(v_mis));                    // keep the two select operands in registers
    auto row = [&](int i, int qa, int t2) {
        const bool ne = qa != t2;
        int s = ne ? v_mis : v_match;
        if (AMBI) s = (qa | t2) > 3 ? -a.ambi : s;
        const int mm = H + s;
        const int f = row_shl1_z(G);                                  // from lane d+1 of the row above
        const int ht = mm > f ? mm : f;
        const int e = row_shr1(row_prefix_max_incl_dpp(ht + gel), 0) - goel;
        const int h = ht > e ? ht : e;
        const int fo = h - goe, fe = f - ge;
        a0 = shift_in(a0, mm == h);
        a1 = shift_in(a1, e >= f);
        a2 = shift_in(a2, e + go > h);
        a3 = shift_in(a3, fe > fo);
        a4 = shift_in(a4, ne);
        G = fo > fe ? fo : fe;
        H = h;
        if (i == m) Hend = h;
    };
    // two rows per trip (the bases of the next two rows are in flight meanwhile); a word of the planes fills up on
    // an even row
    int qa = sq[0], t2 = tp[1];
    int i = 1;
    for (; i < rows; i += 2) {
        const int qb = sq[i], tb = tp[i + 1], qc = sq[i + 1], tc = tp[i + 2];
        row(i, qa, t2);
        row(i + 1, qb, tb);
        qa = qc; t2 = tc;
        if (((i + 1) & 31) == 0) {
            const int c = ((i + 1) >> 5) - 1;
            pl[PL_DIAG][c][lane] = a0; pl[PL_EGEF][c][lane] = a1; pl[PL_EEXT][c][lane] = a2; pl[PL_FEXT][c][lane] = a3;
            pl[PL_NE][c][lane] = a4;
        }
    }
    if (i == rows) row(i, qa, t2);                                    // odd row count (rows is odd: no word boundary here)
    if (rows & 31) {        // partial word: move its first row up to bit 31
        const int c = rows >> 5, up = 32 - (rows & 31);
        pl[PL_DIAG][c][lane] = a0 << up; pl[PL_EGEF][c][lane] = a1 << up; pl[PL_EEXT][c][lane] = a2 << up;
        pl[PL_FEXT][c][lane] = a3 << up; pl[PL_NE][c][lane] = a4 << up;
    }
}

template <int NR_MAX>
__global__ __launch_bounds__(WG) void align_narrow_kernel(AlignArgs a) {
    constexpr int NR_CHUNKS = NR_MAX / 32, NQ_STEPS = NR_MAX / 64, NT_STEPS = NR_MAX / 64 + 1;
    __shared__ uint32_t s_pl[WAVES][N_PLANES][NR_CHUNKS][64];
    __shared__ __attribute__((aligned(4))) uint8_t s_q[WAVES * 4][NR_MAX + 4];
    __shared__ __attribute__((aligned(4))) uint8_t s_t[WAVES * 4][NR_MAX + NT_EXTRA];
    __shared__ uint32_t s_runs[WAVES * 4][RUN_BUF_NARROW];
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15, wv = threadIdx.x >> 6;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t (*pl)[NR_CHUNKS][64] = s_pl[wv];
    uint8_t *sq = s_q[wv * 4 + g], *st = s_t[wv * 4 + g];
    uint32_t chunk_off = 0, chunk_left = 0;
    const size_t n_quads = (a.n_list + 3) / 4;
    for (size_t qd = wave; qd < n_quads; qd += n_waves) {
        const size_t li = qd * 4 + g;
        const bool live = li < a.n_list;
        size_t ti = 0;
        Task tk{};
        if (live) { ti = a.list[li]; tk = a.tasks[ti]; }
        const int m = tk.m, n = tk.n, dlo = tk.dlo;
        bool ambig = false;
        if (live) {     // stage the two windows, 4 bases per lane and step (all loads in flight together)
            const bool rev = (tk.kind & TASK_REV) != 0;
            uint32_t vq[NQ_STEPS], vt[NT_STEPS];
#pragma unroll
            for (int r = 0; r < NQ_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                vq[r] = 0;
                if (x < m) vq[r] = load_window4p(a.qcodes, (long long)tk.qa, rev, rev, x);
            }
#pragma unroll
            for (int r = 0; r < NT_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                vt[r] = 0;
                if (x < n) vt[r] = load_window4p(a.tcodes, (long long)tk.ta, false, false, x);
            }
#pragma unroll
            for (int r = 0; r < NQ_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                if (x < m) { *(uint32_t *)(sq + x) = vq[r]; ambig |= (vq[r] & 0x04040404u) != 0; }
            }
#pragma unroll
            for (int r = 0; r < NT_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                if (x < n) { *(uint32_t *)(st + NT_PAD + x) = vt[r]; ambig |= (vt[r] & 0x04040404u) != 0; }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const int rows = (int)wave_max_u32_dpp(live ? (uint32_t)m : 0u);
        int Hend = 0;
        if (__any(ambig)) narrow_rows<true, NR_CHUNKS>(a, rows, m, dlo, l, sq, st, pl, lane, Hend);
        else narrow_rows<false, NR_CHUNKS>(a, rows, m, dlo, l, sq, st, pl, lane, Hend);
        const int score = __shfl(Hend, g * 16 + ((n - m - dlo) & (NARROW_W - 1)), 64) - DP_BIAS + a.match * tk.trim;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        uint32_t *rbuf = s_runs[wv * 4 + g];
        const uint32_t rcap = a.run_buf_cap < (uint32_t)RUN_BUF_NARROW ? a.run_buf_cap : (uint32_t)RUN_BUF_NARROW;
        uint32_t cp_n = 0, cp_off = 0;                          // runs the group copies out of LDS
        if (live && l == 0) {
            const NarrowWalk w{&pl[0][0][0], NR_CHUNKS, g * 16, m, n, dlo, NARROW_W - 1, false, tk.trim};
            uint32_t ends = 0;
            const uint32_t n_runs = narrow_walk(w, nullptr, 0, rbuf, rcap, nullptr, &ends);
            uint32_t off = 0;
            bool ok = true;
            if (n_runs) off = pool_take(a, n_runs, chunk_off, chunk_left, ok, RUN_CHUNK_SMALL);
            if (ok && n_runs) {
                if (n_runs <= rcap) { cp_n = n_runs; cp_off = off; }
                else narrow_walk(w, a.runs + off, n_runs);
            }
            a.out[ti] = TaskOut{score, m, n, off, ok ? n_runs : 0, ends};
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        cp_n = (uint32_t)__shfl((int)cp_n, g * 16, 64); cp_off = (uint32_t)__shfl((int)cp_off, g * 16, 64);
        for (uint32_t k = (uint32_t)l; k < cp_n; k += 16) a.runs[cp_off + k] = rbuf[cp_n - 1 - k];   // forward order
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- pass 2a, packed: EIGHT tasks per wave -------------------------------------------------------------------------
// Two tasks share every lane of a 16-lane group: their scores are the 16-bit halves of one register (bias 2^13, block
// scores stay within +-4096 of it - the host checks the scoring constants), the recurrences are v_pk_* instructions
// and every DPP move shifts both halves at once; only the ten compare + add-with-carry pairs of the bit planes are per
// task.  22 vector instructions per task row instead of 31, and eight lanes of a wave walk the tracebacks instead of four.
// The score of a task comes out of its walk (sum over the runs); a quad with an ambiguous base keeps H of the last row
// per task instead (a base pair with an ambiguous code scores -ambi whether it reads '=' or 'X').
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 s2_of(int x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ int int_of(s2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ s2 splat2(int v) { return s2{(short)v, (short)v}; }
__device__ __forceinline__ s2 max2(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
// per-half min(x, 1) and a * b + c (the C++ forms of these come out as compares and selects)
__device__ __forceinline__ s2 pk_min_u16(s2 x, s2 y) {
    s2 r = x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
#endif
    return r;
}
__device__ __forceinline__ s2 pk_mad_i16(s2 x, s2 y, s2 z) {
    s2 r = x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
#endif
    return r;
}
// lane l <- lane l-n of its row of 16, zero where there is none (no `old` register to prepare)
template <int CTRL>
__device__ __forceinline__ int dpp_z(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
constexpr int DP_BIAS16 = 1 << 13;
constexpr int PK_T0 = 12;                    // s_t2 index of target element dlo (dlo >= -(NARROW_PAD + NARROW_DELTA) = -10)

// sq2[i] = base i of the query windows of both tasks (A | B << 8); st2[PK_T0 - dlo + x] = target base x of each task,
// i.e. both tasks read their row-i base of lane l at the same index PK_T0 + l - 1 + i
// TB = false: scores only (tasks of stub candidates) - no traceback planes; H(m, n) of both tasks is kept instead
template <bool AMBI, int NR_CHUNKS, bool TB = true>
__device__ __forceinline__ void narrow_rows_pk(const AlignArgs &a, int rows, int mA, int mB, int dloA, int dloB, int l,
                                               const uint16_t *sq2, const uint16_t *st2, uint32_t (*plA)[NR_CHUNKS][64],
                                               uint32_t (*plB)[NR_CHUNKS][64], int lane, int &HendA, int &HendB) {
    const int go = a.go, ge = a.ge, goe = go + ge;
    const s2 v_goe = splat2(goe), v_ge = splat2(ge), v_go = splat2(go), v_gel = splat2(ge * l), v_goel = splat2(go + ge * l);
    auto h0 = [&](int j0) { return j0 == 0 ? DP_BIAS16 : (j0 > 0 ? DP_BIAS16 - (go + ge * j0) : 0); };
    s2 H = s2{(short)h0(dloA + l), (short)h0(dloB + l)};              // row 0 of both tasks
    s2 G = H - v_goe;
    uint32_t a0A = 0, a1A = 0, a2A = 0, a3A = 0, a4A = 0, a0B = 0, a1B = 0, a2B = 0, a3B = 0, a4B = 0;
    const uint16_t *tp = st2 + PK_T0 + l - 1;                         // target bases of row i: tp[i]
    s2 v_match = splat2(a.match), v_dm = splat2(-(a.match + a.mismatch)), v_one = splat2(1);
    const s2 v_ambi = splat2(a.ambi);
    asm volatile("" : "+v"(v_match), "+v"(v_dm), "+v"(v_one));        // keep the operands of the two asm forms in registers
    auto row = [&](int i, int qb, int tb) {                           // qb / tb: base of task A | base of task B << 8
        const s2 x = s2_of((int)__builtin_amdgcn_perm(0u, (uint32_t)(qb ^ tb), 0x0c010c00u));   // A | B << 16
        s2 s = pk_mad_i16(pk_min_u16(x, v_one), v_dm, v_match);       // match, or -mismatch where the bases differ
        if (AMBI) {
            const s2 amb01 = s2_of((int)__builtin_amdgcn_perm(0u, (uint32_t)(((qb | tb) >> 2) & 0x0101), 0x0c010c00u));
            s = s - amb01 * (s + v_ambi);
        }
        const s2 mm = H + s;
        const s2 f = s2_of(dpp_z<0x101>(int_of(G)));                  // from lane d+1 of the row above
        const s2 ht = max2(mm, f);
        // E by a prefix max over the row of 16; the zero a lane without a source receives is this band's -inf
        s2 p = ht + v_gel;
        p = max2(p, s2_of(dpp_z<0x111>(int_of(p))));
        p = max2(p, s2_of(dpp_z<0x112>(int_of(p))));
        p = max2(p, s2_of(dpp_z<0x114>(int_of(p))));
        p = max2(p, s2_of(dpp_z<0x118>(int_of(p))));
        const s2 e = s2_of(dpp_z<0x111>(int_of(p))) - v_goel;
        const s2 h = max2(ht, e);
        const s2 fo = h - v_goe, fe = f - v_ge, eo = e + v_go;
        if (TB) {
            a0A = shift_in(a0A, mm.x == h.x);   a0B = shift_in(a0B, mm.y == h.y);
            a1A = shift_in(a1A, e.x >= f.x);    a1B = shift_in(a1B, e.y >= f.y);
            a2A = shift_in(a2A, eo.x > h.x);    a2B = shift_in(a2B, eo.y > h.y);
            a3A = shift_in(a3A, fe.x > fo.x);   a3B = shift_in(a3B, fe.y > fo.y);
            a4A = shift_in(a4A, x.x != 0);      a4B = shift_in(a4B, x.y != 0);
        }
        G = max2(fo, fe);
        H = h;
        if (AMBI || !TB) { if (i == mA) HendA = h.x; if (i == mB) HendB = h.y; }
    };
    auto flush = [&](int c, int up) {
        if (!TB) return;
        plA[PL_DIAG][c][lane] = a0A << up; plA[PL_EGEF][c][lane] = a1A << up; plA[PL_EEXT][c][lane] = a2A << up;
        plA[PL_FEXT][c][lane] = a3A << up; plA[PL_NE][c][lane] = a4A << up;
        plB[PL_DIAG][c][lane] = a0B << up; plB[PL_EGEF][c][lane] = a1B << up; plB[PL_EEXT][c][lane] = a2B << up;
        plB[PL_FEXT][c][lane] = a3B << up; plB[PL_NE][c][lane] = a4B << up;
    };
    int qa = sq2[0], ta = tp[1];
    int i = 1;
    for (; i < rows; i += 2) {                                        // two rows per trip, the next two rows' bases in flight
        const int qb = sq2[i], tb = tp[i + 1], qc = sq2[i + 1], tc = tp[i + 2];
        row(i, qa, ta);
        row(i + 1, qb, tb);
        qa = qc; ta = tc;
        if (((i + 1) & 31) == 0) flush(((i + 1) >> 5) - 1, 0);
    }
    if (i == rows) row(i, qa, ta);                                    // odd row count (no word boundary here)
    if (rows & 31) flush(rows >> 5, 32 - (rows & 31));                // partial word: its first row up to bit 31
}

constexpr int PK_WAVES = 1;          // waves per workgroup of the packed kernel: its LDS (12 KB per wave with 128 rows) sets the occupancy
template <int NR_MAX, bool TB = true>
__global__ __launch_bounds__(64 * PK_WAVES) void align_narrow_pk_kernel(AlignArgs a) {
    constexpr int NR_CHUNKS = TB ? NR_MAX / 32 : 1, NQ_STEPS = NR_MAX / 64, NT_STEPS = NR_MAX / 64 + 1;
    constexpr int T2_LEN = NR_MAX + PK_T0 + NARROW_W + NARROW_DELTA + 12;
    // the run buffers of the walks share the LDS of the staged sequences (dead once the rows are done)
    constexpr int Q2_LEN = NR_MAX + 4, RUN_BUF = NR_MAX >= 128 ? RUN_BUF_NARROW : 40;
    constexpr int SEQ_WORDS = 4 * (Q2_LEN + T2_LEN + 1) / 2 + 2;
    static_assert(8 * RUN_BUF <= SEQ_WORDS, "run buffers do not fit the sequence area");
    __shared__ uint32_t s_pl[PK_WAVES][2][N_PLANES][NR_CHUNKS][64];
    __shared__ uint32_t s_seq[PK_WAVES][SEQ_WORDS];
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15, wv = threadIdx.x >> 6;
    uint16_t *const q2 = (uint16_t *)s_seq[wv] + g * Q2_LEN, *const t2 = (uint16_t *)s_seq[wv] + 4 * Q2_LEN + g * T2_LEN;
    uint32_t (*const s_runs)[RUN_BUF] = (uint32_t (*)[RUN_BUF])s_seq[wv];
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t chunk_off = 0, chunk_left = 0;
    const size_t n_octs = (a.n_list + 7) / 8;
    uint8_t *q2b = (uint8_t *)q2, *t2b = (uint8_t *)t2;
    for (size_t oc = wave; oc < n_octs; oc += n_waves) {
        // group g works on tasks 8 oc + 2 g (half A of its lanes) and 8 oc + 2 g + 1 (half B)
        const size_t liA = oc * 8 + (size_t)g * 2, liB = liA + 1;
        const bool liveA = liA < a.n_list, liveB = liB < a.n_list;
        size_t tiA = 0, tiB = 0;
        Task tA{}, tB{};
        if (liveA) { tiA = a.list[liA]; tA = a.tasks[tiA]; }
        if (liveB) { tiB = a.list[liB]; tB = a.tasks[tiB]; }
        bool ambig = false;
        auto stage = [&](const Task &t, int hf) {     // 4 bases per lane and step, stored as the bytes `hf` of the pair arrays
            const bool rev = (t.kind & TASK_REV) != 0;
            uint32_t vq[NQ_STEPS], vt[NT_STEPS];
#pragma unroll
            for (int r = 0; r < NQ_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                vq[r] = 0;
                if (x < t.m) vq[r] = load_window4p(a.qcodes, (long long)t.qa, rev, rev, x);
            }
#pragma unroll
            for (int r = 0; r < NT_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                vt[r] = 0;
                if (x < t.n) vt[r] = load_window4p(a.tcodes, (long long)t.ta, false, false, x);
            }
#pragma unroll
            for (int r = 0; r < NQ_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                if (x < t.m) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) q2b[2 * (x + k) + hf] = (uint8_t)(vq[r] >> (8 * k));
                    ambig |= (vq[r] & 0x04040404u) != 0;
                }
            }
            const int t0 = PK_T0 - t.dlo;
#pragma unroll
            for (int r = 0; r < NT_STEPS; ++r) {
                const int x = 4 * (l + 16 * r);
                if (x < t.n) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) t2b[2 * (t0 + x + k) + hf] = (uint8_t)(vt[r] >> (8 * k));
                    ambig |= (vt[r] & 0x04040404u) != 0;
                }
            }
        };
        if (liveA) stage(tA, 0);
        if (liveB) stage(tB, 1);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const int mA = tA.m, mB = tB.m;
        const int rows = (int)wave_max_u32_dpp((uint32_t)(mA > mB ? mA : mB));       // (idle halves: m = 0)
        int HendA = 0, HendB = 0;
        const bool amb = __any(ambig);
        if (amb) narrow_rows_pk<true, NR_CHUNKS, TB>(a, rows, mA, mB, tA.dlo, tB.dlo, l, q2, t2, s_pl[wv][0], s_pl[wv][1], lane, HendA, HendB);
        else narrow_rows_pk<false, NR_CHUNKS, TB>(a, rows, mA, mB, tA.dlo, tB.dlo, l, q2, t2, s_pl[wv][0], s_pl[wv][1], lane, HendA, HendB);
        // H(m, n) of each task sits in the lane of its end diagonal (quads with an ambiguous base only)
        const int endA = __shfl(HendA, g * 16 + ((tA.n - mA - tA.dlo) & (NARROW_W - 1)), 64) - DP_BIAS16;
        const int endB = __shfl(HendB, g * 16 + ((tB.n - mB - tB.dlo) & (NARROW_W - 1)), 64) - DP_BIAS16;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if constexpr (!TB) {                         // the score is H(m, n) (+ the trimmed matches): no walk, no runs
            const int hf = l & 1;
            if (l < 2 && (hf ? liveB : liveA)) {
                const Task &t = hf ? tB : tA;
                a.out[hf ? tiB : tiA] = TaskOut{(hf ? endB : endA) + a.match * t.trim, t.m, t.n, 0, 0, 0};
            }
            __builtin_amdgcn_wave_barrier();         // (the sequence area is the next pair's)
            continue;
        }
        // lanes 0 and 1 of every group walk the two tasks of the group
        const int hf = l & 1;
        const bool walker = l < 2 && (hf ? liveB : liveA);
        const int wm = hf ? tB.m : tA.m, wn = hf ? tB.n : tA.n, wd = hf ? tB.dlo : tA.dlo, wtrim = hf ? tB.trim : tA.trim;
        uint32_t *rbuf = s_runs[g * 2 + hf];
        const uint32_t rcap = a.run_buf_cap < (uint32_t)RUN_BUF ? a.run_buf_cap : (uint32_t)RUN_BUF;
        uint32_t cp_n = 0, cp_off = 0;
        if (walker) {
            const NarrowWalk w{&s_pl[wv][0][0][0][0] + hf * (N_PLANES * NR_CHUNKS * 64), NR_CHUNKS, g * 16, wm, wn, wd, NARROW_W - 1, false, wtrim};
            WalkScore ws{a.match, a.mismatch, a.go, a.ge, 0};
            uint32_t ends = 0;
            const uint32_t n_runs = narrow_walk(w, nullptr, 0, rbuf, rcap, &ws, &ends);
            uint32_t off = 0;
            bool ok = true;
            if (n_runs) off = pool_take(a, n_runs, chunk_off, chunk_left, ok, RUN_CHUNK_SMALL);
            if (ok && n_runs) {
                if (n_runs <= rcap) { cp_n = n_runs; cp_off = off; }
                else narrow_walk(w, a.runs + off, n_runs);
            }
            const int score = amb ? (hf ? endB : endA) + a.match * wtrim : ws.total;
            a.out[hf ? tiB : tiA] = TaskOut{score, wm, wn, off, ok ? n_runs : 0, ends};
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // both run buffers of the group go out through all its lanes (forward order)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const uint32_t n2 = (uint32_t)__shfl((int)cp_n, g * 16 + h2, 64), o2 = (uint32_t)__shfl((int)cp_off, g * 16 + h2, 64);
            const uint32_t *rb = s_runs[g * 2 + h2];
            for (uint32_t k = (uint32_t)l; k < n2; k += 16) a.runs[o2 + k] = rb[n2 - 1 - k];
        }
        __builtin_amdgcn_s_waitcnt(0);               // the buffers are the next pair's sequence area
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- pass 2b: the 64-diagonal band (wide blocks, end extensions), one task per wave ------------------------------
// Same scheme as align_narrow_kernel with the whole wave as one band: bit planes per lane, no validity masks in
// the recurrences.  Extensions additionally rank their cells (best score, then fewest bases, ksw2 end bonus on the
// row that reaches the query end): along one diagonal "fewest bases" is "first row", so every lane keeps its first
// best row and the 64 lanes are compared once at the end; only there the cells outside 0 <= j <= n are masked.
constexpr int WT_PAD = 64;                                  // st index of target offset 0 (dlo >= -51)
constexpr int WT_LEN = WT_PAD + SEQ_T_MAX + 64;

// Planes of the 64-diagonal kernel.  Two-piece gap cost: min(go + ge L, go2 + ge2 L) - every gap state exists once per
// piece, E = max(E1, E2), F = max(F1, F2), the first piece on ties (oracle/ava_oracle.c:band_dp).  With one piece the
// second piece's planes stay zero and its states at "-inf".
//   DIAG  mm == h                 the cell took the diagonal move (ties: M > E > F)
//   EGEF  e >= f                  otherwise E, else F
//   EP/FP e2 > e1 / f2 > f1       the gap state that made E / F is the second piece
//   EX1/2 e_p + go_p > h          E_p of the cell to the RIGHT extends this cell's E_p
//   FX1/2 f_p - ge_p > h - go_p - ge_p   F_p of the cell BELOW extends this cell's F_p
//   NE    q != t
enum { WP_DIAG = 0, WP_EGEF, WP_EP, WP_FP, WP_EX1, WP_EX2, WP_FX1, WP_FX2, WP_NE, N_WPLANES };

template <bool AMBI, bool EXT, bool TWO, int W_CHUNKS, bool TB = true>
__device__ __forceinline__ void wide_rows(const AlignArgs &a, int m, int n, int dlo, int lane, int end_row,
                                          const uint8_t *sq, const uint8_t *st, uint32_t (*pl)[W_CHUNKS][64], int &Hend,
                                          int &best_h, int &best_i) {
    const int go = a.go, ge = a.ge, goe = go + ge, gel = ge * lane, goel = go + ge * lane;
    const int go2 = a.go2, ge2 = a.ge2, goe2 = go2 + ge2, gel2 = ge2 * lane, goel2 = go2 + ge2 * lane;
    const int j0 = dlo + lane;
    // row 0 (biased scores, see narrow_rows): a gap from the corner, the cheaper piece
    int gap0 = go + ge * j0;
    if (TWO && go2 + ge2 * j0 < gap0) gap0 = go2 + ge2 * j0;
    int H = j0 == 0 ? DP_BIAS : (j0 > 0 ? DP_BIAS - gap0 : 0);
    int G = H - goe, G2 = TWO ? H - goe2 : 0;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    int v_match = a.match, v_mis = -a.mismatch;
    asm volatile("" : "+v"(v_match), "+v"(v_mis));
    const uint8_t *tp = st + WT_PAD + dlo + lane - 1;                 // target base of row i: tp[i]
    // cell (i, j) is inside the rectangle for imin <= i <= imax (j = i + dlo + lane)
    const int imin = -j0, span = n - j0 - imin;                       // span < 0: never
    best_h = 0; best_i = 0;                                           // biased like H: 0 = no cell yet
    if (EXT && j0 >= 0 && j0 <= n) best_h = H;                        // row 0 (only (0,0) can win, the rest is a gap from it)
    int qa = sq[0], t2 = tp[1];
    for (int i = 1; i <= m; ++i) {
        const int qa_next = sq[i], t2_next = tp[i + 1];
        const bool ne = qa != t2;
        int s = ne ? v_mis : v_match;
        if (AMBI) s = (qa | t2) > 3 ? -a.ambi : s;
        const int mm = H + s;
        const int f1 = wave_shl1_z(G);
        const int f2 = TWO ? wave_shl1_z(G2) : 0;
        const int f = TWO && f2 > f1 ? f2 : f1;
        const int ht = mm > f ? mm : f;
        const int e1 = wave_shr1(wave_prefix_max_incl_dpp(ht + gel), 0) - goel;
        const int e2 = TWO ? wave_shr1(wave_prefix_max_incl_dpp(ht + gel2), 0) - goel2 : 0;
        const int e = TWO && e2 > e1 ? e2 : e1;
        const int h = ht > e ? ht : e;
        const int fo = h - goe, fe = f1 - ge;
        if (TB) {
            a0 = shift_in(a0, mm == h);
            a1 = shift_in(a1, e >= f);
            a2 = shift_in(a2, e1 + go > h);
            a3 = shift_in(a3, fe > fo);
            a4 = shift_in(a4, ne);
        }
        G = fo > fe ? fo : fe;
        if (TWO) {
            const int fo2 = h - goe2, fe2 = f2 - ge2;
            if (TB) {
                b0 = shift_in(b0, e2 > e1);
                b1 = shift_in(b1, f2 > f1);
                b2 = shift_in(b2, e2 + go2 > h);
                b3 = shift_in(b3, fe2 > fo2);
            }
            G2 = fo2 > fe2 ? fo2 : fe2;
        }
        H = h;
        qa = qa_next; t2 = t2_next;
        if (EXT) {
            const int hb = h + (i == end_row ? a.end_bonus : 0);
            if ((uint32_t)(i - imin) <= (uint32_t)span && span >= 0 && hb > best_h) { best_h = hb; best_i = i; }
        }
        if (TB && (i & 31) == 0) {
            const int c = (i >> 5) - 1;
            pl[WP_DIAG][c][lane] = a0; pl[WP_EGEF][c][lane] = a1; pl[WP_EX1][c][lane] = a2; pl[WP_FX1][c][lane] = a3;
            pl[WP_NE][c][lane] = a4;
            if (TWO) { pl[WP_EP][c][lane] = b0; pl[WP_FP][c][lane] = b1; pl[WP_EX2][c][lane] = b2; pl[WP_FX2][c][lane] = b3; }
        }
    }
    Hend = H;
    if (TB && (m & 31)) {
        const int c = m >> 5, up = 32 - (m & 31);
        pl[WP_DIAG][c][lane] = a0 << up; pl[WP_EGEF][c][lane] = a1 << up; pl[WP_EX1][c][lane] = a2 << up;
        pl[WP_FX1][c][lane] = a3 << up; pl[WP_NE][c][lane] = a4 << up;
        if (TWO) { pl[WP_EP][c][lane] = b0 << up; pl[WP_FP][c][lane] = b1 << up; pl[WP_EX2][c][lane] = b2 << up; pl[WP_FX2][c][lane] = b3 << up; }
    }
}

// Traceback of one 64-diagonal task by one lane (the scheme of narrow_walk with the gap states per piece): runs in
// reverse, the first `cap` kept in `buf`, with `out` written to out[total-1 .. 0] (left extensions in emission order).
template <bool TWO>
__device__ uint32_t wide_walk(const uint32_t *pl, int chunks, int m0, int n0, int dlo, bool keep_order, uint32_t *out, uint32_t total,
                              uint32_t *buf, uint32_t cap, uint32_t *ends = nullptr, int max_it = 4 * (EXT_MAX + SEQ_T_MAX)) {
    auto word = [&](int plane, int c, int lane) { return pl[(plane * chunks + c) * 64 + lane]; };
    int i = m0, j = n0, state = 0;                    // state: 0 H, 1 E1, 2 F1, 3 E2, 4 F2
    uint32_t cur_op = 0, cur_len = 0, n_runs = 0, e_first = 0, e_last = 0;
    auto put = [&]() {
        if (!n_runs) e_first = cur_op;
        e_last = cur_op;
        if (out) out[keep_order ? n_runs : total - 1 - n_runs] = cur_len << 4 | cur_op;
        else if (n_runs < cap) buf[n_runs] = cur_len << 4 | cur_op;
        ++n_runs;
    };
    for (int it = 0; (i > 0 || j > 0) && it < max_it; ++it) {
        uint32_t op, len;
        if (i == 0) {                                          // row 0: H(0,j) is a gap from the corner
            op = OP_D; len = (uint32_t)j; j = 0;
        } else {
            const int d = (j - i - dlo) & (BAND_W - 1);
            const int c = (i - 1) >> 5, sh = 31 - ((i - 1) & 31);
            const int i2 = i > 1 ? i - 2 : 0, c2 = i2 >> 5, sh2 = 31 - (i2 & 31);
            int st = state;
            if (st == 0) {
                const uint32_t dg = word(WP_DIAG, c, d) >> sh;
                if (dg & 1u) st = 0;
                else if ((word(WP_EGEF, c, d) >> sh) & 1u) st = TWO && ((word(WP_EP, c, d) >> sh) & 1u) ? 3 : 1;
                else st = TWO && ((word(WP_FP, c, d) >> sh) & 1u) ? 4 : 2;
                if (st == 0) {
                    const uint32_t ne = word(WP_NE, c, d) >> sh;
                    const uint32_t inv = ~dg;
                    const int r = inv ? __ffs((int)inv) - 1 : 32;                 // diagonal moves in a row (this word)
                    const bool isx = (ne & 1u) != 0;
                    const uint32_t flip = isx ? ~ne : ne;
                    int l = flip ? __ffs((int)flip) - 1 : 32;
                    l = l < r ? l : r;
                    op = isx ? OP_X : OP_EQ; len = (uint32_t)l;
                    i -= l; j -= l;
                    state = 0;
                    if (op == cur_op) cur_len += len;
                    else { if (cur_len) put(); cur_op = op; cur_len = len; }
                    continue;
                }
            }
            if (st == 1 || st == 3) {                          // E_p of this cell extends the E_p of the cell to the left
                const uint32_t ex = word(st == 1 ? WP_EX1 : WP_EX2, c, (d - 1) & 63) >> sh;
                op = OP_D; len = 1;
                state = (ex & 1u) ? st : 0;
                --j;
            } else {                                           // F_p of this cell extends the F_p of the cell above
                const uint32_t fx = word(st == 2 ? WP_FX1 : WP_FX2, c2, (d + 1) & 63) >> sh2;
                op = OP_I; len = 1;
                state = (i > 1 && (fx & 1u)) ? st : 0;
                --i;
            }
        }
        if (op == cur_op) cur_len += len;
        else { if (cur_len) put(); cur_op = op; cur_len = len; }
    }
    if (cur_len) put();
    if (ends) *ends = keep_order ? end_codes(e_first, e_last) : end_codes(e_last, e_first);
    return n_runs;
}

// The same walk over planes in GLOBAL memory (align_long_kernel's scratch area): the nine words an iteration can need are
// requested together - one round trip to the caches per event instead of two or three dependent ones.  Runs in emission
// order into buf (cap entries).
template <bool TWO, int W = BAND_W>
__device__ uint32_t long_walk(const uint32_t *pl, int chunks, int m0, int n0, int dlo, bool keep_order, uint32_t *buf, uint32_t cap,
                              uint32_t *ends, int max_it) {
    auto word = [&](int plane, int c, int lane) { return pl[((size_t)plane * chunks + c) * W + lane]; };
    int i = m0, j = n0, state = 0;                    // state: 0 H, 1 E1, 2 F1, 3 E2, 4 F2
    uint32_t cur_op = 0, cur_len = 0, n_runs = 0, e_first = 0, e_last = 0;
    auto put = [&]() {
        if (!n_runs) e_first = cur_op;
        e_last = cur_op;
        if (n_runs < cap) buf[n_runs] = cur_len << 4 | cur_op;
        ++n_runs;
    };
    for (int it = 0; (i > 0 || j > 0) && it < max_it; ++it) {
        uint32_t op, len;
        if (i == 0) {                                          // row 0: H(0,j) is a gap from the corner
            op = OP_D; len = (uint32_t)j; j = 0;
        } else {
            const int d = (j - i - dlo) & (W - 1);
            const int c = (i - 1) >> 5, sh = 31 - ((i - 1) & 31);
            const int i2 = i > 1 ? i - 2 : 0, c2 = i2 >> 5, sh2 = 31 - (i2 & 31);
            const int dl = (d - 1) & (W - 1), dr = (d + 1) & (W - 1);
            const uint32_t w_dg = word(WP_DIAG, c, d), w_ne = word(WP_NE, c, d), w_eg = word(WP_EGEF, c, d);
            const uint32_t w_ep = TWO ? word(WP_EP, c, d) : 0u, w_fp = TWO ? word(WP_FP, c, d) : 0u;
            const uint32_t w_ex1 = word(WP_EX1, c, dl), w_ex2 = TWO ? word(WP_EX2, c, dl) : 0u;
            const uint32_t w_fx1 = word(WP_FX1, c2, dr), w_fx2 = TWO ? word(WP_FX2, c2, dr) : 0u;
            int st = state;
            if (st == 0) {
                const uint32_t dg = w_dg >> sh;
                if (dg & 1u) st = 0;
                else if ((w_eg >> sh) & 1u) st = TWO && ((w_ep >> sh) & 1u) ? 3 : 1;
                else st = TWO && ((w_fp >> sh) & 1u) ? 4 : 2;
                if (st == 0) {
                    const uint32_t ne = w_ne >> sh;
                    const uint32_t inv = ~dg;
                    const int r = inv ? __ffs((int)inv) - 1 : 32;                 // diagonal moves in a row (this word)
                    const bool isx = (ne & 1u) != 0;
                    const uint32_t flip = isx ? ~ne : ne;
                    int l = flip ? __ffs((int)flip) - 1 : 32;
                    l = l < r ? l : r;
                    op = isx ? OP_X : OP_EQ; len = (uint32_t)l;
                    i -= l; j -= l;
                    state = 0;
                    if (op == cur_op) cur_len += len;
                    else { if (cur_len) put(); cur_op = op; cur_len = len; }
                    continue;
                }
            }
            if (st == 1 || st == 3) {                          // E_p of this cell extends the E_p of the cell to the left
                const uint32_t ex = (st == 1 ? w_ex1 : w_ex2) >> sh;
                op = OP_D; len = 1;
                state = (ex & 1u) ? st : 0;
                --j;
            } else {                                           // F_p of this cell extends the F_p of the cell above
                const uint32_t fx = (st == 2 ? w_fx1 : w_fx2) >> sh2;
                op = OP_I; len = 1;
                state = (i > 1 && (fx & 1u)) ? st : 0;
                --i;
            }
        }
        if (op == cur_op) cur_len += len;
        else { if (cur_len) put(); cur_op = op; cur_len = len; }
    }
    if (cur_len) put();
    if (ends) *ends = keep_order ? end_codes(e_first, e_last) : end_codes(e_last, e_first);
    return n_runs;
}

// TB = false: scores (and, for extensions, the cell they stop in) only: tasks of stub candidates
template <int ROWS_MAX, bool TB = true>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(TB && ROWS_MAX <= 128 ? 4 : 1, 8))) void align_kernel(AlignArgs a) {
    // (the 128-row instance: 4 workgroups per CU need <= 40 KB of LDS and <= 128 registers each - it sat at 41.2 KB and 129)
    constexpr int RUN_BUF = TB && ROWS_MAX <= 128 ? 96 : RUN_BUF_WIDE;
    constexpr int W_CHUNKS = TB ? ROWS_MAX / 32 : 1;
    __shared__ uint32_t s_pl[WAVES][TB ? N_WPLANES : 1][W_CHUNKS][64];
    __shared__ __attribute__((aligned(4))) uint8_t s_q[WAVES][ROWS_MAX + 4];
    __shared__ __attribute__((aligned(4))) uint8_t s_t[WAVES][WT_LEN];
    __shared__ uint32_t s_runs[WAVES][RUN_BUF];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint32_t (*pl)[W_CHUNKS][64] = s_pl[wv];
    uint8_t *sq = s_q[wv], *st = s_t[wv];
    uint32_t chunk_off = 0, chunk_left = 0;
    for (size_t li = wave; li < a.n_list; li += n_waves) {
        const size_t ti = a.list[li];
        const Task tk = a.tasks[ti];
        const int n = tk.n, dlo = tk.dlo;
        // rows i > n - dlo have no cell inside the band (j = i + dlo + lane > n): an extension whose query
        // side is longer than the target side + half band never looks at them
        const int m = tk.m < n - dlo ? tk.m : n - dlo;
        if (m <= 0 || n <= 0) {
            if (lane == 0) a.out[ti] = TaskOut{0, 0, 0, 0, 0, 0};
            continue;
        }
        const int kind = tk.kind & 3;
        bool ambig = false;
        {   // stage the two windows (DP order), 4 bases per lane and step
            const bool rev = (tk.kind & TASK_REV) != 0, left = kind == 1;
            uint32_t vq = 0, vt[2] = {0, 0};
            if (4 * lane < m) vq = load_window4p(a.qcodes, (long long)tk.qa, left != rev, rev, 4 * lane);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (4 * (lane + 64 * r) < n) vt[r] = load_window4p(a.tcodes, (long long)tk.ta, left, false, 4 * (lane + 64 * r));
            if (4 * lane < m) { *(uint32_t *)(sq + 4 * lane) = vq; ambig |= (vq & 0x04040404u) != 0; }
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (4 * (lane + 64 * r) < n) {
                    *(uint32_t *)(st + WT_PAD + 4 * (lane + 64 * r)) = vt[r];
                    ambig |= (vt[r] & 0x04040404u) != 0;
                }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const int end_row = kind != 0 ? (int)(tk.narrow >> 1) - 1 : -1;
        int Hend = 0, best_h = 0, best_i = 0;
        const bool amb = __any(ambig);
        const bool two = a.go2 > 0 && !(tk.kind & TASK_ONE);
        auto rows_of = [&](auto AMB, auto EXTN, auto TW) {
            wide_rows<decltype(AMB)::value, decltype(EXTN)::value, decltype(TW)::value, W_CHUNKS, TB>(a, m, n, dlo, lane, end_row, sq, st, pl, Hend,
                                                                                                    best_h, best_i);
        };
        using T_ = std::true_type; using F_ = std::false_type;
        if (kind == 0) {
            if (two) { if (amb) rows_of(T_{}, F_{}, T_{}); else rows_of(F_{}, F_{}, T_{}); }
            else { if (amb) rows_of(T_{}, F_{}, F_{}); else rows_of(F_{}, F_{}, F_{}); }
        } else {
            if (two) { if (amb) rows_of(T_{}, T_{}, T_{}); else rows_of(F_{}, T_{}, T_{}); }
            else { if (amb) rows_of(T_{}, T_{}, F_{}); else rows_of(F_{}, T_{}, F_{}); }
        }
        int ei, ej, score;
        if (kind == 0) {
            ei = m; ej = n;
            score = __shfl(Hend, n - m - dlo, 64) - DP_BIAS;
        } else {
            // best cell: score (with the bonus), then fewest bases i + j, then fewest rows
            unsigned long long best = 0;
            if (best_h > DP_BIAS / 2)
                best = (unsigned long long)(uint32_t)(best_h - DP_BIAS + (1 << 20)) << 32 |
                       (unsigned long long)(0xffffu - (uint32_t)(2 * best_i + dlo + lane)) << 16 |
                       (unsigned long long)(0xffffu - (uint32_t)best_i);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long u = __shfl_xor(best, o, 64);
                best = u > best ? u : best;
            }
            ei = (int)(0xffffu - (uint32_t)(best & 0xffff));
            ej = (int)(0xffffu - (uint32_t)((best >> 16) & 0xffff)) - ei;
            score = (int)(uint32_t)(best >> 32) - (1 << 20) - (ei == end_row ? a.end_bonus : 0);   // the bonus only ranks
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if constexpr (!TB) {
            if (lane == 0) a.out[ti] = TaskOut{score, ei, ej, 0, 0, kind != 0 && ei == end_row ? 0x80000000u : 0u};
            continue;
        }
        uint32_t *rbuf = s_runs[wv];
        const uint32_t rcap = a.run_buf_cap < (uint32_t)RUN_BUF ? a.run_buf_cap : (uint32_t)RUN_BUF;
        uint32_t cp_n = 0, cp_off = 0;
        if (lane == 0) {
            const uint32_t *planes = &pl[0][0][0];
            uint32_t ends = 0;
            const uint32_t n_runs = two ? wide_walk<true>(planes, W_CHUNKS, ei, ej, dlo, kind == 1, nullptr, 0, rbuf, rcap, &ends)
                                        : wide_walk<false>(planes, W_CHUNKS, ei, ej, dlo, kind == 1, nullptr, 0, rbuf, rcap, &ends);
            uint32_t off = 0;
            bool ok = true;
            if (n_runs) off = pool_take(a, n_runs, chunk_off, chunk_left, ok);
            if (ok && n_runs) {
                if (n_runs <= rcap) { cp_n = n_runs; cp_off = off; }
                else if (two) wide_walk<true>(planes, W_CHUNKS, ei, ej, dlo, kind == 1, a.runs + off, n_runs, nullptr, 0);
                else wide_walk<false>(planes, W_CHUNKS, ei, ej, dlo, kind == 1, a.runs + off, n_runs, nullptr, 0);
            }
            a.out[ti] = TaskOut{score, ei, ej, off, ok ? n_runs : 0, ends | (kind != 0 && ei == end_row ? 0x80000000u : 0u)};
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        cp_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cp_n); cp_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)cp_off);
        for (uint32_t k = (uint32_t)lane; k < cp_n; k += 64)     // left extensions keep the emission order
            a.runs[cp_off + k] = rbuf[kind == 1 ? k : cp_n - 1 - k];
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- pass 2c: LONG tasks - blocks of more than BLOCK_MAX rows or columns, extensions of more than EXT_MAX rows ----------------
// What `minimap2 -c -g10000` buys (script/filter_overlap_slr2.py:51): the gap between two chained anchors is filled whatever
// its length, and a chain end is extended until a z-drop.  Same recurrences, band (64 diagonals = the wave), tie rules and
// best-cell rule as align_kernel; what differs is where things live:
//   * the band walks the task in TILES of LONG_TILE rows: the two sequence windows of a tile are staged in LDS (a few
//     hundred bytes per wave), H and the gap states of the band's last row stay in registers across tiles;
//   * the nine traceback planes go to a scratch area of the wave in global memory, one 256-byte line per plane and 32 rows
//     (9 x 8 bytes per row: a 10 000-row task writes 0.7 MB, re-read by its own walk while still in the caches), the walk's
//     runs to a second scratch area, from where the whole wave copies them into the run pool;
//   * an extension stops at a z-drop: after every ZDROP_STEP-th row the row's best in-rectangle cell is held against the best
//     cell so far (oracle/ava_oracle.c:band_dp);
//   * tasks are handed out through an atomic counter (their lengths span 256 .. 10 000 rows).
// TB = false: scores (and the cell an extension stops in) only - tasks of stub candidates.
constexpr int LONG_TILE = 256;
constexpr int LT_LEN = LONG_TILE + BAND_W + 8;        // staged target elements: index (row in tile - 1) + lane, + 1 look-ahead
struct LongArgs {
    uint32_t *planes;       // per wave: N_WPLANES x chunks_cap x 64 words
    uint32_t *run_scratch;  // per wave: runs_cap words
    uint32_t chunks_cap;    // 32-row words per plane and lane one wave's area holds
    uint32_t runs_cap;
    int zdrop;              // 0: none
    uint32_t *next;         // work counter (zeroed before the launch)
    uint32_t *too_long;     // set when a task does not fit the scratch areas (the host sizes them from the options: never)
    unsigned long long *rows_run;   // sum of the rows the tasks really ran (statistics: a z-drop ends an extension early)
};

template <bool TWO, bool TB>
__global__ __launch_bounds__(WG) void align_long_kernel(AlignArgs a, LongArgs la) {
    __shared__ __attribute__((aligned(4))) uint8_t s_q[WAVES][LONG_TILE + 8];
    __shared__ __attribute__((aligned(4))) uint8_t s_t[WAVES][LT_LEN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    uint32_t *const pl = TB ? la.planes + wave * ((size_t)N_WPLANES * la.chunks_cap * 64) : nullptr;
    uint32_t *const rs = TB ? la.run_scratch + wave * (size_t)la.runs_cap : nullptr;
    uint8_t *sq = s_q[wv], *st = s_t[wv];
    uint32_t chunk_off = 0, chunk_left = 0;
    const int go = a.go, ge = a.ge, goe = go + ge, gel = ge * lane, goel = go + ge * lane;
    const int go2 = a.go2, ge2 = a.ge2, goe2 = go2 + ge2, gel2 = ge2 * lane, goel2 = go2 + ge2 * lane;
    // The few LONG tasks of a batch are its critical path (thousands of dependent rows each) and run beside kernels that fill every
    // SIMD: their waves take the issue slots first
    __builtin_amdgcn_s_setprio(3);
    for (;;) {
        uint32_t li = 0;
        if (lane == 0) li = atomicAdd(la.next, 1u);
        li = (uint32_t)__builtin_amdgcn_readfirstlane((int)li);
        if (li >= a.n_list) break;                                // (every wave gets here: the counter only grows)
        const size_t ti = a.list[li];
        const Task tk = a.tasks[ti];
        const int n = tk.n, dlo = tk.dlo;
        const int m = tk.m < n - dlo ? tk.m : n - dlo;            // rows below have no cell inside the band
        const int kind = tk.kind & 3;
        const int chunks = (m + 31) >> 5;
        if (m <= 0 || n <= 0 || (TB && ((uint32_t)chunks > la.chunks_cap || (uint32_t)(m + n + 2) > la.runs_cap))) {
            if (lane == 0) {
                a.out[ti] = TaskOut{0, 0, 0, 0, 0, 0};
                if (m > 0 && n > 0) *la.too_long = 1;
            }
            continue;
        }
        const bool ext = kind != 0, rev = (tk.kind & TASK_REV) != 0, left = kind == 1;
        const int end_row = ext ? (int)(tk.narrow >> 1) - 1 : -1;
        const int j0 = dlo + lane;
        int gap0 = go + ge * j0;
        if (TWO && go2 + ge2 * j0 < gap0) gap0 = go2 + ge2 * j0;
        int H = j0 == 0 ? DP_BIAS : (j0 > 0 ? DP_BIAS - gap0 : 0);          // row 0 (biased scores: narrow_rows)
        int G = H - goe, G2 = TWO ? H - goe2 : 0;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        int best_h = 0, best_i = 0;                                          // biased like H: 0 = no cell yet
        if (ext && j0 >= 0 && j0 <= n) best_h = H;
        int rows_done = m;                                                   // a z-drop ends the task earlier
        bool stop = false;
        for (int r0 = 0; r0 < m && !stop; r0 += LONG_TILE) {
            const int R = m - r0 < LONG_TILE ? m - r0 : LONG_TILE;
            {   // stage the tile: query elements r0 .. r0 + R - 1, target elements eb .. eb + R + 63 (eb = r0 + dlo: the base
                // of row r0 + 1 in lane 0); elements outside the windows read as the ambiguous code (their cells lie outside
                // the rectangle)
                __builtin_amdgcn_wave_barrier();
                const int xq = r0 + 4 * lane;
                uint32_t vq = 0x04040404u;
                if (4 * lane < R) vq = load_window4p(a.qcodes, (long long)tk.qa, left != rev, rev, xq);
                const int eb = r0 + dlo;
                uint32_t vt[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = 4 * (lane + 64 * r), x = eb + k;
                    vt[r] = 0x04040404u;
                    if (k < R + 64 && x + 3 >= 0 && x < n) vt[r] = load_window4p(a.tcodes, (long long)tk.ta, left, false, x);
                }
                *(uint32_t *)(sq + 4 * lane) = vq;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = 4 * (lane + 64 * r);
                    if (k < LT_LEN - 3) *(uint32_t *)(st + k) = vt[r];
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
            }
            const uint8_t *tp = st + lane;                                   // target base of row r0 + ii: tp[ii - 1]
            int qa = sq[0], t2 = tp[0];
            for (int ii = 1; ii <= R; ++ii) {
                const int i = r0 + ii;
                const int qa_next = sq[ii], t2_next = tp[ii];
                const bool ne = qa != t2;
                int s = ne ? -a.mismatch : a.match;
                s = (qa | t2) > 3 ? -a.ambi : s;
                const int mm = H + s;
                const int f1 = wave_shl1_z(G);
                const int f2 = TWO ? wave_shl1_z(G2) : 0;
                const int f = TWO && f2 > f1 ? f2 : f1;
                const int ht = mm > f ? mm : f;
                const int e1 = wave_shr1(wave_prefix_max_incl_dpp(ht + gel), 0) - goel;
                const int e2 = TWO ? wave_shr1(wave_prefix_max_incl_dpp(ht + gel2), 0) - goel2 : 0;
                const int e = TWO && e2 > e1 ? e2 : e1;
                const int h = ht > e ? ht : e;
                const int fo = h - goe, fe = f1 - ge;
                if (TB) {
                    a0 = shift_in(a0, mm == h);
                    a1 = shift_in(a1, e >= f);
                    a2 = shift_in(a2, e1 + go > h);
                    a3 = shift_in(a3, fe > fo);
                    a4 = shift_in(a4, ne);
                }
                G = fo > fe ? fo : fe;
                if (TWO) {
                    const int fo2 = h - goe2, fe2 = f2 - ge2;
                    if (TB) {
                        b0 = shift_in(b0, e2 > e1);
                        b1 = shift_in(b1, f2 > f1);
                        b2 = shift_in(b2, e2 + go2 > h);
                        b3 = shift_in(b3, fe2 > fo2);
                    }
                    G2 = fo2 > fe2 ? fo2 : fe2;
                }
                H = h;
                qa = qa_next; t2 = t2_next;
                const bool inside = (uint32_t)(i + j0) <= (uint32_t)n;       // 0 <= j <= n
                if (ext) {
                    const int hb = h + (i == end_row ? a.end_bonus : 0);
                    if (inside && hb > best_h) { best_h = hb; best_i = i; }
                }
                if ((i & 31) == 0) {
                    if (TB) {
                        const size_t c = (size_t)(i >> 5) - 1;
                        pl[((size_t)WP_DIAG * chunks + c) * 64 + lane] = a0; pl[((size_t)WP_EGEF * chunks + c) * 64 + lane] = a1;
                        pl[((size_t)WP_EX1 * chunks + c) * 64 + lane] = a2; pl[((size_t)WP_FX1 * chunks + c) * 64 + lane] = a3;
                        pl[((size_t)WP_NE * chunks + c) * 64 + lane] = a4;
                        if (TWO) {
                            pl[((size_t)WP_EP * chunks + c) * 64 + lane] = b0; pl[((size_t)WP_FP * chunks + c) * 64 + lane] = b1;
                            pl[((size_t)WP_EX2 * chunks + c) * 64 + lane] = b2; pl[((size_t)WP_FX2 * chunks + c) * 64 + lane] = b3;
                        }
                    }
                    if (ext && la.zdrop > 0) {                               // z-drop (ZDROP_STEP = 32 = the planes' word)
                        const int row_max = (int)wave_max_u32_dpp(inside ? (uint32_t)h : 0u);
                        const int so_far = (int)wave_max_u32_dpp((uint32_t)best_h);
                        if (row_max == 0 || so_far - row_max > la.zdrop) { rows_done = i; stop = true; break; }
                    }
                }
            }
        }
        if (TB && (rows_done & 31)) {                                        // partial word: its first row up to bit 31
            const size_t c = (size_t)(rows_done >> 5);
            const int up = 32 - (rows_done & 31);
            pl[((size_t)WP_DIAG * chunks + c) * 64 + lane] = a0 << up; pl[((size_t)WP_EGEF * chunks + c) * 64 + lane] = a1 << up;
            pl[((size_t)WP_EX1 * chunks + c) * 64 + lane] = a2 << up; pl[((size_t)WP_FX1 * chunks + c) * 64 + lane] = a3 << up;
            pl[((size_t)WP_NE * chunks + c) * 64 + lane] = a4 << up;
            if (TWO) {
                pl[((size_t)WP_EP * chunks + c) * 64 + lane] = b0 << up; pl[((size_t)WP_FP * chunks + c) * 64 + lane] = b1 << up;
                pl[((size_t)WP_EX2 * chunks + c) * 64 + lane] = b2 << up; pl[((size_t)WP_FX2 * chunks + c) * 64 + lane] = b3 << up;
            }
        }
        int ei, ej, score;
        if (!ext) {
            ei = m; ej = n;
            score = __shfl(H, n - m - dlo, 64) - DP_BIAS;
        } else {
            // best cell: score (with the bonus), then fewest bases i + j, then fewest rows (align_kernel)
            unsigned long long best = 0;
            if (best_h > DP_BIAS / 2)
                best = (unsigned long long)(uint32_t)(best_h - DP_BIAS + (1 << 20)) << 32 |
                       (unsigned long long)(0xffffu - (uint32_t)(2 * best_i + dlo + lane)) << 16 |
                       (unsigned long long)(0xffffu - (uint32_t)best_i);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long u = __shfl_xor(best, o, 64);
                best = u > best ? u : best;
            }
            ei = (int)(0xffffu - (uint32_t)(best & 0xffff));
            ej = (int)(0xffffu - (uint32_t)((best >> 16) & 0xffff)) - ei;
            score = (int)(uint32_t)(best >> 32) - (1 << 20) - (ei == end_row ? a.end_bonus : 0);   // the bonus only ranks
        }
        const uint32_t flag = ext && ei == end_row ? 0x80000000u : 0u;
        if (lane == 0) atomicAdd(la.rows_run, (unsigned long long)rows_done);
        if constexpr (!TB) {
            if (lane == 0) a.out[ti] = TaskOut{score, ei, ej, 0, 0, flag};
            continue;
        } else {
            __threadfence_block();                                           // the planes of all lanes, before lane 0 reads them
            __builtin_amdgcn_wave_barrier();
            uint32_t n_runs = 0, off = 0, ends = 0;
            bool ok = true;
            if (lane == 0) {
                n_runs = long_walk<TWO>(pl, chunks, ei, ej, dlo, left, rs, la.runs_cap, &ends, 2 * (m + n) + 8);
                if (n_runs) off = pool_take(a, n_runs, chunk_off, chunk_left, ok);
                a.out[ti] = TaskOut{score, ei, ej, off, ok ? n_runs : 0, ends | flag};
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            n_runs = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_runs);
            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
            ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
            if (ok) for (uint32_t k = (uint32_t)lane; k < n_runs; k += 64)      // left extensions keep the emission order
                a.runs[off + k] = rs[left ? k : n_runs - 1 - k];
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- pass 2c': LONG tasks in the 32-diagonal band, TWO tasks per wave ------------------------------------------------------
// LONG extensions (band -15 .. +16) and LONG blocks whose diagonal shift is at most HALF_DELTA (32 diagonals centred on the
// corners): lanes 0-31 run one task, lanes 32-63 another - the scheme of align_long_kernel with every cross-lane step
// confined to its half (the prefix maximum stops at the half's last lane, the one-lane shifts take no value across the
// middle).  The list is sorted by rows, so the two tasks of a wave are about equally long; the shorter one idles its
// half for the difference.  Per-half scratch: planes [plane][32-row word][32 lanes], runs.
constexpr int LT32_LEN = LONG_TILE + HALF_W + 8;
__device__ __forceinline__ int half_prefix_max_incl_dpp(int x) {       // inclusive prefix max inside each half of the wave
    constexpr int ident = DPP_MAX_IDENT;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    x = mx(x, dpp_i32<0x111>(ident, x));
    x = mx(x, dpp_i32<0x112>(ident, x));
    x = mx(x, dpp_i32<0x114>(ident, x));
    x = mx(x, dpp_i32<0x118>(ident, x));
    x = mx(x, dpp_i32<0x142, 0xa>(ident, x));                          // lane 15 -> row 1, lane 47 -> row 3
    return x;
}
__device__ __forceinline__ uint32_t half_max_u32(uint32_t v, int hf) {  // max over the lanes of the own half, to every lane of it
    int x = (int)v;
    auto mx = [](int a, int b) { return (int)((uint32_t)a > (uint32_t)b ? (uint32_t)a : (uint32_t)b); };
    x = mx(x, dpp_i32<0x111>(0, x));
    x = mx(x, dpp_i32<0x112>(0, x));
    x = mx(x, dpp_i32<0x114>(0, x));
    x = mx(x, dpp_i32<0x118>(0, x));
    x = mx(x, dpp_i32<0x142, 0xa>(0, x));
    return (uint32_t)__shfl(x, 32 * hf + 31, 64);
}

template <bool TWO, bool TB>
__global__ __launch_bounds__(WG) void align_long32_kernel(AlignArgs a, LongArgs la) {
    __shared__ __attribute__((aligned(4))) uint8_t s_q[WAVES][2][LONG_TILE + 8];
    __shared__ __attribute__((aligned(4))) uint8_t s_t[WAVES][2][LT32_LEN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hf = lane >> 5, hl = lane & 31;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    // (a wave's plane area has room for 64 lanes: each half takes 32 of them)
    uint32_t *const pl = TB ? la.planes + wave * ((size_t)N_WPLANES * la.chunks_cap * 64) + (size_t)hf * ((size_t)N_WPLANES * la.chunks_cap * 32) : nullptr;
    uint32_t *const rs = TB ? la.run_scratch + (wave * 2 + (size_t)hf) * (size_t)la.runs_cap : nullptr;
    uint8_t *sq = s_q[wv][hf], *st = s_t[wv][hf];
    uint32_t chunk_off = 0, chunk_left = 0;
    const int go = a.go, ge = a.ge, goe = go + ge, gel = ge * hl, goel = go + ge * hl;
    const int go2 = a.go2, ge2 = a.ge2, goe2 = go2 + ge2, gel2 = ge2 * hl, goel2 = go2 + ge2 * hl;
    __builtin_amdgcn_s_setprio(3);                                 // (see align_long_kernel)
    for (;;) {
        uint32_t pi = 0;
        if (lane == 0) pi = atomicAdd(la.next, 1u);
        pi = (uint32_t)__builtin_amdgcn_readfirstlane((int)pi);
        if (2 * (size_t)pi >= a.n_list) break;                    // (every wave gets here: the counter only grows)
        const size_t li = 2 * (size_t)pi + (size_t)hf;
        bool live = li < a.n_list;
        size_t ti = 0;
        Task tk{};
        if (live) { ti = a.list[li]; tk = a.tasks[ti]; }
        const int n = tk.n, dlo = tk.dlo;
        int m = live ? (tk.m < n - dlo ? (int)tk.m : n - dlo) : 0;   // rows this half runs (a z-drop lowers it)
        if (m < 0) m = 0;
        const int kind = tk.kind & 3;
        const int chunks = (m + 31) >> 5;
        if (live && (m <= 0 || n <= 0 || (TB && ((uint32_t)chunks > la.chunks_cap || (uint32_t)(m + n + 2) > la.runs_cap)))) {
            if (hl == 0) {
                a.out[ti] = TaskOut{0, 0, 0, 0, 0, 0};
                if (m > 0 && n > 0) *la.too_long = 1;
            }
            live = false;
            m = 0;
        }
        const int m0 = m;
        const bool ext = kind != 0, rev = (tk.kind & TASK_REV) != 0, left = kind == 1;
        const int end_row = ext ? (int)(tk.narrow >> 1) - 1 : -1;
        const int j0 = dlo + hl;
        int gap0 = go + ge * j0;
        if (TWO && go2 + ge2 * j0 < gap0) gap0 = go2 + ge2 * j0;
        int H = j0 == 0 ? DP_BIAS : (j0 > 0 ? DP_BIAS - gap0 : 0);          // row 0 (biased scores: narrow_rows)
        int G = H - goe, G2 = TWO ? H - goe2 : 0, Hend = 0;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        int best_h = 0, best_i = 0;                                          // biased like H: 0 = no cell yet
        if (live && ext && j0 >= 0 && j0 <= n) best_h = H;
        for (int r0 = 0;; r0 += LONG_TILE) {
            // rows the longer of the two tasks still has (a z-drop may have shortened either)
            const int mA = __builtin_amdgcn_readlane(m, 0), mB = __builtin_amdgcn_readlane(m, 32);
            const int rows_max = mA > mB ? mA : mB;
            if (r0 >= rows_max) break;
            const int R_all = rows_max - r0 < LONG_TILE ? rows_max - r0 : LONG_TILE;
            {   // stage the tile of the own task (align_long_kernel): 32 lanes x 4 bases per round
                __builtin_amdgcn_wave_barrier();
                const int R = m - r0 < 0 ? 0 : (m - r0 < LONG_TILE ? m - r0 : LONG_TILE);
                uint32_t vq[2], vt[3];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int k = 4 * (hl + 32 * r);
                    vq[r] = 0x04040404u;
                    if (k < R) vq[r] = load_window4p(a.qcodes, (long long)tk.qa, left != rev, rev, r0 + k);
                }
                const int eb = r0 + dlo;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int k = 4 * (hl + 32 * r), x = eb + k;
                    vt[r] = 0x04040404u;
                    if (R > 0 && k < R + HALF_W && x + 3 >= 0 && x < n) vt[r] = load_window4p(a.tcodes, (long long)tk.ta, left, false, x);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) *(uint32_t *)(sq + 4 * (hl + 32 * r)) = vq[r];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int k = 4 * (hl + 32 * r);
                    if (k < LT32_LEN - 3) *(uint32_t *)(st + k) = vt[r];
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
            }
            const uint8_t *tp = st + hl;                                     // target base of row r0 + ii: tp[ii - 1]
            int qa = sq[0], t2 = tp[0];
            for (int ii = 1; ii <= R_all; ++ii) {
                const int i = r0 + ii;
                const int qa_next = sq[ii], t2_next = tp[ii];
                const bool ne = qa != t2;
                int s = ne ? -a.mismatch : a.match;
                s = (qa | t2) > 3 ? -a.ambi : s;
                const int mm = H + s;
                int f1 = wave_shl1_z(G);
                f1 = hl == 31 ? 0 : f1;                                      // (nothing from the other half's lane 0)
                int f2 = 0;
                if (TWO) { f2 = wave_shl1_z(G2); f2 = hl == 31 ? 0 : f2; }
                const int f = TWO && f2 > f1 ? f2 : f1;
                const int ht = mm > f ? mm : f;
                int p1 = wave_shr1(half_prefix_max_incl_dpp(ht + gel), 0);
                p1 = hl == 0 ? 0 : p1;
                const int e1 = p1 - goel;
                int e2 = 0;
                if (TWO) {
                    int p2 = wave_shr1(half_prefix_max_incl_dpp(ht + gel2), 0);
                    p2 = hl == 0 ? 0 : p2;
                    e2 = p2 - goel2;
                }
                const int e = TWO && e2 > e1 ? e2 : e1;
                const int h = ht > e ? ht : e;
                const int fo = h - goe, fe = f1 - ge;
                if (TB) {
                    a0 = shift_in(a0, mm == h);
                    a1 = shift_in(a1, e >= f);
                    a2 = shift_in(a2, e1 + go > h);
                    a3 = shift_in(a3, fe > fo);
                    a4 = shift_in(a4, ne);
                }
                G = fo > fe ? fo : fe;
                if (TWO) {
                    const int fo2 = h - goe2, fe2 = f2 - ge2;
                    if (TB) {
                        b0 = shift_in(b0, e2 > e1);
                        b1 = shift_in(b1, f2 > f1);
                        b2 = shift_in(b2, e2 + go2 > h);
                        b3 = shift_in(b3, fe2 > fo2);
                    }
                    G2 = fo2 > fe2 ? fo2 : fe2;
                }
                H = h;
                qa = qa_next; t2 = t2_next;
                const bool run = i <= m;                                     // this half's task still has this row
                const bool inside = run && (uint32_t)(i + j0) <= (uint32_t)n;   // 0 <= j <= n
                if (i == m0) Hend = h;
                if (ext) {
                    const int hb = h + (i == end_row ? a.end_bonus : 0);
                    if (inside && hb > best_h) { best_h = hb; best_i = i; }
                }
                if ((i & 31) == 0) {
                    if (TB && run) {
                        const size_t c = (size_t)(i >> 5) - 1;
                        pl[((size_t)WP_DIAG * chunks + c) * 32 + hl] = a0; pl[((size_t)WP_EGEF * chunks + c) * 32 + hl] = a1;
                        pl[((size_t)WP_EX1 * chunks + c) * 32 + hl] = a2; pl[((size_t)WP_FX1 * chunks + c) * 32 + hl] = a3;
                        pl[((size_t)WP_NE * chunks + c) * 32 + hl] = a4;
                        if (TWO) {
                            pl[((size_t)WP_EP * chunks + c) * 32 + hl] = b0; pl[((size_t)WP_FP * chunks + c) * 32 + hl] = b1;
                            pl[((size_t)WP_EX2 * chunks + c) * 32 + hl] = b2; pl[((size_t)WP_FX2 * chunks + c) * 32 + hl] = b3;
                        }
                    }
                    if (la.zdrop > 0) {                                      // z-drop of an extension (uniform per half)
                        const int row_max = (int)half_max_u32(inside ? (uint32_t)h : 0u, hf);
                        const int so_far = (int)half_max_u32((uint32_t)best_h, hf);
                        if (ext && run && i < m && (row_max == 0 || so_far - row_max > la.zdrop)) m = i;
                    }
                } else if (TB && i == m) {
                    // the task's last row is not a word boundary: its partial word now, first row up to bit 31 (the accumulators
                    // go on shifting while the other half's task runs)
                    const size_t c = (size_t)(i >> 5);
                    const int up = 32 - (i & 31);
                    pl[((size_t)WP_DIAG * chunks + c) * 32 + hl] = a0 << up; pl[((size_t)WP_EGEF * chunks + c) * 32 + hl] = a1 << up;
                    pl[((size_t)WP_EX1 * chunks + c) * 32 + hl] = a2 << up; pl[((size_t)WP_FX1 * chunks + c) * 32 + hl] = a3 << up;
                    pl[((size_t)WP_NE * chunks + c) * 32 + hl] = a4 << up;
                    if (TWO) {
                        pl[((size_t)WP_EP * chunks + c) * 32 + hl] = b0 << up; pl[((size_t)WP_FP * chunks + c) * 32 + hl] = b1 << up;
                        pl[((size_t)WP_EX2 * chunks + c) * 32 + hl] = b2 << up; pl[((size_t)WP_FX2 * chunks + c) * 32 + hl] = b3 << up;
                    }
                }
            }
        }
        int ei, ej, score;
        if (!ext) {
            ei = m0; ej = n;
            score = __shfl(Hend, 32 * hf + ((n - m0 - dlo) & (HALF_W - 1)), 64) - DP_BIAS;
        } else {
            // best cell: score (with the bonus), then fewest bases i + j, then fewest rows (align_kernel) - over the own half
            unsigned long long best = 0;
            if (best_h > DP_BIAS / 2)
                best = (unsigned long long)(uint32_t)(best_h - DP_BIAS + (1 << 20)) << 32 |
                       (unsigned long long)(0xffffu - (uint32_t)(2 * best_i + dlo + hl)) << 16 |
                       (unsigned long long)(0xffffu - (uint32_t)best_i);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const unsigned long long u = __shfl_xor(best, o, 64);
                best = u > best ? u : best;
            }
            ei = (int)(0xffffu - (uint32_t)(best & 0xffff));
            ej = (int)(0xffffu - (uint32_t)((best >> 16) & 0xffff)) - ei;
            score = (int)(uint32_t)(best >> 32) - (1 << 20) - (ei == end_row ? a.end_bonus : 0);   // the bonus only ranks
        }
        const uint32_t flag = ext && ei == end_row ? 0x80000000u : 0u;
        if (live && hl == 0) atomicAdd(la.rows_run, (unsigned long long)m);
        if constexpr (!TB) {
            if (live && hl == 0) a.out[ti] = TaskOut{score, ei, ej, 0, 0, flag};
            continue;
        } else {
            __threadfence_block();                                           // the planes of all lanes, before the walkers read them
            __builtin_amdgcn_wave_barrier();
            uint32_t n_runs = 0, off = 0, ends = 0;
            bool ok = true;
            if (live && hl == 0) {
                n_runs = long_walk<TWO, HALF_W>(pl, chunks, ei, ej, dlo, left, rs, la.runs_cap, &ends, 2 * (m0 + n) + 8);
                if (n_runs) off = pool_take(a, n_runs, chunk_off, chunk_left, ok);
                a.out[ti] = TaskOut{score, ei, ej, off, ok ? n_runs : 0, ends | flag};
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            n_runs = (uint32_t)__shfl((int)n_runs, 32 * hf, 64);
            off = (uint32_t)__shfl((int)off, 32 * hf, 64);
            ok = __shfl((int)ok, 32 * hf, 64) != 0;
            if (ok) for (uint32_t k = (uint32_t)hl; k < n_runs; k += 32)       // left extensions keep the emission order
                a.runs[off + k] = rs[left ? k : n_runs - 1 - k];
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- pass 2d: ungapped tasks ---------------------------------------------------------------------------------------------
// bandwidth 0 = `minimap2 -r 0` of the contig-vs-contig call (script/HyLight.py:309): the DP band is the diagonal alone
// (oracle/ava_oracle.c: W = 1).  A block (square: the chain's links have no diagonal shift) is its diagonal, an extension the
// best prefix of its diagonal (score + end bonus on the row that reaches the query end, first row on ties, z-drop test every
// ZDROP_STEP rows).  One lane per task, two passes over the bases (count the runs, then write them); the calls that take this
// form have 10^3 .. 10^5 rows.
template <bool TB>
__global__ __launch_bounds__(WG) void align_ungapped_kernel(AlignArgs a, int zdrop) {
    const int lane = threadIdx.x & 63;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, n_thr = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (a.n_list + n_thr - 1) / n_thr;             // uniform trip count: the allocation is per wave
    uint32_t chunk_off = 0, chunk_left = 0;
    for (size_t r = 0; r < rounds; ++r) {
        const size_t u = r * n_thr + tid;
        const bool live = u < a.n_list;
        size_t ti = 0;
        Task tk{};
        if (live) { ti = a.list[u]; tk = a.tasks[ti]; }
        const int kind = tk.kind & 3;
        const bool ext = kind != 0, rev = (tk.kind & TASK_REV) != 0, left = kind == 1;
        const int L = live ? (tk.m < tk.n ? tk.m : tk.n) : 0;
        const int end_row = ext ? (int)(tk.narrow >> 1) - 1 : -1;
        auto pair_at = [&](int x, uint64_t &q8, uint64_t &t8) {
            q8 = load_window8p(a.qcodes, (long long)tk.qa, left != rev, rev, x);
            t8 = load_window8p(a.tcodes, (long long)tk.ta, left, false, x);
        };
        int score = 0, best = 0, best_i = 0, best_s = 0, cur = -1;
        uint32_t nr = 0, nr_best = 0, first_op = 0, last_op_best = 0;
        bool stop = false;
        for (int x = 0; x < L && !stop; x += 8) {
            uint64_t q8, t8;
            pair_at(x, q8, t8);
            const int lim = L - x < 8 ? L - x : 8;
            for (int y = 0; y < lim; ++y) {
                const int qc = (int)(q8 >> (8 * y)) & 0xff, tc = (int)(t8 >> (8 * y)) & 0xff, i = x + y + 1;
                score += (qc | tc) > 3 ? -a.ambi : (qc == tc ? a.match : -a.mismatch);
                const int op = qc == tc ? (int)OP_EQ : (int)OP_X;
                if (op != cur) { if (!nr) first_op = (uint32_t)op; ++nr; cur = op; }
                if (ext) {
                    const int rank = score + (i == end_row ? a.end_bonus : 0);
                    if (rank > best) { best = rank; best_i = i; best_s = score; nr_best = nr; last_op_best = (uint32_t)op; }
                    if (zdrop > 0 && (i & (ZDROP_STEP - 1)) == 0 && best - score > zdrop) { stop = true; break; }
                }
            }
        }
        const int take = ext ? best_i : L;
        const int out_score = ext ? best_s : score;
        uint32_t n_runs = ext ? nr_best : nr, last_op = ext ? last_op_best : (uint32_t)cur;
        if (!TB || take == 0) n_runs = 0;
        uint32_t wave_total;
        const uint32_t mine = wave_excl_sum_u32(n_runs, lane, wave_total);
        uint32_t base = 0;
        bool ok = true;
        if (wave_total) {
            if (lane == 0) base = pool_take(a, wave_total, chunk_off, chunk_left, ok);
            base = (uint32_t)__shfl((int)base, 0, 64);
            ok = __shfl((int)ok, 0, 64) != 0;
        }
        if (live && ok && n_runs) {                                   // second pass: the runs of [0, take), in sequence order
            uint32_t k = 0, len = 0;
            int op_cur = -1;
            uint32_t *dst = a.runs + base + mine;
            auto put = [&]() { dst[left ? n_runs - 1 - k : k] = len << 4 | (uint32_t)op_cur; ++k; };
            for (int x = 0; x < take; x += 8) {
                uint64_t q8, t8;
                pair_at(x, q8, t8);
                const int lim = take - x < 8 ? take - x : 8;
                for (int y = 0; y < lim; ++y) {
                    const int op = ((q8 >> (8 * y)) & 0xff) == ((t8 >> (8 * y)) & 0xff) ? (int)OP_EQ : (int)OP_X;
                    if (op != op_cur) { if (len) put(); op_cur = op; len = 0; }
                    ++len;
                }
            }
            if (len) put();
        }
        if (live) {
            // element 0 of a left extension is the base next to the fixed point: sequence order is the reverse
            const uint32_t ends = n_runs ? (left ? end_codes(last_op, first_op) : end_codes(first_op, last_op)) : 0u;
            a.out[ti] = TaskOut{out_score, ext ? take : (int)tk.m, ext ? take : (int)tk.n, base + mine, ok ? n_runs : 0,
                                ends | (ext && take == end_row ? 0x80000000u : 0u)};
        }
    }
}

// ---- assembly of the task results into PAF rows ----------------------------------------------------------
struct AsmArgs {
    const Piece *pieces;
    const FixPt *fps;
    const uint32_t *task_off;
    const TaskOut *tout;
    const uint32_t *runs;
    size_t n_pieces;
    const uint32_t *qlen, *tlen, *rank_q, *rank_t, *chunk_of_t;
    int min_dp_score, end_bonus;
    // counting pass only -----------------------------------------------------------------------------------------------
    const uint32_t *plist;      // pieces to count (n_pieces of them); null: all
    const PieceGeom *pg;        // stub rule: with `late` set, a stub candidate (pg[i].stub_cand) whose blocks alone do not reach
    uint8_t *late;              // stub_score is flagged here - its end extensions, held back so far, have to run after all
    int stub_score;
    int bare;                   // 1: the rows of stub candidates are bare (no CIGAR, columns 10 / 11 zero): their tasks report scores only
    uint32_t stage_cap;         // slots of one 64-task step that go through the LDS buffer (ASM_STAGE; test hook HLMI_ASM_STAGE_CAP lowers it)
};

// One wavefront per piece, one lane per task (64 tasks per step).  The row's CIGAR is the concatenation of the
// kept tasks' runs with equal neighbours merged, so with S = runs of the kept tasks before a task and
// M = boundaries up to and including its own where the first code equals the last code of the previous
// non-empty kept task, run x of a task lands in slot S - M + x: a merged first run shares the slot of the run
// it extends.  Slots are zero-initialised and the first / last run of a task are added atomically (they are the
// only ones other tasks can touch); pass <false> needs just the first and last code of each task.
constexpr uint32_t ASM_MID = 4;
constexpr int ASM_STAGE = 1024;      // slots of one 64-task step staged in LDS (per wave)
template <bool WRITE>
__global__ __launch_bounds__(WG) void assemble_kernel(AsmArgs a, uint32_t *n_ops, uint8_t *valid, const uint64_t *ops_off,
                                                       uint32_t *ops, PafRec *recs, uint64_t *ord_hi, uint64_t *ord_lo) {
    __shared__ uint32_t s_stage[WRITE ? WAVES : 1][WRITE ? ASM_STAGE : 1];
    uint32_t *const stage = s_stage[WRITE ? (threadIdx.x >> 6) : 0];
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t idx = wave; idx < a.n_pieces; idx += n_waves) {
        const size_t i = !WRITE && a.plist ? (size_t)a.plist[idx] : idx;
        if (WRITE && !valid[i]) continue;
        const Piece p = a.pieces[i];
        const FixPt *fp = a.fps + p.fp_off;
        const TaskOut *to = a.tout + a.task_off[i];
        const uint32_t n_tasks = p.n_fp + 1;
        const bool bare = a.bare && a.pg[i].stub_cand != 0;
        uint32_t *w = WRITE ? ops + ops_off[i] : nullptr;
        long long score = 0;
        unsigned long long nmatch = 0, blen = 0;
        uint32_t slots = 0;                       // S - M carried over the 64-task steps
        uint32_t carry_code = 99;                 // last code of the last non-empty kept task so far
        constexpr uint32_t NO_SLOT = 0xffffffffu;
        uint32_t open_idx = NO_SLOT, open_val = 0; // WRITE: the slot still taking merges (wave-uniform)
        int ext_l_i = 0, ext_l_j = 0, ext_r_i = 0, ext_r_j = 0;
        TaskOut r_next{0, 0, 0, 0, 0, 0};
        if ((uint32_t)lane < n_tasks) r_next = to[lane];
        for (uint32_t t0 = 0; t0 < n_tasks; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            TaskOut r = r_next;
            r_next = TaskOut{0, 0, 0, 0, 0, 0};
            if (t + 64 < n_tasks) r_next = to[t + 64];        // in flight while this step's runs are copied
            bool keep = false;
            if (t < n_tasks) {
                keep = true;
                if (t == 0 || t == n_tasks - 1)   // an extension counts when it gains something (bonus: decision only)
                    keep = !(r.score + ((r.pad & 0x80000000u) ? a.end_bonus : 0) <= 0 || (r.n_runs == 0 && !bare));
            }
            if (keep && t == 0) { ext_l_i = r.bi; ext_l_j = r.bj; }
            if (keep && t == n_tasks - 1 && t != 0) { ext_r_i = r.bi; ext_r_j = r.bj; }
            long long sc = keep ? r.score : 0;
            const uint32_t nr = keep && !bare ? r.n_runs : 0;
            const uint32_t first = r.pad & 15u, last = (r.pad >> 4) & 15u;     // codes of the first / last run (TaskOut::pad)
            const unsigned long long ne_mask = __ballot(nr != 0);
            const unsigned long long below = ne_mask & ((1ull << lane) - 1ull);
            const int prev_lane = below ? 63 - __clzll((long long)below) : 0;
            const uint32_t prev_last = (uint32_t)__shfl((int)last, prev_lane, 64);
            const uint32_t prev_code = below ? (prev_last & 15u) : carry_code;
            const bool mrg = nr != 0 && (first & 15u) == prev_code;
            uint32_t tot_runs, tot_mrg;
            const uint32_t S = wave_excl_sum_u32(nr, lane, tot_runs);
            const uint32_t Mx = wave_excl_sum_u32(mrg ? 1u : 0u, lane, tot_mrg);
            if (WRITE) {
                // No atomics, no zeroed slots: every slot has ONE writer.  A slot is opened by a task's last run (or its only
                // run when that does not merge into the slot before) and takes the first runs of the tasks after it while they
                // merge; it closes at the next opener.  With P = inclusive prefix sum of the merged first-run lengths, the
                // opener at lane a collects P(b) - P(a), b = the next opener: one scan and one shuffle per 64 tasks.  The
                // slot open at the end of a step is carried in (open_idx, open_val) and written when it closes.
                // (Round 3 until here: the slots were zeroed and the first / last run of every task added atomically - C3's rows
                //  are mostly exact-match tasks of one run each, ~50 adds in a row on ONE address.)
                const uint32_t F = nr ? a.runs[r.runs_off] : 0u;
                const uint32_t L = nr >= 2 ? a.runs[r.runs_off + nr - 1] : F;
                // the first interior runs travel with F and L: a load behind this step's stores would wait for them
                // (loads and stores share one counter), and that wait was 63 % of the kernel (phase timers)
                uint32_t mid[ASM_MID];
#pragma unroll
                for (uint32_t u = 0; u < ASM_MID; ++u) mid[u] = u + 3 <= nr ? a.runs[r.runs_off + 1 + u] : 0u;
                const uint32_t P = wave_prefix_sum_incl_dpp(mrg ? F >> 4 : 0u);
                const bool opens = nr != 0 && (!mrg || nr >= 2);
                const unsigned long long om = __ballot(opens);
                const uint32_t P_end = (uint32_t)__builtin_amdgcn_readlane((int)P, 63);
                const uint32_t T0 = om ? (uint32_t)__shfl((int)P, __ffsll((long long)om) - 1, 64) : P_end;
                open_val += T0 << 4;
                if (om && open_idx != NO_SLOT && lane == 0) w[open_idx] = open_val;
                const unsigned long long above = lane == 63 ? 0ull : om & ~((2ull << lane) - 1ull);
                const uint32_t P_next = (uint32_t)__shfl((int)P, above ? __ffsll((long long)above) - 1 : 63, 64);
                const uint32_t base = slots + S - (Mx + (mrg ? 1u : 0u));
                const uint32_t own_slot = base + (nr ? nr - 1 : 0u);
                const uint32_t own_val = L + ((P_next - P) << 4);
                // The step's slots [slots, slots + n_out) are put together in LDS and leave as whole lines: a lane's own
                // stores are 4 bytes each at a stride of its neighbours' run counts (~10 partial lines per store
                // instruction; the kernel spent 63 % of its time issuing them).  A step with more slots than the buffer
                // holds stores directly.
                const uint32_t n_out = tot_runs - tot_mrg;
                const bool staged = n_out <= a.stage_cap;
                auto put = [&](uint32_t slot, uint32_t v) { if (staged) stage[slot - slots] = v; else w[slot] = v; };
                if (om) {
                    const int lb = 63 - __clzll((long long)om);              // the step's last opener: its slot stays open
                    open_idx = (uint32_t)__builtin_amdgcn_readlane((int)own_slot, lb);
                    open_val = (uint32_t)__builtin_amdgcn_readlane((int)own_val, lb);
                    if (opens && lane != lb) put(own_slot, own_val);
                }
                if (nr) {
                    blen += F >> 4;
                    if ((F & 15u) == OP_EQ) nmatch += F >> 4;
                }
                if (nr >= 2) {
                    blen += L >> 4;
                    if ((L & 15u) == OP_EQ) nmatch += L >> 4;
                    if (!mrg) put(base, F);
                    // (tried: four loads at a time, then their stores - 45 -> 47 ms; the head of the next piece requested
                    //  a piece ahead - 45 -> 48.5 ms)
#pragma unroll
                    for (uint32_t u = 0; u < ASM_MID; ++u)
                        if (u + 3 <= nr) {
                            blen += mid[u] >> 4;
                            if ((mid[u] & 15u) == OP_EQ) nmatch += mid[u] >> 4;
                            put(base + 1 + u, mid[u]);
                        }
                    for (uint32_t x = 1 + ASM_MID; x + 1 < nr; ++x) {
                        const uint32_t run = a.runs[r.runs_off + x];
                        blen += run >> 4;
                        if ((run & 15u) == OP_EQ) nmatch += run >> 4;
                        put(base + x, run);
                    }
                }
                if (staged) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);                      // the LDS writes (lgkmcnt 0)
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t k = (uint32_t)lane; k < n_out; k += 64)
                        if (slots + k != open_idx) w[slots + k] = stage[k];    // (the open slot is written when it closes)
                    __builtin_amdgcn_s_waitcnt(0xc07f);                      // read before the next step overwrites it
                    __builtin_amdgcn_wave_barrier();
                }
            }
            slots += tot_runs - tot_mrg;
            if (ne_mask) carry_code = (uint32_t)__shfl((int)last, 63 - __clzll((long long)ne_mask), 64) & 15u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o, 64);
            score += sc;
        }
        // extension shifts live in lane 0 (left) and in the lane of the last task (right)
        const int li = __shfl(ext_l_i, 0, 64), lj = __shfl(ext_l_j, 0, 64);
        const int last_lane = (int)((n_tasks - 1) & 63u);
        const int ri = __shfl(ext_r_i, last_lane, 64), rj = __shfl(ext_r_j, last_lane, 64);
        if (!WRITE) {
            if (lane == 0) {
                const bool ok = (slots > 0 || bare) && score >= a.min_dp_score;
                valid[i] = ok ? 1 : 0;
                n_ops[i] = ok ? slots : 0;
                // (the extensions of a stub candidate have not run: `score` is the score of its blocks)
                if (a.late) a.late[i] = a.pg[i].stub_cand && !((slots > 0 || bare) && score >= a.stub_score) ? 1 : 0;
            }
            continue;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { nmatch += __shfl_xor(nmatch, o, 64); blen += __shfl_xor(blen, o, 64); }
        if (lane == 0) {
            if (open_idx != NO_SLOT) w[open_idx] = open_val;
            const int qs = (int)fp[0].q - li, ts = (int)fp[0].t - lj;
            const int qe = (int)fp[p.n_fp - 1].q + ri, te = (int)fp[p.n_fp - 1].t + rj;
            const uint32_t ql = a.qlen[p.q];
            PafRec r{};
            r.qid = a.rank_q[p.q]; r.tid = a.rank_t[p.t];
            r.qlen = ql; r.tlen = a.tlen[p.t];
            r.qs = p.strand ? ql - (uint32_t)qe : (uint32_t)qs;
            r.qe = p.strand ? ql - (uint32_t)qs : (uint32_t)qe;
            r.ts = (uint32_t)ts; r.te = (uint32_t)te;
            r.nmatch = (uint32_t)nmatch; r.blen = (uint32_t)blen;
            r.flags = p.strand ? PF_REV : 0;
            r.chunk = a.chunk_of_t[p.t];
            r.cig_off = ops_off[i]; r.cig_n = slots; r.tie = 0;
            recs[i] = r;
            ord_hi[i] = (uint64_t)r.chunk << 32 | p.q;
            ord_lo[i] = (uint64_t)p.t << 43 | (uint64_t)p.strand << 42 | (uint64_t)(p.chain & 0x3fffffu) << 20 | (p.piece & 0xfffffu);
        }
    }
}

// The same with a LANE per piece: the pieces of short reads have half a dozen tasks of one or two runs each, and a wave per piece
// (62 M waves a step on the short-read calls) spends its time on scans over six lanes.  The lane walks its piece's tasks in
// order with the row's definition as it stands above: kept tasks' runs concatenated, a task's first run merged into the run
// before it when the codes agree.  The counting pass reads the codes of the first / last run from TaskOut::pad like the wave
// form; the writing pass reads the runs.
template <bool WRITE>
__global__ __launch_bounds__(WG) void assemble_lane_kernel(AsmArgs a, uint32_t *n_ops, uint8_t *valid, const uint64_t *ops_off,
                                                            uint32_t *ops, PafRec *recs, uint64_t *ord_hi, uint64_t *ord_lo) {
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= a.n_pieces) return;
    const size_t i = !WRITE && a.plist ? (size_t)a.plist[idx] : idx;
    if (WRITE && !valid[i]) return;
    const Piece p = a.pieces[i];
    const FixPt *fp = a.fps + p.fp_off;
    const TaskOut *to = a.tout + a.task_off[i];
    const uint32_t n_tasks = p.n_fp + 1;
    const bool bare = a.bare && a.pg[i].stub_cand != 0;
    uint32_t *w = WRITE ? ops + ops_off[i] : nullptr;
    long long score = 0;
    unsigned long long nmatch = 0, blen = 0;
    uint32_t slots = 0;                       // ops written (WRITE) / counted so far, the pending one excluded
    uint32_t pend = 0;                        // WRITE: the op still taking merges (len << 4 | code), 0: none
    uint32_t carry_code = 99;                 // last code of the last non-empty kept task so far
    int li = 0, lj = 0, ri = 0, rj = 0;
    for (uint32_t t = 0; t < n_tasks; ++t) {
        const TaskOut r = to[t];
        bool keep = true;
        if (t == 0 || t == n_tasks - 1)       // an extension counts when it gains something (bonus: decision only)
            keep = !(r.score + ((r.pad & 0x80000000u) ? a.end_bonus : 0) <= 0 || (r.n_runs == 0 && !bare));
        if (!keep) continue;
        if (t == 0) { li = r.bi; lj = r.bj; }
        else if (t == n_tasks - 1) { ri = r.bi; rj = r.bj; }
        score += r.score;
        const uint32_t nr = bare ? 0u : r.n_runs;
        if (!nr) continue;
        const uint32_t first = r.pad & 15u, last = (r.pad >> 4) & 15u;     // codes of the first / last run (TaskOut::pad)
        const bool mrg = first == carry_code;  // (both passes decide by these codes: the writer fills exactly the counted slots)
        carry_code = last;
        if (!WRITE) {
            slots += nr - (mrg ? 1u : 0u);
            continue;
        }
        for (uint32_t x = 0; x < nr; ++x) {
            const uint32_t run = a.runs[r.runs_off + x];
            blen += run >> 4;
            if ((run & 15u) == OP_EQ) nmatch += run >> 4;
            if (x == 0 && mrg) pend += (run >> 4) << 4;
            else {
                if (pend) w[slots++] = pend;
                pend = run;
            }
        }
    }
    if (!WRITE) {
        const bool ok = (slots > 0 || bare) && score >= a.min_dp_score;
        valid[i] = ok ? 1 : 0;
        n_ops[i] = ok ? slots : 0;
        if (a.late) a.late[i] = a.pg[i].stub_cand && !((slots > 0 || bare) && score >= a.stub_score) ? 1 : 0;
        return;
    }
    if (pend) w[slots++] = pend;
    const int qs = (int)fp[0].q - li, ts = (int)fp[0].t - lj;
    const int qe = (int)fp[p.n_fp - 1].q + ri, te = (int)fp[p.n_fp - 1].t + rj;
    const uint32_t ql = a.qlen[p.q];
    PafRec r{};
    r.qid = a.rank_q[p.q]; r.tid = a.rank_t[p.t];
    r.qlen = ql; r.tlen = a.tlen[p.t];
    r.qs = p.strand ? ql - (uint32_t)qe : (uint32_t)qs;
    r.qe = p.strand ? ql - (uint32_t)qs : (uint32_t)qe;
    r.ts = (uint32_t)ts; r.te = (uint32_t)te;
    r.nmatch = (uint32_t)nmatch; r.blen = (uint32_t)blen;
    r.flags = p.strand ? PF_REV : 0;
    r.chunk = a.chunk_of_t[p.t];
    r.cig_off = ops_off[i]; r.cig_n = slots; r.tie = 0;
    recs[i] = r;
    ord_hi[i] = (uint64_t)r.chunk << 32 | p.q;
    ord_lo[i] = (uint64_t)p.t << 43 | (uint64_t)p.strand << 42 | (uint64_t)(p.chain & 0x3fffffu) << 20 | (p.piece & 0xfffffu);
}

// the two end extensions of every listed piece, as task ids (stub rule: pieces whose extensions were held back and are
// needed after all)
__global__ void late_tasks_kernel(const uint32_t *plist, size_t n, const Piece *pieces, const uint32_t *task_off, uint32_t *tasks,
                                  uint32_t *count) {
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j == 0) *count = (uint32_t)(2 * n);
    if (j >= n) return;
    const uint32_t i = plist[j];
    tasks[2 * j] = task_off[i];
    tasks[2 * j + 1] = task_off[i] + pieces[i].n_fp;
}

// (rows + columns) of the DP tasks per kernel that runs them, after a classification pass: the algorithmic bytes of the DP
// launches (bench.py's roofline table).  Slots: LB_NAMES in align_span.
__global__ __launch_bounds__(WG) void class_bases_kernel(const uint8_t *cls, const uint8_t *cls_bare, const Task *tasks, size_t n,
                                                          unsigned long long *out) {
    __shared__ unsigned long long s_sum[14];
    if (threadIdx.x < 14) s_sum[threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = cls[i], cb = cls_bare ? cls_bare[i] : 0;
        if (!c && !cb) continue;
        const Task &t = tasks[i];
        int slot;
        if (c) slot = c == 1 ? ((int)t.m < NR_SMALL ? 0 : 1) : c == 3 ? 2 : c == 2 ? 3 : c == 4 ? 4 : c == CLS_LONG ? 5 : c == CLS_LONG32 ? 12 : 11;
        else slot = cb == 1 ? 6 : cb == 3 ? 7 : cb == 2 ? 8 : cb == 4 ? 9 : cb == CLS_LONG ? 10 : cb == CLS_LONG32 ? 13 : 11;
        atomicAdd(&s_sum[slot], (unsigned long long)((int)t.m + (int)t.n));
    }
    __syncthreads();
    if (threadIdx.x < 14 && s_sum[threadIdx.x]) atomicAdd(&out[threadIdx.x], s_sum[threadIdx.x]);
}

// LONG task list: how many of the tasks are end extensions, and their rows (statistics: align_long_ext / align_long_ext_rows)
__global__ __launch_bounds__(WG) void list_ext_kernel(const Task *tasks, const uint32_t *list, size_t n, unsigned long long *out) {
    unsigned long long c = 0, r = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Task &t = tasks[list[i]];
        if (t.kind & 3) { ++c; r += (unsigned long long)(int)t.m; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); r += __shfl_xor(r, o, 64); }
    if ((threadIdx.x & 63) == 0 && c) { atomicAdd(out, c); atomicAdd(out + 1, r); }
}

// flag[i] = (cls[i] == v): the list of one class outside the four of select_classes4_async
__global__ void class_flag_kernel(const uint8_t *cls, size_t n, uint8_t v, uint8_t *flag) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) flag[i] = cls[i] == v ? 1 : 0;
}

__global__ void compact_rows_kernel(const uint32_t *idx, size_t n, const PafRec *recs, const uint64_t *hi,
                                    const uint64_t *lo, PafRec *orecs, uint64_t *ohi, uint64_t *olo) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    orecs[i] = recs[idx[i]];
    ohi[i] = hi[idx[i]];
    olo[i] = lo[idx[i]];
}

// ---- deferring the pieces with LONG tasks ----------------------------------------------------------------------------------
// A LONG task is thousands of dependent rows; a query batch of a clean read set holds a few thousand of them (C3: 1.6 % of the
// pieces), and a launch over so few lasts as long as its longest task whatever the card could do meanwhile.  The pieces
// that have one are therefore set aside batch after batch and aligned together at the end of the run - one launch with two
// orders of magnitude more tasks.  Rows find their place by their stream-order keys, not by the batch they were aligned in.
// flag[i] = piece i has a LONG block or a LONG end extension (the predicates of build_task / classify_kernel)
__global__ __launch_bounds__(WG) void piece_long_flag_kernel(const Piece *pieces, size_t n, const FixPt *fps, const uint32_t *qlen,
                                                              const uint32_t *tlen, int ext_max, uint8_t *flag) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < n; i += n_waves) {
        const Piece p = pieces[i];
        const FixPt *fp = fps + p.fp_off;
        bool lng = false;
        if (lane == 0) {                                      // the two end extensions: rows their 64-diagonal band runs
            const FixPt a = fp[0], b = fp[p.n_fp - 1];
            const int ql = (int)qlen[p.q], tl = (int)tlen[p.t];
            auto rows = [&](int aq, int at) {
                const int m = aq < ext_max ? aq : ext_max, nn = at < m + BAND_W ? at : m + BAND_W;
                return m < nn + (BAND_W / 2 - 1) ? m : nn + (BAND_W / 2 - 1);
            };
            lng = rows((int)a.q, (int)a.t) > EXT_MAX || rows(ql - (int)b.q, tl - (int)b.t) > EXT_MAX;
        }
        for (uint32_t k = (uint32_t)lane; k + 1 < p.n_fp && !__any(lng); k += 64) {
            const FixPt f0 = fp[k], f1 = fp[k + 1];
            lng = lng || (int)(f1.q - f0.q) > BLOCK_MAX || (int)(f1.t - f0.t) > BLOCK_MAX;
        }
        const bool any = __any(lng);
        if (lane == 0) flag[i] = any ? 1 : 0;
    }
}
// the same with a lane per piece: pieces of a handful of fixed points (short reads on contigs: 62 M pieces of ~6 a step) would
// each occupy a wave above
__global__ __launch_bounds__(WG) void piece_long_flag_lane_kernel(const Piece *pieces, size_t n, const FixPt *fps, const uint32_t *qlen,
                                                                   const uint32_t *tlen, int ext_max, uint8_t *flag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Piece p = pieces[i];
    const FixPt *fp = fps + p.fp_off;
    FixPt f0 = fp[0];
    const FixPt b = fp[p.n_fp - 1];
    const int ql = (int)qlen[p.q], tl = (int)tlen[p.t];
    auto rows = [&](int aq, int at) {
        const int m = aq < ext_max ? aq : ext_max, nn = at < m + BAND_W ? at : m + BAND_W;
        return m < nn + (BAND_W / 2 - 1) ? m : nn + (BAND_W / 2 - 1);
    };
    bool lng = rows((int)f0.q, (int)f0.t) > EXT_MAX || rows(ql - (int)b.q, tl - (int)b.t) > EXT_MAX;
    for (uint32_t k = 1; k < p.n_fp && !lng; ++k) {
        const FixPt f1 = fp[k];
        lng = (int)(f1.q - f0.q) > BLOCK_MAX || (int)(f1.t - f0.t) > BLOCK_MAX;
        f0 = f1;
    }
    flag[i] = lng ? 1 : 0;
}
__global__ void gather_pieces_kernel(const Piece *src, const uint32_t *idx, size_t n, Piece *dst, uint32_t *n_fp) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Piece p = src[idx[i]];
    dst[i] = p;
    if (n_fp) n_fp[i] = p.n_fp;
}
// the fixed points of the set-aside pieces move into the deferred array: piece i's at fp_base + off[i]
__global__ __launch_bounds__(WG) void move_fixed_points_kernel(Piece *pieces, size_t n, const uint32_t *off, uint32_t fp_base, const FixPt *src,
                                                                FixPt *dst) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < n; i += n_waves) {
        const Piece p = pieces[i];
        const uint32_t to = off[i];
        for (uint32_t k = (uint32_t)lane; k < p.n_fp; k += 64) dst[to + k] = src[p.fp_off + k];
        if (lane == 0) pieces[i].fp_off = fp_base + to;
    }
}
}  // namespace

// pieces [0, P) of `pieces` with n_fp fixed points between them
struct PieceSpan { struct { const Piece *p; } pieces; size_t n_pieces, n_fp; struct { const FixPt *p; } fps; };
static void align_span(const AvaInput &in, const hlmi_ava_opts &o, const uint32_t *d_qlen, const uint32_t *d_tlen,
                       const PieceSpan &ch, AlignOut &out) {
    out = AlignOut();
    const size_t P = ch.n_pieces;
    if (!P) return;
    DBuf<uint32_t> tcnt(P), toff(P);
    hipLaunchKernelGGL(piece_task_count_kernel, grid1(P), dim3(WG), 0, stream(), ch.pieces.p, P, tcnt.p);
    exclusive_scan_u32(tcnt.p, toff.p, P);
    const size_t NT = ch.n_fp + P;              // a piece of n fixed points has n - 1 blocks and two extensions
    DBuf<Task> tasks(NT);
    if (P >= (1u << 30)) fail(HLMI_EINVAL, "more than 2^30 alignment pieces in one batch");
    DBuf<TaskRef> task_ref(NT);
    DBuf<PieceGeom> pgeom(P);
    if (NT <= 17 * P)                  // pieces of a handful of tasks: a lane each
        hipLaunchKernelGGL(task_ref_lane_kernel, grid1(P), dim3(WG), 0, stream(), ch.pieces.p, toff.p, P, d_qlen, d_tlen, in.Q->off.p, in.T->off.p,
                           ch.fps.p, o.stub_oh, ext_rows(o), task_ref.p, pgeom.p);
    else
    hipLaunchKernelGGL(task_ref_kernel, dim3((unsigned)std::min<size_t>(cdiv(P, (size_t)WAVES), 256 * 32)), dim3(WG), 0, stream(),
                       ch.pieces.p, toff.p, P, d_qlen, d_tlen, in.Q->off.p, in.T->off.p, ch.fps.p, o.stub_oh, ext_rows(o), task_ref.p, pgeom.p);
    HIP_CHECK(hipGetLastError());
    DBuf<TaskOut> tout(NT);
    DBuf<uint32_t> counters(2);
    DBuf<unsigned long long> astats(N_ALIGN_STATS);
    // runs + one open chunk per allocating lane of every launch: classify and the two align_kernel (one per wave), the two
    // align_narrow_kernel instances (one per 16-lane group)
    constexpr size_t MAX_BLOCKS = 256 * 16;
    // (a pool that overflows costs a second run of every DP kernel: 12 runs per task cover read sets with a few per cent
    // of errors, where the average is 8)
    const size_t open_chunks = MAX_BLOCKS * (4 * WAVES * RUN_CHUNK + 2 * 4 * WAVES * RUN_CHUNK_SMALL) +
                               (size_t)3 * 256 * 32 * 8 * PK_WAVES * RUN_CHUNK_SMALL;  // (three packed launches: 8 allocating lanes per wave)
    size_t run_share = std::max<size_t>(NT * 12, 1 << 16);
    DBuf<uint32_t> runs;
    AsmArgs as{};
    as.pieces = ch.pieces.p; as.fps = ch.fps.p; as.task_off = toff.p; as.tout = tout.p; as.n_pieces = P;
    as.min_dp_score = o.min_dp_score; as.end_bonus = o.end_bonus;
    as.qlen = d_qlen; as.tlen = d_tlen; as.rank_q = in.d_rank_q; as.rank_t = in.d_rank_t; as.chunk_of_t = in.d_chunk_of_t;
    as.pg = pgeom.p; as.stub_score = o.min_dp_score + std::max(0, o.end_bonus);
    as.stage_cap = (uint32_t)ASM_STAGE;
    if (const char *e = hook("HLMI_ASM_STAGE_CAP")) as.stage_cap = (uint32_t)std::min(ASM_STAGE, std::max(0, atoi(e)));
    DBuf<uint32_t> nops(P);
    DBuf<uint8_t> valid(P), late(o.stub_oh >= 0 ? P : 0);
    DBuf<uint32_t> late_idx(o.stub_oh >= 0 ? P : 0);
    size_t n_late = 0;
    const unsigned nba = (unsigned)std::min<size_t>(cdiv(P, (size_t)WAVES), 256 * 32);
    const bool small_pieces = NT <= 17 * P && !hook("HLMI_ASM_WAVE");      // half a dozen tasks per piece: the lane forms of the piece kernels
    // packed form of the near-diagonal DP (two tasks per lane, 16-bit scores): block scores must stay within +-4096 of the bias
    const int worst = std::max(std::max(o.match, o.mismatch), std::max(o.ambi, o.gap_open + o.gap_ext));
    const bool packed = worst > 0 && (long long)worst * (BLOCK_MAX + NARROW_W + 2) <= 4000 && o.match >= 0 && o.mismatch >= 0 &&
                        o.ambi >= 0 && o.gap_open >= 0 && o.gap_ext >= 0 && !hook("HLMI_NARROW_UNPACKED");
    // stub rule, second half: nobody reads the content of a stub candidate's row, so its tasks report scores only (score-only
    // instances of the packed and the 64-diagonal kernel; HLMI_STUB_FULL_ROWS keeps the candidates' CIGARs: test hook)
    const bool bare = o.stub_oh >= 0 && packed && !hook("HLMI_STUB_FULL_ROWS");
    as.bare = bare ? 1 : 0;
    // LONG tasks (align_long_kernel).  Sizes of a wave's scratch areas from the options: a chain link spans at most max_gap
    // bases and an extension ext_rows(o) rows.
    if (long_rows_cap(o) > 32000) fail(HLMI_EINVAL, "max_gap %d: alignment tasks are limited to 32 000 rows", o.max_gap);
    // An extension of <= EXT_MAX rows runs in the 64-diagonal kernels, which know no z-drop: it cannot matter there when a
    // drop of more than zdrop (at most D per row, D = the dearest single step) plus the climb back above the old best cell
    // (at most `match` per row) need more rows than EXT_MAX - otherwise every extension is a LONG task.
    bool ext_all_long = false;
    if (o.zdrop > 0) {
        const int D = std::max(std::max(o.mismatch, o.ambi), o.gap_open + o.gap_ext);
        const int rows_down = D > 0 ? (o.zdrop + D - 1) / D : EXT_MAX;
        // (the best cell is ranked with the end bonus: a row that reaches the query end counts end_bonus more)
        ext_all_long = (long long)o.match * (EXT_MAX - rows_down) + std::max(o.end_bonus, 0) > o.zdrop;
    }
    const uint32_t long_chunks_cap = (uint32_t)(long_rows_cap(o) + 31) / 32, long_runs_cap = (uint32_t)(2 * long_rows_cap(o) + 64);
    // 8 workgroups of 4 waves per CU: the walks of the tasks wait on memory.  (Fewer - to leave wave slots to the kernels on the
    // compute stream - was measured on a C5 chunk: 1.46 s with 2 048 workgroups, 1.45 with 1 024, 1.40 with 512, 1.43 with the long
    // kernels in line on the compute stream: the card is busy either way, what runs beside a kernel runs that much slower.)
    constexpr unsigned LONG_BLOCKS_MAX = 256 * 8;
    // The LONG tasks of a batch are few (C3: ~10 000 per batch) and serial in their rows: alone on the card their launch lasts as
    // long as its longest task.  They run on the side stream beside the batch's other DP kernels; every launch keeps its own
    // list / scratch / control words until the streams are joined (join_long) in front of whatever reads the task results.
    struct LongLaunch { DBuf<uint32_t> list, planes, runs, ctl; };
    std::vector<std::unique_ptr<LongLaunch>> long_launches;
    struct LongReady { AlignArgs al; LongArgs la; unsigned nb; bool half, tb; const char *timer; };
    std::vector<LongReady> long_ready;                   // prepared, not yet launched
    hipEvent_t long_done = nullptr;
    bool long_pending = false, long_on_side = false;     // launches not yet joined / some of them on the second stream
    struct SideGuard {                                   // an exception on the way out must not free buffers the side stream still uses
        ~SideGuard() { if (hipStream_t s = side_stream_if_created()) (void)hipStreamSynchronize(s); }      // (never creates it, never throws)
    } side_guard;
    auto join_long = [&]() {
        if (!long_pending) return;
        if (long_on_side) {                              // (in line on the compute stream - the default - there is nothing to join)
            if (!long_done) HIP_CHECK(hipEventCreateWithFlags(&long_done, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(long_done, side_stream()));
            HIP_CHECK(hipStreamWaitEvent(stream(), long_done, 0));
            long_on_side = false;
        }
        long_pending = false;
    };
    DBuf<uint8_t> long_flag;
    DBuf<uint32_t> long_list, long_n(1);
    size_t n_long_tb = 0, n_long_bare = 0;
    double long_rows_run = 0;
    // per DP launch: tasks and bases (rows + columns) - "align_n.<timer>", "align_bases.<timer>" of the statistics
    static const char *const LB_NAMES[] = {"align_narrow_small", "align_narrow", "align_narrow_long", "align_wide", "align_wide_short",
                                           "align_long", "align_score_narrow", "align_score_narrow_long", "align_score_wide",
                                           "align_score_wide_short", "align_score_long", "align_ungapped", "align_long32",
                                           "align_score_long32"};
    constexpr int N_LB = 14;
    DBuf<unsigned long long> lb_bases(N_LB + 2);              // (+ 2: end extensions among the LONG tasks, their rows)
    std::vector<double> lb_n(N_LB, 0.0);
    lb_bases.zero();
    auto note_list = [&](const char *timer, const Task *tk, const uint32_t *lst, size_t n) {
        if (!n) return;
        int slot = 0;
        while (slot < N_LB && strcmp(LB_NAMES[slot], timer) != 0) ++slot;
        if (slot == N_LB) return;
        lb_n[slot] += (double)n;             // (their bases: class_bases_kernel, once per classification pass)
        (void)tk; (void)lst;
    };
    for (int attempt = 0;; ++attempt) {
        const size_t cap_runs = run_share + open_chunks;
        if (cap_runs >= (1ull << 32)) fail(HLMI_ENOMEM, "CIGAR run pool exceeds 4G entries");
        runs.alloc(cap_runs);
        counters.zero();
        lb_bases.zero();
        std::fill(lb_n.begin(), lb_n.end(), 0.0);
        AlignArgs aa{};
        aa.tasks = tasks.p; aa.n_tasks = NT;
        aa.geom = TaskGeom{task_ref.p, pgeom.p, ch.fps.p};
        aa.qcodes = in.Q->codes(); aa.tcodes = in.T->codes();
        aa.q_total = (long long)in.Q->total; aa.t_total = (long long)in.T->total;
        aa.end_bonus = o.end_bonus;
        aa.ext_max = ext_rows(o);
        aa.ext_all_long = ext_all_long ? 1 : 0;
        aa.ungapped = o.bandwidth == 0 ? 1 : 0;
        aa.match = o.match; aa.mismatch = o.mismatch; aa.go = o.gap_open; aa.ge = o.gap_ext; aa.ambi = o.ambi;
        aa.go2 = o.gap_open2 > 0 ? o.gap_open2 : 0; aa.ge2 = o.gap_open2 > 0 ? o.gap_ext2 : 0;
        // the 16-diagonal kernels (gaps of at most 15 bases) know the first piece only
        if (aa.go2 && aa.go2 + aa.ge2 * (NARROW_W - 1) < aa.go + aa.ge * (NARROW_W - 1))
            fail(HLMI_EINVAL, "two-piece gap cost: the second piece must not be the cheaper one below %d bases", NARROW_W);
        {   // largest k with k*(match+mismatch) < match + 2*(open+ext), capped at 3 (7 runs)
            const int den = o.match + o.mismatch, num = o.match + 2 * (o.gap_open + o.gap_ext);
            aa.kmax = den > 0 && num > 0 ? std::min(3, (num - 1) / den) : 0;
            if (o.match <= 0 || o.gap_open < 0 || o.gap_ext <= 0) aa.kmax = -1;   // proof needs sane scores
            // fourth certificate: with k = kmax + 1 substitutions the only gapped paths that reach the diagonal's score
            // have one inserted + one deleted base and no mismatch; paths with a further gap base pair lose another
            // match + 2 ext, paths with a further gap another open + ext + match (three positions fit the run buffer)
            const int k = aa.kmax + 1, U = den, T = num;
            // fifth certificate: one substitution gains less than a further inserted + deleted base costs
            aa.kgap1 = aa.kmax >= 0 && U < o.gap_open + 2 * o.gap_ext + o.match && !hook("HLMI_NO_GAP1_CERT") ? 1 : 0;
            {   // seventh certificate: see classify_kernel (c(L) = the cheaper piece's cost of a gap of L bases)
                auto c = [&](int L) { int v = o.gap_open + o.gap_ext * L; if (o.gap_open2 > 0) v = std::min(v, o.gap_open2 + o.gap_ext2 * L); return v; };
                aa.kgap2 = aa.kgap1 && U < o.match + c(2) && 2 * U < o.match + 2 * c(1) && 2 * U < 2 * o.match + c(2) + c(3) - c(1) &&
                           c(2) == o.gap_open + 2 * o.gap_ext && c(1) == o.gap_open + o.gap_ext && !hook("HLMI_NO_GAP2_CERT") ? 1 : 0;
            }
            aa.kshift = aa.kmax >= 0 && k <= 3 && k * U > T && k * U < T + o.match + std::min(2 * o.gap_ext, o.gap_open + o.gap_ext) &&
                        !hook("HLMI_NO_SHIFT_CERT") ? 1 : 0;
        }
        aa.trim_ok = hook("HLMI_NO_SUFFIX_TRIM") ? 0 : 1;
        aa.one_ok = hook("HLMI_NO_ONE_PIECE_CERT") ? 0 : 1;
        {   // sixth certificate (classify_kernel: extensions with one or two substitutions); G = cheapest one-base gap
            const int U = o.match + o.mismatch;
            int G = o.gap_open + o.gap_ext;
            if (aa.go2 > 0) G = std::min(G, aa.go2 + aa.ge2);
            const bool sane = aa.kmax >= 0 && o.match > 0 && o.mismatch >= 0 && !hook("HLMI_NO_EXT_CERT");
            aa.kext_plain = sane && G >= U && U < 2 * G + o.match ? 1 : 0;
            aa.kext_bonus = 0;
            for (int k = 1; k <= 2; ++k)
                if (sane && o.end_bonus > 0 && k * U < G && k * U < o.end_bonus + o.match && k * U < 2 * G + o.match) aa.kext_bonus = k;
        }
        aa.out = tout.p; aa.runs = runs.p; aa.cap_runs = (uint32_t)cap_runs; aa.counters = counters.p;
        as.runs = runs.p;
        aa.run_buf_cap = 0xffffffffu;
        if (const char *e = hook("HLMI_RUN_BUF_CAP")) aa.run_buf_cap = (uint32_t)std::max(0, atoi(e));
        // pass 1: classify every task, finish the diagonal fast path right away
        DBuf<uint8_t> cls(NT), f1(NT), cls_bare(bare ? NT : 0);
        DBuf<uint32_t> list1(NT), list2(NT), list3(NT), list4(NT);
        DBuf<uint32_t> list_n(4);
        aa.cls_bare = bare ? cls_bare.p : nullptr;
        // LONG tasks of one class array (cls: with traceback, cls_bare: scores only): list, then align_long_kernel
        // the tasks of one class outside the four of select_classes4_async -> long_list; returns their number
        auto class_list = [&](const uint8_t *cls_arr, uint8_t v) -> size_t {
            if (long_flag.n < NT) { long_flag.alloc(NT); long_list.alloc(NT); }
            hipLaunchKernelGGL(class_flag_kernel, grid1(NT), dim3(WG), 0, stream(), cls_arr, NT, v, long_flag.p);
            select_flagged_indices_async(long_flag.p, long_list.p, NT, long_n.p);
            return (size_t)long_n.download(1)[0];
        };
        // bandwidth 0: everything the certificates left over, along the diagonal
        auto run_ungapped = [&](const uint8_t *cls_arr, bool tb) -> size_t {
            if (!aa.ungapped) return 0;
            const size_t nl = class_list(cls_arr, CLS_UNGAPPED);
            if (!nl) return 0;
            aa.list = long_list.p; aa.n_list = nl;
            note_list("align_ungapped", tasks.p, long_list.p, nl);
            KTimer kt("align_ungapped");
            const unsigned nb = (unsigned)std::min<size_t>(cdiv(nl, (size_t)WG), MAX_BLOCKS);
            if (tb) hipLaunchKernelGGL(align_ungapped_kernel<true>, dim3(nb), dim3(WG), 0, stream(), aa, o.zdrop);
            else hipLaunchKernelGGL(align_ungapped_kernel<false>, dim3(nb), dim3(WG), 0, stream(), aa, o.zdrop);
            HIP_CHECK(hipGetLastError());
            return nl;
        };
        // LONG tasks of class cv (CLS_LONG: 64 diagonals, a task per wave; CLS_LONG32: 32 diagonals, two tasks per wave)
        auto run_long_class = [&](const uint8_t *cls_arr, bool tb, uint8_t cv) -> size_t {
            const bool half = cv == CLS_LONG32;
            const char *timer = half ? (tb ? "align_long32" : "align_score_long32") : (tb ? "align_long" : "align_score_long");
            const size_t nl = class_list(cls_arr, cv);               // (waits for the stream: the task records are written)
            if (!nl) return 0;
            note_list(timer, tasks.p, long_list.p, nl);
            hipLaunchKernelGGL(list_ext_kernel, dim3((unsigned)std::min<size_t>(cdiv(nl, (size_t)WG), 1024)), dim3(WG), 0, stream(), tasks.p,
                               long_list.p, nl, lb_bases.p + N_LB);
            long_launches.emplace_back(new LongLaunch());
            LongLaunch &L = *long_launches.back();
            {   // longest first: the work queue then ends with the short tasks (and the two tasks of a wave are equally long)
                DBuf<uint32_t> key(nl);
                hipLaunchKernelGGL(task_rows_desc_key_kernel, grid1(nl), dim3(WG), 0, stream(), tasks.p, long_list.p, nl, key.p);
                L.list.alloc(nl);
                HIP_CHECK(hipMemcpyAsync(L.list.p, long_list.p, nl * 4, hipMemcpyDeviceToDevice, stream()));
                sort_pairs_u32_u32(key, L.list, nl, 0, 16);
            }
            const size_t units = half ? (nl + 1) / 2 : nl;           // waves' worth of work
            const unsigned nb = (unsigned)std::min<size_t>((units + WAVES - 1) / WAVES, LONG_BLOCKS_MAX);
            const size_t n_waves = (size_t)nb * WAVES;
            LongArgs la{};
            if (tb) {
                L.planes.alloc(n_waves * (size_t)N_WPLANES * long_chunks_cap * 64);
                L.runs.alloc(n_waves * (size_t)long_runs_cap * (half ? 2 : 1));
                la.planes = L.planes.p; la.run_scratch = L.runs.p;
            }
            la.chunks_cap = long_chunks_cap; la.runs_cap = long_runs_cap; la.zdrop = o.zdrop;
            L.ctl.alloc(4);
            L.ctl.zero();
            la.next = L.ctl.p; la.too_long = L.ctl.p + 1; la.rows_run = (unsigned long long *)(L.ctl.p + 2);
            AlignArgs al = aa;
            al.list = L.list.p; al.n_list = nl;
            long_ready.push_back(LongReady{al, la, nb, half, tb, timer});
            return nl;
        };
        // every prepared LONG launch starts after ALL their lists are built: on the second stream (HLMI_LONG_SIDE_STREAM) a selection
        // pass queued behind a running long kernel waits for the whole of it (its first round of tasks fills every SIMD; measured:
        // a 40 us scan took 31 ms), and with it everything the compute stream was to run beside the long kernel
        auto launch_long_ready = [&]() {
            if (long_ready.empty()) return;
            sync();                                                  // lists, keys and control words are in place
            // in line on the compute stream.  (HLMI_LONG_SIDE_STREAM: on the second stream, beside the DP kernels that follow -
            // measured equal, C3 slice 957 vs 961 ms, C5 chunk 3.55 vs 3.56 s: the card is busy either way, and kernels that
            // share it stretch each other's timers, which is why the default is the one stream)
            const bool on_side = hook("HLMI_LONG_SIDE_STREAM") != nullptr;
            hipStream_t ls = on_side ? side_stream() : stream();
            long_on_side = long_on_side || on_side;
            for (const LongReady &r : long_ready) {
                KTimer kt(r.timer, ls);
                const bool two = aa.go2 > 0;
                auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(r.nb), dim3(WG), 0, ls, r.al, r.la); };
                if (r.half) {      // (the 32-diagonal band knows the first piece of the gap cost alone: section 5 of DESIGN.md)
                    if (r.tb) go(align_long32_kernel<false, true>); else go(align_long32_kernel<false, false>);
                } else {
                    if (r.tb) { if (two) go(align_long_kernel<true, true>); else go(align_long_kernel<false, true>); }
                    else { if (two) go(align_long_kernel<true, false>); else go(align_long_kernel<false, false>); }
                }
            }
            HIP_CHECK(hipGetLastError());
            long_ready.clear();
            long_pending = true;
        };
        auto run_long = [&](const uint8_t *cls_arr, bool tb) -> size_t {
            if (aa.ungapped) return run_ungapped(cls_arr, tb);       // (no LONG class there: every DP task is CLS_UNGAPPED)
            // (prepared only: launch_long_ready() starts them)
            const size_t n32 = run_long_class(cls_arr, tb, CLS_LONG32);
            return n32 + run_long_class(cls_arr, tb, CLS_LONG);
        };
        // after a join and a wait for the stream: did a LONG task exceed its scratch area?
        auto check_long = [&]() {
            for (auto &L : long_launches) {
                if (!L->ctl.n) continue;
                const std::vector<uint32_t> c = L->ctl.download(4);
                if (c[1]) fail(HLMI_EINVAL, "an alignment task exceeds the scratch area of align_long_kernel");
                long_rows_run += (double)((unsigned long long)c[3] << 32 | c[2]);
                // (read once; the launch is over - joined and waited for -: its planes (GBs at max_gap 10 000), runs and list go
                //  back to the pool now, not when align_span returns: a retry of the span allocates them again)
                L->ctl.release(); L->planes.release(); L->runs.release(); L->list.release();
            }
        };
        // the tasks of stub candidates (cls_bare): score-only kernels over their own four lists
        size_t n_bare_tasks = 0;
        auto run_bare_long = [&]() {
            if (!bare) return;
            const size_t nl = run_long(cls_bare.p, false);
            n_bare_tasks += nl; n_long_bare += nl;
        };
        auto run_bare = [&](bool with_long) {             // (the lists of the four classes are rebuilt here: after the kernels that use them)
            if (!bare) return;
            if (with_long) { run_bare_long(); launch_long_ready(); }
            select_classes4_async(cls_bare.p, NT, list1.p, list2.p, list3.p, list4.p, list_n.p);
            const std::vector<uint32_t> hb = list_n.download(4);
            n_bare_tasks += (size_t)hb[0] + hb[1] + hb[2] + hb[3];
            for (int which = 0; which < 2; ++which) {            // near-diagonal lists in ascending row order (octets run equally long)
                DBuf<uint32_t> &lst = which ? list3 : list1;
                const size_t nl = which ? hb[2] : hb[0];
                if (nl < 8) continue;
                DBuf<uint32_t> key(nl);
                hipLaunchKernelGGL(task_rows_key_kernel, grid1(nl), dim3(WG), 0, stream(), tasks.p, lst.p, nl, key.p);
                sort_pairs_u32_u32(key, lst, nl, 0, 7);
            }
            auto pk_grid = [](size_t n) { return dim3((unsigned)std::max<size_t>(1, std::min<size_t>(((n + 7) / 8 + PK_WAVES - 1) / PK_WAVES, 256 * 32))); };
            auto w_grid = [](size_t n) { return dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n + WAVES - 1) / WAVES, 256 * 16))); };
            if (hb[0]) {
                note_list("align_score_narrow", tasks.p, list1.p, hb[0]);
                KTimer kt("align_score_narrow");
                aa.list = list1.p; aa.n_list = hb[0];
                hipLaunchKernelGGL((align_narrow_pk_kernel<NR_SHORT, false>), pk_grid(hb[0]), dim3(64 * PK_WAVES), 0, stream(), aa);
            }
            if (hb[2]) {
                note_list("align_score_narrow_long", tasks.p, list3.p, hb[2]);
                KTimer kt("align_score_narrow_long");
                aa.list = list3.p; aa.n_list = hb[2];
                hipLaunchKernelGGL((align_narrow_pk_kernel<BLOCK_MAX, false>), pk_grid(hb[2]), dim3(64 * PK_WAVES), 0, stream(), aa);
            }
            if (hb[1]) {
                note_list("align_score_wide", tasks.p, list2.p, hb[1]);
                KTimer kt("align_score_wide");
                aa.list = list2.p; aa.n_list = hb[1];
                hipLaunchKernelGGL((align_kernel<EXT_MAX, false>), w_grid(hb[1]), dim3(WG), 0, stream(), aa);
            }
            if (hb[3]) {
                note_list("align_score_wide_short", tasks.p, list4.p, hb[3]);
                KTimer kt("align_score_wide_short");
                aa.list = list4.p; aa.n_list = hb[3];
                hipLaunchKernelGGL((align_kernel<WIDE_SHORT, false>), w_grid(hb[3]), dim3(WG), 0, stream(), aa);
            }
        };
        {
            KTimer kt("align_classify");
            const unsigned nbc = (unsigned)std::min<size_t>(cdiv(NT, (size_t)WG), MAX_BLOCKS);
            astats.zero();
            aa.defer_flag = f1.p;
            aa.defer_list = list1.p; aa.defer_count = list_n.p;
            hipLaunchKernelGGL(classify_kernel<1>, dim3(nbc ? nbc : 1), dim3(WG), 0, stream(), aa, cls.p, astats.p);
            select_flagged_indices_async(f1.p, list1.p, NT, list_n.p);
            const unsigned nb2 = std::max(1u, nbc / 4);
            hipLaunchKernelGGL(classify_kernel<2>, dim3(nb2), dim3(WG), 0, stream(), aa, cls.p, astats.p);
        }
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(class_bases_kernel, dim3((unsigned)std::min<size_t>(cdiv(NT, (size_t)WG), 2048)), dim3(WG), 0, stream(), cls.p,
                           bare ? cls_bare.p : nullptr, tasks.p, NT, lb_bases.p);
        select_classes4_async(cls.p, NT, list1.p, list2.p, list3.p, list4.p, list_n.p);      // the four DP task lists
        const std::vector<uint32_t> hn = list_n.download(4);
        const size_t n1 = hn[0], n2 = hn[1], n3 = hn[2], n4 = hn[3];
        for (int which = 0; which < 2; ++which) {
            DBuf<uint32_t> &lst = which ? list3 : list1;
            const size_t nl = which ? n3 : n1;
            if (nl < 8) continue;
            DBuf<uint32_t> key(nl);
            hipLaunchKernelGGL(task_rows_key_kernel, grid1(nl), dim3(WG), 0, stream(), tasks.p, lst.p, nl, key.p);
            sort_pairs_u32_u32(key, lst, nl, 0, 7);
        }
        // the LONG tasks first, on the side stream, and the score-only tasks of the stub candidates: everything below runs beside them
        // (the candidates' few LONG tasks first: list building behind a running align_long_kernel is slowed down tenfold - the
        //  long kernel's first round of tasks fills every SIMD)
        // (the classifier counted them: most batches of a clean read set have none left once their pieces are set aside, and
        //  building an empty list is four passes over the class arrays)
        unsigned long long long_seen = astats.download(N_ALIGN_STATS)[ST_LONG];
        if (long_seen || aa.ungapped) {
            run_bare_long();
            n_long_tb += run_long(cls.p, true);
            launch_long_ready();
        }
        // pass 2a: near-diagonal blocks, four per wave in the 16-diagonal band
        if (n1 && packed) {
            // the list is in ascending row order: its head (fewer than NR_SMALL rows, counted by the classifier)
            // runs in the instance with half the plane LDS and twice the resident waves
            size_t n_small = 0;
            if (n1 >= 8) n_small = std::min<size_t>(n1, (size_t)astats.download(N_ALIGN_STATS)[ST_NARROW_SMALL]);
            if (n_small) {
                note_list("align_narrow_small", tasks.p, list1.p, n_small);
                KTimer kt("align_narrow_small");
                aa.list = list1.p; aa.n_list = n_small;
                const unsigned nb = (unsigned)std::min<size_t>(((n_small + 7) / 8 + PK_WAVES - 1) / PK_WAVES, 256 * 32);
                hipLaunchKernelGGL(align_narrow_pk_kernel<NR_SMALL>, dim3(nb ? nb : 1), dim3(64 * PK_WAVES), 0, stream(), aa);
            }
            if (n1 > n_small) {
                note_list("align_narrow", tasks.p, list1.p + n_small, n1 - n_small);
                KTimer kt("align_narrow");
                aa.list = list1.p + n_small; aa.n_list = n1 - n_small;
                const unsigned nb = (unsigned)std::min<size_t>(((aa.n_list + 7) / 8 + PK_WAVES - 1) / PK_WAVES, 256 * 32);
                hipLaunchKernelGGL(align_narrow_pk_kernel<NR_SHORT>, dim3(nb ? nb : 1), dim3(64 * PK_WAVES), 0, stream(), aa);
            }
        } else if (n1) {
            note_list("align_narrow", tasks.p, list1.p, n1);
            KTimer kt("align_narrow");
            aa.list = list1.p; aa.n_list = n1;
            const unsigned nb = (unsigned)std::min<size_t>(((n1 + 3) / 4 + WAVES - 1) / WAVES, 256 * 16);
            hipLaunchKernelGGL(align_narrow_kernel<NR_SHORT>, dim3(nb ? nb : 1), dim3(WG), 0, stream(), aa);
        }
        if (n3 && packed && !hook("HLMI_NARROW_LONG_UNPACKED")) {
            // blocks of more than NR_SHORT rows (divergent reads: C5 has as many of these as of the short ones) in the packed
            // form as well: eight tasks per wave at twice the plane LDS (25 KB per wave) instead of four
            note_list("align_narrow_long", tasks.p, list3.p, n3);
            KTimer kt("align_narrow_long");
            aa.list = list3.p; aa.n_list = n3;
            const unsigned nb = (unsigned)std::min<size_t>(((n3 + 7) / 8 + PK_WAVES - 1) / PK_WAVES, 256 * 32);
            hipLaunchKernelGGL(align_narrow_pk_kernel<BLOCK_MAX>, dim3(nb ? nb : 1), dim3(64 * PK_WAVES), 0, stream(), aa);
        } else if (n3) {
            note_list("align_narrow_long", tasks.p, list3.p, n3);
            KTimer kt("align_narrow_long");
            aa.list = list3.p; aa.n_list = n3;
            const unsigned nb = (unsigned)std::min<size_t>(((n3 + 3) / 4 + WAVES - 1) / WAVES, 256 * 16);
            hipLaunchKernelGGL(align_narrow_kernel<BLOCK_MAX>, dim3(nb ? nb : 1), dim3(WG), 0, stream(), aa);
        }
        // pass 2b: the rest (wide blocks, end extensions) in the 64-diagonal band
        auto run_wide = [&](size_t n_long, size_t n_short) {
            if (n_long) {
                note_list("align_wide", tasks.p, list2.p, n_long);
                KTimer kt("align_wide");
                aa.list = list2.p; aa.n_list = n_long;
                const unsigned nb = (unsigned)std::min<size_t>((n_long + WAVES - 1) / WAVES, 256 * 16);
                hipLaunchKernelGGL(align_kernel<EXT_MAX>, dim3(nb ? nb : 1), dim3(WG), 0, stream(), aa);
            }
            if (n_short) {
                note_list("align_wide_short", tasks.p, list4.p, n_short);
                KTimer kt("align_wide_short");
                aa.list = list4.p; aa.n_list = n_short;
                const unsigned nb = (unsigned)std::min<size_t>((n_short + WAVES - 1) / WAVES, 256 * 16);
                hipLaunchKernelGGL(align_kernel<WIDE_SHORT>, dim3(nb ? nb : 1), dim3(WG), 0, stream(), aa);
            }
        };
        run_wide(n2, n4);
        run_bare(false);
        size_t n_wide_late = 0;
        if (o.stub_oh >= 0) {
            // Stub rule (hlmi_ava_opts::stub_oh; proof at oracle/ava_oracle.c:is_stub).  The end extensions of the stub
            // candidates were held back.  Their blocks are scored now: a candidate that is certain to be reported without them
            // (blocks >= min_dp_score + end_bonus: an extension adds a positive score, or at least 1 - end_bonus when it
            // reaches the query end) stays without - its row can only be dropped by the consumer's overhang test; the others
            // get their extensions in a second round.
            as.plist = nullptr; as.n_pieces = P; as.late = late.p;
            join_long();
            {
                KTimer kt("assemble_count");
                if (small_pieces) hipLaunchKernelGGL(assemble_lane_kernel<false>, grid1(as.n_pieces), dim3(WG), 0, stream(), as, nops.p, valid.p,
                                                     nullptr, nullptr, nullptr, nullptr, nullptr);
                else
                hipLaunchKernelGGL(assemble_kernel<false>, dim3(nba), dim3(WG), 0, stream(), as, nops.p, valid.p, nullptr, nullptr,
                                   nullptr, nullptr, nullptr);
            }
            as.late = nullptr;
            n_late = select_flagged_indices(late.p, late_idx.p, P);
            if (n_late) {
                hipLaunchKernelGGL(late_tasks_kernel, grid1(n_late), dim3(WG), 0, stream(), late_idx.p, n_late, ch.pieces.p, toff.p,
                                   list1.p, list_n.p);
                HIP_CHECK(hipMemsetAsync(cls.p, 0, NT, stream()));
                if (bare) HIP_CHECK(hipMemsetAsync(cls_bare.p, 0, NT, stream()));
                {
                    KTimer kt("align_classify");
                    aa.defer_list = list1.p; aa.defer_count = list_n.p;
                    const unsigned nb2 = (unsigned)std::min<size_t>(cdiv(2 * n_late, (size_t)WG), MAX_BLOCKS);
                    hipLaunchKernelGGL(classify_kernel<2>, dim3(nb2 ? nb2 : 1), dim3(WG), 0, stream(), aa, cls.p, astats.p);
                }
                hipLaunchKernelGGL(class_bases_kernel, dim3((unsigned)std::min<size_t>(cdiv(NT, (size_t)WG), 2048)), dim3(WG), 0, stream(), cls.p,
                                   bare ? cls_bare.p : nullptr, tasks.p, NT, lb_bases.p);
                if (bare) {                        // (every late task belongs to a candidate)
                    const size_t before = n_bare_tasks;
                    const bool more_long = astats.download(N_ALIGN_STATS)[ST_LONG] > long_seen || aa.ungapped;
                    run_bare(more_long);
                    n_wide_late = n_bare_tasks - before;
                } else {
                    select_classes4_async(cls.p, NT, list1.p, list2.p, list3.p, list4.p, list_n.p);
                    const std::vector<uint32_t> hl = list_n.download(4);
                    run_wide(hl[1], hl[3]);
                    const bool more_long = astats.download(N_ALIGN_STATS)[ST_LONG] > long_seen || aa.ungapped;
                    const size_t nl = more_long ? run_long(cls.p, true) : 0;
                    launch_long_ready();
                    n_long_tb += nl;
                    n_wide_late = (size_t)hl[1] + hl[3] + nl;
                }
                as.plist = late_idx.p; as.n_pieces = n_late;
                const unsigned nbl = (unsigned)std::min<size_t>(cdiv(n_late, (size_t)WAVES), 256 * 32);
                join_long();
                {
                    KTimer kt("assemble_count");
                    if (small_pieces) hipLaunchKernelGGL(assemble_lane_kernel<false>, grid1(as.n_pieces), dim3(WG), 0, stream(), as, nops.p, valid.p,
                                                         nullptr, nullptr, nullptr, nullptr, nullptr);
                    else
                    hipLaunchKernelGGL(assemble_kernel<false>, dim3(nbl), dim3(WG), 0, stream(), as, nops.p, valid.p, nullptr, nullptr,
                                       nullptr, nullptr, nullptr);
                }
                as.plist = nullptr; as.n_pieces = P;
            }
        }
        if (attempt == 0) {
            stat_add("align_tasks_narrow", (double)(n1 + n3)); stat_add("align_tasks_wide", (double)(n2 + n4 + (bare ? 0 : n_wide_late)));
            stat_add("align_tasks_score_only", (double)n_bare_tasks);
            stat_add("align_tasks_long", (double)n_long_tb); stat_add("align_tasks_long_score_only", (double)n_long_bare);
        }
        HIP_CHECK(hipGetLastError());
        join_long();
        std::vector<uint32_t> hc = counters.download(2);
        check_long();
        if (!hc[1]) break;
        if (attempt >= 3) fail(HLMI_ENOMEM, "CIGAR run pool overflow");
        run_share *= 4;
    }
    stat_add("align_tasks", (double)NT);
    {
        std::vector<unsigned long long> h = astats.download(N_ALIGN_STATS);
        stat_add("align_tasks_fast", (double)h[ST_FAST]);
        stat_add("align_tasks_dp", (double)h[ST_DP]);
        stat_add("align_dp_rows", (double)h[ST_DP_ROWS]);
        stat_add("align_dp_bases", (double)h[ST_BASES]);              // Lq + Lt over all tasks
        stat_add("align_bases_classify", (double)h[ST_BASES_SQUARE]);
        stat_add("align_bases_narrow", (double)h[ST_BASES_NARROW]);
        stat_add("align_bases_wide", (double)h[ST_BASES_WIDE]);
        stat_add("align_bases_long", (double)h[ST_BASES_LONG]);
        const std::vector<unsigned long long> lb = lb_bases.download(N_LB + 2);
        stat_add("align_long_rows_run", long_rows_run);
        stat_add("align_long_ext", (double)lb[N_LB]);
        stat_add("align_long_ext_rows", (double)lb[N_LB + 1]);
        for (int k = 0; k < N_LB; ++k)
            if (lb_n[k] > 0) {
                stat_add(std::string("align_n.") + LB_NAMES[k], lb_n[k]);
                stat_add(std::string("align_bases.") + LB_NAMES[k], (double)lb[k]);
            }
        stat_add("align_tasks_wide_one_piece", (double)h[ST_WIDE_ONE]);
        stat_add("align_ext_certified", (double)h[ST_EXT_CERT]);
        stat_add("align_ext_dp_bonus_row_elsewhere", (double)h[ST_EXT_DP_ROW]);
        stat_add("align_ext_dp_substitutions", (double)h[ST_EXT_DP_K]);
        stat_add("align_ext_held", (double)h[ST_STUB_EXT]);            // end extensions of stub candidates ...
        stat_add("align_ext_late", (double)(2 * n_late));              // ... of which these had to run after all
    }
    // assemble
    if (o.stub_oh < 0) {
        KTimer kt("assemble_count");
        if (small_pieces) hipLaunchKernelGGL(assemble_lane_kernel<false>, grid1(as.n_pieces), dim3(WG), 0, stream(), as, nops.p, valid.p,
                                             nullptr, nullptr, nullptr, nullptr, nullptr);
        else
        hipLaunchKernelGGL(assemble_kernel<false>, dim3(nba), dim3(WG), 0, stream(), as, nops.p, valid.p, nullptr, nullptr,
                           nullptr, nullptr, nullptr);
    }
    HIP_CHECK(hipGetLastError());
    DBuf<uint64_t> ooff(P);
    exclusive_scan_u32_to_u64(nops.p, ooff.p, P);
    const size_t n_ops = (size_t)(download_one(ooff.p + (P - 1)) + download_one(nops.p + (P - 1)));
    DBuf<uint32_t> vidx(P);
    const size_t R = select_flagged_indices(valid.p, vidx.p, P);
    if (!R) return;
    out.ops.alloc(n_ops ? n_ops : 1);
    DBuf<PafRec> recs(P);
    DBuf<uint64_t> hi(P), lo(P);
    {
        KTimer kt("assemble_write");
        if (small_pieces) hipLaunchKernelGGL(assemble_lane_kernel<true>, grid1(as.n_pieces), dim3(WG), 0, stream(), as, nullptr, valid.p, ooff.p,
                                             out.ops.p, recs.p, hi.p, lo.p);
        else
        hipLaunchKernelGGL(assemble_kernel<true>, dim3(nba), dim3(WG), 0, stream(), as, nullptr, valid.p, ooff.p, out.ops.p,
                           recs.p, hi.p, lo.p);
    }
    HIP_CHECK(hipGetLastError());
    out.recs.alloc(R);
    out.ord_hi.alloc(R);
    out.ord_lo.alloc(R);
    hipLaunchKernelGGL(compact_rows_kernel, grid1(R), dim3(WG), 0, stream(), vidx.p, R, recs.p, hi.p, lo.p, out.recs.p,
                       out.ord_hi.p, out.ord_lo.p);
    HIP_CHECK(hipGetLastError());
    sync();
    out.n_rows = R;
    out.n_ops = n_ops;
}

}  // namespace hlmi

namespace hlmi {
// The CIGAR run pool and the task lists of one alignment pass are indexed with 32 bits: a batch with more tasks than
// that allows (many short pieces per anchor: divergent read sets) is aligned in spans of consecutive pieces.
void align_pieces(const AvaInput &in, const hlmi_ava_opts &o, const uint32_t *d_qlen, const uint32_t *d_tlen,
                  const ChainOut &ch_in, std::vector<AlignOut> &outs, DeferredPieces *defer) {
    if (!ch_in.n_pieces) return;
    // Pieces with LONG tasks are set aside (see piece_long_flag_kernel) when they are few: a batch where most pieces have one
    // (divergent read sets: C5) keeps them - its long launches are large enough by themselves.
    ChainOut kept;
    const ChainOut *chp = &ch_in;
    if (defer && o.bandwidth != 0) {
        const size_t P0 = ch_in.n_pieces;
        DBuf<uint8_t> flag(P0), nflag(P0);
        if (ch_in.n_fp <= 16 * P0)         // few fixed points per piece: a lane each
            hipLaunchKernelGGL(piece_long_flag_lane_kernel, grid1(P0), dim3(WG), 0, stream(), ch_in.pieces.p, P0, ch_in.fps.p, d_qlen, d_tlen,
                               ext_rows(o), flag.p);
        else
        hipLaunchKernelGGL(piece_long_flag_kernel, dim3((unsigned)std::min<size_t>(cdiv(P0, (size_t)WAVES), 256 * 32)), dim3(WG), 0, stream(),
                           ch_in.pieces.p, P0, ch_in.fps.p, d_qlen, d_tlen, ext_rows(o), flag.p);
        DBuf<uint32_t> lidx(P0);
        const size_t n_long = select_flagged_indices(flag.p, lidx.p, P0);
        if (n_long && n_long * 10 <= P0) {
            // the set-aside pieces, with their fixed points, behind what earlier batches left
            DeferredPieces::Part part;
            part.pieces.alloc(n_long);
            DBuf<uint32_t> nfp(n_long), off(n_long);
            hipLaunchKernelGGL(gather_pieces_kernel, grid1(n_long), dim3(WG), 0, stream(), ch_in.pieces.p, lidx.p, n_long, part.pieces.p, nfp.p);
            exclusive_scan_u32(nfp.p, off.p, n_long);
            const size_t n_fp_long = (size_t)download_one(off.p + (n_long - 1)) + (size_t)download_one(nfp.p + (n_long - 1));
            part.fps.alloc(n_fp_long);
            if (defer->n_fp + n_fp_long >= (1ull << 32)) fail(HLMI_EINVAL, "more than 2^32 fixed points in the set-aside alignment pieces");
            hipLaunchKernelGGL(move_fixed_points_kernel, dim3((unsigned)std::min<size_t>(cdiv(n_long, (size_t)WAVES), 256 * 32)), dim3(WG), 0, stream(),
                               part.pieces.p, n_long, off.p, (uint32_t)defer->n_fp, ch_in.fps.p, part.fps.p);
            part.n_pieces = n_long; part.n_fp = n_fp_long; part.fp_base = defer->n_fp;
            defer->n_pieces += n_long; defer->n_fp += n_fp_long;
            defer->parts.push_back(std::move(part));
            // the others, in their order (the fixed points stay where they are)
            hipLaunchKernelGGL(class_flag_kernel, grid1(P0), dim3(WG), 0, stream(), flag.p, P0, (uint8_t)0, nflag.p);
            DBuf<uint32_t> kidx(P0);
            const size_t n_keep = select_flagged_indices(nflag.p, kidx.p, P0);
            kept.pieces.alloc(n_keep ? n_keep : 1);
            if (n_keep) hipLaunchKernelGGL(gather_pieces_kernel, grid1(n_keep), dim3(WG), 0, stream(), ch_in.pieces.p, kidx.p, n_keep, kept.pieces.p,
                                           (uint32_t *)nullptr);
            HIP_CHECK(hipGetLastError());
            sync();
            kept.n_pieces = n_keep; kept.n_fp = ch_in.n_fp - n_fp_long;
            stat_add("align_pieces_deferred", (double)n_long);
            if (!n_keep) return;
            chp = &kept;
        }
    }
    const ChainOut &ch = *chp;
    const FixPt *fps_base = ch_in.fps.p;                      // (kept pieces still index the batch's fixed points)
    const size_t P = ch.n_pieces;
    size_t span_tasks = 64u << 20;               // (x 12 runs, x 4 once when a span overflows its pool: below 2^32)
    if (const char *e = hook("HLMI_ALIGN_SPAN_TASKS")) span_tasks = (size_t)std::max(64, atoi(e));      // test hook
    const size_t NT = ch.n_fp + P;
    if (NT <= span_tasks) {
        AlignOut ao;
        align_span(in, o, d_qlen, d_tlen, PieceSpan{{ch.pieces.p}, P, ch.n_fp, {fps_base}}, ao);
        if (ao.n_rows) outs.push_back(std::move(ao));
        return;
    }
    DBuf<uint32_t> tcnt(P), toff(P);
    hipLaunchKernelGGL(piece_task_count_kernel, dim3((unsigned)cdiv(P, (size_t)256)), dim3(256), 0, stream(), ch.pieces.p, P, tcnt.p);
    exclusive_scan_u32(tcnt.p, toff.p, P);
    const size_t n_spans = (NT + span_tasks - 1) / span_tasks;
    size_t p0 = 0, t0 = 0;                       // first piece / first task of the span
    for (size_t k = 1; k <= n_spans && p0 < P; ++k) {
        size_t p1 = P, t1 = NT;
        if (k < n_spans) {                       // first piece whose tasks start at or behind k / n_spans of all tasks
            const size_t want = NT / n_spans * k;
            size_t lo = p0 + 1, hi = P;
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if ((size_t)download_one(toff.p + mid) < want) lo = mid + 1; else hi = mid;
            }
            p1 = lo;
            t1 = p1 < P ? (size_t)download_one(toff.p + p1) : NT;
        }
        AlignOut ao;
        align_span(in, o, d_qlen, d_tlen, PieceSpan{{ch.pieces.p + p0}, p1 - p0, (t1 - t0) - (p1 - p0), {fps_base}}, ao);
        if (ao.n_rows) outs.push_back(std::move(ao));
        p0 = p1; t0 = t1;
    }
    stat_add("align_spans", (double)n_spans);
}
}  // namespace hlmi
