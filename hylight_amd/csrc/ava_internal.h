// ava_internal.h - structures shared by the overlapper's translation units.
#pragma once
#include "ava.h"

namespace hlmi {

// ---- S2: index over the targets of this run (all chunks together, per-chunk semantics kept) ----
struct DevIndex {
    size_t n = 0;
    DBuf<uint64_t> key;      // minimizer hash (x >> 8), ascending; equal keys ordered by (rank word, target position)
    DBuf<uint64_t> y;        // target << 32 | pos << 1 | strand
    DBuf<uint32_t> occ;      // occurrences of this key inside the entry's chunk (order of the first sort; build only)
    DBuf<uint32_t> mid_occ;  // per chunk: occurrence cut-off
    DBuf<uint32_t> rk;       // rank word: 0 = too frequent in its chunk, else name rank of the entry's target + 1;
                             // ascending inside a key, so the partners of a query are a suffix of the key's run
    DBuf<uint64_t> ck;       // key << rank_bits | rank word: ONE ascending word per entry for the counting searches
    int rank_bits = 0;       // (0: key and rank word do not fit 64 bits together, the searches compare both arrays)
    DBuf<uint32_t> bucket;   // first entry of every value of the top bucket_bits key bits (2^bits + 1 offsets)
    int bucket_shift = 0, bucket_bits = 0;
    int pair_once = 1;       // seeding rule carried with the index (hlmi_ava_opts::pair_once)
    // seed_group.hip: the entries once more as 32-bit words dense target << y32_sh | position << 1 | strand (dense = place of
    // the target in the order (name rank, target), the order of a hash's entries); y32_sh = 0: not built
    DBuf<uint32_t> y32, t_of_dense;
    int y32_sh = 0;
    std::vector<uint32_t> rank_of_dense;   // host: name ranks of the targets in dense order (ascending)
};
void build_index(const DevSketch &tsk, const uint32_t *d_chunk_of_t, const uint32_t *d_rank_t, uint32_t n_chunks,
                 uint64_t n_names, const hlmi_ava_opts &o, DevIndex &ix);      // n_names: bound of the name ranks

// ---- S3/S4 output: alignment pieces = lists of fixed points -----------------------------------
struct Piece {               // 32 B
    uint32_t q, t;           // global query index, local target index
    uint32_t strand;
    uint32_t chain;          // start anchor of the chain inside its (q,t,strand) group
    uint32_t piece;          // index of the piece inside the chain
    uint32_t fp_off, n_fp;   // fixed points: (q,t) exclusive-end coordinates, n_fp >= 2
    uint32_t pad;
};
struct FixPt { uint32_t q, t; };

struct ChainOut {
    size_t n_pieces = 0, n_fp = 0;
    DBuf<Piece> pieces;
    DBuf<FixPt> fps;
};
struct SeedStats { uint64_t anchors = 0, groups = 0; };
// one counting pass over ALL query minimizers: anchors per minimizer (device) and per query (host)
struct SeedPlan {
    DBuf<uint32_t> cnt;                 // per query minimizer
    DBuf<uint32_t> lo, len;             // its occurrence run in the index (found once, by the counting pass)
    std::vector<uint64_t> per_query;    // per query read
};
void plan_seeds(const AvaInput &in, const DevIndex &ix, SeedPlan &plan);
// Anchors of a query batch grouped by (query, target, strand) without a device-wide sort (seed_group.hip): one 64-bit word per
// anchor (tpos << vb | qpos << 8 | span), every group contiguous and in generation order, one record per group of at least
// min_cnt anchors.
struct GroupedAnchors {
    DBuf<uint64_t> key;
    DBuf<uint32_t> gstart, gsize, gq, gts;   // first anchor, anchors, query inside the batch, target << 1 | strand
    size_t G = 0, G_all = 0;                 // records; groups of any size (statistics)
};
void seed_group_prepare(const AvaInput &in, DevIndex &ix);       // once per index
// can the batch take that path (widths)?
bool seed_group_supported(const DevIndex &ix, uint64_t anchors, int pb, int qpb);
// false: the kernel met a group its counters do not hold - the caller sorts the batch instead
bool seed_group(const AvaInput &in, const DevIndex &ix, const SeedPlan &plan, const uint32_t *d_qlen, size_t q_lo, size_t q_hi,
                int vb, int min_cnt, size_t A, GroupedAnchors &out);
// seeds + chains queries [q_lo,q_hi)
void seed_and_chain(const AvaInput &in, const DevIndex &ix, const hlmi_ava_opts &o, const SeedPlan &plan,
                    const uint32_t *d_qlen, const uint32_t *d_tlen, size_t q_lo, size_t q_hi, ChainOut &out,
                    SeedStats &st);

// ---- S5: alignment of the pieces -> PAF rows -------------------------------------------------------
struct AlignOut {
    size_t n_rows = 0, n_ops = 0;
    DBuf<PafRec> recs;       // qid/tid = name ranks; chunk = chunk slot; tie unset
    DBuf<uint32_t> ops;
    DBuf<uint64_t> ord_hi, ord_lo;   // stream-order keys of each row
};
// Pieces with LONG alignment tasks (blocks of more than BLOCK_MAX rows or columns, extensions of more than EXT_MAX rows), set
// aside batch after batch: they are aligned together at the end of the run (ava_align.hip: piece_long_flag_kernel).  The
// fp_off of a set-aside piece counts from the start of the concatenated fixed points of all parts.
struct DeferredPieces {
    // fp_base: where the part's pieces expect their fixed points (Piece::fp_off counts from the start of the set the part was
    // added to: a batch's own set; ava_device moves the offsets when it puts the batches' sets together)
    struct Part { size_t n_pieces = 0, n_fp = 0, fp_base = 0; DBuf<Piece> pieces; DBuf<FixPt> fps; };
    std::vector<Part> parts;
    size_t n_pieces = 0, n_fp = 0;
};
// appends the rows of the batch's pieces (one AlignOut, or one per span of pieces when the batch has too many tasks);
// defer != nullptr: the pieces with LONG tasks go there instead (when they are a small share of the batch)
void align_pieces(const AvaInput &in, const hlmi_ava_opts &o, const uint32_t *d_qlen, const uint32_t *d_tlen,
                  const ChainOut &ch, std::vector<AlignOut> &outs, DeferredPieces *defer = nullptr);

}  // namespace hlmi
