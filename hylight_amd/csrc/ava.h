// ava.h - the overlapper: sketch -> index -> seed -> chain -> align -> PAF rows.
// Specification: DESIGN.md "Overlapper spec" (restated sequentially in oracle/ava_oracle.c, which the
// kernels must match bit for bit).  Replaces the external minimap2 call of
// script/filter_overlap_slr2.py:51.
#pragma once
#include <vector>

#include "common.h"
#include "paf_io.h"

namespace hlmi {

// ---- spec constants (same values as oracle/ava_oracle.c) -------------------------------------
constexpr int CHAIN_PRED = 64;
constexpr int BLOCK_MIN = 32;
constexpr int BLOCK_MAX = 256;      // rows / columns up to which a block takes the 16 / 64-diagonal kernels; longer: LONG blocks
constexpr int BAND_W = 64;
constexpr int BAND_PAD = 12;
constexpr int NARROW_W = 16;        // blocks with |delta| <= NARROW_DELTA use a 16-diagonal band
constexpr int NARROW_PAD = 5;
constexpr int NARROW_DELTA = 5;
constexpr int EXT_MAX = 256;        // an end extension runs over up to max(EXT_MAX, max_gap) rows (ext_rows)
constexpr int HALF_W = 32;          // diagonals of a LONG extension and of a LONG block with |shift| <= HALF_DELTA
constexpr int HALF_DELTA = 7;
constexpr int ZDROP_STEP = 32;      // the z-drop test of an extension runs after rows 32, 64, ...
constexpr int SHIFT_MAX = BAND_W - 2 * BAND_PAD - 1;     // 39: largest diagonal shift of one block
constexpr int MAX_MID_OCC = 1000000;
constexpr int NEG_INF = -(1 << 29);

// rows an end extension may run over: minimap2 extends a chain end by up to max_gap bases (or to a z-drop)
inline int ext_rows(const hlmi_ava_opts &o) { return o.max_gap > EXT_MAX ? o.max_gap : EXT_MAX; }
// longest block / extension a chain can ask for (a link spans <= max_gap; fixed points are >= BLOCK_MIN apart)
inline int long_rows_cap(const hlmi_ava_opts &o) { return ext_rows(o) + BLOCK_MIN + SHIFT_MAX + BAND_W + 1; }

hlmi_ava_opts ava_opts_long();    // filter_overlap_slr2.py:51
hlmi_ava_opts ava_opts_short();   // filter_overlap_slr2.py:55

struct Mz {                 // one minimizer, 16 B (the unit all-gathered between GPUs)
    uint64_t x;             // hash << 8 | span
    uint64_t y;             // read << 32 | pos << 1 | strand
};

struct DevReads {           // a read set resident in HBM: 1 B/base codes (0..3 ACGT, 4 other)
    static constexpr size_t PAD = 64;   // bytes of code 4 in front of the first and behind the last base: 8-byte loads that
                                        // start a few bases outside the array need no bounds test
    size_t n = 0;
    uint64_t total = 0;
    DBuf<uint8_t> store;    // PAD + total + PAD bytes
    uint8_t *codes() const { return store.p + PAD; }
    void alloc_codes() {
        store.alloc(total + 2 * PAD);
        HIP_CHECK(hipMemsetAsync(store.p, 4, PAD, stream()));
        HIP_CHECK(hipMemsetAsync(store.p + PAD + total, 4, PAD, stream()));
    }
    DBuf<uint64_t> off;     // n+1 base offsets
    std::vector<uint64_t> h_off;
};
// reads [lo,hi) of s, in order
void upload_reads(const SeqSet &s, size_t lo, size_t hi, DevReads &out);
void upload_reads(const SeqSet &s, const std::vector<uint32_t> &ids, DevReads &out);
// device-to-device subset (reads already resident in HBM)
void subset_reads_device(const DevReads &all, const std::vector<uint32_t> &ids, DevReads &out);

struct DevSketch {
    size_t n = 0;
    DBuf<Mz> mz;            // read-major, position order; y carries rid = rid_base + local index
    DBuf<uint32_t> counts;  // per read
};
void sketch_device(const DevReads &r, int k, int w, int hpc, uint32_t rid_base, DevSketch &out);
// same, into caller-owned buffers (multi-GPU staging); returns the number of minimizers
int64_t sketch_device_into(const DevReads &r, int k, int w, int hpc, uint32_t rid_base, Mz *d_out, int64_t cap,
                           uint32_t *d_counts);

struct AvaInput {
    const DevReads *T = nullptr;           // targets of this run (local ids 0..nT), grouped by chunk
    const DevReads *Q = nullptr;           // all queries
    const uint32_t *d_rank_t = nullptr;    // strcmp rank of each target / query name (equal names, equal rank)
    const uint32_t *d_rank_q = nullptr;
    uint64_t n_ranks = 0;                  // every rank is below this (number of distinct names)
    const uint32_t *d_chunk_of_t = nullptr;  // chunk slot (0..n_chunks) of each local target
    uint32_t n_chunks = 1;
    const Mz *d_qmz = nullptr;             // complete query sketch
    std::vector<uint64_t> qmz_off;         // host: nQ+1 offsets into d_qmz
    // all-vs-all on one read set: local target t is query t_query[t], so its minimizers are already in d_qmz and
    // the target sketch is a gather instead of a second pass over the bases (empty: targets are sketched)
    std::vector<uint32_t> t_query;
    // more anchors than this in the run: ava_device stops after the counting pass (AvaRows::refused_anchors is set) so
    // that the caller can come back with fewer chunks (0: no limit; a run of a single chunk is never refused)
    uint64_t max_anchors = 0;
    // same for the run's output: after the first query batch the rows + CIGARs of the whole run are projected from that
    // batch's bytes per anchor (divergent read sets carry five times the CIGAR ops per anchor of clean ones)
    uint64_t max_out_bytes = 0;
    int max_lanes = 0;                 // query batches in flight (runtime.cpp: lanes); 0: the library's default (lane_count()), 1: one
                                       // after the other - the stage says so when the card has no room for a second batch's buffers
};
// HBM a run's output occupies until the caller has filtered it: CIGAR ops once (they stay in their batch buffers), the
// 64-byte records three times (per batch, concatenated, stream-ordered) plus their order keys
inline uint64_t ava_out_bytes(uint64_t rows, uint64_t ops) { return 4 * ops + 216 * rows; }

struct AvaRows {            // overlapper output in stream order (chunk, query, target, strand, chain, piece)
    size_t n_rows = 0, n_ops = 0;
    DBuf<PafRec> recs;
    // The CIGAR ops stay where the alignment passes of the query batches wrote them (one buffer per batch or span): a row's
    // cig_off counts 4-byte words from `ops_base`, the lowest of those buffers.  (Round 2 copied them into one array: 92 GB
    // per bench step on the full C4, and the rows' ops sat in HBM twice while that happened.)
    std::vector<DBuf<uint32_t>> ops_parts;
    std::vector<uint64_t> ops_part_len;      // ops in each part
    const uint32_t *ops_base = nullptr;
    std::vector<uint64_t> chunk_row_start;   // n_chunks+1
    uint64_t refused_anchors = 0;            // != 0: the run was given up, it would have had this many anchors ...
    double refused_shrink = 1.0;             // ... and fits when the targets shrink by this factor
};
void ava_device(const AvaInput &in, const hlmi_ava_opts &o, AvaRows &out);

// one target file vs one query file, PAF text out (the minimap2 call of filter_overlap_slr2.py:51)
void ava_files(const char *target_fa, const char *query_fa, const hlmi_ava_opts &o, const char *out_paf);

// strcmp ranks of the names of two sets over their union
void name_ranks(const std::vector<std::string> &a, const std::vector<std::string> &b, std::vector<uint32_t> &ra,
                std::vector<uint32_t> &rb, std::vector<std::string> &name_of_rank);

// PAF text of one overlapper row (12 columns + NM, tp, cg)
void format_ava_row(const PafRec &r, const uint32_t *ops, const std::string &qname, const std::string &tname,
                    std::string &out);

}  // namespace hlmi
