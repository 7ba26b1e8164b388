// graph.h - overlap-graph build (what `miniasm -d D -n N -e E -c C -f reads in.paf`
// computes for HyLight, script/HyLight.py:137,140,171) and sfo2overlaps (a18).
//
//   device (graph_dev.hip):  a9  PAF text -> overlap records, read ids in order of first appearance
//                            a10 per-read coverage window        a11 clipping, crude arc filter
//                            a12 containment, read renumbering   a13 overlap -> arc, string graph
//                            a14 transitive reduction, duplicate and unpaired arcs
//   host (graph_host.cpp):   the two order-defining sorts (tie order of the reference's in-place sort), a15 the
//                            order-dependent cleaners, a16 unitigs, sequences and the GFA text
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace hlmi {

// option set of tools/miniasm/common.c:5-23 after main.c's argument handling
struct GraphOpt {
    int min_span = 2000, min_match = 100, min_dp = 3;
    float min_iden = .05f;
    int max_hang = 1000, min_ovlp = 2000;
    float int_frac = .8f;
    int gap_fuzz = 1000, n_rounds = 2, bub_dist = 50000, max_ext = 4;
    float min_drop = .5f, max_drop = .7f, final_drop = .8f;
};

// one overlap as seen from its query read (the reference's 32-byte ma_hit_t carries the same fields,
// tools/miniasm/miniasm.h:29-34)
struct Ovl {
    uint32_t q, qs, qe;       // read id, interval on it
    uint32_t t, ts, te;
    uint32_t ml_rev;          // matches | reverse strand << 31
    uint32_t bl;              // alignment length (31 bits)
};
static_assert(sizeof(Ovl) == 32, "Ovl is 32 bytes");

struct ReadWin {              // kept sub-region of a read (ma_sub_t: s is a 31-bit field there)
    uint32_t s, e, del, pad;
};

// string-graph arc: 16 bytes, `ul` = source vertex << 32 | distance to the next vertex (tools/miniasm/asg.h:7-11)
struct Arc {
    uint64_t ul;
    uint32_t v;               // target vertex
    uint32_t ol_del;          // overlap length | deleted << 31
    uint32_t src() const { return (uint32_t)(ul >> 32); }
    uint32_t len() const { return (uint32_t)ul; }
    uint32_t ol() const { return ol_del & 0x7fffffffu; }
    bool del() const { return ol_del >> 31; }
    void set_del(bool d) { ol_del = (ol_del & 0x7fffffffu) | (d ? 0x80000000u : 0u); }
};
static_assert(sizeof(Arc) == 16, "Arc is 16 bytes");

struct NameRef { uint64_t off; uint32_t len; };     // a read name as a slice of the PAF text

// everything the host half needs, as left by the device half
struct GraphState {
    std::string paf;                    // the file (names are slices of it)
    std::vector<NameRef> name;          // per read (ids after renumbering)
    std::vector<ReadWin> win;           // per read
    std::vector<Ovl> ovl;               // surviving overlaps (filled for the "paf" dump only)
    std::vector<uint32_t> seq_len;      // string graph: per read length | deleted << 31
    std::vector<Arc> arc;               // sorted by ul, after a14
    bool have_graph = false;
    bool symmetric = false;             // asg_symm has run (it does after a reduction that removed arcs, asg.c:187-190)
    std::string read_name(uint32_t id) const { return paf.substr(name[id].off, name[id].len); }
};

// a9-a14 on the GPU.  until: "bed" / "paf" stop after a12, anything else builds and reduces the graph.
void graph_device(const char *paf_path, const GraphOpt &o, const std::string &until, GraphState &out);

// permutation that the reference's in-place radix sort (ksort.h:132-184) applies to records with these keys:
// perm[i] = input index of the record that ends up at position i.  Defines the order among equal keys.
void reference_sort_order(const std::vector<uint64_t> &keys, std::vector<uint32_t> &perm);

void miniasm_run(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg, int max_ext, int min_dp,
                 const char *outfmt, const char *out_path);
void sfo2overlaps_run(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs);

// vq_front.hip: SURVEY 8f rank 3 (started)
void vq_parse_overlaps(const char *path, uint32_t min_len, uint32_t min_perc, int relax_pe, uint64_t max_overlaps,
                       hlmi_vq_overlap *out, uint64_t cap, uint64_t *n_out, uint64_t *n_nonedge, uint64_t *n_skipped);
void vq_overlap_scores(const char *fastq, const hlmi_vq_overlap *ov, uint64_t n, double mismatch, uint32_t min_read_len,
                       double *score, double *mismatch_rate, int64_t *pos3);
void vq_transitive_edges(uint32_t n_vertices, uint64_t n_edges, const uint32_t *src, const uint32_t *dst, const uint32_t *ovlen,
                         int remove_trans, uint8_t *flags, uint64_t *n_transitive);

}  // namespace hlmi
