/* hylight_mi.h - C ABI of libhylight_mi.so: the MI355X-native overlap -> filter -> graph hot path
 * of HyLight.
 *
 * The reference (kangxiongbin/HyLight @ 2024_10_08) has no FFI: the path sits behind process
 * boundaries.  Each entry point below replaces one of those
 * boundaries and cites it; `INTEGRATION.md` shows the ctypes stub a maintainer would add on the
 * reference side.  Conventions:
 *   - plain C types only; strings are caller-owned NUL-terminated paths; outputs are files,
 *     so no memory ownership crosses the boundary (device pointers, where they appear, are
 *     caller-owned HIP allocations passed as void*);
 *   - every call returns 0 on success or a negative HLMI_E* code; the message is available
 *     from hlmi_last_error() (thread-local);
 *   - calls need a HIP device: the library has NO CPU fallback and fails with HLMI_ENODEV
 *     when none is usable.
 */
#ifndef HYLIGHT_MI_H
#define HYLIGHT_MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HLMI_OK        0
#define HLMI_EINVAL   -1   /* bad argument / malformed input file */
#define HLMI_EIO      -2   /* cannot open / read / write a file */
#define HLMI_ENODEV   -3   /* no usable HIP device (there is no CPU fallback) */
#define HLMI_EHIP     -4   /* a HIP runtime call or kernel failed */
#define HLMI_ENOMEM   -5
#define HLMI_ESTATE   -6   /* call sequence error (e.g. job step called out of order) */

/* ---- library state -------------------------------------------------------------------- */
/* Select the HIP device used by this process (one process per GPU).  device < 0 keeps the
 * current device.  host_threads bounds host-side helper threads (text formatting, sorting,
 * the output writer); <= 0: the CPUs this process may run on, 16 at most. */
/* ABI version: bumped whenever a public struct grows or an entry point changes (5: hlmi_ava_opts ends in zdrop).  A caller
 * compares hlmi_abi_version() with the HLMI_ABI_VERSION of the header it was built against BEFORE passing structs: a caller of
 * an older header would hand over a shorter hlmi_ava_opts than the library reads. */
#define HLMI_ABI_VERSION 5
int         hlmi_abi_version(void);
int         hlmi_init(int device, int host_threads);
void        hlmi_shutdown(void);
const char *hlmi_last_error(void);
const char *hlmi_version(void);

/* ---- B1: whole stage = utils.split_reads2 (script/utils.py:41-71) ----------------------- */
/* reads_fa = query reads (`-r`), ref_fa = targets that get --nsplit-chunked (`-c`); writes the
 * merged, score-sorted 14-column PAF (trailing TAB before newline, filter_overlap_slr2.py:151)
 * to out_paf.  `threads` is accepted for signature parity (the reference uses it for xargs -P
 * and minimap2 -t; here the GPU does the work).  rank/world shard the target chunks
 * (chunk i -> rank i % world); with world > 1 each rank writes `out_paf` for its own chunks and
 * hlmi_merge_scored_paf() combines them. */
int hlmi_split_reads2(const char *reads_fa, const char *ref_fa, int nsplit, const char *out_dir,
                      const char *out_paf, int threads, int len_over, int mc, double iden,
                      int long_mode);
int hlmi_split_reads2_shard(const char *reads_fa, const char *ref_fa, int nsplit, const char *out_dir,
                            const char *out_paf, int threads, int len_over, int mc, double iden,
                            int long_mode, int rank, int world);
/* merge of per-rank outputs = the final `sort -k12 -nr` of utils.py:69 */
int hlmi_merge_scored_paf(const char *const *in_pafs, int n_in, const char *out_paf);

/* ---- B2: one chunk of filter_overlap_slr2.main after the overlapper (slr2:57-152) ------- */
/* paf_in: overlapper output (PAF with cg:Z: as LAST field, slr2:312).  Runs the v4 window
 * filter (-len 30 -oh 3, slr2:51), the intermediate order (slr2:57), the SNP pile-up
 * (slr2:229-367), mutation_re (slr2:370-405) and pass 2 (slr2:77-152) on the GPU and writes
 * <chunk>_tmp_overlap4.paf-format rows (unsorted, in pass-2 order) to out_paf. */
int hlmi_filter_chunk(const char *paf_in, const char *out_overlap4_paf, int len_over, int mc,
                      double iden, double thre /*0.0025*/, int min_o /*4*/, int long_mode);

/* ---- a4 / a17: filter_trans_ovlp_inline_v4.py:31-85 and _v3.py:39-102 ------------------- */
/* variant 4: rows are copied through unchanged; variant 3: sfo != 0 writes SFO rows, else
 * "q t score" rows.  min_iden < 0 selects the script default (0.6 for v4, 0.8 for v3). */
int hlmi_paf_window_filter(int variant, int min_len, double min_iden, int min_o, int sfo,
                           const char *in_paf, const char *out_path);

/* ---- the short-read cluster path's filter and converter ----------------- */
/* filter_ovlp_inline.py <min_ovlp_len> <min_identity> <o> <r> (script/filter_ovlp_inline.py:12-106,
 * called at polyte.tune_params.py:507-511): 1000-row windows, internal-match test, longest overlap per
 * pair.  Runs on the GPU. */
int hlmi_filter_ovlp_inline(const char *in_paf, const char *out_paf, int min_ovlp_len, double min_identity,
                            int o, double r);
/* minimap22sfo.py --in --out -m <min_overlap_len> -p <min_pident> (script/minimap22sfo.py:28-75): PAF -> SFO with
 * the ids in string order.  Pure text conversion (host). */
int hlmi_minimap22sfo(const char *in_paf, const char *out_sfo, int min_overlap_len, double min_pident);

/* ---- the line-oriented text passes either side of the path (host I/O, no GPU work) ----
 * Same text conventions as the Python originals (universal newlines, str.strip()/split() white space).
 *
 * utils.filter_non_atcg(fq, out_dir, model) (script/utils.py:81-114): sequences upper-cased with every character
 * outside ATGCN replaced by N, headers cut at the first space; `is_fastq` = (model == "fastq").  The reference
 * derives the output path (<out_dir>/1.split_fastx/s1.fa); here the caller passes it. */
int hlmi_filter_non_atcg(const char *fastx, const char *out_fa, int is_fastq);
/* HyLight.gfa2fa(gfa, fa) (script/HyLight.py:328-337): S lines -> FASTA records.  An empty line is an error
 * (the reference raises IndexError on it). */
int hlmi_gfa2fa(const char *gfa, const char *out_fa);
/* HyLight.pick_up(ovlap, outdir, fq) (script/HyLight.py:347-378): the records of `fastx` whose name (text before
 * the first '/', without the leading '@' or '>') appears in neither column 1 nor column 6 of the PAF.  The
 * reference names the output <outdir>/sub<clock digits>_remain.fq; here the caller passes the path.  As in the
 * reference an existing file is removed first and no file is created when nothing is kept. */
int hlmi_pick_up(const char *ovlap_paf, const char *fastx, const char *out_fastx, int is_fastq);

/* ---- a3: the overlapper (replaces the external minimap2 call, slr2:51 / slr2:55) -------- */
typedef struct {
    int k;                 /* 19 (long, -Hk19) */
    int w;                 /* 5  (ava-pb)      */
    int hpc;               /* 1  (-H)          */
    int min_chain_score;   /* 100   (-m100)    */
    int max_gap;           /* 10000 (-g10000)  */
    int bandwidth;         /* chaining band, 2000 */
    int min_cnt;           /* 3 minimizers per chain (-n default) */
    int min_mid_occ;       /* 10 */
    double mid_occ_frac;   /* 2e-4; <= 0: the cut-off is min_mid_occ itself (-f INT) */
    int match, mismatch, gap_open, gap_ext, ambi;   /* 2 4 4 2 1 */
    int min_dp_score;      /* alignment pieces below this DP score are dropped: 80 (long), 60 (-s 60, short) */
    int end_bonus;         /* bonus for an end extension that reaches the query end: 0 (long), 100 (short) */
    int pair_once;         /* 1: report a pair only with strcmp(qname,tname) < 0 (ava-pb -X); 0: every non-self pair */
    int gap_open2, gap_ext2;   /* second piece of the gap cost: a gap of L bases costs min(gap_open + gap_ext L, gap_open2 +
                                  gap_ext2 L): 24, 1 (long: the preset's -O4,24 -E2,1), 32, 1 (short: --sr's -O12,32 -E2,1);
                                  gap_open2 <= 0: one piece.  The second piece must not be the cheaper one for gaps of
                                  fewer than 16 bases (the 16-diagonal kernels work with the first piece alone). */
    int stub_oh;           /* >= 0: the rows go to a consumer that drops internal matches with this overhang bound
                              (filter_trans_ovlp_inline_v4.py:52-64 with -oh 3, slr2:51,55).  An alignment piece that is
                              certain to be reported (block score >= min_dp_score + end_bonus) and certain to fail that test
                              whatever its end extensions find (an end more than X + stub_oh query and X + 64 + stub_oh target
                              bases inside both sequences, X = max(256, max_gap) = the reach of an end extension) is then
                              reported WITHOUT end extensions: the consumer only counts it as a line of its 1000-line windows.
                              < 0: every piece is extended (hlmi_ava's default; the stage entry points use 3). */
    int zdrop;             /* > 0: an end extension (up to max(256, max_gap) rows) stops at the first of its rows 32, 64, ...
                              whose best cell lies more than this below the best cell so far: 400 (long: minimap2's -z of the
                              ava-pb preset); 0: no z-drop (short: extensions stay within 256 rows) */
} hlmi_ava_opts;
void hlmi_ava_opts_long(hlmi_ava_opts *o);    /* the constants of slr2:51 (ava-pb -Hk19 -m100 -g10000) */
void hlmi_ava_opts_short(hlmi_ava_opts *o);   /* the constants of slr2:55 (--sr -k21 -w11 -s60 -m30 -n2 -A4 -B2 --end-bonus=100) */
/* target_fa = one chunk, query_fa = all reads; writes minimap2-style PAF rows (12 columns +
 * NM, tp, cg:Z: tags, cg last) in query-file order. */
int hlmi_ava(const char *target_fa, const char *query_fa, const hlmi_ava_opts *opts,
             const char *out_paf);

/* ---- B3: miniasm (tools/miniasm/main.c:32-211) with the flags HyLight passes ------------- */
/* HyLight.py:137,140,171:  miniasm -d <bub_dist> -n <n_rounds_arg> -e <max_ext> -c <min_dp>
 * -f <reads_fa> <paf> > <out_gfa>.  reads_fa may be NULL (S lines then carry '*').
 * outfmt: "ug" (GFA, default when NULL), "sg", "paf", "bed" (main.c:146-150). */
int hlmi_miniasm(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg,
                 int max_ext, int min_dp, const char *outfmt, const char *out_path);

/* ---- a18: sfo2overlaps.py (--num_pairs 0 branch, HyLight.py:315-318) --------------------- */
int hlmi_sfo2overlaps(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs);

/* ---- f3 (STARTED): front end of the SAVAGE overlap-graph assembler ------ */
/* tools/HaploConduct/src (ViralQuasispecies) needs Boost and cannot be built here: these two entry points restate its
 * text of record and are checked against oracle/vq.py only (parity unpinned).  Not built: the quality-aware overlap score
 * (EdgeCalculator.cpp:26-139, log / pow / exp thresholds), edge orientation (Edge.h), everything after the graph. */
typedef struct {
    uint64_t id1, id2;                 /* strtoul(..., 0) of columns 1, 2                        (Overlap.h:39-40, Types.h:99)  */
    uint32_t pos1, pos2, perc1, perc2, len1, len2;   /* atoi; pos2 = perc2 = len2 = 0 when column 4 is "-" (Overlap.h:53-57) */
    char ord, ori1, ori2, type1, type2; /* '1' '2' '-';  '+' '-';  's' 'p'                                                  */
    char pad[3];
} hlmi_vq_overlap;
/* EdgeCalculator.cpp:561-666 construct_edges, the part in front of process_overlaps: lines are trimmed of outer tabs and
 * blanks and split at tabs; a line without 13 fields is skipped ("incorrect overlap"), as is id1 == id2; an overlap is an
 * edge candidate when (s,s: len1 >= min_len) or (a 'p' type: len1, len2 >= min_len / 2, or with relax_pe len1 + len2 >=
 * min_len) and perc >= min_perc, where perc = (perc1 + perc2) / 2 truncated when perc2 > 0, else perc1 (Overlap.h:196).
 * Reads at most max_overlaps lines.  Candidates go to out[0 .. min(*n_out, cap)) in file order; *n_nonedge counts the rows
 * the reference writes back to nonedge_overlaps.txt, *n_skipped the other ones.  out may be NULL (cap 0) to count. */
int hlmi_vq_parse_overlaps(const char *savage_path, uint32_t min_overlap_len, uint32_t min_overlap_perc, int relax_pe,
                           uint64_t max_overlaps, hlmi_vq_overlap *out, uint64_t cap, uint64_t *n_out,
                           uint64_t *n_nonedge, uint64_t *n_skipped);
/* GraphAlgos.cpp:746-795 (findTransEdges, nonemptyIntersect) and :938-993 (removeTransitiveEdges up to the deletion):
 * edge k = src[k] -> dst[k] of a directed graph on n_vertices vertices.  An edge u -> v is transitive when some w has
 * u -> w and w -> v.  remove_trans = 1: flags bit 0 marks the transitive edges; 2 / 3: the search is repeated on the graph
 * of the edges found so far ("double" / "triple" transitive), bit 0 marks the last round's set - the edges the reference
 * then removes.  With remove_trans == 1 and ovlen != NULL (branch_reduction > 0, :970-993) bit 1 marks the edges
 * scheduled for deletion: for every transitive edge u -> v of overlap length L, every out-edge of u and every in-edge of v
 * whose length is <= L.  GPU: one wave per vertex, its out-neighbours in an LDS hash table, the in-lists streamed by the
 * lanes. */
int hlmi_vq_transitive_edges(uint32_t n_vertices, uint64_t n_edges, const uint32_t *src, const uint32_t *dst,
                             const uint32_t *ovlen, int remove_trans, uint8_t *flags, uint64_t *n_transitive);
/* EdgeCalculator::overlap_score (EdgeCalculator.cpp:26-139) for the single-single overlaps of an overlaps file
 * (compute_overlap's "s"-"s" branch, :186-222 - the only one HyLight reaches: --num_pairs 0): for overlap k of `ov`
 * (as returned by hlmi_vq_parse_overlaps) score[k] = exp(mean log-probability that the overlapping bases of the two reads
 * of fastq_singles come from one sequence, given their phred qualities), 0 when a position's probability falls below
 * `mismatch`, a read is shorter than `min_read_len` or pos1 lies behind read 1; mismatch_rate[k] as the reference sets it
 * (1.0 where it returns early); pos3[k] = len1 - pos1 - len2 (Edge::set_extra_pos).  The per-position sums are taken in
 * sequence order in double precision, log / pow / exp are the host's libm (tables by quality pair): the same arithmetic as
 * the reference's.  Reads are looked up by the integer id of their "@<id>" line (strtoul base 0, FastqStorage.cpp:109-117). */
int hlmi_vq_overlap_scores(const char *fastq_singles, const hlmi_vq_overlap *ov, uint64_t n, double mismatch,
                           uint32_t min_read_len, double *score, double *mismatch_rate, int64_t *pos3);

/* ---- staged multi-GPU job: sketch shard -> (RCCL all-gather by the caller) -> run -------- */
/* One process per GPU.  Every rank opens the same files, sketches its slice of the query
 * reads into a caller-owned device buffer (16 B per minimizer: two uint64), the caller
 * all-gathers the buffers over RCCL (torch.distributed), hands the gathered sketch back and
 * runs its share of the target chunks. */
typedef struct hlmi_job hlmi_job;
hlmi_job *hlmi_job_open(const char *reads_fa, const char *ref_fa, int nsplit, int long_mode);
void      hlmi_job_close(hlmi_job *j);
int64_t   hlmi_job_num_queries(const hlmi_job *j);
int64_t   hlmi_job_num_chunks(const hlmi_job *j);
/* upper bound of minimizers for query reads [lo,hi): capacity needed by hlmi_job_sketch */
int64_t   hlmi_job_sketch_bound(const hlmi_job *j, int64_t lo, int64_t hi);
/* sketch query reads [lo,hi) on the GPU into dev_mz (capacity cap entries of 16 B), sorted by
 * (read, position); per-read counts go to dev_counts[hi-lo] (uint32).  *n_out = entries. */
int       hlmi_job_sketch(hlmi_job *j, int64_t lo, int64_t hi, void *dev_mz, int64_t cap,
                          void *dev_counts, int64_t *n_out);
/* install the complete query sketch (all reads, read-major): dev_mz[n] + dev_counts[nq] */
/* single-GPU form of the two calls around it: sketches ALL query reads into buffers of the job's own, sized exactly
 * (hlmi_job_sketch needs room for the bound - one entry per base - which is 80 GB for 5 Gbases of reads), and installs them */
int       hlmi_job_sketch_own(hlmi_job *j);
int       hlmi_job_set_query_sketch(hlmi_job *j, const void *dev_mz, int64_t n, const void *dev_counts);
/* overlap + filter the chunks {c : c % world == rank}; writes the rank's score-sorted PAF */
int       hlmi_job_run(hlmi_job *j, int rank, int world, int len_over, int mc, double iden,
                       const char *out_paf);

/* ---- measurement hooks (bench.py) -------------------------------------------------------- */
/* Counters of the last stage run in this process: name -> value, written as JSON to buf. */
int hlmi_last_stats_json(char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* HYLIGHT_MI_H */
