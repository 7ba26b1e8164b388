// stage.cpp - one split_reads2 stage (script/utils.py:41-71) on the GPU: --nsplit chunking of the
// target file, overlapper, filter chain, score-sorted 14-column PAF.  Also the staged entry
// points used by the multi-GPU driver (sketch shard / install gathered sketch / run chunk share).
#include "stage.h"

#include <memory>
#include <string_view>
#include <thread>

#include <algorithm>
#include <chrono>

#include "ava.h"
#include "filter_stage.h"
#include "paf_io.h"
#include "row_text.h"

namespace hlmi {

struct Job::Impl {
    SeqSet Q, T_own;
    SeqSet &T;                        // the target set: Q itself when both paths name the same file
    explicit Impl(bool same) : T(same ? Q : T_own) {}
    size_t n_bases_q = 0;
    bool long_mode = true;
    hlmi_ava_opts opts{};
    std::vector<std::pair<uint32_t, uint32_t>> chunks;   // target read ranges [lo,hi) per chunk
    std::vector<uint32_t> rank_q, rank_t;
    std::vector<std::string> name_of_rank;
    DevReads dQ, dT_own;              // reads resident in HBM from job_open on
    DevReads *dT = nullptr;           // all targets (aliases dQ when both paths name the same file)
    DBuf<uint32_t> d_rank_q;
    DevSketch own;                    // sketch owned by the job (single-GPU path)
    const Mz *d_qmz = nullptr;        // installed query sketch (own.mz or caller memory)
    std::vector<uint64_t> qmz_off;
    // anchors / output bytes per target base seen by the last pass over this job: the next pass sizes its first sub-run
    // from them instead of probing with 128 Mbases (and being refused on deep read sets)
    double est_anchors_per_base = 0, est_out_per_base = 0;
    // text buffers of the last pass, kept for the next one (capacity only): a short-read call formats ~400 MB of rows per
    // pass, and fresh memory costs a page fault per 4 KB under a lock all formatting threads share
    std::vector<std::unique_ptr<std::string>> text_cache;
    DevNames d_names;                 // the reads' names in HBM (uploaded when the first pass formats its rows on the device)
};

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Job::Job(const char *reads_fa, const char *ref_fa, int nsplit, bool long_mode)
    : impl_(new Impl(std::string(reads_fa) == ref_fa)) {
    Impl &m = *impl_;
    m.long_mode = long_mode;
    m.opts = long_mode ? ava_opts_long() : ava_opts_short();   // filter_overlap_slr2.py:51 / :55
    // both command lines pipe the rows into `filter_trans_ovlp_inline_v4.py -len 30 -oh 3`: pieces that are certain to
    // fail its overhang test are reported without end extensions (hlmi_ava_opts::stub_oh; HLMI_NO_STUB: test hook)
    m.opts.stub_oh = hook("HLMI_NO_STUB") ? -1 : FilterCfg().v4_min_o;
    read_seqs(reads_fa, m.Q);
    const bool same = std::string(reads_fa) == ref_fa;
    if (!same) read_seqs(ref_fa, m.T);
    m.n_bases_q = m.Q.bases.size();
    name_ranks(m.T.names, m.Q.names, m.rank_t, m.rank_q, m.name_of_rank);
    // --nsplit chunking by LINES (utils.py:44-47): nu = `wc -l`, per = int(nu/(8*nsplit)+1)*8
    const uint64_t nu = m.T.n_lines;
    const uint64_t per = (nu / (8ull * (uint64_t)nsplit) + 1) * 8;
    for (size_t r = 0; r < m.T.size();) {
        const uint64_t c = m.T.first_line[r] / per;
        size_t e = r;
        while (e < m.T.size() && m.T.first_line[e] / per == c) ++e;
        // a record must not straddle a chunk boundary (`split -l` would cut it and the reference would
        // silently work on the truncated pieces)
        const uint64_t end_line = e < m.T.size() ? m.T.first_line[e] : std::max<uint64_t>(nu, m.T.first_line[e - 1] + 1);
        if ((end_line - 1) / per != c)
            fail(HLMI_EINVAL, "record %s straddles an --nsplit chunk boundary", m.T.names[e - 1].c_str());
        m.chunks.emplace_back((uint32_t)r, (uint32_t)e);
        r = e;
    }
    upload_reads(m.Q, 0, m.Q.size(), m.dQ);
    if (same) m.dT = &m.dQ;
    else { upload_reads(m.T, 0, m.T.size(), m.dT_own); m.dT = &m.dT_own; }
    // the bases live in HBM from here on: the host copies (a gigabyte at 100 k reads) go
    std::string().swap(m.Q.bases);
    std::string().swap(m.T.bases);
    m.d_rank_q.upload(m.rank_q);
    if (m.rank_q.empty()) m.d_rank_q.alloc(1);
}

Job::~Job() = default;
size_t Job::num_queries() const { return impl_->Q.size(); }
size_t Job::num_chunks() const { return impl_->chunks.size(); }

int64_t Job::sketch_bound(int64_t lo, int64_t hi) const {
    const Impl &m = *impl_;
    if (lo < 0 || hi < lo || (size_t)hi > m.Q.size()) fail(HLMI_EINVAL, "sketch range out of bounds");
    return (int64_t)(m.Q.off[hi] - m.Q.off[lo]);
}

int64_t Job::sketch_range(int64_t lo, int64_t hi, void *dev_mz, int64_t cap, void *dev_counts) {
    Impl &m = *impl_;
    if (lo < 0 || hi < lo || (size_t)hi > m.Q.size()) fail(HLMI_EINVAL, "sketch range out of bounds");
    if (lo == hi) return 0;
    // The sketch kernels index bases with 32 bits: a range of more than SKETCH_PART bases (BASELINE configs[3]: a million
    // long reads are 10 Gbases) is sketched in parts of consecutive reads, each appended to the output.
    uint64_t part_bases = 3ull << 30;
    if (const char *e = hook("HLMI_SKETCH_PART_MBASES")) part_bases = (uint64_t)std::max(1, atoi(e)) << 20;   // test hook
    const uint64_t bases = m.Q.off[hi] - m.Q.off[lo];
    if (lo == 0 && (size_t)hi == m.Q.size() && bases <= part_bases)     // the reads are already resident in HBM
        return sketch_device_into(m.dQ, m.opts.k, m.opts.w, m.opts.hpc, 0, (Mz *)dev_mz, cap, (uint32_t *)dev_counts);
    int64_t total = 0;
    for (int64_t a = lo; a < hi;) {
        int64_t b = a + 1;
        while (b < hi && m.Q.off[b + 1] - m.Q.off[a] <= part_bases) ++b;
        DevReads part;
        std::vector<uint32_t> ids((size_t)(b - a));
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)(a + (int64_t)i);
        subset_reads_device(m.dQ, ids, part);
        total += sketch_device_into(part, m.opts.k, m.opts.w, m.opts.hpc, (uint32_t)a, (Mz *)dev_mz + total, cap - total,
                                    (uint32_t *)dev_counts + (a - lo));
        a = b;
    }
    return total;
}

void Job::set_query_sketch(const void *dev_mz, int64_t n, const void *dev_counts) {
    Impl &m = *impl_;
    std::vector<uint32_t> cnt(m.Q.size());
    if (!cnt.empty()) {
        HIP_CHECK(hipMemcpyAsync(cnt.data(), dev_counts, cnt.size() * 4, hipMemcpyDeviceToHost, stream()));
        sync();
    }
    m.qmz_off.assign(m.Q.size() + 1, 0);
    for (size_t i = 0; i < cnt.size(); ++i) m.qmz_off[i + 1] = m.qmz_off[i] + cnt[i];
    if ((int64_t)m.qmz_off.back() != n) fail(HLMI_EINVAL, "query sketch has %lld entries but counts sum to %llu", (long long)n,
                                             (unsigned long long)m.qmz_off.back());
    m.d_qmz = (const Mz *)dev_mz;
}

void Job::sketch_all_queries() {
    Impl &m = *impl_;
    uint64_t part_bases = 3ull << 30;
    if (const char *e = hook("HLMI_SKETCH_PART_MBASES")) part_bases = (uint64_t)std::max(1, atoi(e)) << 20;
    if (m.Q.off.back() <= part_bases) {
        sketch_device(m.dQ, m.opts.k, m.opts.w, m.opts.hpc, 0, m.own);
    } else {                                     // in parts of consecutive reads, each sketched to its exact size, then joined
        const size_t nq = m.Q.size();
        std::vector<DevSketch> parts;
        std::vector<size_t> first;
        size_t total = 0;
        for (size_t a = 0; a < nq;) {
            size_t b = a + 1;
            while (b < nq && m.Q.off[b + 1] - m.Q.off[a] <= part_bases) ++b;
            DevReads part;
            std::vector<uint32_t> ids(b - a);
            for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)(a + i);
            subset_reads_device(m.dQ, ids, part);
            parts.emplace_back();
            sketch_device(part, m.opts.k, m.opts.w, m.opts.hpc, (uint32_t)a, parts.back());
            first.push_back(a);
            total += parts.back().n;
            a = b;
        }
        m.own.mz.alloc(total ? total : 1);
        m.own.counts.alloc(nq ? nq : 1);
        size_t at = 0;
        for (size_t p = 0; p < parts.size(); ++p) {
            const size_t n_reads = (p + 1 < parts.size() ? first[p + 1] : nq) - first[p];
            if (parts[p].n) HIP_CHECK(hipMemcpyAsync(m.own.mz.p + at, parts[p].mz.p, parts[p].n * sizeof(Mz), hipMemcpyDeviceToDevice, stream()));
            HIP_CHECK(hipMemcpyAsync(m.own.counts.p + first[p], parts[p].counts.p, n_reads * 4, hipMemcpyDeviceToDevice, stream()));
            at += parts[p].n;
        }
        sync();
        m.own.n = total;
    }
    set_query_sketch(m.own.mz.p, (int64_t)m.own.n, m.own.counts.p);
}

void Job::run(int rank, int world, int len_over, int mc, double iden, const char *out_paf) {
    Impl &m = *impl_;
    if (!m.d_qmz && m.Q.size()) fail(HLMI_ESTATE, "hlmi_job_run before a query sketch was installed");
    ktimer_discard();            // timers of earlier calls (sketching) do not belong to this pass
    stat_reset();
    (void)dev_peak_bytes(true);  // the high-water mark of this pass starts from what is resident now (reads, sketch)
    const double t0 = now_s();
    struct ExitStamp { double t0; ~ExitStamp() { stat_set("t_with_cleanup_s", now_s() - t0); } } exit_stamp{t0};   // runs after the locals are gone
    // ---- this rank's chunks, in sub-runs ----------------------------------------------------------------------
    // The chunks are independent (the reference runs one worker per chunk), so the rank's share is processed in
    // sub-runs of consecutive chunks: every structure of a sub-run (index, anchors, candidate rows with their CIGARs)
    // lives in HBM only while the sub-run is worked on, and the hard limits of one overlapper run (2^21 targets, 2^32
    // index entries) are never met whatever the size of the share.  The first sub-run is small; its anchors per target
    // base set the size of the following ones (~SUBRUN_ANCHORS anchors: rows + CIGARs of a sub-run are ~2-4 B/anchor).
    std::vector<uint32_t> my_chunks;
    for (size_t c = 0; c < m.chunks.size(); ++c)
        if ((int)(c % (size_t)world) == rank) my_chunks.push_back((uint32_t)c);
    constexpr double SUBRUN_ANCHORS = 2.0e10;    // (every sub-run counts the seeds of ALL queries again: fewer, larger ones)
    // The rows of a sub-run sit in HBM twice while they are put into stream order (their CIGARs stay where the batches
    // wrote them), beside ~45 GB of batch buffers, and a sub-run is only refused at 1.5 x its budget: with the reads, the
    // query sketch and the seed plan of a million long reads resident (84 GB on the full C4) the usual 64 GB do not fit
    // and the pool thrashes - the budget follows what is free when the pass starts
    double SUBRUN_OUT_BYTES = 52e9;
    bool lanes_fit = false;
    if (const size_t avail = dev_available_bytes()) {
        const double plan = 12.0 * (m.qmz_off.empty() ? 0.0 : (double)m.qmz_off.back());        // count + run of every query minimizer
        // (round 4: whole-overlap CIGARs of divergent reads - a complete C5 pass peaked at 275 of the 288 GB with 64e9 and a
        //  divisor of 1.3 x 1.5)
        SUBRUN_OUT_BYTES = std::min(52e9, std::max(8e9, ((double)avail - plan - 60e9) / (1.5 * 1.5)));
        // a second query batch in flight (lanes, runtime.cpp) brings its own batch buffers - ~45 GB, and as much again in LONG
        // scratch on divergent reads: the full C5 (265e9 left here of the card's 309e9 bytes) peaked at 298e9 with two lanes
        // against ~240e9 with one.  Lanes only where nearly the whole card is free: C2, C3, the short-read calls; not the full
        // C4 and C5, whose resident sketches and plans take 40-100 GB.
        lanes_fit = (double)avail - plan >= 285e9;
        stat_set("lanes_fit", lanes_fit ? 1 : 0);
    }
    stat_set("subrun_out_budget_gb", SUBRUN_OUT_BYTES / 1e9);
    constexpr uint64_t SUBRUN_MAX_TARGETS = 1u << 20, SUBRUN_MAX_BASES = 3ull << 30;
    double subrun_anchors_max = 4.0e10, subrun_out_max = 1.5 * SUBRUN_OUT_BYTES;
    // test hooks: the refusal / retry path below with small inputs (millions of anchors / MB of output)
    if (const char *e = hook("HLMI_SUBRUN_MAX_MANCHORS")) subrun_anchors_max = 1e6 * std::max(1e-3, atof(e));
    if (const char *e = hook("HLMI_SUBRUN_MAX_OUT_MB")) subrun_out_max = 1e6 * std::max(1e-3, atof(e));
    const double subrun_anchors = std::min(SUBRUN_ANCHORS, subrun_anchors_max / 2), subrun_out = std::min(SUBRUN_OUT_BYTES, subrun_out_max / 1.5);
    uint64_t budget_bases = 128ull << 20;
    if (const char *e = hook("HLMI_SUBRUN_MBASES")) budget_bases = (uint64_t)std::max(1, atoi(e)) << 20;   // test hook
    const bool fixed_budget = hook("HLMI_SUBRUN_MBASES") != nullptr;
    if (!fixed_budget && m.est_anchors_per_base > 0) {
        const double by_anchors = subrun_anchors / m.est_anchors_per_base;
        const double by_bytes = m.est_out_per_base > 0 ? subrun_out / m.est_out_per_base : by_anchors;
        budget_bases = std::min<uint64_t>(SUBRUN_MAX_BASES, std::max<uint64_t>(4ull << 20, (uint64_t)(0.9 * std::min(by_anchors, by_bytes))));
    }
    // the rows' text: every formatting thread appends to a buffer of its own, `lines` are views into those buffers
    std::vector<std::string_view> lines;
    std::vector<uint32_t> line_keys;             // column 12 of every line as an integer (format_scored_row)
    std::vector<std::unique_ptr<std::string>> text;
    std::vector<uint32_t> tids;
    double t_ava = 0, t_flt = 0, t_fmt = 0, t_dl = 0;
    size_t n_v4 = 0, n_ev = 0, n_pairs = 0, n_subruns = 0;
    uint64_t done_bases = 0;
    double done_anchors = 0, done_out_bytes = 0;
    for (size_t ci = 0; ci < my_chunks.size() && m.Q.size();) {
        std::vector<uint32_t> sub_tids, chunk_of_t;
        uint32_t n_my = 0;
        // The budget counts WORK bases: with every pair taken once (strcmp(qname, tname) < 0, the all-vs-all calls) a
        // target only meets the queries that rank below it, so a chunk's anchors grow with the name ranks of its reads -
        // chunks of equal size differ by a factor of two and more, and estimates in plain bases made every other sub-run
        // of the full C4 come back refused.  Weight 2 rank / names keeps the mean at one.
        uint64_t sub_bases = 0, sub_real_bases = 0;
        const double n_names = (double)std::max<size_t>(1, m.name_of_rank.size());
        while (ci < my_chunks.size()) {
            const auto &ch = m.chunks[my_chunks[ci]];
            const uint64_t cb_real = m.T.off[ch.second] - m.T.off[ch.first];
            uint64_t cb = cb_real;
            if (m.opts.pair_once) {
                double w = 0;
                for (uint32_t t = ch.first; t < ch.second; ++t) w += (double)m.T.len(t) * 2.0 * ((double)m.rank_t[t] + 0.5) / n_names;
                cb = (uint64_t)w + 1;
            }
            if (n_my && (sub_bases + cb > budget_bases || sub_real_bases + cb_real > SUBRUN_MAX_BASES ||
                         sub_tids.size() + (ch.second - ch.first) > SUBRUN_MAX_TARGETS)) break;
            for (uint32_t t = ch.first; t < ch.second; ++t) { sub_tids.push_back(t); chunk_of_t.push_back(n_my); }
            sub_bases += cb;
            sub_real_bases += cb_real;
            ++n_my; ++ci;
        }
        if (sub_tids.empty()) continue;
        const double ts0 = now_s();
        DevReads dT_sub;
        const bool all_in_order = sub_tids.size() == m.T.size();   // every chunk, file order
        if (!all_in_order) subset_reads_device(*m.dT, sub_tids, dT_sub);
        DevReads &dT = all_in_order ? *m.dT : dT_sub;
        std::vector<uint32_t> rt(sub_tids.size());
        for (size_t i = 0; i < sub_tids.size(); ++i) rt[i] = m.rank_t[sub_tids[i]];
        DBuf<uint32_t> d_rt, d_ct;
        d_rt.upload(rt);
        d_ct.upload(chunk_of_t);
        AvaInput in;
        in.T = &dT; in.Q = &m.dQ; in.d_rank_t = d_rt.p; in.d_rank_q = m.d_rank_q.p; in.d_chunk_of_t = d_ct.p;
        in.n_chunks = n_my; in.d_qmz = m.d_qmz; in.qmz_off = m.qmz_off;
        in.n_ranks = m.name_of_rank.size();
        if (m.dT == &m.dQ) in.t_query = sub_tids;     // reads vs themselves: the targets' minimizers are in the query sketch
        in.max_anchors = (uint64_t)subrun_anchors_max;
        in.max_out_bytes = (uint64_t)subrun_out_max;
        in.max_lanes = lanes_fit ? 0 : 1;
        AvaRows rows;
        const double a0 = stats()["anchors"];
        // a run that is given up (it may already have aligned its first query batch) leaves no trace in the pass's
        // counts and kernel timers: bench.py's roofline bytes are computed from them
        const std::map<std::string, double> stats_before = stats();
        const size_t timers_before = ktimer_mark();
        ava_device(in, m.opts, rows);
        if (rows.refused_anchors) {           // deeper than the estimate: come back with fewer chunks
            budget_bases = std::max<uint64_t>(1, (uint64_t)(rows.refused_shrink * (double)sub_bases));
            ci -= n_my;
            stats() = stats_before;
            ktimer_rollback(timers_before);
            stat_add("subruns_refused", 1);
            continue;
        }
        tids.insert(tids.end(), sub_tids.begin(), sub_tids.end());
        ++n_subruns;
        done_bases += sub_bases;
        done_anchors += stats()["anchors"] - a0;
        done_out_bytes += (double)ava_out_bytes(rows.n_rows, rows.n_ops);
        if (!fixed_budget && done_anchors > 0) {        // next sub-run: as many target bases as give ~SUBRUN_ANCHORS anchors and
            const double by_anchors = subrun_anchors * (double)done_bases / done_anchors;      // ~SUBRUN_OUT_BYTES of output
            const double by_bytes = done_out_bytes > 0 ? subrun_out * (double)done_bases / done_out_bytes : by_anchors;
            budget_bases = std::min<uint64_t>(SUBRUN_MAX_BASES, std::max<uint64_t>(subrun_anchors < SUBRUN_ANCHORS || subrun_out < SUBRUN_OUT_BYTES ? 1 : 4ull << 20, (uint64_t)std::min(by_anchors, by_bytes)));
        }
        const double t1 = now_s();
        t_ava += t1 - ts0;
        FilterCfg cfg;
        cfg.len_over = len_over; cfg.mc = mc; cfg.long_mode = m.long_mode;
        cfg.chunk_id_bound = n_my;
        cfg.reference_order = false;          // the rows are sorted by (score, text) below: one total order
        // filter the chunks in groups whose SNP events (<= 2 per X op) stay below the 32-bit offsets the event arrays use
        const std::vector<uint64_t> chunk_ops = ops_per_chunk(rows.recs.p, rows.n_rows, n_my);
        FilterOut fo;
        double t_fmt_sub = 0;
        for (uint32_t c0 = 0; c0 < n_my;) {
            uint32_t c1 = c0;
            uint64_t ops = 0;
            while (c1 < n_my && (c1 == c0 || ops + chunk_ops[c1] <= (1ull << 30))) ops += chunk_ops[c1++];
            const uint64_t r0 = rows.chunk_row_start[c0], r1 = rows.chunk_row_start[c1];
            std::vector<uint64_t> crs(rows.chunk_row_start.begin() + c0, rows.chunk_row_start.begin() + c1 + 1);
            for (auto &v : crs) v -= r0;
            filter_stage_device(rows.recs.p + r0, (size_t)(r1 - r0), rows.ops_base, crs, cfg, fo);
            n_v4 += fo.n_after_v4; n_ev += fo.n_events; n_pairs += fo.n_pairs;
            const double tf = now_s();
            // Rows -> text.  Many rows (the short-read calls keep millions per pass): on the device (row_text.hip), the host only
            // takes the few rows the device's "%.4f" does not cover.  Few rows: on the host threads as before (HLMI_TEXT_GPU /
            // HLMI_TEXT_HOST: test hooks for the two forms).
            const bool text_on_gpu = !hook("HLMI_TEXT_HOST") && (fo.rows.size() >= 65536 || hook("HLMI_TEXT_GPU"));
            if (text_on_gpu) {
                if (!m.d_names.n) m.d_names.upload(m.name_of_rank);
                RowText rt;
                format_rows_device(rows.recs.p + r0, fo.rows, fo.x_digit_sum, m.d_names, iden, rt);
                t_dl += now_s() - tf;
                std::unique_ptr<std::string> extra(new std::string());        // rows formatted by the host
                std::vector<std::pair<size_t, std::pair<size_t, size_t>>> extra_at;     // row -> (offset, length) in *extra
                for (size_t i = 0; i < fo.rows.size(); ++i)
                    if (rt.len[i] == ROW_TEXT_TO_HOST) {
                        const std::vector<PafRec> one = download_rows(rows.recs.p + r0, std::vector<uint32_t>(1, fo.rows[i]));
                        std::string tmp;
                        uint32_t k = 0;
                        if (format_scored_row(one[0], m.name_of_rank[one[0].qid], m.name_of_rank[one[0].tid], fo.x_digit_sum[i], iden, tmp, &k)) {
                            extra_at.push_back({i, {extra->size(), tmp.size()}});
                            extra->append(tmp);
                            rt.key[i] = k;
                        }
                    }
                stat_add("rows_text_on_gpu", (double)fo.rows.size());
                stat_add("rows_text_left_to_host", (double)extra_at.size());
                text.emplace_back(new std::string(std::move(rt.text)));
                const std::string &buf = *text.back();
                text.emplace_back(std::move(extra));
                const std::string &ebuf = *text.back();
                size_t e = 0;
                lines.reserve(lines.size() + fo.rows.size());
                line_keys.reserve(line_keys.size() + fo.rows.size());
                for (size_t i = 0; i < fo.rows.size(); ++i) {               // stream order kept (the final sort is total anyway)
                    if (rt.len[i] == ROW_TEXT_TO_HOST) {
                        if (e < extra_at.size() && extra_at[e].first == i) {
                            lines.emplace_back(ebuf.data() + extra_at[e].second.first, extra_at[e].second.second);
                            line_keys.push_back(rt.key[i]);
                            ++e;
                        }
                    } else if (rt.len[i]) { lines.emplace_back(buf.data() + rt.at[i], rt.len[i]); line_keys.push_back(rt.key[i]); }
                }
            } else {
            std::vector<PafRec> kept = download_rows(rows.recs.p + r0, fo.rows);
            t_dl += now_s() - tf;
            {   // rows -> text on the host threads (three %.4f conversions per row dominate), order kept
                const size_t nk = kept.size();
                std::vector<uint32_t> at(nk), len(nk, 0), key(nk); // span of row i in its thread's buffer (len 0: dropped), sort key
                const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), nk / 2048));
                const size_t first_buf = text.size();
                for (int t = 0; t < nt; ++t) {
                    if (!m.text_cache.empty()) { text.emplace_back(std::move(m.text_cache.back())); m.text_cache.pop_back(); text.back()->clear(); }
                    else text.emplace_back(new std::string());
                }
                auto work = [&](int t) {
                    std::string tmp, &buf = *text[first_buf + (size_t)t];
                    buf.reserve((nk / nt + 1) * 160);
                    for (size_t i = nk * (size_t)t / nt; i < nk * (size_t)(t + 1) / nt; ++i)
                        if (format_scored_row(kept[i], m.name_of_rank[kept[i].qid], m.name_of_rank[kept[i].tid],
                                              fo.x_digit_sum[i], iden, tmp, &key[i])) {
                            at[i] = (uint32_t)buf.size(); len[i] = (uint32_t)tmp.size();
                            buf.append(tmp);
                        }
                };
                std::vector<std::thread> pool;
                for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
                work(0);
                for (auto &th : pool) th.join();
                lines.reserve(lines.size() + nk);
                line_keys.reserve(line_keys.size() + nk);
                for (int t = 0; t < nt; ++t) {                      // the buffers are final: views are safe now
                    const std::string &buf = *text[first_buf + (size_t)t];
                    for (size_t i = nk * (size_t)t / nt; i < nk * (size_t)(t + 1) / nt; ++i)
                        if (len[i]) { lines.emplace_back(buf.data() + at[i], len[i]); line_keys.push_back(key[i]); }
                }
            }
            }
            t_fmt_sub += now_s() - tf;
            c0 = c1;
        }
        t_fmt += t_fmt_sub;
        t_flt += now_s() - t1 - t_fmt_sub;
    }
    if (done_bases && done_anchors > 0) {
        m.est_anchors_per_base = done_anchors / (double)done_bases;
        m.est_out_per_base = done_out_bytes / (double)done_bases;
    }
    stat_set("subruns", (double)n_subruns);
    stat_set("rows_after_v4", (double)n_v4);
    stat_set("snp_events", (double)n_ev);
    stat_set("pairs", (double)n_pairs);
    stat_set("t_ava_s", t_ava);
    stat_set("t_filter_s", t_flt);
    stat_set("t_rows_to_text_s", t_fmt);
    stat_set("t_rows_download_s", t_dl);            // (part of t_rows_to_text_s)
    const double t3 = now_s();
    sort_scored_lines(lines, &line_keys);     // per-chunk sort + merged sort of utils.py:54,69 collapse into one total order
    stat_set("t_final_sort_s", now_s() - t3);
    write_lines(out_paf, lines);
    {   // the buffers go back to the job, largest first, 2 GB of capacity at most
        std::sort(text.begin(), text.end(), [](const auto &a, const auto &b) { return a->capacity() > b->capacity(); });
        size_t held = 0;
        for (auto &b : m.text_cache) held += b->capacity();
        for (auto &b : text)
            if (held + b->capacity() <= (2ull << 30) && m.text_cache.size() < 64) { held += b->capacity(); m.text_cache.emplace_back(std::move(b)); }
        lines.clear();
        text.clear();
    }
    stat_set("rows_out", (double)line_keys.size());
    stat_set("t_format_sort_write_s", now_s() - t3);
    ktimer_flush();
    stat_set("t_total_s", now_s() - t0);
    stat_set("hbm_peak_in_use_gb", (double)dev_peak_bytes(false) / 1e9);      // reads + sketch + the largest sub-run

    // the counts SURVEY.md 8d's byte formula is evaluated on (bench.py: roofline.stage)
    stat_set("bases_q", (double)m.n_bases_q);
    uint64_t bt = 0;
    for (uint32_t t : tids) bt += m.T.len(t);
    stat_set("bases_t", (double)bt);
    stat_set("targets", (double)tids.size());
    stat_set("queries", (double)m.Q.size());
    stat_set("minimizers_q", m.qmz_off.empty() ? 0.0 : (double)m.qmz_off.back());
    stat_set("chunks_run", (double)my_chunks.size());
}

}  // namespace hlmi
