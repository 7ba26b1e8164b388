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

namespace hlmi {

struct Job::Impl {
    SeqSet Q, T;
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
};

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Job::Job(const char *reads_fa, const char *ref_fa, int nsplit, bool long_mode) : impl_(new Impl) {
    Impl &m = *impl_;
    m.long_mode = long_mode;
    m.opts = long_mode ? ava_opts_long() : ava_opts_short();   // filter_overlap_slr2.py:51 / :55
    read_seqs(reads_fa, m.Q);
    if (std::string(reads_fa) == ref_fa) m.T = m.Q; else read_seqs(ref_fa, m.T);
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
    if (std::string(reads_fa) == ref_fa) m.dT = &m.dQ;
    else { upload_reads(m.T, 0, m.T.size(), m.dT_own); m.dT = &m.dT_own; }
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
    if (lo == 0 && (size_t)hi == m.Q.size())     // the reads are already resident in HBM
        return sketch_device_into(m.dQ, m.opts.k, m.opts.w, m.opts.hpc, 0, (Mz *)dev_mz, cap, (uint32_t *)dev_counts);
    DevReads part;
    std::vector<uint32_t> ids((size_t)(hi - lo));
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = (uint32_t)(lo + (int64_t)i);
    subset_reads_device(m.dQ, ids, part);
    return sketch_device_into(part, m.opts.k, m.opts.w, m.opts.hpc, (uint32_t)lo, (Mz *)dev_mz, cap, (uint32_t *)dev_counts);
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
    sketch_device(m.dQ, m.opts.k, m.opts.w, m.opts.hpc, 0, m.own);
    set_query_sketch(m.own.mz.p, (int64_t)m.own.n, m.own.counts.p);
}

void Job::run(int rank, int world, int len_over, int mc, double iden, const char *out_paf) {
    Impl &m = *impl_;
    if (!m.d_qmz && m.Q.size()) fail(HLMI_ESTATE, "hlmi_job_run before a query sketch was installed");
    stat_reset();
    const double t0 = now_s();
    struct ExitStamp { double t0; ~ExitStamp() { stat_set("t_with_cleanup_s", now_s() - t0); } } exit_stamp{t0};   // runs after the locals are gone
    // ---- this rank's chunks and targets ---------------------------------------------------------------
    std::vector<uint32_t> tids, chunk_of_t;
    uint32_t n_my = 0;
    for (size_t c = 0; c < m.chunks.size(); ++c) {
        if ((int)(c % (size_t)world) != rank) continue;
        for (uint32_t t = m.chunks[c].first; t < m.chunks[c].second; ++t) { tids.push_back(t); chunk_of_t.push_back(n_my); }
        ++n_my;
    }
    // the rows' text: every formatting thread appends to a buffer of its own, `lines` are views into those buffers
    std::vector<std::string_view> lines;
    std::vector<std::unique_ptr<std::string>> text;
    if (!tids.empty() && m.Q.size()) {
        DevReads dT_sub;
        const bool all_in_order = tids.size() == m.T.size();   // world == 1: every chunk, file order
        if (!all_in_order) subset_reads_device(*m.dT, tids, dT_sub);
        DevReads &dT = all_in_order ? *m.dT : dT_sub;
        std::vector<uint32_t> rt(tids.size());
        for (size_t i = 0; i < tids.size(); ++i) rt[i] = m.rank_t[tids[i]];
        DBuf<uint32_t> d_rt, d_ct;
        d_rt.upload(rt);
        d_ct.upload(chunk_of_t);
        AvaInput in;
        in.T = &dT; in.Q = &m.dQ; in.d_rank_t = d_rt.p; in.d_rank_q = m.d_rank_q.p; in.d_chunk_of_t = d_ct.p;
        in.n_chunks = n_my; in.d_qmz = m.d_qmz; in.qmz_off = m.qmz_off;
        in.n_ranks = m.name_of_rank.size();
        if (m.dT == &m.dQ) in.t_query = tids;         // reads vs themselves: the targets' minimizers are in the query sketch
        AvaRows rows;
        ava_device(in, m.opts, rows);
        const double t1 = now_s();
        FilterCfg cfg;
        cfg.len_over = len_over; cfg.mc = mc; cfg.long_mode = m.long_mode;
        cfg.chunk_id_bound = n_my;
        cfg.reference_order = false;          // the rows are sorted by (score, text) below: one total order
        // The chunks are independent (the reference runs one worker per chunk); filter them in groups whose SNP
        // events (<= 2 per X op) stay below the 32-bit offsets the event arrays use.
        const std::vector<uint64_t> chunk_ops = ops_per_chunk(rows.recs.p, rows.n_rows, n_my);
        FilterOut fo;
        std::string s;
        size_t n_v4 = 0, n_ev = 0, n_pairs = 0;
        double t_fmt = 0;
        for (uint32_t c0 = 0; c0 < n_my;) {
            uint32_t c1 = c0;
            uint64_t ops = 0;
            while (c1 < n_my && (c1 == c0 || ops + chunk_ops[c1] <= (1ull << 30))) ops += chunk_ops[c1++];
            const uint64_t r0 = rows.chunk_row_start[c0], r1 = rows.chunk_row_start[c1];
            std::vector<uint64_t> crs(rows.chunk_row_start.begin() + c0, rows.chunk_row_start.begin() + c1 + 1);
            for (auto &v : crs) v -= r0;
            filter_stage_device(rows.recs.p + r0, (size_t)(r1 - r0), rows.ops.p, crs, cfg, fo);
            n_v4 += fo.n_after_v4; n_ev += fo.n_events; n_pairs += fo.n_pairs;
            const double tf = now_s();
            std::vector<PafRec> kept = download_rows(rows.recs.p + r0, fo.rows);
            {   // rows -> text on the host threads (three %.4f conversions per row dominate), order kept
                const size_t nk = kept.size();
                std::vector<uint32_t> at(nk), len(nk, 0);          // span of row i in its thread's buffer (len 0: dropped)
                const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), nk / 2048));
                const size_t first_buf = text.size();
                for (int t = 0; t < nt; ++t) text.emplace_back(new std::string());
                auto work = [&](int t) {
                    std::string tmp, &buf = *text[first_buf + (size_t)t];
                    buf.reserve((nk / nt + 1) * 160);
                    for (size_t i = nk * (size_t)t / nt; i < nk * (size_t)(t + 1) / nt; ++i)
                        if (format_scored_row(kept[i], m.name_of_rank[kept[i].qid], m.name_of_rank[kept[i].tid],
                                              fo.x_digit_sum[i], iden, tmp)) {
                            at[i] = (uint32_t)buf.size(); len[i] = (uint32_t)tmp.size();
                            buf.append(tmp);
                        }
                };
                std::vector<std::thread> pool;
                for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
                work(0);
                for (auto &th : pool) th.join();
                for (int t = 0; t < nt; ++t) {                      // the buffers are final: views are safe now
                    const std::string &buf = *text[first_buf + (size_t)t];
                    for (size_t i = nk * (size_t)t / nt; i < nk * (size_t)(t + 1) / nt; ++i)
                        if (len[i]) lines.emplace_back(buf.data() + at[i], len[i]);
                }
            }
            t_fmt += now_s() - tf;
            c0 = c1;
        }
        const double t2 = now_s() - t_fmt;
        fo.n_after_v4 = n_v4; fo.n_events = n_ev; fo.n_pairs = n_pairs;
        stat_set("rows_after_v4", (double)fo.n_after_v4);
        stat_set("snp_events", (double)fo.n_events);
        stat_set("pairs", (double)fo.n_pairs);
        stat_set("t_ava_s", t1 - t0);
        stat_set("t_filter_s", t2 - t1);
        stat_set("t_rows_to_text_s", t_fmt);
    }
    const double t3 = now_s();
    sort_scored_lines(lines);     // per-chunk sort + merged sort of utils.py:54,69 collapse into one total order
    stat_set("t_final_sort_s", now_s() - t3);
    write_lines(out_paf, lines);
    stat_set("rows_out", (double)lines.size());
    stat_set("t_format_sort_write_s", now_s() - t3);
    ktimer_flush();
    stat_set("t_total_s", now_s() - t0);

    stat_set("bases_q", (double)m.Q.bases.size());
}

}  // namespace hlmi
