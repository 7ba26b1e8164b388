// capi.cpp - the extern "C" boundary of libhylight_mi.so (include/hylight_mi.h).
#include <charconv>
#include <chrono>

#include "ava.h"
#include "common.h"
#include "filter_stage.h"
#include "graph.h"
#include "paf_io.h"
#include "stage.h"
#include "textpass.h"

namespace hlmi {
void init_device(int device, int threads);
void shutdown_device();
const std::string &last_error();
}  // namespace hlmi

using namespace hlmi;

namespace {
template <typename F>
int guarded(F &&f) {
    try {
        hooks_refresh();         // test hooks / tuning switches of this call (the one place the environment is read)
        f();
        ktimer_flush();          // kernel timers of this call end here: none leaks into the next call's statistics
        return HLMI_OK;
    } catch (const Error &e) {
        ktimer_discard();
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        ktimer_discard();
        set_last_error("out of host memory");
        return HLMI_ENOMEM;
    } catch (const std::exception &e) {
        ktimer_discard();
        set_last_error(e.what());
        return HLMI_EINVAL;
    }
}

std::string py_float_repr(double v) {   // Python str(float) for the magnitudes that occur here
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos &&
        s.find("nan") == std::string::npos)
        s += ".0";
    return s;
}
}  // namespace

extern "C" {

int hlmi_init(int device, int host_threads) {
    return guarded([&] { init_device(device, host_threads); });
}
int hlmi_abi_version(void) { return HLMI_ABI_VERSION; }
void hlmi_shutdown(void) { shutdown_device(); }
const char *hlmi_last_error(void) { return last_error().c_str(); }
const char *hlmi_version(void) { return "hylight-mi355x 0.1 (gfx950)"; }

int hlmi_filter_chunk(const char *paf_in, const char *out_paf, int len_over, int mc, double iden, double thre,
                      int min_o, int long_mode) {
    return guarded([&] {
        if (!paf_in || !out_paf) fail(HLMI_EINVAL, "hlmi_filter_chunk: NULL path");
        require_device();
        stat_reset();
        PafText pt;
        read_paf(paf_in, pt, true);
        DBuf<PafRec> d_recs;
        DBuf<uint32_t> d_ops;
        d_recs.upload(pt.recs);
        d_ops.upload(pt.ops.data(), pt.ops.size());
        if (pt.ops.empty()) d_ops.alloc(1);
        FilterCfg cfg;
        cfg.len_over = len_over; cfg.mc = mc; cfg.thre = thre; cfg.min_o = min_o; cfg.long_mode = long_mode != 0;
        FilterOut fo;
        filter_stage_device(d_recs.p, pt.recs.size(), d_ops.p, {0, (uint64_t)pt.recs.size()}, cfg, fo);
        std::vector<std::string> lines;
        std::string s;
        for (size_t i = 0; i < fo.rows.size(); ++i) {
            const PafRec &r = pt.recs[fo.rows[i]];
            if (format_scored_row(r, pt.dict.names[r.qid], pt.dict.names[r.tid], fo.x_digit_sum[i], iden, s))
                lines.push_back(s);
        }
        write_lines(out_paf, lines);
        stat_set("rows_in", (double)pt.recs.size());
        stat_set("rows_after_v4", (double)fo.n_after_v4);
        stat_set("snp_events", (double)fo.n_events);
        stat_set("pairs", (double)fo.n_pairs);
        stat_set("rows_out", (double)lines.size());
    });
}

int hlmi_paf_window_filter(int variant, int min_len, double min_iden, int min_o, int sfo, const char *in_paf,
                           const char *out_path) {
    return guarded([&] {
        if (!in_paf || !out_path) fail(HLMI_EINVAL, "hlmi_paf_window_filter: NULL path");
        if (variant != 3 && variant != 4) fail(HLMI_EINVAL, "variant must be 3 or 4");
        require_device();
        if (min_iden < 0) min_iden = variant == 4 ? 0.6 : 0.8;
        PafText pt;
        read_paf(in_paf, pt, false);
        size_t n = pt.recs.size();
        std::vector<std::string> lines;
        if (n) {
            DBuf<PafRec> d_recs;
            d_recs.upload(pt.recs);
            DBuf<uint8_t> keep(n);
            window_filter_device(d_recs.p, n, {0, (uint64_t)n}, variant, min_len, min_iden, min_o, keep.p);
            std::vector<uint8_t> hk = keep.download(n);
            for (size_t i = 0; i < n; ++i) {
                if (!hk[i]) continue;
                if (variant == 4) {
                    lines.emplace_back(pt.line(i));
                    continue;
                }
                const PafRec &r = pt.recs[i];
                const std::string &q = pt.dict.names[r.qid], &t = pt.dict.names[r.tid];
                if (sfo) {   // filter_trans_ovlp_inline_v3.py:83-102
                    int64_t ql = r.qlen, qs = r.qs, tl = r.tlen, ts = r.ts, te = r.te, oha, ohb;
                    bool rev = r.flags & PF_REV;
                    if (!rev) { oha = qs - ts; ohb = tl - ts - (ql - qs); }
                    else { oha = qs - (tl - te); ohb = te - (ql - qs); }
                    int64_t ola = oha >= 0 ? std::min(ql - oha, tl) : std::min(tl + oha, ql);
                    char buf[160];
                    snprintf(buf, sizeof buf, "\t%c\t%lld\t%lld\t%lld\t%lld\t%lld", rev ? 'I' : 'N', (long long)oha,
                             (long long)ohb, (long long)ola, (long long)ola, (long long)r.blen - (long long)r.nmatch);
                    lines.push_back(q + "\t" + t + buf);
                } else {     // filter_trans_ovlp_inline_v3.py:79
                    double mlen = (double)((uint64_t)r.qlen + r.tlen) / 2.0;
                    double a = 0.1 * ((double)r.blen / mlen), b = 0.9 * ((double)r.nmatch / (double)r.blen);
                    lines.push_back(q + "\t" + t + "\t" + py_float_repr(a + b));
                }
            }
        }
        write_lines(out_path, lines);
    });
}

int hlmi_filter_ovlp_inline(const char *in_paf, const char *out_paf, int min_ovlp_len, double min_identity, int o, double r) {
    return guarded([&] {
        if (!in_paf || !out_paf) fail(HLMI_EINVAL, "hlmi_filter_ovlp_inline: NULL path");
        require_device();
        PafText pt;
        read_paf(in_paf, pt, false);
        const size_t n = pt.recs.size();
        std::vector<std::string> lines;
        if (n) {
            DBuf<PafRec> d_recs;
            d_recs.upload(pt.recs);
            DBuf<uint8_t> keep(n);
            DBuf<uint32_t> first_of(n);
            ovlp_inline_device(d_recs.p, n, min_ovlp_len, min_identity, o, r, keep.p, first_of.p);
            std::vector<uint8_t> hk = keep.download(n);
            std::vector<uint32_t> hf = first_of.download(n);
            std::vector<std::pair<uint32_t, uint32_t>> order;      // (print position, row)
            for (size_t i = 0; i < n; ++i) if (hk[i]) order.emplace_back(hf[i], (uint32_t)i);
            std::sort(order.begin(), order.end());
            for (auto &pr : order) lines.emplace_back(pt.line(pr.second));
        }
        write_lines(out_paf, lines);
    });
}

int hlmi_filter_non_atcg(const char *fastx, const char *out_fa, int is_fastq) {
    return guarded([&] {
        if (!fastx || !out_fa) fail(HLMI_EINVAL, "hlmi_filter_non_atcg: NULL path");
        filter_non_atcg_run(fastx, out_fa, is_fastq != 0);
    });
}

int hlmi_gfa2fa(const char *gfa, const char *out_fa) {
    return guarded([&] {
        if (!gfa || !out_fa) fail(HLMI_EINVAL, "hlmi_gfa2fa: NULL path");
        gfa2fa_run(gfa, out_fa);
    });
}

int hlmi_pick_up(const char *ovlap_paf, const char *fastx, const char *out_fastx, int is_fastq) {
    return guarded([&] {
        if (!ovlap_paf || !fastx || !out_fastx) fail(HLMI_EINVAL, "hlmi_pick_up: NULL path");
        pick_up_run(ovlap_paf, fastx, out_fastx, is_fastq != 0);
    });
}

int hlmi_minimap22sfo(const char *in_paf, const char *out_sfo, int min_overlap_len, double min_pident) {
    return guarded([&] {                                         // script/minimap22sfo.py:28-75 - a text converter
        if (!in_paf || !out_sfo) fail(HLMI_EINVAL, "hlmi_minimap22sfo: NULL path");
        PafText pt;
        read_paf(in_paf, pt, false);
        std::vector<std::string> lines;
        for (const PafRec &rc : pt.recs) {
            if ((int64_t)rc.blen < (int64_t)min_overlap_len) continue;
            if ((double)rc.nmatch / (double)rc.blen < min_pident / 100.0) continue;
            const bool rev = rc.flags & PF_REV;
            int64_t ql = rc.qlen, qs = rc.qs, tl = rc.tlen, ts = rc.ts, te = rc.te, oha, ohb;
            if (!rev) { oha = qs - ts; ohb = tl - ts - (ql - qs); }
            else { oha = qs - (tl - te); ohb = te - (ql - qs); }
            const int64_t ola = oha >= 0 ? std::min(ql - oha, tl) : std::min(tl + oha, ql);
            const std::string *a = &pt.dict.names[rc.qid], *b = &pt.dict.names[rc.tid];
            if (*a > *b) {                                       // ids in string order; read A forward
                std::swap(a, b);
                if (!rev) { oha = -oha; ohb = -ohb; } else std::swap(oha, ohb);
            }
            char buf[160];
            snprintf(buf, sizeof buf, "\t%c\t%lld\t%lld\t%lld\t%lld\t%lld", rev ? 'I' : 'N', (long long)oha, (long long)ohb,
                     (long long)ola, (long long)ola, (long long)rc.blen - (long long)rc.nmatch);
            lines.push_back(*a + "\t" + *b + buf);
        }
        write_lines(out_sfo, lines);
    });
}

int hlmi_merge_scored_paf(const char *const *in_pafs, int n_in, const char *out_paf) {
    return guarded([&] {
        std::vector<std::string> data((size_t)n_in);         // the parts stay whole, the lines are views into them
        std::vector<std::string_view> lines;
        for (int i = 0; i < n_in; ++i) {
            data[(size_t)i] = read_file(in_pafs[i]);
            const std::string &d = data[(size_t)i];
            size_t pos = 0;
            while (pos < d.size()) {
                size_t e = d.find('\n', pos);
                if (e == std::string::npos) e = d.size();
                lines.emplace_back(d.data() + pos, e - pos);
                pos = e + 1;
            }
        }
        sort_scored_lines(lines);
        write_lines(out_paf, lines);
    });
}

int hlmi_split_reads2(const char *reads_fa, const char *ref_fa, int nsplit, const char *out_dir, const char *out_paf,
                      int threads, int len_over, int mc, double iden, int long_mode) {
    return hlmi_split_reads2_shard(reads_fa, ref_fa, nsplit, out_dir, out_paf, threads, len_over, mc, iden, long_mode,
                                   0, 1);
}

int hlmi_split_reads2_shard(const char *reads_fa, const char *ref_fa, int nsplit, const char *out_dir,
                            const char *out_paf, int threads, int len_over, int mc, double iden, int long_mode,
                            int rank, int world) {
    (void)out_dir; (void)threads;
    return guarded([&] {
        if (!reads_fa || !ref_fa || !out_paf) fail(HLMI_EINVAL, "hlmi_split_reads2: NULL path");
        if (nsplit < 1 || world < 1 || rank < 0 || rank >= world) fail(HLMI_EINVAL, "bad nsplit/rank/world");
        require_device();
        Job job(reads_fa, ref_fa, nsplit, long_mode != 0);
        job.sketch_all_queries();
        job.run(rank, world, len_over, mc, iden, out_paf);
    });
}

void hlmi_ava_opts_long(hlmi_ava_opts *o) {
    if (!o) return;
    *o = ava_opts_long();
}
void hlmi_ava_opts_short(hlmi_ava_opts *o) {
    if (!o) return;
    *o = ava_opts_short();
}

int hlmi_ava(const char *target_fa, const char *query_fa, const hlmi_ava_opts *opts, const char *out_paf) {
    return guarded([&] {
        if (!target_fa || !query_fa || !out_paf) fail(HLMI_EINVAL, "hlmi_ava: NULL path");
        require_device();
        hlmi_ava_opts o = opts ? *opts : ava_opts_long();
        ava_files(target_fa, query_fa, o, out_paf);
    });
}

int hlmi_miniasm(const char *paf, const char *reads_fa, int bub_dist, int n_rounds_arg, int max_ext, int min_dp,
                 const char *outfmt, const char *out_path) {
    return guarded([&] {
        if (!paf || !out_path) fail(HLMI_EINVAL, "hlmi_miniasm: NULL path");
        require_device();
        miniasm_run(paf, reads_fa, bub_dist, n_rounds_arg, max_ext, min_dp, outfmt ? outfmt : "ug", out_path);
    });
}

int hlmi_sfo2overlaps(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs) {
    return guarded([&] {
        if (!in_sfo || !out_savage) fail(HLMI_EINVAL, "hlmi_sfo2overlaps: NULL path");
        sfo2overlaps_run(in_sfo, out_savage, num_singles, num_pairs);
    });
}

int hlmi_vq_parse_overlaps(const char *savage_path, uint32_t min_overlap_len, uint32_t min_overlap_perc, int relax_pe,
                           uint64_t max_overlaps, hlmi_vq_overlap *out, uint64_t cap, uint64_t *n_out,
                           uint64_t *n_nonedge, uint64_t *n_skipped) {
    return guarded([&] {
        if (!savage_path || !n_out || !n_nonedge || !n_skipped || (cap && !out)) fail(HLMI_EINVAL, "hlmi_vq_parse_overlaps: NULL argument");
        vq_parse_overlaps(savage_path, min_overlap_len, min_overlap_perc, relax_pe, max_overlaps, out, cap, n_out, n_nonedge, n_skipped);
    });
}

int hlmi_vq_transitive_edges(uint32_t n_vertices, uint64_t n_edges, const uint32_t *src, const uint32_t *dst,
                             const uint32_t *ovlen, int remove_trans, uint8_t *flags, uint64_t *n_transitive) {
    return guarded([&] {
        if (!n_transitive || (n_edges && (!src || !dst || !flags))) fail(HLMI_EINVAL, "hlmi_vq_transitive_edges: NULL argument");
        require_device();
        vq_transitive_edges(n_vertices, n_edges, src, dst, ovlen, remove_trans, flags, n_transitive);
    });
}

int hlmi_vq_overlap_scores(const char *fastq_singles, const hlmi_vq_overlap *ov, uint64_t n, double mismatch,
                           uint32_t min_read_len, double *score, double *mismatch_rate, int64_t *pos3) {
    return guarded([&] {
        if (!fastq_singles || (n && (!ov || !score || !mismatch_rate || !pos3))) fail(HLMI_EINVAL, "hlmi_vq_overlap_scores: NULL argument");
        require_device();
        vq_overlap_scores(fastq_singles, ov, n, mismatch, min_read_len, score, mismatch_rate, pos3);
    });
}

hlmi_job *hlmi_job_open(const char *reads_fa, const char *ref_fa, int nsplit, int long_mode) {
    hlmi_job *j = nullptr;
    guarded([&] {
        if (!reads_fa || !ref_fa || nsplit < 1) fail(HLMI_EINVAL, "hlmi_job_open: bad arguments");
        require_device();
        j = reinterpret_cast<hlmi_job *>(new Job(reads_fa, ref_fa, nsplit, long_mode != 0));
    });
    return j;
}
void hlmi_job_close(hlmi_job *j) { delete reinterpret_cast<Job *>(j); }
int64_t hlmi_job_num_queries(const hlmi_job *j) { return j ? (int64_t) reinterpret_cast<const Job *>(j)->num_queries() : -1; }
int64_t hlmi_job_num_chunks(const hlmi_job *j) { return j ? (int64_t) reinterpret_cast<const Job *>(j)->num_chunks() : -1; }
int64_t hlmi_job_sketch_bound(const hlmi_job *j, int64_t lo, int64_t hi) {
    int64_t v = -1;
    guarded([&] { v = reinterpret_cast<const Job *>(j)->sketch_bound(lo, hi); });
    return v;
}
int hlmi_job_sketch(hlmi_job *j, int64_t lo, int64_t hi, void *dev_mz, int64_t cap, void *dev_counts, int64_t *n_out) {
    return guarded([&] {
        if (!j || !dev_mz || !dev_counts || !n_out) fail(HLMI_EINVAL, "hlmi_job_sketch: NULL argument");
        *n_out = reinterpret_cast<Job *>(j)->sketch_range(lo, hi, dev_mz, cap, dev_counts);
    });
}
int hlmi_job_sketch_own(hlmi_job *j) {
    return guarded([&] {
        if (!j) fail(HLMI_EINVAL, "hlmi_job_sketch_own: NULL job");
        reinterpret_cast<Job *>(j)->sketch_all_queries();
    });
}
int hlmi_job_set_query_sketch(hlmi_job *j, const void *dev_mz, int64_t n, const void *dev_counts) {
    return guarded([&] {
        if (!j || !dev_mz || !dev_counts) fail(HLMI_EINVAL, "hlmi_job_set_query_sketch: NULL argument");
        reinterpret_cast<Job *>(j)->set_query_sketch(dev_mz, n, dev_counts);
    });
}
int hlmi_job_run(hlmi_job *j, int rank, int world, int len_over, int mc, double iden, const char *out_paf) {
    return guarded([&] {
        if (!j || !out_paf || world < 1 || rank < 0 || rank >= world) fail(HLMI_EINVAL, "hlmi_job_run: bad arguments");
        reinterpret_cast<Job *>(j)->run(rank, world, len_over, mc, iden, out_paf);
    });
}

int hlmi_last_stats_json(char *buf, int64_t cap) {
    return guarded([&] {
        std::string s = "{";
        bool first = true;
        for (auto &kv : stats()) {
            char tmp[128];
            snprintf(tmp, sizeof tmp, "%s\"%s\": %.17g", first ? "" : ", ", kv.first.c_str(), kv.second);
            s += tmp;
            first = false;
        }
        s += "}";
        if ((int64_t)s.size() + 1 > cap) fail(HLMI_EINVAL, "stats buffer too small");
        memcpy(buf, s.c_str(), s.size() + 1);
    });
}

}  // extern "C"
