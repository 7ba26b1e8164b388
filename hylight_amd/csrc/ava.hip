// ava.hip - overlapper orchestration: index once, then query batches of seed -> sort -> chain ->
// align, rows gathered in HBM and put into stream order (chunk, query, target, strand, chain, piece).
#include "ava.h"

#include <algorithm>
#include <numeric>

#include "ava_internal.h"

#include <time.h>
#include <atomic>
#include <exception>
#include <memory>
#include <thread>
#include "dev_prims.h"

namespace hlmi {

namespace {
constexpr int WG = 256;
inline dim3 grid1(size_t n) { return dim3(cdiv(n ? n : 1, WG)); }
constexpr size_t ANCHOR_BATCH = 384u << 20;  // anchors per query batch: ~80 B of buffers each (anchor x2 for the sort, chain scratch, fixed
                                             // points, pieces) = 30 GB of the 288; on C3 128 M costs 8 % more per step (per-batch fixed
                                             // costs: scans, fills, list selections), 512 M outgrows the buffer pool's cap
constexpr size_t QUERY_BATCH = 1u << 16;     // queries per batch (ava_chain.hip: QL_BITS)

__global__ void lens_kernel(const uint64_t *off, size_t n, uint32_t *len) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) len[i] = (uint32_t)(off[i + 1] - off[i]);
}
__global__ void iota_kernel(uint32_t *a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ void gather_u64_kernel(const uint64_t *src, const uint32_t *idx, uint64_t *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void gather_rows_kernel(const PafRec *src, const uint32_t *idx, PafRec *dst, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    PafRec r = src[idx[i]];
    r.tie = (uint32_t)i;    // stream position (only identical lines fall back to it)
    r.flags |= PF_GEN;      // whole-line order of these rows comes from their fields (filter_stage.hip: row_text_less)
    dst[i] = r;
}
__global__ void shift_cigar_kernel(PafRec *recs, size_t n, uint64_t base) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) recs[i].cig_off += base;
}
}  // namespace

hlmi_ava_opts ava_opts_long() {
    // script/filter_overlap_slr2.py:51: -x ava-pb -Hk19 -m100 -g10000 (+ ava-pb: w=5; defaults -n3, A2 B4 O4 E2)
    hlmi_ava_opts o{};
    o.k = 19; o.w = 5; o.hpc = 1;
    o.min_chain_score = 100; o.max_gap = 10000; o.bandwidth = 2000; o.min_cnt = 3;
    o.min_mid_occ = 10; o.mid_occ_frac = 2e-4;
    o.match = 2; o.mismatch = 4; o.gap_open = 4; o.gap_ext = 2; o.ambi = 1;
    o.min_dp_score = 80; o.end_bonus = 0; o.pair_once = 1;
    o.gap_open2 = 24; o.gap_ext2 = 1;          // -O4,24 -E2,1: the preset's two-piece gap cost
    o.stub_oh = -1;                            // every piece extended (the stage sets the bound of its v4 filter)
    o.zdrop = 400;                             // -z 400 (preset default): end extensions run up to max_gap rows or to a z-drop
    return o;
}

hlmi_ava_opts ava_opts_short() {
    // script/filter_overlap_slr2.py:55: --sr -DP --no-long-join -k21 -w11 -s60 -m30 -n2 -A4 -B2 --end-bonus=100
    // (+ what the --sr preset leaves in place: no HPC, -g200, -r50, -O12 -E2, -f1000; no -X, so both directions
    // of a pair are reported)
    hlmi_ava_opts o{};
    o.k = 21; o.w = 11; o.hpc = 0;
    o.min_chain_score = 30; o.max_gap = 200; o.bandwidth = 50; o.min_cnt = 2;
    o.min_mid_occ = 1000; o.mid_occ_frac = 0.0;
    o.match = 4; o.mismatch = 2; o.gap_open = 12; o.gap_ext = 2; o.ambi = 1;
    o.min_dp_score = 60; o.end_bonus = 100; o.pair_once = 0;
    o.gap_open2 = 32; o.gap_ext2 = 1;          // --sr: -O12,32 -E2,1
    o.stub_oh = -1;
    o.zdrop = 0;                               // extensions stay within 256 rows (max_gap 200): no z-drop
    return o;
}

void name_ranks(const std::vector<std::string> &a, const std::vector<std::string> &b, std::vector<uint32_t> &ra,
                std::vector<uint32_t> &rb, std::vector<std::string> &name_of_rank) {
    std::vector<const std::string *> all;
    all.reserve(a.size() + b.size());
    for (auto &s : a) all.push_back(&s);
    for (auto &s : b) all.push_back(&s);
    std::sort(all.begin(), all.end(), [](const std::string *x, const std::string *y) { return *x < *y; });
    name_of_rank.assign(all.size(), std::string());
    auto rank_of = [&](const std::string &s) {
        size_t lo = 0, hi = all.size();
        while (lo < hi) {
            size_t m = (lo + hi) >> 1;
            if (*all[m] < s) lo = m + 1; else hi = m;
        }
        return (uint32_t)lo;
    };
    ra.resize(a.size());
    rb.resize(b.size());
    for (size_t i = 0; i < a.size(); ++i) { ra[i] = rank_of(a[i]); name_of_rank[ra[i]] = a[i]; }
    for (size_t i = 0; i < b.size(); ++i) { rb[i] = rank_of(b[i]); name_of_rank[rb[i]] = b[i]; }
}

// one wave per target: copy its minimizers out of the query sketch, read id rewritten to the local target index
__global__ __launch_bounds__(WG) void gather_sketch_kernel(const Mz *qmz, const uint64_t *src_off, const uint64_t *dst_off, size_t n_t,
                                                            Mz *out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t t = wave; t < n_t; t += n_waves) {
        const uint64_t s = src_off[t], d = dst_off[t], cnt = dst_off[t + 1] - d;
        for (uint64_t k = lane; k < cnt; k += 64) {
            Mz z = qmz[s + k];
            z.y = (uint64_t)t << 32 | (z.y & 0xffffffffull);
            out[d + k] = z;
        }
    }
}

// start[c] = first row of the sorted keys whose chunk field (the upper half) is >= c; start[n_chunks] = n
__global__ __launch_bounds__(WG) void chunk_row_start_kernel(const uint64_t *hi, size_t n, uint32_t n_chunks, uint64_t *start) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chunks) return;
    size_t lo = 0, up = n;
    if (c == n_chunks) lo = n;
    for (int it = 0; it < 64 && lo < up; ++it) {
        const size_t mid = lo + (up - lo) / 2;
        if ((uint32_t)(hi[mid] >> 32) < c) lo = mid + 1; else up = mid;
    }
    start[c] = lo;
}

// pieces of a set-aside part behind the parts of earlier batches: their fixed points moved by `delta` (mod 2^32)
__global__ __launch_bounds__(WG) void shift_fp_off_kernel(Piece *pieces, size_t n, uint32_t delta) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) pieces[i].fp_off += delta;
}

void ava_device(const AvaInput &in, const hlmi_ava_opts &o, AvaRows &out) {
    timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    const double w_start = (double)ts0.tv_sec + 1e-9 * (double)ts0.tv_nsec;
    out = AvaRows();
    out.chunk_row_start.assign(in.n_chunks + 1, 0);
    const size_t nQ = in.Q->n, nT = in.T->n;
    if (!nQ || !nT) return;
    DBuf<uint32_t> qlen(nQ), tlen(nT);
    hipLaunchKernelGGL(lens_kernel, grid1(nQ), dim3(WG), 0, stream(), in.Q->off.p, nQ, qlen.p);
    hipLaunchKernelGGL(lens_kernel, grid1(nT), dim3(WG), 0, stream(), in.T->off.p, nT, tlen.p);

    // ---- S1 (targets) + S2 ---------------------------------------------------------------------------
    DevSketch tsk;
    DevIndex ix;
    {
        HostTimer ht("index");
        if (in.t_query.size() == nT && in.d_qmz) {
            std::vector<uint64_t> src(nT), dst(nT + 1, 0);
            for (size_t t = 0; t < nT; ++t) {
                src[t] = in.qmz_off[in.t_query[t]];
                dst[t + 1] = dst[t] + (in.qmz_off[in.t_query[t] + 1] - src[t]);
            }
            tsk.n = dst[nT];
            tsk.mz.alloc(tsk.n ? tsk.n : 1);
            DBuf<uint64_t> d_src, d_dst;
            d_src.upload(src);
            d_dst.upload(dst);
            hipLaunchKernelGGL(gather_sketch_kernel, dim3((unsigned)std::min<size_t>(cdiv(nT, (size_t)(WG / 64)), 256 * 32)), dim3(WG), 0,
                               stream(), in.d_qmz, d_src.p, d_dst.p, nT, tsk.mz.p);
            HIP_CHECK(hipGetLastError());
        } else {
            sketch_device(*in.T, o.k, o.w, o.hpc, 0, tsk);
        }
        if (!in.n_ranks) fail(HLMI_EINVAL, "AvaInput::n_ranks not set");
        build_index(tsk, in.d_chunk_of_t, in.d_rank_t, in.n_chunks, in.n_ranks, o, ix);
        tsk.mz.release();
        if (hook("HLMI_SEED_GROUP")) seed_group_prepare(in, ix);
    }

    // ---- query batches ----------------------------------------------------------------------------------
    std::vector<AlignOut> parts;
    SeedStats st;
    SeedPlan plan;
    {
        HostTimer ht("plan_seeds");
        plan_seeds(in, ix, plan);
    }
    size_t anchor_batch = ANCHOR_BATCH;
    if (const char *e = hook("HLMI_ANCHOR_BATCH_M")) anchor_batch = (size_t)std::max(1, atoi(e)) << 20;   // tuning hook
    // batches of equal anchor counts (not full ones and a remainder: the last launches of a step would be small ones),
    // and no more queries than keep the anchor a single 64-bit word (seed_and_chain: "fits")
    uint64_t total_anchors = 0;
    for (size_t i = 0; i < nQ; ++i) total_anchors += plan.per_query[i];
    if (in.max_anchors && total_anchors > in.max_anchors && in.n_chunks > 1) {
        out.refused_anchors = total_anchors;
        out.refused_shrink = 0.5 * (double)in.max_anchors / (double)total_anchors;
        return;
    }
    const uint64_t n_batches = std::max<uint64_t>(1, (total_anchors + anchor_batch - 1) / anchor_batch);
    const uint64_t batch_target = (total_anchors + n_batches - 1) / n_batches;
    size_t q_cap = QUERY_BATCH;
    {
        uint64_t max_tlen = 1;
        for (size_t t = 0; t < nT; ++t) max_tlen = std::max<uint64_t>(max_tlen, in.T->h_off[t + 1] - in.T->h_off[t]);
        // query position bits of a typical batch: the 98th percentile of the query lengths (a batch that holds one of
        // the few longer reads takes the key + value form, the others are not made smaller for its sake)
        std::vector<uint64_t> ql(nQ);
        for (size_t i = 0; i < nQ; ++i) ql[i] = in.Q->h_off[i + 1] - in.Q->h_off[i];
        const size_t k98 = (nQ - 1) * 98 / 100;
        std::nth_element(ql.begin(), ql.begin() + (std::ptrdiff_t)k98, ql.end());
        const int spare = 64 - (bits_for(nT > 1 ? nT - 1 : 1) + 1 + bits_for(max_tlen) + bits_for(std::max<uint64_t>(ql[k98], 1)) + 8);
        // (never below 8192: a batch of many short reads rather takes the wider anchor form than ends early)
        size_t q_floor = 8192;
        if (const char *e = hook("HLMI_QCAP_MIN")) q_floor = (size_t)std::max(1, atoi(e));      // tuning hook
        // (when even q_floor queries do not fit the one-word form, the batch takes the word + small-key form anyway - the
        //  (target, strand) bits travel apart - and only its own widths bound the batch: on the full C4, 4 000 targets per
        //  sub-run, batches of 8 192 queries held a third of the anchors a batch is sized for)
        const int spare_split = 64 - (bits_for(max_tlen) + bits_for(std::max<uint64_t>(ql[k98], 1)) + 8);
        const int use = ((size_t)1 << std::max(0, std::min(spare, 16))) >= q_floor ? spare : spare_split;
        q_cap = std::min<size_t>(q_cap, std::max<size_t>(q_floor, (size_t)1 << std::max(0, std::min(use, 16))));
        // the widest anchor form (key + value) still keeps query, target, strand and target position in one 64-bit key
        const int key_spare = 64 - (bits_for(nT > 1 ? nT - 1 : 1) + 1 + bits_for(max_tlen));
        q_cap = std::min<size_t>(q_cap, (size_t)1 << std::max(0, std::min(key_spare, 16)));
    }
    // wall clock of the pass's phases (no extra synchronisation: each phase ends drained)
    auto wall = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; };
    const double w_batches = wall();
    stat_add("wall_s.ava_index_plan", w_batches - w_start);
    // the batches of the pass: [q, hi) and their anchors
    struct Batch { size_t q, hi; uint64_t acc; };
    std::vector<Batch> batches;
    for (size_t q = 0; q < nQ;) {
        uint64_t acc = 0;
        size_t hi = q;
        while (hi < nQ && hi - q < q_cap && (hi == q || acc < batch_target) && (hi == q || acc + plan.per_query[hi] <= anchor_batch + anchor_batch / 4))
            acc += plan.per_query[hi++];
        batches.push_back(Batch{q, hi, acc});
        q = hi;
    }
    // every batch leaves its rows, its deferred pieces and its counts in a slot of its own: whatever order the lanes finish them
    // in, they are put together in batch order
    const size_t nb = batches.size();
    std::vector<std::vector<AlignOut>> parts_of(nb);
    std::vector<DeferredPieces> deferred_of(nb);
    std::vector<SeedStats> st_of(nb);
    auto run_batch = [&](size_t b) {
        ChainOut ch;
        {
            HostTimer ht("seed_and_chain");
            seed_and_chain(in, ix, o, plan, qlen.p, tlen.p, batches[b].q, batches[b].hi, ch, st_of[b]);
        }
        stat_add("pieces", (double)ch.n_pieces);
        stat_add("fixed_points", (double)ch.n_fp);
        if (ch.n_pieces) {
            HostTimer ht("align_pieces");
            align_pieces(in, o, qlen.p, tlen.p, ch, parts_of[b], &deferred_of[b]);
        }
    };
    size_t done = 0;
    bool projected_once = false;
    // one after the other up to the first batch with output (the projection of the run's size wants it)
    while (done < nb && !projected_once) {
        run_batch(done);
        const auto &first_parts = parts_of[done];
        const uint64_t acc = batches[done].acc;
        ++done;
        if (first_parts.empty()) continue;
        projected_once = true;
        if (in.max_out_bytes && in.n_chunks > 1 && acc) {
            size_t rows = 0, ops = 0;
            for (const auto &p : first_parts) { rows += p.n_rows; ops += p.n_ops; }
            const double projected = (double)ava_out_bytes(rows, ops) / (double)acc * (double)total_anchors;
            if (projected > (double)in.max_out_bytes) {
                out.refused_anchors = total_anchors;
                out.refused_shrink = 0.8 * (double)in.max_out_bytes / projected;
                return;
            }
        }
    }
    // The set-aside pieces (LONG tasks: few waves, each busy for milliseconds) of the batches [b0, b1) aligned together.  With
    // lanes, the batches are cut into groups and the lane that finishes a group's last batch aligns the group's set at once,
    // beside the other lane's batches; only the last group's set is left for the end of the pass (a C3 step spent 41 ms there
    // with the device nearly idle).
    std::vector<std::vector<AlignOut>> parts_late;          // one per group, in group order
    auto align_set_aside = [&](size_t b0, size_t b1, std::vector<AlignOut> &outs) {
        ChainOut dc;
        for (size_t b = b0; b < b1; ++b) { dc.n_pieces += deferred_of[b].n_pieces; dc.n_fp += deferred_of[b].n_fp; }
        if (!dc.n_pieces) return;
        if (dc.n_fp >= (1ull << 32)) fail(HLMI_EINVAL, "more than 2^32 fixed points in the set-aside alignment pieces");
        HostTimer ht("align_deferred");
        dc.pieces.alloc(dc.n_pieces);
        dc.fps.alloc(dc.n_fp);
        size_t p0 = 0, f0 = 0;
        for (size_t b = b0; b < b1; ++b) {
            for (auto &part : deferred_of[b].parts) {
                HIP_CHECK(hipMemcpyAsync(dc.pieces.p + p0, part.pieces.p, part.n_pieces * sizeof(Piece), hipMemcpyDeviceToDevice, stream()));
                HIP_CHECK(hipMemcpyAsync(dc.fps.p + f0, part.fps.p, part.n_fp * sizeof(FixPt), hipMemcpyDeviceToDevice, stream()));
                if (f0 != part.fp_base)       // (Piece::fp_off counts from the start of the batch's own set)
                    hipLaunchKernelGGL(shift_fp_off_kernel, grid1(part.n_pieces), dim3(WG), 0, stream(), dc.pieces.p + p0, part.n_pieces,
                                       (uint32_t)(f0 - part.fp_base));
                p0 += part.n_pieces; f0 += part.n_fp;
            }
        }
        HIP_CHECK(hipGetLastError());
        sync();
        for (size_t b = b0; b < b1; ++b) deferred_of[b].parts.clear();
        align_pieces(in, o, qlen.p, tlen.p, dc, outs);
    };
    // the rest in lanes: each lane a host thread with a stream of its own, taking the next batch when it is done with one
    std::vector<size_t> cut{0};                            // group g = batches [cut[g], cut[g + 1])
    {
        const int n_lanes = (int)std::min<size_t>((size_t)(in.max_lanes > 0 ? std::min(in.max_lanes, lane_count()) : lane_count()), nb - done);
        if (n_lanes > 1) {
            std::vector<size_t> pcts{60};
            if (const char *e = hook("HLMI_SET_ASIDE_CUTS")) {               // tuning hook: "50,85"; "" = no groups
                pcts.clear();
                size_t v = 0;
                bool have = false;
                for (const char *c = e; *c && pcts.size() < 16; ++c) {       // digits and commas; anything else ends the list
                    if (*c >= '0' && *c <= '9') { v = std::min<size_t>(v * 10 + (size_t)(*c - '0'), 100); have = true; }
                    else { if (have) pcts.push_back(v); v = 0; have = false; if (*c != ',') break; }
                }
                if (have && pcts.size() < 16) pcts.push_back(v);
            }
            for (const size_t pct : pcts) {
                const size_t c = done + (nb - done) * std::min<size_t>(pct, 100) / 100;
                if (c > done && c > cut.back() && c < nb) cut.push_back(c);      // (every group has a batch the lanes still run)
            }
        }
        cut.push_back(nb);
        const size_t n_groups = cut.size() - 1;
        parts_late.resize(n_groups);
        std::vector<std::atomic<size_t>> left(n_groups);
        for (size_t g = 0; g < n_groups; ++g) left[g].store(cut[g + 1] - std::max(cut[g], done));
        std::atomic<size_t> next(done);
        std::atomic<bool> stop(false);
        std::exception_ptr errors[MAX_LANES];
        auto work = [&](int lane_id) {
            try {
                std::unique_ptr<LaneScope> scope;
                if (lane_id) scope.reset(new LaneScope(lane_id));
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= nb || stop.load()) break;
                    run_batch(b);
                    sync();                                  // (another lane may read what this batch set aside)
                    const size_t g = (size_t)(std::upper_bound(cut.begin(), cut.end(), b) - cut.begin()) - 1;
                    if (left[g].fetch_sub(1) == 1 && g + 1 < n_groups) align_set_aside(cut[g], cut[g + 1], parts_late[g]);
                }
                if (lane_id) sync();
            } catch (...) {
                errors[lane_id] = std::current_exception();
                stop.store(true);
            }
        };
        std::vector<std::thread> workers;
        try {
            for (int l = 1; l < n_lanes; ++l) workers.emplace_back(work, l);
        } catch (...) {                                   // no thread to be had: the caller's lane does what the others would have
            stat_add("ava_lane_threads_refused", 1);
        }
        work(0);
        for (auto &w : workers) w.join();
        for (int l = 0; l < MAX_LANES; ++l) if (errors[l]) std::rethrow_exception(errors[l]);
        if (n_lanes > 1) stat_set("ava_lanes", (double)n_lanes);
    }
    const double w_tail = wall();
    stat_add("wall_s.ava_batches", w_tail - w_batches);
    align_set_aside(cut[cut.size() - 2], nb, parts_late.back());      // the last group's (without lanes: everything)
    stat_add("wall_s.ava_set_aside_tail", wall() - w_tail);
    for (size_t b = 0; b < nb; ++b) {
        for (auto &p : parts_of[b]) parts.push_back(std::move(p));
        st.anchors += st_of[b].anchors; st.groups += st_of[b].groups;
    }
    for (auto &late : parts_late) for (auto &p : late) parts.push_back(std::move(p));
    parts_of.clear(); deferred_of.clear(); parts_late.clear();
    stat_add("index_entries", (double)ix.n);
    stat_add("anchors", (double)st.anchors);
    stat_add("chain_groups", (double)st.groups);

    // ---- concatenate + stream order ---------------------------------------------------------------------
    const double w_concat = wall();
    HostTimer ht_concat("concat_order");
    HostTimer *ht_part = new HostTimer("concat_alloc");
    size_t R = 0, E = 0;
    for (auto &p : parts) { R += p.n_rows; E += p.n_ops; }
    stat_add("ava_rows", (double)R);
    stat_add("cigar_ops", (double)E);
    if (!R) return;
    DBuf<PafRec> recs(R);
    DBuf<uint64_t> hi64(R), lo64(R);
    uintptr_t base_addr = ~(uintptr_t)0;
    for (auto &p : parts) if (p.n_ops) base_addr = std::min(base_addr, (uintptr_t)p.ops.p);
    if (base_addr == ~(uintptr_t)0) base_addr = 0;
    out.ops_base = (const uint32_t *)base_addr;
    delete ht_part; ht_part = new HostTimer("concat_copy");
    size_t r0 = 0;
    for (auto &p : parts) {
        HIP_CHECK(hipMemcpyAsync(recs.p + r0, p.recs.p, p.n_rows * sizeof(PafRec), hipMemcpyDeviceToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(hi64.p + r0, p.ord_hi.p, p.n_rows * 8, hipMemcpyDeviceToDevice, stream()));
        HIP_CHECK(hipMemcpyAsync(lo64.p + r0, p.ord_lo.p, p.n_rows * 8, hipMemcpyDeviceToDevice, stream()));
        const uint64_t shift = p.n_ops ? (uint64_t)(((uintptr_t)p.ops.p - base_addr) / 4) : 0;     // the part's ops stay in place
        if (shift) hipLaunchKernelGGL(shift_cigar_kernel, grid1(p.n_rows), dim3(WG), 0, stream(), recs.p + r0, p.n_rows, shift);
        r0 += p.n_rows;
        out.ops_part_len.push_back(p.n_ops);
        out.ops_parts.push_back(std::move(p.ops));
    }
    sync();
    delete ht_part; ht_part = new HostTimer("concat_free_parts");
    parts.clear();
    delete ht_part; ht_part = new HostTimer("concat_order_sort");
    DBuf<uint32_t> perm(R);
    hipLaunchKernelGGL(iota_kernel, grid1(R), dim3(WG), 0, stream(), perm.p, R);
    sort_pairs_u64_u32(lo64.p, perm.p, R);
    DBuf<uint64_t> hi_g(R);
    hipLaunchKernelGGL(gather_u64_kernel, grid1(R), dim3(WG), 0, stream(), hi64.p, perm.p, hi_g.p, R);
    sort_pairs_u64_u32(hi_g.p, perm.p, R);
    out.recs.alloc(R);
    hipLaunchKernelGGL(gather_rows_kernel, grid1(R), dim3(WG), 0, stream(), recs.p, perm.p, out.recs.p, R);
    HIP_CHECK(hipGetLastError());
    delete ht_part; ht_part = new HostTimer("concat_download");
    // rows of chunk c: [first row whose key's chunk field is >= c, ...) - found on the device (the keys themselves, 8 bytes per
    // row, took 21 ms of a C3 step to bring over)
    DBuf<uint64_t> d_start(in.n_chunks + 1);
    hipLaunchKernelGGL(chunk_row_start_kernel, grid1(in.n_chunks + 1), dim3(WG), 0, stream(), hi_g.p, R, in.n_chunks, d_start.p);
    HIP_CHECK(hipGetLastError());
    const std::vector<uint64_t> h_start = d_start.download(in.n_chunks + 1);
    delete ht_part;
    out.n_rows = R;
    out.n_ops = E;
    for (uint32_t c = 0; c < in.n_chunks; ++c) out.chunk_row_start[c] = (size_t)h_start[c];
    out.chunk_row_start[in.n_chunks] = R;
    stat_add("wall_s.ava_concat", wall() - w_concat);
}

void format_ava_row(const PafRec &r, const uint32_t *ops, const std::string &qname, const std::string &tname,
                    std::string &out) {
    char buf[256];
    out.assign(qname);
    int m = snprintf(buf, sizeof buf, "\t%u\t%u\t%u\t%c\t", r.qlen, r.qs, r.qe, (r.flags & PF_REV) ? '-' : '+');
    out.append(buf, m);
    out.append(tname);
    m = snprintf(buf, sizeof buf, "\t%u\t%u\t%u\t%u\t%u\t0\tNM:i:%u\ttp:A:S\tcg:Z:", r.tlen, r.ts, r.te, r.nmatch, r.blen,
                 r.blen - r.nmatch);
    out.append(buf, m);
    static const char opc[16] = {'?', 'I', 'D', '?', '?', '?', '?', '=', 'X', '?', '?', '?', '?', '?', '?', '?'};
    for (uint32_t i = 0; i < r.cig_n; ++i) {
        m = snprintf(buf, sizeof buf, "%u%c", ops[i] >> 4, opc[ops[i] & 15]);
        out.append(buf, m);
    }
    if (!r.cig_n) out.push_back('*');          // the bare row of a stub candidate (hlmi_ava_opts::stub_oh)
}

void ava_files(const char *target_fa, const char *query_fa, const hlmi_ava_opts &o, const char *out_paf) {
    stat_reset();
    SeqSet T, Q;
    read_seqs(target_fa, T);
    read_seqs(query_fa, Q);
    std::vector<uint32_t> rt, rq;
    std::vector<std::string> name_of_rank;
    name_ranks(T.names, Q.names, rt, rq, name_of_rank);
    DevReads dT, dQ;
    upload_reads(T, 0, T.size(), dT);
    upload_reads(Q, 0, Q.size(), dQ);
    DBuf<uint32_t> d_rt, d_rq, d_chunk(T.size() ? T.size() : 1);
    d_rt.upload(rt);
    d_rq.upload(rq);
    if (rt.empty()) d_rt.alloc(1);
    if (rq.empty()) d_rq.alloc(1);
    d_chunk.zero();
    DevSketch qsk;
    sketch_device(dQ, o.k, o.w, o.hpc, 0, qsk);
    std::vector<uint32_t> qc = qsk.counts.download(Q.size());
    AvaInput in;
    in.T = &dT; in.Q = &dQ; in.d_rank_t = d_rt.p; in.d_rank_q = d_rq.p; in.d_chunk_of_t = d_chunk.p; in.n_chunks = 1;
    in.n_ranks = name_of_rank.size();
    in.d_qmz = qsk.mz.p;
    in.qmz_off.assign(Q.size() + 1, 0);
    for (size_t i = 0; i < Q.size(); ++i) in.qmz_off[i + 1] = in.qmz_off[i] + qc[i];
    AvaRows rows;
    ava_device(in, o, rows);
    ktimer_flush();
    std::vector<PafRec> hr = rows.recs.download(rows.n_rows);
    // the parts of the CIGAR array, on the host one after the other: a row's offset from ops_base -> its place there
    std::vector<uint32_t> hops;
    std::vector<std::pair<uint64_t, uint64_t>> where;          // (first word of the part counted from ops_base, its place in hops)
    for (size_t k = 0; k < rows.ops_parts.size(); ++k) {
        if (!rows.ops_part_len[k]) continue;
        where.emplace_back((uint64_t)(rows.ops_parts[k].p - rows.ops_base), hops.size());
        const std::vector<uint32_t> part = rows.ops_parts[k].download(rows.ops_part_len[k]);
        hops.insert(hops.end(), part.begin(), part.end());
    }
    std::sort(where.begin(), where.end());
    std::vector<std::string> lines(rows.n_rows);
    static const uint32_t no_ops[1] = {0};
    for (size_t i = 0; i < rows.n_rows; ++i) {
        const uint32_t *ops = no_ops;
        if (hr[i].cig_n) {
            auto it = std::upper_bound(where.begin(), where.end(), std::pair<uint64_t, uint64_t>(hr[i].cig_off, ~(uint64_t)0));
            --it;
            ops = hops.data() + it->second + (hr[i].cig_off - it->first);
        }
        format_ava_row(hr[i], ops, name_of_rank[hr[i].qid], name_of_rank[hr[i].tid], lines[i]);
    }
    write_lines(out_paf, lines);
}

}  // namespace hlmi
