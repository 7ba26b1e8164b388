// paf_io.cpp - host text I/O (see paf_io.h).  Nothing here is on the timed device path; it is
// the text boundary the reference pipes between its processes.
#include "paf_io.h"
#include <atomic>

#include <functional>

#include <unistd.h>

#include <sys/stat.h>

#include <sys/mman.h>

#include <fcntl.h>

#include <cmath>

#include <charconv>

#include <algorithm>
#include <thread>
#include <cerrno>
#include <cstring>
#include <cstdlib>
#include <numeric>

namespace hlmi {

uint32_t NameDict::put(std::string_view s) {
    auto it = index.find(std::string(s));
    if (it != index.end()) return it->second;
    uint32_t id = (uint32_t)names.size();
    names.emplace_back(s);
    index.emplace(names.back(), id);
    return id;
}

std::string read_file(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) fail(HLMI_EIO, "cannot open %s: %s", path, strerror(errno));
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

void write_lines(const char *path, const std::vector<std::string> &lines) {
    std::vector<std::string_view> v(lines.begin(), lines.end());
    write_lines(path, v);
}
void write_lines(const char *path, const std::vector<std::string_view> &lines) {
    // A regular-file target is written next to itself under a name of this process's own and renamed on success: a short
    // write (ENOSPC, quota) never leaves a truncated file under the final name for the next tool of the pipeline to read,
    // and two writers of one target do not share a temporary.  Anything else (/dev/stdout, a FIFO) is written directly.
    struct stat st;
    const bool direct = stat(path, &st) == 0 && !S_ISREG(st.st_mode);
    std::string tmp = path;
    int fd = -1;
    if (direct) {
        fd = open(path, O_WRONLY);
    } else {
        tmp += ".XXXXXX";
        fd = mkstemp(&tmp[0]);
        if (fd >= 0) fchmod(fd, 0666 & ~[] { const mode_t m = umask(0); umask(m); return m; }());
    }
    if (fd < 0) fail(HLMI_EIO, "cannot write %s: %s", tmp.c_str(), strerror(errno));
    // Regular file: the size is known, so the blocks are reserved (a full disk is reported here, as an error), the file is
    // mapped and the lines are copied straight into the page cache by a few threads - a third of the time of write() from a
    // staging buffer for the millions of rows of a short-read call.  Anything that cannot be mapped (and every special file)
    // takes the plain loop: 8 MB buffers, write().
    const size_t n = lines.size();
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(std::min(host_threads(), 4), n / 65536));
    std::vector<size_t> first(nt + 1), off(nt + 1, 0);
    for (int t = 0; t <= nt; ++t) first[t] = n * (size_t)t / (size_t)nt;
    auto each_span = [&](auto &&fn) {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(fn, t);
        fn(0);
        for (auto &th : pool) th.join();
    };
    each_span([&](int t) { size_t b = 0; for (size_t i = first[t]; i < first[t + 1]; ++i) b += lines[i].size() + 1; off[t + 1] = b; });
    for (int t = 0; t < nt; ++t) off[t + 1] += off[t];
    const size_t total = off[nt];
    std::atomic<int> err{0};                     // errno of the first failing call
    bool done = total == 0;
    if (!direct && total) {
        const int fe = posix_fallocate(fd, 0, (off_t)total);
        if (fe == ENOSPC || fe == EFBIG || fe == EDQUOT || fe == EIO) err = fe;
        else if (fe == 0) {
            void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) {
                each_span([&](int t) {
                    char *p = (char *)m + off[t];
                    for (size_t i = first[t]; i < first[t + 1]; ++i) {
                        memcpy(p, lines[i].data(), lines[i].size());
                        p += lines[i].size();
                        *p++ = '\n';
                    }
                });
                if (munmap(m, total) != 0) err = errno ? errno : EIO;
                done = true;
            } else if (ftruncate(fd, 0) != 0) err = errno ? errno : EIO;      // (back to an empty file for the plain loop)
        }
    }
    if (!done && !err.load()) {
        std::string buf;
        buf.reserve(8u << 20);
        auto flush = [&]() {
            const char *p = buf.data();
            size_t left = buf.size();
            while (left && !err.load()) {
                const ssize_t w = write(fd, p, left);
                if (w < 0) { if (errno == EINTR) continue; err = errno ? errno : EIO; break; }
                p += w; left -= (size_t)w;
            }
            buf.clear();
        };
        for (auto &l : lines) {
            buf.append(l);
            buf.push_back('\n');
            if (buf.size() > (7u << 20)) flush();
        }
        flush();
    }
    int e = err.load();
    if (close(fd) != 0 && !e) e = errno ? errno : EIO;
    if (e) {
        if (!direct) remove(tmp.c_str());
        fail(HLMI_EIO, "write error on %s: %s", tmp.c_str(), strerror(e));
    }
    if (!direct && rename(tmp.c_str(), path) != 0) {
        const int e2 = errno;
        remove(tmp.c_str());
        fail(HLMI_EIO, "cannot rename %s to %s: %s", tmp.c_str(), path, strerror(e2));
    }
}

static bool parse_u32(std::string_view s, uint32_t &v) {
    if (s.empty() || s.size() > 10) return false;
    uint64_t x = 0;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        x = x * 10 + (uint64_t)(c - '0');
    }
    if (x > 0xffffffffull) return false;
    v = (uint32_t)x;
    return true;
}

void read_paf(const char *path, PafText &out, bool need_tie_rank) {
    out.data = read_file(path);
    const std::string &d = out.data;
    size_t pos = 0, N = d.size();
    while (pos < N) {
        size_t e = d.find('\n', pos);
        if (e == std::string::npos) e = N;
        out.line_off.push_back(pos);
        out.line_len.push_back((uint32_t)(e - pos));
        pos = e + 1;
    }
    size_t n = out.line_off.size();
    out.recs.resize(n);
    for (size_t i = 0; i < n; ++i) {
        std::string_view L = out.line(i);
        PafRec r{};
        // split into fields
        size_t fs[12];
        size_t nf = 0, p = 0, last_start = 0;
        while (true) {
            if (nf < 12) fs[nf] = p;
            last_start = p;
            ++nf;
            size_t t = L.find('\t', p);
            if (t == std::string_view::npos) break;
            p = t + 1;
        }
        if (nf < 11) fail(HLMI_EINVAL, "%s:%zu: PAF row has %zu columns (< 11)", path, i + 1, nf);
        auto field = [&](int k) {
            size_t b = fs[k];
            size_t e2 = (k + 1 < (int)nf && k + 1 < 12) ? fs[k + 1] - 1 : L.find('\t', b);
            if (e2 == std::string_view::npos) e2 = L.size();
            return L.substr(b, e2 - b);
        };
        uint32_t *dst[] = {&r.qlen, &r.qs, &r.qe, nullptr, nullptr, &r.tlen, &r.ts, &r.te, &r.nmatch, &r.blen};
        for (int k = 1; k <= 10; ++k) {
            if (!dst[k - 1]) continue;
            if (!parse_u32(field(k), *dst[k - 1]))
                fail(HLMI_EINVAL, "%s:%zu: column %d is not an unsigned integer", path, i + 1, k + 1);
        }
        r.qid = out.dict.put(field(0));
        r.tid = out.dict.put(field(5));
        std::string_view strand = field(4);
        r.flags = (strand == "+") ? 0u : PF_REV;   // the reference tests `qori == '-'` / `flag == "+"`
        if (strand != "+" && strand != "-") fail(HLMI_EINVAL, "%s:%zu: strand must be + or -", path, i + 1);
        // last field: CIGAR (filter_overlap_slr2.py:312 reads len_sp[-1])
        std::string_view lf = L.substr(last_start);
        r.cig_off = out.ops.size();
        if (lf == "*") {
            r.flags |= PF_STAR;
        } else if (lf.size() > 5 && lf.substr(0, 5) == "cg:Z:") {
            uint64_t num = 0;
            bool have = false;
            for (size_t q = 5; q < lf.size(); ++q) {
                char c = lf[q];
                if (c >= '0' && c <= '9') {
                    num = num * 10 + (uint64_t)(c - '0');
                    have = true;
                    if (num >= (1ull << 28)) fail(HLMI_EINVAL, "%s:%zu: CIGAR op too long", path, i + 1);
                } else {
                    if (!have) fail(HLMI_EINVAL, "%s:%zu: malformed cg:Z: field", path, i + 1);
                    uint32_t code = c == '=' ? OP_EQ : c == 'X' ? OP_X : c == 'I' ? OP_I : c == 'D' ? OP_D : OP_OTHER;
                    out.ops.push_back((uint32_t)num << 4 | code);
                    num = 0;
                    have = false;
                }
            }
        }
        r.cig_n = (uint32_t)(out.ops.size() - r.cig_off);
        r.chunk = 0;
        r.tie = (uint32_t)i;
        out.recs[i] = r;
    }
    if (need_tie_rank && n) {
        std::vector<uint32_t> idx(n);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return out.line(a) < out.line(b); });
        for (size_t k = 0; k < n; ++k) out.recs[idx[k]].tie = (uint32_t)k;
    }
}

// ------------------------------------------------------------------------------------------
// FASTA / FASTQ
// ------------------------------------------------------------------------------------------
namespace {
// read-only view of a whole file: mapped when the system allows it (no copy of a read set of hundreds of MB),
// else read into memory
struct FileView {
    const char *p = nullptr;
    size_t n = 0;
    bool mapped = false;
    std::string owned;
    explicit FileView(const char *path) {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) fail(HLMI_EIO, "cannot open %s: %s", path, strerror(errno));
        struct stat st;
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { p = (const char *)m; n = (size_t)st.st_size; mapped = true; }
        }
        close(fd);
        if (!mapped) { owned = read_file(path); p = owned.data(); n = owned.size(); }
    }
    ~FileView() { if (mapped) munmap((void *)p, n); }
    FileView(const FileView &) = delete;
    FileView &operator=(const FileView &) = delete;
};
}  // namespace

void read_seqs(const char *path, SeqSet &out) { read_seqs_subset(path, nullptr, out); }

void read_seqs_subset(const char *path, const std::function<bool(std::string_view)> *want, SeqSet &out) {
    const FileView file(path);
    const std::string_view d(file.p, file.n);
    size_t N = d.size(), pos = 0;
    uint32_t line_no = 0;
    out.off.push_back(0);
    auto next_line = [&](std::string_view &L) -> bool {
        if (pos >= N) return false;
        size_t e = d.find('\n', pos);
        if (e == std::string::npos) e = N;
        size_t end = e;
        if (end > pos && d[end - 1] == '\r') --end;
        L = d.substr(pos, end - pos);
        pos = e + 1;
        ++line_no;
        return true;
    };
    out.n_lines = want ? 0 : (uint64_t)std::count(d.begin(), d.end(), '\n');     // (only the stage's chunking asks for it)
    if (!want) {                                     // one allocation instead of a gigabyte grown by doubling
        out.bases.reserve(N);
        out.names.reserve(out.n_lines / 2 + 1);
        out.off.reserve(out.n_lines / 2 + 2);
        out.first_line.reserve(out.n_lines / 2 + 1);
    }
    std::string_view L;
    bool have = next_line(L);
    while (have) {
        if (L.empty() || (L[0] != '>' && L[0] != '@')) {  // kseq skips to the next header
            have = next_line(L);
            continue;
        }
        bool fq = L[0] == '@';
        size_t ws = L.find_first_of(" \t", 1);
        const std::string_view name = L.substr(1, (ws == std::string_view::npos ? L.size() : ws) - 1);
        const bool keep = !want || (*want)(name);             // a record that is not wanted keeps its name, with no bases
        out.names.emplace_back(name);
        out.first_line.push_back(line_no - 1);
        size_t slen = 0;
        have = next_line(L);
        while (have && !(L.size() && (L[0] == '>' || L[0] == '@' || L[0] == '+'))) {
            if (keep) out.bases.append(L);
            slen += L.size();
            have = next_line(L);
        }
        if (fq && have && L.size() && L[0] == '+') {
            size_t q = 0;
            have = next_line(L);
            while (have && q < slen) {
                q += L.size();
                have = next_line(L);
            }
        }
        out.off.push_back(out.bases.size());
    }
}

// ------------------------------------------------------------------------------------------
// final rows
// ------------------------------------------------------------------------------------------
// "%.4f" of v into dst (at least 64 bytes), the bytes printf writes.  A finite v in [0, 2^40) - every score of a row - is
// converted exactly in integers: v = m x 2^e, so v x 10^4 = (m x 10^4) >> -e, rounded to nearest, ties to even (the
// rounding glibc's printf applies to the exact binary value in the default rounding mode); three conversions per row were
// most of the stage's text time with printf and still a third of it with std::to_chars.  Anything else goes through
// std::to_chars / printf.
size_t format_fixed4(double v, char *dst) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const int be = (int)((bits >> 52) & 0x7ff);
    if (!(bits >> 63) && be != 0x7ff && v < 1099511627776.0) {
        uint64_t m = bits & ((1ull << 52) - 1);
        int e = be - 1075;                                   // v = m x 2^e
        if (be) m |= 1ull << 52; else e = -1074;
        uint64_t q;
        if (e >= 0) q = (m << e) * 10000ull;                 // (an integer below 2^40)
        else {
            const unsigned __int128 N = (unsigned __int128)m * 10000u;      // < 2^67
            const int sh = -e;
            if (sh > 68) q = 0;                              // below a half: rounds to zero
            else {
                q = (uint64_t)(N >> sh);
                const unsigned __int128 rem = N & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
                if (rem > half || (rem == half && (q & 1))) ++q;
            }
        }
        const uint64_t ip = q / 10000u;
        uint32_t fp = (uint32_t)(q % 10000u);
        auto res = std::to_chars(dst, dst + 24, ip);
        char *p = res.ptr;
        *p++ = '.';
        p[3] = (char)('0' + fp % 10); fp /= 10;
        p[2] = (char)('0' + fp % 10); fp /= 10;
        p[1] = (char)('0' + fp % 10); fp /= 10;
        p[0] = (char)('0' + fp);
        return (size_t)(p + 4 - dst);
    }
    if (std::isfinite(v)) {
        auto res = std::to_chars(dst, dst + 48, v, std::chars_format::fixed, 4);
        if (res.ec == std::errc()) return (size_t)(res.ptr - dst);
    }
    return (size_t)snprintf(dst, 64, "%.4f", v);
}

bool format_scored_row(const PafRec &r, const std::string &qname, const std::string &tname,
                       uint32_t x_digit_sum, double iden, std::string &out, uint32_t *sort_key) {
    // filter_overlap_slr2.py:113,138-146 - plain IEEE doubles in the reference's evaluation order
    double mc = (double)r.nmatch, ln = (double)r.blen;
    double mlen = (double)((uint64_t)r.qlen + r.tlen) / 2.0;
    double t1 = mc / mlen, t2 = mc / ln;
    double a = 0.4 * t1, b = 0.6 * t2;
    double score = a + b;
    double mis = (double)x_digit_sum / mc;
    double score2 = 1.0 - mis;
    // "%.4f" (format_fixed4).  float("0.9876") is the correctly rounded 9876 / 10^4, which is what the division below gives.
    auto f4 = [](double v, char *dst) -> size_t { return format_fixed4(v, dst); };
    char s1[64], s2[64], s3[64];
    const size_t n1 = f4(score, s1), n2 = f4(score2, s2), n3 = f4(t2, s3);
    {
        double shown;
        const char *p = s2;
        const bool neg = *p == '-';
        if (neg) ++p;
        uint64_t digits = 0;
        size_t nd = 0;
        bool plain = std::isfinite(score2);
        for (; plain && p < s2 + n2; ++p) {
            if (*p == '.') continue;
            if (*p < '0' || *p > '9' || ++nd > 18) plain = false;
            else digits = digits * 10 + (uint64_t)(*p - '0');
        }
        if (plain && digits < (1ull << 53)) { shown = (double)digits / 10000.0; if (neg) shown = -shown; }
        else { s2[n2] = 0; shown = strtod(s2, nullptr); }
        if (shown < iden) return false;
    }
    auto put_u32 = [&](uint32_t v) {
        char b[16];
        auto res = std::to_chars(b, b + sizeof b, v);
        out.append(b, (size_t)(res.ptr - b));
    };
    out.assign(qname);
    out.push_back('\t'); put_u32(r.qlen);
    out.push_back('\t'); put_u32(r.qs);
    out.push_back('\t'); put_u32(r.qe);
    out.push_back('\t'); out.push_back((r.flags & PF_REV) ? '-' : '+');
    out.push_back('\t'); out.append(tname);
    out.push_back('\t'); put_u32(r.tlen);
    out.push_back('\t'); put_u32(r.ts);
    out.push_back('\t'); put_u32(r.te);
    out.push_back('\t'); put_u32(r.nmatch);
    out.push_back('\t'); put_u32(r.blen);
    out.push_back('\t'); out.append(s1, n1);
    out.push_back('\t'); out.append(s2, n2);
    out.push_back('\t'); out.append(s3, n3);
    out.push_back('\t');
    if (sort_key) {                              // column 12 as sort_scored_lines reads it: value x 10^4 when it is plain digits
        uint64_t v = 0;
        size_t frac = 0;
        bool plain = n1 > 0, dot = false;
        for (size_t i = 0; plain && i < n1; ++i) {
            if (s1[i] == '.' && !dot) dot = true;
            else if (s1[i] >= '0' && s1[i] <= '9' && v < (1ull << 40)) { v = v * 10 + (uint64_t)(s1[i] - '0'); frac += dot ? 1 : 0; }
            else plain = false;
        }
        *sort_key = plain && dot && frac == 4 && v < SCORE_KEY_LIMIT ? (uint32_t)v : 0xffffffffu;
    }
    return true;
}

namespace {
// exact view of the number GNU `sort -n` reads at p: sign, integer digits, fraction digits
struct NumKey {
    bool neg;
    std::string_view ip, fp;
};
NumKey gnu_num(std::string_view s) {
    size_t p = 0;
    while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) ++p;
    NumKey k{false, {}, {}};
    if (p < s.size() && s[p] == '-') { k.neg = true; ++p; }
    size_t b = p;
    while (p < s.size() && s[p] >= '0' && s[p] <= '9') ++p;
    k.ip = s.substr(b, p - b);
    while (!k.ip.empty() && k.ip[0] == '0') k.ip.remove_prefix(1);
    if (p < s.size() && s[p] == '.') {
        ++p;
        b = p;
        while (p < s.size() && s[p] >= '0' && s[p] <= '9') ++p;
        k.fp = s.substr(b, p - b);
        while (!k.fp.empty() && k.fp.back() == '0') k.fp.remove_suffix(1);
    }
    if (k.ip.empty() && k.fp.empty()) k.neg = false;  // -0 == 0
    return k;
}
int cmp_mag(const NumKey &a, const NumKey &b) {
    if (a.ip.size() != b.ip.size()) return a.ip.size() < b.ip.size() ? -1 : 1;
    int c = a.ip.compare(b.ip);
    if (c) return c < 0 ? -1 : 1;
    c = a.fp.compare(b.fp);   // digit strings without trailing zeros: lexicographic == numeric
    return c < 0 ? -1 : c > 0 ? 1 : 0;
}
std::string_view field_tail(std::string_view L, int k) {
    size_t p = 0;
    for (int i = 1; i < k; ++i) {
        size_t t = L.find('\t', p);
        if (t == std::string_view::npos) return {};
        p = t + 1;
    }
    return L.substr(p);
}
}  // namespace

void sort_scored_lines(std::vector<std::string> &lines) {
    std::vector<std::string_view> v(lines.begin(), lines.end());
    sort_scored_lines(v);
    std::vector<std::string> out;
    out.reserve(lines.size());
    for (std::string_view x : v) out.emplace_back(x);
    lines.swap(out);
}
void sort_scored_lines(std::vector<std::string_view> &lines, const std::vector<uint32_t> *keys) {
    const size_t n = lines.size();
    std::vector<uint32_t> idx(n);
    // The rows this library writes carry "%.4f" scores: non-negative, few integer digits, at most 4 decimals.  Such a
    // key is an integer (value x 10^4): a counting sort on it, then the runs of equal scores by the text rule (the
    // 8-byte head of a line settles nearly every tie without touching the strings; memcmp order = big-endian integer
    // order).  Parsing and the tie runs go over the host threads.  Anything else in the column takes the general
    // comparator.  `keys` (the stage: format_scored_row hands them out) spares the parse of column 12.
    struct Rec { uint32_t score, idx; uint64_t head; };
    constexpr uint32_t SCORE_LIMIT = SCORE_KEY_LIMIT;         // 419.4304
    std::vector<Rec> rec(n);
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), n / 4096));
    std::vector<uint8_t> bad(nt, 0);
    auto each_slice = [&](auto &&fn) {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(fn, t);
        fn(0);
        for (auto &th : pool) th.join();
    };
    const bool have_keys = keys && keys->size() == n;
    each_slice([&](int t) {
        for (size_t i = n * (size_t)t / nt; i < n * (size_t)(t + 1) / nt; ++i) {
            uint64_t v;
            if (have_keys) {
                v = (*keys)[i];
            } else {
                const NumKey k = gnu_num(field_tail(lines[i], 12));
                v = 0;
                for (char c : k.ip) v = v * 10 + (uint64_t)(c - '0');
                for (size_t d = 0; d < 4; ++d) v = v * 10 + (d < k.fp.size() ? (uint64_t)(k.fp[d] - '0') : 0u);
                if (k.neg || k.ip.size() > 4 || k.fp.size() > 4) v = SCORE_LIMIT;
            }
            if (v >= SCORE_LIMIT) { bad[t] = 1; return; }
            uint64_t h = 0;
            for (size_t d = 0; d < 8; ++d) h = h << 8 | (d < lines[i].size() ? (uint64_t)(unsigned char)lines[i][d] : 0u);
            rec[i] = Rec{(uint32_t)v, (uint32_t)i, h};
        }
    });
    bool simple = true;
    for (uint8_t x : bad) simple = simple && !x;
    if (simple && n) {
        uint32_t hi = 0;
        for (const Rec &r : rec) hi = std::max(hi, r.score);
        std::vector<uint32_t> start(hi + 2, 0);               // descending: slot of score s = #records with a larger score
        for (const Rec &r : rec) ++start[hi - r.score + 1];
        for (uint32_t v = 1; v <= hi + 1; ++v) start[v] += start[v - 1];
        std::vector<Rec> byscore(n);
        {
            std::vector<uint32_t> at(start.begin(), start.end() - 1);
            for (const Rec &r : rec) byscore[at[hi - r.score]++] = r;
        }
        auto less = [&](const Rec &a, const Rec &b) {
            if (a.head != b.head) return a.head > b.head;
            return lines[a.idx] > lines[b.idx];
        };
        // Tie runs.  Scores of one read set crowd into a few hundred values (C4s: 3.5 M rows), so the runs are handed out by
        // ROWS, not by score values: thread t takes the runs that start in its n / nt rows; a run longer than that is sorted
        // afterwards by all threads together (pieces, then merges).
        const size_t big = std::max<size_t>(n / (size_t)nt, 1u << 16);
        std::vector<std::pair<uint32_t, uint32_t>> big_runs;
        for (uint32_t v = 0; v <= hi; ++v)
            if ((size_t)(start[v + 1] - start[v]) > big && nt > 1) big_runs.emplace_back(start[v], start[v + 1]);
        each_slice([&](int t) {
            const uint32_t r0 = (uint32_t)(n * (size_t)t / nt), r1 = (uint32_t)(n * (size_t)(t + 1) / nt);
            // first score value whose run starts at or after r0
            uint32_t v = (uint32_t)(std::lower_bound(start.begin(), start.end() - 1, r0) - start.begin());
            for (; v <= hi && start[v] < r1; ++v) {
                const uint32_t lo = start[v], up = start[v + 1];
                if (up - lo > 1 && !((size_t)(up - lo) > big && nt > 1)) std::sort(byscore.begin() + lo, byscore.begin() + up, less);
            }
        });
        for (auto &run : big_runs) {
            const size_t lo = run.first, len = run.second - run.first;
            auto cut = [&](int k) { return lo + len * (size_t)k / (size_t)nt; };
            each_slice([&](int t) { std::sort(byscore.begin() + cut(t), byscore.begin() + cut(t + 1), less); });
            for (int w = 1; w < nt; w *= 2) {                 // merge neighbours: pieces [k, k + w) and [k + w, k + 2w)
                std::vector<std::thread> pool;
                for (int k = 0; k + w < nt; k += 2 * w)
                    pool.emplace_back([&, k, w] {
                        std::inplace_merge(byscore.begin() + cut(k), byscore.begin() + cut(k + w), byscore.begin() + cut(std::min(k + 2 * w, nt)), less);
                    });
                for (auto &th : pool) th.join();
            }
        }
        for (size_t i = 0; i < n; ++i) idx[i] = byscore[i].idx;
    } else {
        std::iota(idx.begin(), idx.end(), 0u);
        std::vector<NumKey> key(n);             // the numeric prefix of column 12.., parsed once per line
        for (size_t i = 0; i < n; ++i) key[i] = gnu_num(field_tail(lines[i], 12));
        std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            const NumKey &x = key[a], &y = key[b];
            int c;
            if (x.neg != y.neg) c = x.neg ? -1 : 1;
            else { c = cmp_mag(x, y); if (x.neg) c = -c; }
            if (c) return c > 0;               // -r: descending
            return lines[a] > lines[b];        // last resort, reversed as well
        });
    }
    std::vector<std::string_view> out;
    out.reserve(n);
    for (uint32_t i : idx) out.push_back(lines[i]);
    lines.swap(out);
}

}  // namespace hlmi
