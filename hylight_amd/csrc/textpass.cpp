// textpass.cpp - the line-oriented text passes either side of the hot path, restated from the
// reference's Python so that a C4-scale run (10^7 reads) does not spend minutes in per-line interpreter loops:
//   filter_non_atcg   script/utils.py:81-114
//   gfa2fa            script/HyLight.py:328-337
//   pick_up           script/HyLight.py:347-378
// Host-only (no GPU work): pure I/O.  Python's text-mode conventions are reproduced: universal newlines on input
// ("\r\n" and "\r" end a line and read as "\n"), str.strip() / str.split() white space, str.split(" ") on the
// single space, ASCII upper-casing; a multi-byte UTF-8 character counts as ONE character (it becomes one 'N').
#include <cstdio>
#include <string>
#include <string_view>
#include <unordered_set>
#include <vector>

#include "common.h"
#include "textpass.h"

namespace hlmi {
namespace {

std::string slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) fail(HLMI_EIO, "cannot open %s", path);
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

// Python text-mode iteration: one line per call, terminator translated to "\n" (has_nl), last line may lack it
struct Lines {
    std::string_view s;
    size_t pos = 0;
    explicit Lines(std::string_view text) : s(text) {}
    bool next(std::string_view &body, bool &has_nl) {
        if (pos >= s.size()) return false;
        size_t e = pos;
        while (e < s.size() && s[e] != '\n' && s[e] != '\r') ++e;
        body = s.substr(pos, e - pos);
        has_nl = e < s.size();
        pos = e;
        if (has_nl) pos += (s[e] == '\r' && e + 1 < s.size() && s[e + 1] == '\n') ? 2 : 1;
        return true;
    }
};

inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

std::string_view py_strip(std::string_view v) {
    size_t b = 0, e = v.size();
    while (b < e && py_space((unsigned char)v[b])) ++b;
    while (e > b && py_space((unsigned char)v[e - 1])) --e;
    return v.substr(b, e - b);
}

void py_split(std::string_view v, std::vector<std::string_view> &out) {
    out.clear();
    size_t i = 0;
    while (i < v.size()) {
        while (i < v.size() && py_space((unsigned char)v[i])) ++i;
        size_t b = i;
        while (i < v.size() && !py_space((unsigned char)v[i])) ++i;
        if (i > b) out.push_back(v.substr(b, i - b));
    }
}

inline std::string_view before(std::string_view v, char sep) {
    const size_t p = v.find(sep);
    return p == std::string_view::npos ? v : v.substr(0, p);
}

struct Out {
    FILE *f = nullptr;
    std::string path;
    std::string buf;
    explicit Out(const char *p, bool open_now) : path(p) { if (open_now) open(); }
    void open() {
        f = fopen(path.c_str(), "wb");
        if (!f) fail(HLMI_EIO, "cannot write %s", path.c_str());
    }
    void put(std::string_view v) {
        buf.append(v);
        if (buf.size() > (1u << 20)) flush();
    }
    void flush() {
        if (buf.empty()) return;
        if (!f) open();
        if (fwrite(buf.data(), 1, buf.size(), f) != buf.size()) fail(HLMI_EIO, "short write to %s", path.c_str());
        buf.clear();
    }
    ~Out() {
        if (!buf.empty() && !std::uncaught_exceptions()) flush();
        if (f) fclose(f);
    }
};

// re.sub(r'[^ATGCN\n]', "N", line.upper())
void sanitise(std::string_view body, bool has_nl, std::string &o) {
    o.clear();
    for (unsigned char c : body) {
        if (c >= 0x80) {                      // UTF-8: lead byte -> one 'N', continuation bytes belong to it
            if ((c & 0xc0) != 0x80) o.push_back('N');
            continue;
        }
        if (c >= 'a' && c <= 'z') c = (unsigned char)(c - 32);
        o.push_back((c == 'A' || c == 'T' || c == 'G' || c == 'C' || c == 'N') ? (char)c : 'N');
    }
    if (has_nl) o.push_back('\n');
}

}  // namespace

void filter_non_atcg_run(const char *fastx, const char *out_fa, bool fastq) {
    const std::string text = slurp(fastx);
    Out out(out_fa, true);
    Lines it(text);
    std::string_view body;
    bool nl;
    std::string tmp;
    for (uint64_t num = 0; it.next(body, nl); ++num) {
        if (fastq) {
            if (num % 4 == 1) { sanitise(body, nl, tmp); out.put(tmp); }
            else if (num % 4 == 0 && !body.empty() && body[0] == '@') {            // utils.py:99-103
                const std::string_view id = before(py_strip(body), ' ');
                out.put(">"); out.put(id.substr(1)); out.put("\n");
            }
        } else {
            if (num % 2 == 1) { sanitise(body, nl, tmp); out.put(tmp); }
            else { out.put(before(py_strip(body), ' ')); out.put("\n"); }          // utils.py:109-112
        }
    }
    out.flush();
}

void gfa2fa_run(const char *gfa, const char *out_fa) {
    const std::string text = slurp(gfa);
    Out out(out_fa, true);
    Lines it(text);
    std::string_view body;
    bool nl;
    std::vector<std::string_view> f;
    for (uint64_t num = 0; it.next(body, nl); ++num) {
        py_split(body, f);
        if (f.empty()) fail(HLMI_EINVAL, "%s line %llu: empty line (the reference raises IndexError here)", gfa, (unsigned long long)num + 1);
        if (f[0] != "S") continue;
        if (f.size() < 3) fail(HLMI_EINVAL, "%s line %llu: S line with fewer than 3 fields", gfa, (unsigned long long)num + 1);
        out.put(">"); out.put(f[1]); out.put("\n"); out.put(f[2]); out.put("\n");
    }
    out.flush();
}

void pick_up_run(const char *paf, const char *fastx, const char *out_fastx, bool fastq) {
    const std::string ptext = slurp(paf);
    std::unordered_set<std::string_view> seen;
    {
        Lines it(ptext);
        std::string_view body;
        bool nl;
        std::vector<std::string_view> f;
        for (uint64_t num = 0; it.next(body, nl); ++num) {
            py_split(body, f);
            if (f.size() < 6) fail(HLMI_EINVAL, "%s line %llu: fewer than 6 columns", paf, (unsigned long long)num + 1);
            seen.insert(before(f[0], '/'));
            seen.insert(before(f[5], '/'));
        }
    }
    remove(out_fastx);                               // HyLight.py:359-361; the file only exists if something is kept
    const std::string text = slurp(fastx);
    Out out(out_fastx, false);
    Lines it(text);
    std::string_view body;
    bool nl, keep = false;
    const uint64_t nu = fastq ? 4 : 2;
    for (uint64_t num = 0; it.next(body, nl); ++num) {
        if (num % nu == 0) {
            const std::string_view k = before(py_strip(body), '/');
            keep = seen.find(k.empty() ? k : k.substr(1)) == seen.end();
        }
        if (keep) { out.put(body); if (nl) out.put("\n"); }
    }
    out.flush();
}

}  // namespace hlmi
