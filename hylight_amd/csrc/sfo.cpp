// sfo.cpp - a18: sfo2overlaps.py as HyLight calls it (`--num_pairs 0`, script/HyLight.py:315-318): SFO rows with the
// smaller id first, `sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq`, single-end SAVAGE rows (script/sfo2overlaps.py:31-200).
// Contig-scale text (10^3 - 10^5 rows): host.
#include <array>
#include <cctype>
#include <cmath>

#include "graph.h"
#include "paf_io.h"

namespace hlmi {

// ---- a18: sfo2overlaps.py, --num_pairs 0 branch ----------------------------------------------------------
void sfo2overlaps_run(const char *in_sfo, const char *out_savage, int num_singles, int num_pairs) {
    (void)num_singles;
    if (num_pairs != 0) fail(HLMI_ESTATE, "sfo2overlaps: paired-end branch (--num_pairs > 0) is not on HyLight's path "
                                           "(HyLight.py:317 passes 0)");
    std::string data = read_file(in_sfo);
    struct Row { long long ia, ib; std::string line; };
    std::vector<Row> rows;
    size_t pos = 0;
    auto split_ws = [](const std::string &l) {
        std::vector<std::string> f;
        size_t p = 0;
        while (p < l.size()) {
            while (p < l.size() && isspace((unsigned char)l[p])) ++p;
            size_t e = p;
            while (e < l.size() && !isspace((unsigned char)l[e])) ++e;
            if (e > p) f.emplace_back(l, p, e - p);
            p = e;
        }
        return f;
    };
    while (pos < data.size()) {
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        std::string line = data.substr(pos, e - pos);
        pos = e + 1;
        std::vector<std::string> f = split_ws(line);
        if (f.size() != 8) fail(HLMI_EINVAL, "%s: SFO row needs 8 fields", in_sfo);
        long long ia = atoll(f[0].c_str()), ib = atoll(f[1].c_str());
        std::string body;
        if (ia > ib) {                                          // sfo2overlaps.py:41-47,112-122
            std::vector<std::string> g;
            if (f[2] == "I") g = {f[1], f[0], f[2], f[4], f[3], f[6], f[5], f[7]};
            else g = {f[1], f[0], f[2], std::to_string(-atoll(f[3].c_str())), std::to_string(-atoll(f[4].c_str())), f[6], f[5], f[7]};
            for (size_t i = 0; i < g.size(); ++i) { if (i) body += '\t'; body += g[i]; }
            std::swap(ia, ib);
        } else body = line;
        rows.push_back(Row{ia, ib, std::to_string(ia) + "\t" + std::to_string(ib) + "\t" + body});
    }
    // sort -k1,1n -k2,2n -k3,3n -k4,4n | uniq   (fields 3,4 are the SFO ids again)
    auto num = [&](const Row &r, int k) { return atoll(split_ws(r.line)[k].c_str()); };
    std::vector<std::array<long long, 4>> keys(rows.size());
    for (size_t i = 0; i < rows.size(); ++i) keys[i] = {num(rows[i], 0), num(rows[i], 1), num(rows[i], 2), num(rows[i], 3)};
    std::vector<uint32_t> idx(rows.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        if (keys[a] != keys[b]) return keys[a] < keys[b];
        return rows[a].line < rows[b].line;
    });
    std::vector<std::string> out;
    const std::string *prev = nullptr;
    for (uint32_t i : idx) {
        const std::string &l = rows[i].line;
        if (prev && *prev == l) continue;
        prev = &l;
        std::vector<std::string> c = split_ws(l);
        if (c.size() != 10) fail(HLMI_EINVAL, "sfo2overlaps: internal row needs 10 fields");
        const long long ida = atoll(c[0].c_str()), idb = atoll(c[1].c_str());
        if (ida == idb) continue;
        const long long oha = atoll(c[5].c_str()), ohb = atoll(c[6].c_str()), ola = atoll(c[7].c_str()), olb = atoll(c[8].c_str());
        const char ori = c[4] == "N" ? '+' : '-';
        const long long ovlen = std::min(ola, olb);
        long long lena, lenb, pos1;
        std::string id1, id2;
        char ori1, ori2;
        if (oha >= 0) {
            lena = ola + oha + (ohb >= 0 ? 0 : -ohb);
            lenb = ohb >= 0 ? olb + ohb : olb;
            id1 = c[0]; id2 = c[1]; pos1 = oha; ori1 = '+'; ori2 = ori;
        } else {
            lena = ohb >= 0 ? ola : ola - ohb;
            lenb = -oha + olb + (ohb >= 0 ? ohb : 0);
            id1 = c[1]; id2 = c[0]; pos1 = -oha; ori1 = ori; ori2 = '+';
        }
        const long long minlen = std::min(lena, lenb);
        if (minlen <= 0) fail(HLMI_EINVAL, "sfo2overlaps: non-positive read length");
        // Python round(): half to even on the exact double 100*ovlen/minlen
        const double x = (double)(100 * ovlen) / (double)minlen;
        long long perc = (long long)std::nearbyint(x);          // FE_TONEAREST = ties to even
        if (perc > 100) perc = 100;
        char buf[256];
        snprintf(buf, sizeof buf, "%s\t%s\t%lld\t-\t-\t%c\t%c\t%lld\t-\t%lld\t-\ts\ts", id1.c_str(), id2.c_str(), pos1, ori1, ori2, perc, ovlen);
        if (out.empty() || out.back() != buf) out.emplace_back(buf);
    }
    write_lines(out_savage, out);
}

}  // namespace hlmi
