// format_fixed4 (hylight_amd/csrc/paf_io.cpp) against printf("%.4f") itself: the three score columns of every final row
// (script/filter_overlap_slr2.py:142-151 writes them with "%.4f") go through it.  Built and run by tests/test_host_sort.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
namespace hlmi { size_t format_fixed4(double v, char *dst); }
static long bad = 0;
static void check(double v) {
    char a[96], b[96];
    const size_t n = hlmi::format_fixed4(v, a);
    a[n] = 0;
    snprintf(b, sizeof b, "%.4f", v);
    if (strcmp(a, b) != 0 && ++bad < 20) fprintf(stderr, "MISMATCH %a: got %s want %s\n", v, a, b);
}
int main() {
    std::mt19937_64 rng(12345);
    // exact ties of the fourth decimal (k + 0.5) / 10^4 are representable only for k/10^4 dyadic: walk the dyadic grid
    for (int j = 0; j <= 20; ++j)
        for (uint64_t k = 0; k < 4096; ++k) check((double)k / (double)(1ull << j));
    for (int i = 0; i < 4000000; ++i) {                       // scores as the rows have them: ratios of small integers
        const double a = (double)(rng() % 100000), b = (double)(rng() % 100000 + 1);
        check(a / b); check(0.4 * (a / b) + 0.6 * (b / (a + b))); check(1.0 - a / b);
    }
    for (int i = 0; i < 4000000; ++i) {                       // every exponent, random mantissas, both signs, specials
        uint64_t bits = rng();
        double v;
        memcpy(&v, &bits, 8);
        if (std::isfinite(v) && std::fabs(v) < 1e15) check(v);
        check(std::ldexp((double)(rng() >> 11), -(int)(rng() % 90)));
    }
    const double sp[] = {0.0, -0.0, 0.00005, 0.00015, 0.99995, 1.0, 0.5, 1e-300, 5e-324, 1099511627775.99995, 1099511627776.0, 1e14,
                         NAN, INFINITY, -INFINITY, -0.00004, -1.5};
    for (double v : sp) check(v);
    printf("mismatches %ld\n", bad);
    return bad ? 1 : 0;
}
