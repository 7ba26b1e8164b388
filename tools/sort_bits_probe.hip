// Radix width of the anchor-batch sort: tools/sort_bits_probe.hip [n] [distinct keys] (hipcc --offload-arch=gfx950 -O3; needs the GPU).
// A batch is sorted on its (target, strand) bits alone - 10 bits against a C3 chunk - as a 16-bit key with the 64-bit anchor
// word as its value; rocPRIM's onesweep takes 8 bits per scatter pass by default, i.e. two passes.  This times one pass of
// 9-12 bits in several block shapes against that, on keys that arrive like anchors do (runs of ascending targets).
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
template <int BITS, int BS, int IPT, int HBS = 1024>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HBS, 16>, rocprim::kernel_config<BS, IPT>, BITS,
                                        rocprim::block_radix_rank_algorithm::match>>;
struct Bufs { uint16_t *k[2]; uint64_t *v[2]; };
template <class C>
double run(Bufs &B, size_t n, int b0, int b1, std::vector<uint64_t> *out) {
    rocprim::double_buffer<uint16_t> dk(B.k[0], B.k[1]);
    rocprim::double_buffer<uint64_t> dv(B.v[0], B.v[1]);
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs<C>(nullptr, tmp, dk, dv, n, b0, b1, 0);
    void *t; (void)hipMalloc(&t, tmp);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    (void)rocprim::radix_sort_pairs<C>(t, tmp, dk, dv, n, b0, b1, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (out) { out->resize(n); (void)hipMemcpy(out->data(), dv.current(), n * 8, hipMemcpyDeviceToHost); }
    (void)hipFree(t);
    return ms;
}
int main(int argc, char **argv) {
    const size_t n = argc > 1 ? atol(argv[1]) : (384u << 20);
    const uint64_t mod = argc > 2 ? atol(argv[2]) : 1008;
    Bufs B;
    uint16_t *ksrc; uint64_t *vsrc;
    for (int i = 0; i < 2; ++i) { (void)hipMalloc(&B.k[i], n * 2); (void)hipMalloc(&B.v[i], n * 8); }
    (void)hipMalloc(&ksrc, n * 2); (void)hipMalloc(&vsrc, n * 8);
    std::vector<uint16_t> hk(n);
    std::vector<uint64_t> hv(n);
    uint64_t x = 88172645463325252ull;
    // a query's minimizer meets ~88 of the ~200 partners of the query, in ascending order; a new query every ~88 000 anchors
    std::vector<uint32_t> partners(200);
    size_t i = 0;
    while (i < n) {
        if (i % 88000 < 88) for (auto &p : partners) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; p = (uint32_t)((x >> 20) % mod); }
        uint32_t cur = 0;
        for (int h = 0; h < 88 && i < n; ++h, ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            cur += 1 + (uint32_t)(x >> 60) % 4;
            hk[i] = (uint16_t)((partners[cur % 200] + cur / 200 * 7) % mod);
            hv[i] = i;
        }
    }
    (void)hipMemcpy(ksrc, hk.data(), n * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(vsrc, hv.data(), n * 8, hipMemcpyHostToDevice);
    auto reset = [&]() { (void)hipMemcpy(B.k[0], ksrc, n * 2, hipMemcpyDeviceToDevice); (void)hipMemcpy(B.v[0], vsrc, n * 8, hipMemcpyDeviceToDevice); };
    int bits = 1;
    while ((1ull << bits) < mod) ++bits;
    std::vector<uint64_t> r_def, r_one;
    for (int rep = 0; rep < 3; ++rep) {
        const bool last = rep == 2;
        reset(); const double t_def = run<rocprim::default_config>(B, n, 0, bits, last ? &r_def : nullptr);
        reset(); const double t8 = run<Cfg<8, 1024, 8>>(B, n, 0, bits, nullptr);
        double t[6] = {0, 0, 0, 0, 0, 0};
        if (bits <= 10) {
            reset(); t[0] = run<Cfg<10, 1024, 16>>(B, n, 0, bits, last ? &r_one : nullptr);
            reset(); t[1] = run<Cfg<10, 1024, 8>>(B, n, 0, bits, nullptr);
            reset(); t[2] = run<Cfg<10, 512, 16>>(B, n, 0, bits, nullptr);
            reset(); t[3] = run<Cfg<10, 1024, 12>>(B, n, 0, bits, nullptr);
        } else {
            reset(); t[0] = run<Cfg<12, 256, 16>>(B, n, 0, bits, last ? &r_one : nullptr);
            reset(); t[1] = run<Cfg<12, 256, 32>>(B, n, 0, bits, nullptr);
            reset(); t[2] = run<Cfg<11, 512, 16>>(B, n, 0, bits, nullptr);
            reset(); t[3] = run<Cfg<6, 1024, 8>>(B, n, 0, bits, nullptr);
        }
        printf("n=%zu, %d key bits: library default %.2f ms, 8 bits per pass (1024x8) %.2f | %s: 1024x16 %.2f  1024x8 %.2f  512x16 %.2f  %s %.2f\n",
               n, bits, t_def, t8, bits <= 10 ? "10 bits per pass" : "12 bits per pass (256x16, 256x32, 11 bits 512x16)", t[0], t[1], t[2], bits <= 10 ? "1024x12" : "6 bits per pass", t[3]);
    }
    printf("identical: %d\n", (int)(r_def == r_one));
    return 0;
}
