/* A C-ABI caller without PyTorch: drives the staged multi-GPU job of include/hylight_mi.h for
 * world = 2 the way a cgo / JNI / plain-C host would - every rank opens the job, sketches ITS slice of the query reads into
 * a device buffer of its own, the buffers are gathered (here: device-to-device copies into one buffer, standing in for
 * ncclAllGather - both "ranks" live in this one process and share the card), every rank installs the gathered sketch and
 * runs its share of the --nsplit chunks; the per-rank files are merged (the `sort -k12 -nr` of script/utils.py:69) and
 * compared byte for byte with the single-rank entry point hlmi_split_reads2.
 *
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/capi/job_two_ranks.c -o job_two_ranks \
 *       -Lhylight_amd -lhylight_mi -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/hylight_amd -Wl,-rpath,/opt/rocm/lib
 *   ./job_two_ranks reads.fa <nsplit> <workdir>
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hylight_mi.h"

#define WORLD 2
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (%s)\n", hlmi_last_error()); exit(1); } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static long file_bytes(const char *p, char **out) {
    FILE *f = fopen(p, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    *out = (char *)malloc(n + 1);
    if (fread(*out, 1, n, f) != (size_t)n) n = -1;
    fclose(f);
    return n;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s reads.fa nsplit workdir\n", argv[0]); return 2; }
    const char *fa = argv[1], *dir = argv[3];
    const int nsplit = atoi(argv[2]), len_over = 1000, mc = 2;
    const double iden = 0.95;
    if (hlmi_init(0, 0) != 0) DIE("hlmi_init");

    hlmi_job *job[WORLD];
    void *d_mz[WORLD], *d_cnt[WORLD];
    int64_t n_of[WORLD], lo[WORLD], hi[WORLD];
    int64_t nq = 0, total = 0;
    for (int r = 0; r < WORLD; ++r) {                 /* every rank: open, sketch its slice of the queries */
        job[r] = hlmi_job_open(fa, fa, nsplit, 1);
        if (!job[r]) DIE("hlmi_job_open");
        nq = hlmi_job_num_queries(job[r]);
        lo[r] = r * nq / WORLD; hi[r] = (r + 1) * nq / WORLD;
        const int64_t cap = hlmi_job_sketch_bound(job[r], lo[r], hi[r]);
        HIP(hipMalloc(&d_mz[r], (size_t)(cap > 0 ? cap : 1) * 16));
        HIP(hipMalloc(&d_cnt[r], (size_t)(hi[r] - lo[r] > 0 ? hi[r] - lo[r] : 1) * 4));
        if (hlmi_job_sketch(job[r], lo[r], hi[r], d_mz[r], cap, d_cnt[r], &n_of[r]) != 0) DIE("hlmi_job_sketch");
        total += n_of[r];
    }
    /* the exchange: what ncclAllGather (variable sizes: counts first, then the slabs) leaves on every rank */
    void *all_mz, *all_cnt;
    HIP(hipMalloc(&all_mz, (size_t)(total > 0 ? total : 1) * 16));
    HIP(hipMalloc(&all_cnt, (size_t)nq * 4));
    int64_t off = 0;
    for (int r = 0; r < WORLD; ++r) {
        HIP(hipMemcpy((char *)all_mz + off * 16, d_mz[r], (size_t)n_of[r] * 16, hipMemcpyDeviceToDevice));
        HIP(hipMemcpy((char *)all_cnt + lo[r] * 4, d_cnt[r], (size_t)(hi[r] - lo[r]) * 4, hipMemcpyDeviceToDevice));
        off += n_of[r];
    }
    HIP(hipDeviceSynchronize());
    char part[WORLD][1024], merged[1024], single[1024];
    const char *parts[WORLD];
    for (int r = 0; r < WORLD; ++r) {                 /* every rank: install, run its chunks */
        if (hlmi_job_set_query_sketch(job[r], all_mz, total, all_cnt) != 0) DIE("hlmi_job_set_query_sketch");
        snprintf(part[r], sizeof part[r], "%s/out.paf.part%d", dir, r);
        if (hlmi_job_run(job[r], r, WORLD, len_over, mc, iden, part[r]) != 0) DIE("hlmi_job_run");
        parts[r] = part[r];
        hlmi_job_close(job[r]);
    }
    snprintf(merged, sizeof merged, "%s/out.paf", dir);
    if (hlmi_merge_scored_paf(parts, WORLD, merged) != 0) DIE("hlmi_merge_scored_paf");
    snprintf(single, sizeof single, "%s/single.paf", dir);
    if (hlmi_split_reads2(fa, fa, nsplit, dir, single, 8, len_over, mc, iden, 1) != 0) DIE("hlmi_split_reads2");
    char *a, *b;
    const long na = file_bytes(merged, &a), nb = file_bytes(single, &b);
    if (na < 0 || nb < 0 || na != nb || memcmp(a, b, (size_t)na) != 0) {
        fprintf(stderr, "two ranks: %ld bytes, one rank: %ld bytes - the files differ\n", na, nb);
        return 1;
    }
    long rows = 0;
    for (long i = 0; i < na; ++i) rows += a[i] == '\n';
    printf("OK world=%d queries=%lld minimizers=%lld exchange_bytes=%lld rows=%ld\n", WORLD, (long long)nq, (long long)total,
           (long long)(total * 16 + nq * 4), rows);
    hlmi_shutdown();
    return rows > 0 ? 0 : 1;
}
