// valu_calib.hip - calibration point for tools/summarize_sq.py: a kernel that does nothing but issue independent
// `v_add_u32` instructions at 8 resident waves per SIMD.  Under
//     rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- ./valu_calib
// the SIMDs' vector issue is saturated by construction, so whatever formula turns the counters into a "VALU issue
// fraction" has to read 1.0 here (MI355X_MICROARCH.md: a wave64 VALU instruction occupies its SIMD-32 for 2 cycles).
// Also prints its own estimate: instructions x 2 cycles / (SIMDs x wall time x clock) with the clock from s_memtime.
//     hipcc --offload-arch=gfx950 -O3 -o valu_calib tools/valu_calib.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 64;      // v_add_u32 per loop trip and lane (eight independent chains)

__global__ __launch_bounds__(256) void valu_loop(unsigned *out, int trips, unsigned long long *clk) {
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; ++u)
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(1u));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

int main(int argc, char **argv) {
    const int trips = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 8;                 // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    unsigned *out;
    unsigned long long *clk, h_clk = 0;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&clk, 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(valu_loop, dim3(blocks), dim3(256), 0, 0, out, trips, clk);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        CHECK(hipMemcpy(&h_clk, clk, 8, hipMemcpyDeviceToHost));
        const double insts = (double)blocks * 4 * (double)trips * UNROLL;          // wave instructions
        const double clock_ghz = (double)h_clk / (ms * 1e6);                        // shader cycles of one wave / wall
        const double simd_cycles = (double)cus * 4 * ms * 1e-3 * clock_ghz * 1e9;
        printf("{\"cus\": %d, \"waves\": %d, \"ms\": %.3f, \"wave_valu_insts\": %.0f, \"clock_ghz\": %.3f, "
               "\"issue_frac_at_2_cycles\": %.4f, \"cycles_per_inst_per_simd\": %.3f}\n",
               cus, blocks * 4, ms, insts, clock_ghz, insts * 2.0 / simd_cycles, simd_cycles / insts);
    }
    return 0;
}
