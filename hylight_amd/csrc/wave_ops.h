// wave_ops.h - wave64 cross-lane primitives on DPP (gfx950 / GFX9 DPP controls).
//
// `__shfl*` lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~50-100 cycles of latency each); the
// reductions / scans / one-lane shifts the DP kernels need per row are dependent chains of those.  The DPP
// forms below are plain VALU instructions with a lane-select modifier.
//   row_shr:n   0x110+n   lane i reads lane i-n inside its row of 16
//   row_bcast15 0x142     lane 15 of each row -> every lane of the next row   (row_mask selects rows)
//   row_bcast31 0x143     lane 31 -> rows 2,3
//   wave_shr:1  0x138     lane i reads lane i-1 across the whole wave
//   wave_shl:1  0x130     lane i reads lane i+1
// Lanes whose source does not exist, or whose row/bank is masked off, keep `old`.
#pragma once
#include <hip/hip_runtime.h>

namespace hlmi {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

constexpr int DPP_MAX_IDENT = (int)0x80000000u;   // identity of signed max: lets a dpp move fold into v_max_i32_dpp

// max over the 64 lanes (unsigned, identity 0), returned to every lane
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
    int x = (int)v;
    auto mx = [](int a, int b) { return (int)((uint32_t)a > (uint32_t)b ? (uint32_t)a : (uint32_t)b); };
    x = mx(x, dpp_i32<0x111>(0, x));
    x = mx(x, dpp_i32<0x112>(0, x));
    x = mx(x, dpp_i32<0x114>(0, x));
    x = mx(x, dpp_i32<0x118>(0, x));
    x = mx(x, dpp_i32<0x142, 0xa>(0, x));
    x = mx(x, dpp_i32<0x143, 0xc>(0, x));
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// min over the 64 lanes (unsigned, identity ~0), returned to every lane
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
    int x = (int)v;
    auto mn = [](int a, int b) { return (int)((uint32_t)a < (uint32_t)b ? (uint32_t)a : (uint32_t)b); };
    x = mn(x, dpp_i32<0x111>(-1, x));
    x = mn(x, dpp_i32<0x112>(-1, x));
    x = mn(x, dpp_i32<0x114>(-1, x));
    x = mn(x, dpp_i32<0x118>(-1, x));
    x = mn(x, dpp_i32<0x142, 0xa>(-1, x));
    x = mn(x, dpp_i32<0x143, 0xc>(-1, x));
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

// max over the 64 lanes (signed), returned to every lane.  `old` = INT_MIN is the identity of signed max, which
// is what lets the compiler fold every step into one v_max_i32_dpp (any other filler costs three instructions).
__device__ __forceinline__ int wave_max_i32_dpp(int x) {
    auto mx = [](int a, int b) { return a > b ? a : b; };
    x = mx(x, dpp_i32<0x111>(DPP_MAX_IDENT, x));
    x = mx(x, dpp_i32<0x112>(DPP_MAX_IDENT, x));
    x = mx(x, dpp_i32<0x114>(DPP_MAX_IDENT, x));
    x = mx(x, dpp_i32<0x118>(DPP_MAX_IDENT, x));
    x = mx(x, dpp_i32<0x142, 0xa>(DPP_MAX_IDENT, x));
    x = mx(x, dpp_i32<0x143, 0xc>(DPP_MAX_IDENT, x));
    return __builtin_amdgcn_readlane(x, 63);
}

// inclusive prefix max over the 64 lanes (signed)
__device__ __forceinline__ int wave_prefix_max_incl_dpp(int x) {
    constexpr int ident = DPP_MAX_IDENT;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    x = mx(x, dpp_i32<0x111>(ident, x));
    x = mx(x, dpp_i32<0x112>(ident, x));
    x = mx(x, dpp_i32<0x114>(ident, x));
    x = mx(x, dpp_i32<0x118>(ident, x));
    x = mx(x, dpp_i32<0x142, 0xa>(ident, x));
    x = mx(x, dpp_i32<0x143, 0xc>(ident, x));
    return x;
}

// inclusive prefix sum over the 64 lanes: four row_shr steps inside every row of 16, then the row totals travel down
// (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3); lanes without a source add 0
__device__ __forceinline__ uint32_t wave_prefix_sum_incl_dpp(uint32_t v) {
    int x = (int)v;
    x += dpp_i32<0x111>(0, x);
    x += dpp_i32<0x112>(0, x);
    x += dpp_i32<0x114>(0, x);
    x += dpp_i32<0x118>(0, x);
    x += dpp_i32<0x142, 0xa>(0, x);
    x += dpp_i32<0x143, 0xc>(0, x);
    return (uint32_t)x;
}

// ---- the same inside rows of 16 lanes (four independent 16-lane groups per wave) -------------------
// lane l <- lane l+1 of its row (l == 15 keeps `last`)
__device__ __forceinline__ int row_shl1(int x, int last) { return dpp_i32<0x101>(last, x); }
// lane l <- lane l-1 of its row (l == 0 keeps `first`)
__device__ __forceinline__ int row_shr1(int x, int first) { return dpp_i32<0x111>(first, x); }
// inclusive prefix max inside every row of 16
__device__ __forceinline__ int row_prefix_max_incl_dpp(int x) {
    constexpr int ident = DPP_MAX_IDENT;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    x = mx(x, dpp_i32<0x111>(ident, x));
    x = mx(x, dpp_i32<0x112>(ident, x));
    x = mx(x, dpp_i32<0x114>(ident, x));
    x = mx(x, dpp_i32<0x118>(ident, x));
    return x;
}

// `old` with lane `sel` replaced by the wave-uniform `v`: v_writelane_b32.  clang has no builtin for it, so the
// LLVM intrinsic is bound by name (the way the ROCm device libraries bind theirs).
extern "C" __device__ int hlmi_llvm_writelane(int v, int sel, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int writelane_i32(int old, int v, int sel) { return hlmi_llvm_writelane(v, sel, old); }

// per-lane select by a wave-uniform 64-bit lane mask held in SGPRs: bit l set -> lane l takes `yes`.  (A C++
// expression of this needs a per-lane compare first; the mask of "lane == j" is just 1 << j on the scalar unit.)
__device__ __forceinline__ int select_by_mask(unsigned long long mask, int yes, int no) {
    int r = no;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(no), "v"(yes), "s"(mask));
#endif
    return r;
}

// |a - b| + c on unsigned operands in one instruction
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
#endif
    return r;
}

// a + b on the scalar unit, opaque to the optimiser (keeps it from re-associating the sum into per-lane adds)
__device__ __forceinline__ int scalar_add(int a, int b) {
    int r = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_add_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
#endif
    return r;
}

// zero-filling one-lane shifts: a single v_mov_b32_dpp ... bound_ctrl:1 (no `old` register to prepare)
__device__ __forceinline__ int row_shl1_z(int x) { return __builtin_amdgcn_mov_dpp(x, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_shl1_z(int x) { return __builtin_amdgcn_mov_dpp(x, 0x130, 0xf, 0xf, true); }

// lane i <- lane i-1 (lane 0 keeps `lane0`)
__device__ __forceinline__ int wave_shr1(int x, int lane0) { return dpp_i32<0x138>(lane0, x); }
// lane i <- lane i+1 (lane 63 keeps `lane63`)
__device__ __forceinline__ int wave_shl1(int x, int lane63) { return dpp_i32<0x130>(lane63, x); }

}  // namespace hlmi
