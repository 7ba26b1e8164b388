"""The build's register-spill guard (hylight_amd/build.py): hipcc's resource remarks are read, a kernel that spills more vector
registers than MAX_VGPR_SPILL is refused, warnings still get through."""
from hylight_amd import build as B

SAMPLE = """x.hip:322:1: remark: Function Name: _ZN4hlmi15classify_kernelILi1EEEv [-Rpass-analysis=kernel-resource-usage]
  322 | __global__ void classify_kernel() {
      | ^
x.hip:322:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]
x.hip:322:1: remark:     VGPRs Spill: 305 [-Rpass-analysis=kernel-resource-usage]
x.hip:845:1: remark: Function Name: _ZN4hlmi12chain_kernelILi3ELi2EEEv [-Rpass-analysis=kernel-resource-usage]
x.hip:845:1: remark:     VGPRs Spill: 4 [-Rpass-analysis=kernel-resource-usage]
x.hip:10:5: warning: unused variable 'y' [-Wunused-variable]
   10 |     int y;
      |         ^
1 warning generated when compiling for gfx950.
"""


def test_spilling_kernels_are_named_and_small_spills_pass():
    spilled, other = B.resource_remarks(SAMPLE)
    assert spilled == [("_ZN4hlmi15classify_kernelILi1EEEv", 305)]
    assert any("unused variable" in l for l in other) and not any("remark" in l or "|" in l for l in other)


def test_no_remarks_no_findings():
    assert B.resource_remarks("") == ([], [])
