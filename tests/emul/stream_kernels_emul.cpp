// TEST INFRASTRUCTURE ONLY -- the barrier-free kernels of the hot path's helpers compiled as host C++ from the product's own
// sources (hipshim/hip/hip_runtime.h turns a launch into a loop over blocks and threads): the C-ABI entry points of resample.hip
// and conv_head.hip, on host memory.  Built by tests/test_stream_kernels_emul.py:
//     g++ -O1 -std=c++17 -ffp-contract=off -Itests/emul/hipshim -shared -fPIC
// (-ffp-contract=off as emoportraits_amd/build.py: no multiply-add is fused that the source does not spell __fmaf_rn).
#include "../../emoportraits_amd/csrc/resample.hip"
#include "../../emoportraits_amd/csrc/conv_head.hip"
