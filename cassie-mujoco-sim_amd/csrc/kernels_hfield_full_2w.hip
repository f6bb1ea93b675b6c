/* cassie_hfield.xml, the 63-row instantiation in its two-wave form as the pass behind the two-wave fast kernel (kernels_cassie_full_2w.hip) */
#include "step_launch.h"
namespace ck {
bool launch_mid_cassie_hfield_2w(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_HFIELD, MID_ROWS, 2, true>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
