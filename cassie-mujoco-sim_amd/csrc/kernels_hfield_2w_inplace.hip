/* cassie_hfield.xml, the row-capped fast instantiation in its two-wave form with the 63-row code behind it in the same kernel
 * (see kernels_cassie_2w_inplace.hip) */
#include "step_launch.h"
namespace ck {
bool launch_fast_cassie_hfield_2w_inplace(dim3 grid, hipStream_t s, PhysIO io) {
    hipLaunchKernelGGL((cassie_step_kernel<32, TopoCassie32, FEAT_HFIELD, FAST_ROWS, 2, false, 2, MID_ROWS>), grid, dim3(2 * WV_WAVE), 0, s, io);
    return hipGetLastError() == hipSuccess;
}
}  // namespace ck
