// gymrs_step_cartpole.hip -- the per-step kernel table of one env type (see gymrs_step_impl.h).
#include "gymrs_step_impl.h"

namespace gymrs {

hipError_t launch_step_cartpole(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    return launch_vec<CartPoleT>(vec, flags, a, consts, stream);
}

} // namespace gymrs
