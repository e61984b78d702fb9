// gymrs_step_mountain_car.hip -- the per-step kernel table of one env type (see gymrs_step_impl.h).
#include "gymrs_step_impl.h"

namespace gymrs {

hipError_t launch_step_mountain_car(int vec, uint32_t flags, const StepArgs& a, const void* consts, hipStream_t stream)
{
    return launch_vec<MountainCarT>(vec, flags, a, consts, stream);
}

} // namespace gymrs
