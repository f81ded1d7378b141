// matrix row-split ELBO kernel, conditional posterior with the experts' sums formed in the kernel (XM == 3), fp32 rows in order
#include "vibo_msplit_kernel.hpp"
#include "vibo_launch.hpp"
namespace vibo {
hipError_t launch_elbo_msplit_xa(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s) {
    return launch_msplit_fused<0>(p, irt, grad, nw, grid, s);
}
}  // namespace vibo
#ifdef VIBO_MS_TIMING
extern "C" int vibo_debug_ms_timing_xa(long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ms_timing), (size_t)n * sizeof(long long));
}
#endif
