// vibo_launch.hpp -- per-ability-width launchers of the fused ELBO kernel.
// One translation unit per template ability width keeps hipcc compile times
// parallel (see Makefile); vibo_capi.hip dispatches to these.
#pragma once
#include <hip/hip_runtime.h>
#include "vibo_params.hpp"

namespace vibo {

// waves per workgroup: 2 (I <= 144), 4 (I <= 304), 8 (I <= 512), 16 (I <= 1024)
struct LaunchGeom {
    int waves, grid;
    size_t lds_bytes;
};

hipError_t launch_elbo_a1(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a2(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a4(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);
hipError_t launch_elbo_a8(const ElboParams& p, int irt, bool grad, const LaunchGeom& g, hipStream_t s);

// wave-per-row kernel (vibo_row_kernel.hip): A in {1,2}, 1PL/2PL, I <= 1024, 16-byte aligned rows
hipError_t launch_elbo_rows(const ElboParams& p, int irt, bool grad, int grid, hipStream_t s);

// row-split kernel (vibo_split_kernel.hpp): template ability width 2 / 4 / 8, 1PL/2PL/3PL, optional planar flows,
// I <= 1024, I % 4 == 0, 16-byte aligned rows, mask u8 or none; nq = ceil(I / 256) waves per workgroup
hipError_t launch_elbo_split_a2(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
hipError_t launch_elbo_split_a4(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
hipError_t launch_elbo_split_a8(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
// the same kernels gathering the fp32 rows through row_index
hipError_t launch_elbo_split_g2(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
hipError_t launch_elbo_split_g4(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
hipError_t launch_elbo_split_g8(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
// the same kernels reading 1-byte cell codes (VIBO_MASK_CODES)
hipError_t launch_elbo_split_c2(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
hipError_t launch_elbo_split_c4(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);
hipError_t launch_elbo_split_c8(const ElboParams& p, int irt, bool grad, int nq, int grid, hipStream_t s);

// matrix-pipe row-split kernel (vibo_msplit_kernel.hpp): any ability_dim <= 8, 1PL/2PL/3PL, I <= 1024, rows chunkable
// in 4 cells; nw = ceil(I / 128) waves per workgroup; fp32 rows in order / through row_index / 1-byte cell codes
hipError_t launch_elbo_msplit_a(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);
hipError_t launch_elbo_msplit_g(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);
hipError_t launch_elbo_msplit_c(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);
// ... the same with planar flows on the ability sample
hipError_t launch_elbo_msplit_fa(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);
hipError_t launch_elbo_msplit_fg(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);
hipError_t launch_elbo_msplit_fc(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);

// ... the conditional posterior of one panel at ability_dim 1 with its first pass folded in (p.cond_table / p.codes_out / p.post_coef)
hipError_t launch_elbo_msplit_xa(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);
hipError_t launch_elbo_msplit_xg(const ElboParams& p, int irt, bool grad, int nw, int grid, hipStream_t s);

// narrow-row kernel: waves per SIMD each instantiation is compiled for (registers: items x (parameters + gradient accumulators) + one
// unit of rows; 3PL with gradients at 8 items per lane: 128 registers were 4-12 short -- spilled, and a spill reload waits behind the
// row loads).  ONE definition for the kernel's launch bounds (vibo_narrow.hip) and the planner's grid (vibo_capi.hip: workgroups per
// CU = waves per SIMD): the grid has to match the occupancy the kernel was compiled for.
constexpr int narrow_waves_per_simd(int at, int il, bool g3 = false) {
    return at == 1 ? ((g3 && il == 8) ? 3 : 4) : at == 2 ? (il == 4 ? 4 : 3) : (il == 4 ? 3 : 2);
}

// narrow-row kernel (vibo_narrow.hip): 4 <= I <= 128, ability_dim <= 4, unconditional posterior, no flows; a row per 16 lanes;
// fp32 rows in order / through p.row_index / 1-byte cell codes (codes: through p.mask); grid workgroups of 4 independent waves
hipError_t launch_elbo_narrow(const ElboParams& p, bool codes, int irt, bool grad, int grid, hipStream_t s);

// the folded train step's epilogue (vibo_trainer.hip): finalize + loss + MLP / item backward + Adam + the next step's noise
hipError_t launch_train_epilogue_fused(const EpiParams& e, hipStream_t s);
int train_epilogue_item_blocks(int n_item_entries);

}  // namespace vibo
