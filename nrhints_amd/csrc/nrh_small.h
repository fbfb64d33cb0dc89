// Host-side interface between nrh_api.hip and the separately compiled small-batch translation unit (nrh_small.hip): the SDF
// network's three training kernels built with FOUR waves per workgroup instead of eight.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace nrh4s {

// `args`: the bytes of nrh::SdfArgs (forward) / nrh::SdfTrainArgs (the two sweeps) as nrh_api.hip fills them - the structures of
// this unit are the same source compiled under another namespace; ntile_groups is recomputed for the smaller workgroup.
// max_grid: workgroups that fit the device at once.  0 ok, -1 bad size, -2 launch / attribute error.
int launch_sdf_train_forward(int precision, const void* args, size_t bytes, int max_grid, hipStream_t st);
int launch_sdf_train_sweeps(int precision, const void* args, size_t bytes, int max_grid, hipStream_t st);   // tangent, then value sweep
constexpr int WAVES = 4;

}  // namespace nrh4s
