// wave_lanes.hpp -- lane-level helpers of the one-wavefront-per-transform link kernels (siso_tdl_wave.hpp, mimo_tdl_wave.hpp):
// the DPP neighbour swap that hands over half of a Philox NOISE block, and v_readlane reads of values parked across the lanes.
#pragma once
#include "common.hpp"

namespace mcle {

template <typename T> __device__ __forceinline__ T dpp_swap1(T v);
template <> __device__ __forceinline__ float dpp_swap1<float>(float v) {         // the value of lane l ^ 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// lane j's value of a VGPR as a wave-uniform scalar (j wave-uniform)
__device__ __forceinline__ float lane_value(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
__device__ __forceinline__ double lane_value(double v, int j) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
}
template <> __device__ __forceinline__ double dpp_swap1<double>(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

}  // namespace mcle
