// bone_chain.hip -- joint rotations -> global bone transforms (scope row a-3), forward and backward, ONE launch each.
//
// Reference (networks/sk_gs.py:1069-1107 kinematic, :193-206 skeleton_warp_SE3, lietorch SE3 product lie.h:242-246):
//   sk_r  = normalize(raw + [0,0,0,1])                         joint rotation, xyzw
//   L_i   = (j_i + R(sk_r_i)(-j_i), sk_r_i)                    rotation about the joint position; L_root = identity
//   A_i   = L_{a1} o L_{a2} o ... o L_i   (a1..: path root -> i, evaluated by pointer jumping over a 2^l-ancestor table)
//   T_i   = G o A_i                                            G = global transform of the frame
// executed by torch/lietorch as ~150 tiny kernels forward and ~300 backward for M = 20 bones (measured: 45 % of the
// training step once everything else is fused).  Here the tree is walked level by level inside ONE workgroup;
// the association order of the products differs from pointer jumping, the value does not (up to fp32 rounding).
//
// Gradients are plain Euclidean gradients w.r.t. the stored numbers.  Quaternions are unit by construction inside the
// chain, multiplication by a unit quaternion is an isometry of R^4, so the radial part of any incoming gradient stays
// radial and is removed where a real normalisation happens: at raw -> sk_r and at G's quaternion.
#include "bone_chain.inl"
#include "skgs_common.h"

namespace skgs {
namespace {

constexpr int CHAIN_THREADS = 256;

// (the bodies live in bone_chain.inl: the deform network's launches run them too)
__global__ void __launch_bounds__(CHAIN_THREADS) bone_chain_forward_kernel(const chain::ChainArgs c) {
  extern __shared__ float s_mem[];
  const chain::Prefetch pf = chain::prefetch(c, true, false);  // every global load of the pass in one round trip
  chain::forward_body(s_mem, c, pf);
}
__global__ void __launch_bounds__(CHAIN_THREADS) bone_chain_backward_kernel(const chain::ChainArgs c) {
  extern __shared__ float s_mem[];
  const chain::Prefetch pf = chain::prefetch(c, true, true);
  chain::backward_body(s_mem, c, c.g_sk_r_raw, true, pf);
}

}  // namespace
}  // namespace skgs

using namespace skgs;

extern "C" {

int skgs_bone_chain_forward(int32_t M, int32_t root, const int32_t* parents, const int32_t* level_nodes,
    const int32_t* level_start, int32_t num_levels, const float* sk_r_raw, const float* joints, const float* global_T,
    float* bone_T, float* chain_A, const int32_t* frame_index, skgs_stream_t stream) {
  SKGS_REQUIRE(M >= 1 && root >= 0 && root < M && num_levels >= 1, "bone_chain: bad skeleton sizes");
  SKGS_REQUIRE(parents && level_nodes && level_start && sk_r_raw && joints && bone_T, "bone_chain: NULL argument");
  const size_t lds = chain::forward_scratch_floats(M, num_levels) * 4;
  SKGS_REQUIRE(lds <= 64 * 1024, "bone_chain: skeleton too large for the LDS staging (about 740 bones)");
  chain::ChainArgs c{};
  c.M = M, c.root = root, c.num_levels = num_levels, c.parents = parents, c.level_nodes = level_nodes;
  c.level_start = level_start, c.sk_r_raw = sk_r_raw, c.joints = joints, c.global_T = global_T, c.frame_index = frame_index;
  c.bone_T = bone_T, c.chain_A = chain_A;
  hipLaunchKernelGGL(bone_chain_forward_kernel, dim3(1), dim3(CHAIN_THREADS), lds, (hipStream_t) stream, c);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

int skgs_bone_chain_backward(int32_t M, int32_t root, const int32_t* parents, const int32_t* level_nodes,
    const int32_t* level_start, int32_t num_levels, const float* sk_r_raw, const float* joints, const float* global_T,
    const float* chain_A, const float* g_bone_T, float* g_sk_r_raw, float* g_joints, float* g_global_T,
    const int32_t* frame_index, skgs_stream_t stream) {
  SKGS_REQUIRE(M >= 1 && root >= 0 && root < M && num_levels >= 1, "bone_chain: bad skeleton sizes");
  SKGS_REQUIRE(parents && level_nodes && level_start && sk_r_raw && joints && chain_A && g_bone_T && g_sk_r_raw,
      "bone_chain: NULL argument");
  const size_t lds = chain::backward_scratch_floats(M, num_levels) * 4;
  SKGS_REQUIRE(lds <= 64 * 1024, "bone_chain backward: skeleton too large for the LDS staging (about 560 bones)");
  chain::ChainArgs c{};
  c.M = M, c.root = root, c.num_levels = num_levels, c.parents = parents, c.level_nodes = level_nodes;
  c.level_start = level_start, c.sk_r_raw = sk_r_raw, c.joints = joints, c.global_T = global_T, c.frame_index = frame_index;
  c.chain_A = const_cast<float*>(chain_A), c.g_bone_T = g_bone_T, c.g_sk_r_raw = g_sk_r_raw, c.g_joints = g_joints;
  c.g_global_T = g_global_T;
  hipLaunchKernelGGL(bone_chain_backward_kernel, dim3(1), dim3(CHAIN_THREADS), lds, (hipStream_t) stream, c);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
