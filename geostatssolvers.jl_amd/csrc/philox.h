// Philox4x32-10 counter RNG (device side).  Contract shared with oracle/philox.py:
//   key = (seed_lo, seed_hi), counter = (block_lo, block_hi, realisation, stream)
//   ua = ((x0 << 32 | x1) >> 11) * 2^-53, ub = ((x2 << 32 | x3) >> 11) * 2^-53
// Replaces `rand(rng, V, dims)` (/root/reference/src/simulation/fft.jl:163) and
// `randn(rng, n)` (/root/reference/src/simulation/lu.jl:209); see DESIGN.md section 6 for why the
// Julia RNG streams themselves cannot be reproduced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gss {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;
enum { STREAM_UNIFORM = 0, STREAM_NORMAL = 1 };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply (v_mad_u64_u32) gives both halves
    const uint64_t p0 = (uint64_t)PHILOX_M0 * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // three-input xor in one instruction (v_bitop3_b32, truth table 0x96)
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n1 = lo1;
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96), n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void philox_pair(uint64_t seed, uint32_t real, uint32_t stream, uint64_t block,
                                            double& ua, double& ub) {
  uint32_t x[4];
  philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), real, stream, (uint32_t)seed, (uint32_t)(seed >> 32), x);
  const double two53 = 1.0 / 9007199254740992.0;
  ua = (double)((((uint64_t)x[0] << 32) | x[1]) >> 11) * two53;
  ub = (double)((((uint64_t)x[2] << 32) | x[3]) >> 11) * two53;
}

__device__ __forceinline__ double philox_normal(uint64_t seed, uint32_t real, uint64_t e) {
  double ua, ub;
  philox_pair(seed, real, STREAM_NORMAL, e, ua, ub);
  return sqrt(-2.0 * log(1.0 - ua)) * cos(6.283185307179586476925286766559 * ub);
}

}  // namespace gss
