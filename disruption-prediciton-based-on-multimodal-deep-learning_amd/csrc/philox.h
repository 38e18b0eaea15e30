// Philox4x32-10 (Salmon et al., SC'11), the counter-based generator behind the mask-free dropout kernels: four 32-bit words per
// (counter, key), no state to carry -- the backward pass regenerates the words of the forward pass from the same (seed, step, tag, index).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void md_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep / drop decisions of the four elements 4*i4 .. 4*i4+3 of a tensor: state = {seed, step} (device int64[2]), tag = call site
struct MdDropKey { uint32_t k0, k1, step; };
__device__ __forceinline__ MdDropKey md_drop_key(const int64_t* __restrict__ state) {
  const uint64_t seed = (uint64_t)state[0], step = (uint64_t)state[1];
  MdDropKey k; k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32); k.step = (uint32_t)step;
  return k;
}
__device__ __forceinline__ void md_drop_keep4(const MdDropKey& k, int tag, int64_t i4, float keep, float m[4]) {
  uint32_t w[4];
  md_philox4x32_10((uint32_t)i4, (uint32_t)((uint64_t)i4 >> 32), (uint32_t)tag, k.step, k.k0, k.k1, w);
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] = (float)(w[e] >> 8) * (1.0f / 16777216.0f) < keep ? 1.f : 0.f;
}
