// Micro-benchmark (round 3): does v_mfma_f32_32x32x16 leave more vector-issue room than v_mfma_f32_16x16x32 at equal FLOPs?
// Two waves per SIMD (512-thread workgroups, one per CU), a loop of MFMAs interleaved with independent VALU work.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_issue.hip -o /tmp/mfma_issue && /tmp/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NVALU>      // SHAPE 16: 12 x 16x16x32 per iteration; 32: 6 x 32x32x16 (same FLOPs)
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
  const int t = threadIdx.x;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(in[(t + i) & 255]); b[i] = (__bf16)(in[(t * 3 + i) & 255]); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[(t + 17 * i) & 255];
  f32x4 c16[12]; f32x16 c32[6];
  for (int i = 0; i < 12; ++i) c16[i] = (f32x4){0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) c32[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (SHAPE == 16) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        c16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c16[i], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < (NVALU + 11 - i) / 12; ++u) v[(i + u) & 7] = __builtin_fmaf(v[(i + u) & 7], 1.0001f, 0.5f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c32[i], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < (NVALU + 5 - i) / 6; ++u) v[(i + u) & 7] = __builtin_fmaf(v[(i + u) & 7], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += c16[i][0] + c16[i][3];
  for (int i = 0; i < 6; ++i) s += c32[i][0] + c32[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + t] = s;
}

template <int SHAPE, int NVALU>
static float run(const float* in, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, NVALU>), dim3(256), dim3(512), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SHAPE, NVALU>), dim3(256), dim3(512), 0, 0, in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5 * 1e3f;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 1024); hipMalloc(&out, 256 * 512 * 4);
  std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = 0.001f * (i % 37) - 0.01f;
  hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
  const int iters = 4000;
  printf("iters %d, per iteration 12 x 16x16x32 or 6 x 32x32x16 bf16 MFMA (192 pipe cycles), two waves per SIMD\n", iters);
#define ROW(NV) printf("VALU per iteration %3d : 16x16x32 %8.1f us   32x32x16 %8.1f us\n", NV, run<16, NV>(in, out, iters), run<32, NV>(in, out, iters));
  ROW(0) ROW(12) ROW(24) ROW(36) ROW(48) ROW(72)
  return 0;
}
