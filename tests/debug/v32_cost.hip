// microbenchmark: 32-bit in-register butterflies (k_vit64_fwd body) without memory traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#include "../../abracadabra_amd/csrc/dabx_vit64_steps.inc"
#define VIT64_PUSH(d0, d1) { aL[q >> 3] = __builtin_amdgcn_alignbit(aL[q >> 3], (uint32_t)(d0), 31); aH[q >> 3] = __builtin_amdgcn_alignbit(aH[q >> 3], (uint32_t)(d1), 31); ++q; }
#define VIT64_BFP(a, b, w) { const int E = R[a], O = R[b]; const int A0 = E + W[w], B0 = O - W[w], A1 = O + W[w], B1 = E - W[w]; VIT64_PUSH(A0 - B0, A1 - B1) R[a] = max(A0, B0); R[b] = max(A1, B1); }
#define VIT64_BFM(a, b, w) { const int E = R[a], O = R[b]; const int A0 = E - W[w], B0 = O + W[w], A1 = O - W[w], B1 = E + W[w]; VIT64_PUSH(A0 - B0, A1 - B1) R[a] = max(A0, B0); R[b] = max(A1, B1); }
template <int MEM> __global__ __launch_bounds__(64) void k(unsigned* out, uint2* dec, int n) {
  int lane = threadIdx.x;
  int R[64];
  for (int i = 0; i < 64; ++i) R[i] = (lane * 2654435761u + i * 40503u) & 0xfff;
  int W[8];
  for (int i = 0; i < 8; ++i) W[i] = 17 * (i + 1) + lane;
  unsigned sink = 0;
  uint2* d = dec + (size_t)blockIdx.x * n * 6 * 64 + lane;
  for (int t = 0; t < n; ++t) {
#define ONE(PH) { uint32_t aL[4] = {0,0,0,0}, aH[4] = {0,0,0,0}; int q = 0; VIT64_STEP_##PH \
      uint32_t wlo = (aL[0] << 24) | (aL[1] << 16) | (aL[2] << 8) | aL[3], whi = (aH[0] << 24) | (aH[1] << 16) | (aH[2] << 8) | aH[3]; \
      if (MEM) d[(size_t)(t * 6 + PH) * 64] = make_uint2(wlo, whi); else sink ^= wlo ^ whi; for (int i = 0; i < 8; ++i) W[i] += 1; }
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5)
  }
  unsigned r = sink;
  for (int i = 0; i < 64; ++i) r ^= R[i];
  out[blockIdx.x * 64 + lane] = r;
}
template <int MEM> void run(unsigned* d, uint2* dec) {
  for (int waves : {1024, 2048, 2432, 3072, 4096, 8192}) {
    int n = 257;  // x6 steps = 1542
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MEM>, dim3(waves), dim3(64), 0, 0, d, dec, 4); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k<MEM>, dim3(waves), dim3(64), 0, 0, d, dec, n); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double wave_steps = (double)waves * n * 6;
    printf("mem=%d waves %5d (%.2f/SIMD): %.3f ms, SIMD time per wave-step %.0f cycles@2.4GHz, %.2e trellis-steps/s\n", MEM, waves, waves / 1024.0, ms,
           ms * 1e-3 * 2.4e9 / (wave_steps / 1024), wave_steps * 64 / (ms * 1e-3));
  }
}
int main() { unsigned* d; uint2* dec; hipMalloc(&d, 8192 * 64 * 4); hipMalloc(&dec, (size_t)8192 * 1542 * 64 * 8);
  run<0>(d, dec); run<1>(d, dec); return 0; }
