// microbenchmark: packed-i16 butterfly + decision extraction (lane-per-codeword Viterbi), VALU rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(unsigned x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ unsigned as_u(s2 x) { return __builtin_bit_cast(unsigned, x); }
__global__ __launch_bounds__(256) void k(unsigned* out, int n) {
  int lane = threadIdx.x;
  s2 R[64];
  for (int i = 0; i < 64; ++i) R[i] = as_s2((lane * 2654435761u + i * 40503u) & 0x0fff0fffu);
  s2 W[8];
  for (int i = 0; i < 8; ++i) W[i] = as_s2(0x00110007u * (i + 1));
  unsigned acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, sink = 0;
  for (int t = 0; t < n; ++t) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      s2 E = R[2 * j], O = R[2 * j + 1], M = W[j & 7];
      s2 A0 = E + M, B0 = O - M, A1 = O + M, B1 = E - M;
      s2 n0 = __builtin_elementwise_max(A0, B0), n1 = __builtin_elementwise_max(A1, B1);
      unsigned d0 = as_u(A0 - B0), d1 = as_u(A1 - B1);
      if (j < 16) { acc0 = (acc0 >> 1) | (d0 & 0x80008000u); acc2 = (acc2 >> 1) | (d1 & 0x80008000u); }
      else { acc1 = (acc1 >> 1) | (d0 & 0x80008000u); acc3 = (acc3 >> 1) | (d1 & 0x80008000u); }
      R[2 * j] = n0; R[2 * j + 1] = n1;
    }
    sink ^= acc0 ^ acc1 ^ acc2 ^ acc3;
    for (int i = 0; i < 8; ++i) W[i] = W[i] + as_s2(0x00010001u);
  }
  unsigned r = sink;
  for (int i = 0; i < 64; ++i) r ^= as_u(R[i]);
  out[blockIdx.x * 256 + lane] = r;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 256 * 4);
  for (int wps : {1, 2, 4}) {
    int grid = 256 * wps * 2, n = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 10); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double wave_steps = (double)grid * 4 * n;            // each wave-step = 128 trellis-steps
    printf("blocks/CU %d: %.3f ms, %.1f ns per wave-step per SIMD (=%.2f cycles@2.4GHz per trellis-step), %.2e trellis-steps/s\n", wps * 2, ms,
           ms * 1e6 / (wave_steps / 1024), ms * 1e6 / (wave_steps / 1024) * 2.4 / 128, wave_steps * 128 / (ms * 1e-3));
  }
  return 0;
}
