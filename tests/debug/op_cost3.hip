// microbenchmark 3: SALU issue rate per SIMD and its overlap with VALU work (8 waves/SIMD) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define SCLOB "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","scc","v10","v11","v12","v13"
#define S4 "s_xor_b32 s20, s24, s25\ns_xor_b32 s21, s25, s26\ns_xor_b32 s22, s26, s27\ns_xor_b32 s23, s27, s24\n"
#define SB4 "s_bitcmp1_b64 s[24:25], s26\ns_cselect_b32 s20, 7, 0\ns_xor_b32 s21, s21, s20\ns_bitcmp1_b64 s[26:27], s21\n"
#define V4 "v_add_u32 v10, v10, v11\nv_add_u32 v11, v11, v12\nv_add_u32 v12, v12, v13\nv_add_u32 v13, v13, v10\n"
#define D4 "v_dot4_i32_i8 v10, v11, v12, v13\nv_dot4_i32_i8 v11, v12, v13, v10\nv_dot4_i32_i8 v12, v13, v10, v11\nv_dot4_i32_i8 v13, v10, v11, v12\n"
template <int V> __global__ __launch_bounds__(256) void k(int* out, int n) {
  for (int i = 0; i < n; ++i) {
    if (V == 0) asm volatile(REP8(S4) ::: SCLOB);                 // 32 SALU
    if (V == 1) asm volatile(REP8(V4) ::: SCLOB);                 // 32 VALU full rate
    if (V == 2) asm volatile(REP8(S4 V4) ::: SCLOB);              // 32 + 32 interleaved
    if (V == 3) asm volatile(REP8(SB4) ::: SCLOB);                // traceback-like SALU chain (dependent)
    if (V == 4) asm volatile(REP8(D4) ::: SCLOB);                 // 32 dot4
    if (V == 5) asm volatile(REP8(D4 S4) ::: SCLOB);              // 32 dot4 + 32 SALU
    if (V == 6) asm volatile(REP8(D4 S4 S4) ::: SCLOB);           // 32 dot4 + 64 SALU
  }
  out[blockIdx.x * 256 + threadIdx.x] = n;
}
template <int V> void run(const char* name, int* d, int wps) {
  int n = 2048, grid = 256 * wps;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-34s %d waves/SIMD: %.2f cycles@2.4GHz per group of 32 per wave-slot per SIMD /32 = %.2f\n", name, wps,
         ms * 1e-3 * 2.4e9 / ((double)wps * n), ms * 1e-3 * 2.4e9 / ((double)wps * n) / 32);
}
int main() { int* d; hipMalloc(&d, 2048 * 256 * 4);
  for (int wps : {1, 4, 8}) {
    run<0>("32 SALU (independent)", d, wps); run<1>("32 VALU add", d, wps); run<2>("32 SALU + 32 VALU add", d, wps); run<3>("32 SALU traceback-like", d, wps);
    run<4>("32 dot4", d, wps); run<5>("32 dot4 + 32 SALU", d, wps); run<6>("32 dot4 + 64 SALU", d, wps); }
  return 0; }
