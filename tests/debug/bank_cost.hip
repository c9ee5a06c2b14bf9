// microbenchmark: does VALU cost depend on VGPR bank (index mod 4) of the two sources / dest?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int V> __global__ __launch_bounds__(256) void k(int* out, int n) {
  int acc = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    if (V == 0)      asm volatile(REP8("v_add_u32 v10, v20, v24\n v_add_u32 v11, v21, v25\n v_add_u32 v12, v22, v26\n v_add_u32 v13, v23, v27\n") ::: "v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27");   // sources same bank
    else if (V == 1) asm volatile(REP8("v_add_u32 v10, v20, v25\n v_add_u32 v11, v21, v26\n v_add_u32 v12, v22, v27\n v_add_u32 v13, v23, v24\n") ::: "v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27");   // sources different banks
    else if (V == 2) asm volatile(REP8("v_add_u32 v20, v20, v25\n v_add_u32 v21, v21, v26\n v_add_u32 v22, v22, v27\n v_add_u32 v23, v23, v24\n") ::: "v20","v21","v22","v23","v24","v25","v26","v27");   // in place, different banks
    else if (V == 3) asm volatile(REP8("v_add_u32 v10, v20, v20\n v_add_u32 v11, v21, v21\n v_add_u32 v12, v22, v22\n v_add_u32 v13, v23, v23\n") ::: "v10","v11","v12","v13","v20","v21","v22","v23");   // same register twice
    else if (V == 4) asm volatile(REP8("v_add_u32 v10, s4, v20\n v_add_u32 v11, s4, v21\n v_add_u32 v12, s4, v22\n v_add_u32 v13, s4, v23\n") ::: "v10","v11","v12","v13","v20","v21","v22","v23","s4");   // one VGPR source
    else if (V == 5) asm volatile(REP8("v_alignbit_b32 v10, v20, v25, 31\n v_alignbit_b32 v11, v21, v26, 31\n v_alignbit_b32 v12, v22, v27, 31\n v_alignbit_b32 v13, v23, v24, 31\n") ::: "v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27");
    else             asm volatile(REP8("v_max_i32 v10, v20, v25\n v_sub_u32 v11, v21, v26\n v_max_i32 v12, v22, v27\n v_sub_u32 v13, v23, v24\n") ::: "v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27");
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int V> void run(const char* name, int* d) {
  int n = 4096, grid = 256 * 4;     // 4 blocks/CU = 4 waves/SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)grid * 4 / 1024 * n * 32;
  printf("%-28s %.3f ms  %.2f cycles@2.4GHz per VALU instr per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { int* d; hipMalloc(&d, 1024 * 256 * 4);
  run<0>("srcs same bank", d); run<1>("srcs different banks", d); run<2>("in place, diff banks", d); run<3>("same reg twice", d);
  run<4>("sgpr + vgpr", d); run<5>("alignbit diff banks", d); run<6>("max/sub mix", d); return 0; }
