// microbenchmark: issue cost per VALU opcode on gfx950 (4 waves/SIMD, independent instructions)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define CLOB "v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27","s4"
#define K4(OP1, OP2) REP8(OP1 " v10, v20, v25\n" OP2 " v11, v21, v26\n" OP1 " v12, v22, v27\n" OP2 " v13, v23, v24\n")
#define K4_3(OP) REP8(OP " v10, v20, v25, v26\n" OP " v11, v21, v26, v27\n" OP " v12, v22, v27, v24\n" OP " v13, v23, v24, v25\n")
template <int V> __global__ __launch_bounds__(256) void k(int* out, int n) {
  for (int i = 0; i < n; ++i) {
    if (V == 0) asm volatile(K4("v_add_u32", "v_add_u32") ::: CLOB);
    if (V == 1) asm volatile(K4("v_sub_u32", "v_sub_u32") ::: CLOB);
    if (V == 2) asm volatile(K4("v_max_i32", "v_max_i32") ::: CLOB);
    if (V == 3) asm volatile(K4("v_add_u32", "v_sub_u32") ::: CLOB);
    if (V == 4) asm volatile(K4("v_add_u32", "v_max_i32") ::: CLOB);
    if (V == 5) asm volatile(K4("v_and_b32", "v_xor_b32") ::: CLOB);
    if (V == 6) asm volatile(K4("v_min_i32", "v_min_i32") ::: CLOB);
    if (V == 7) asm volatile(K4_3("v_add3_u32") ::: CLOB);
    if (V == 8) asm volatile(K4_3("v_dot4_i32_i8") ::: CLOB);
    if (V == 9) asm volatile(K4_3("v_max3_i32") ::: CLOB);
    if (V == 10) asm volatile(K4("v_add_f32", "v_add_f32") ::: CLOB);
    if (V == 11) asm volatile(K4("v_max_f32", "v_max_f32") ::: CLOB);
    if (V == 12) asm volatile(K4("v_sub_f32", "v_sub_f32") ::: CLOB);
    if (V == 13) asm volatile(K4("v_max_u32", "v_max_u32") ::: CLOB);
    if (V == 14) asm volatile(K4("v_sub_i32", "v_sub_i32") ::: CLOB);
    if (V == 15) asm volatile(K4("v_pk_add_i16", "v_pk_max_i16") ::: CLOB);
    if (V == 16) asm volatile(K4("v_subrev_u32", "v_subrev_u32") ::: CLOB);
    if (V == 17) asm volatile(K4("v_max_i16", "v_max_i16") ::: CLOB);
  }
  out[blockIdx.x * 256 + threadIdx.x] = n;
}
template <int V> void run(const char* name, int* d) {
  int n = 4096, grid = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-24s %.2f cycles@2.4GHz per instr per SIMD\n", name, ms * 1e-3 * 2.4e9 / ((double)grid * 4 / 1024 * n * 32));
}
int main() { int* d; hipMalloc(&d, 1024 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_sub_u32", d); run<2>("v_max_i32", d); run<3>("add/sub mix", d); run<4>("add/max mix", d); run<5>("and/xor", d);
  run<6>("v_min_i32", d); run<7>("v_add3_u32", d); run<8>("v_dot4_i32_i8", d); run<9>("v_max3_i32", d); run<10>("v_add_f32", d); run<11>("v_max_f32", d);
  run<12>("v_sub_f32", d); run<13>("v_max_u32", d); run<14>("v_sub_i32?", d); run<15>("pk_add/pk_max i16", d); run<16>("v_subrev_u32", d); run<17>("v_max_i16", d); return 0; }
