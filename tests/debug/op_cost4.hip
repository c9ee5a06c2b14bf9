// microbenchmark 4: issue cost of single VALU instruction forms on gfx950 (8 waves/SIMD, independent registers) + the clock
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","vcc","s20","s21","s22","s23"
#define Q(op, suf) op " v10, v11, v12" suf "\n" op " v11, v12, v13" suf "\n" op " v12, v13, v10" suf "\n" op " v13, v10, v11" suf "\n"
#define Q3(op) op " v10, v11, v12, v13\n" op " v11, v12, v13, v10\n" op " v12, v13, v10, v11\n" op " v13, v10, v11, v12\n"
#define M(op, suf) op " v10, v11" suf "\n" op " v11, v12" suf "\n" op " v12, v13" suf "\n" op " v13, v10" suf "\n"
#define QP " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define RR " row_ror:8 row_mask:0xf bank_mask:0xf"
template <int V> __global__ __launch_bounds__(256) void k(int* out, int n) {
  for (int i = 0; i < n; ++i) {
    if (V == 0) asm volatile(REP8(Q("v_add_u32", "")) ::: CLOB);
    if (V == 1) asm volatile(REP8(Q("v_max_i32", "")) ::: CLOB);
    if (V == 2) asm volatile(REP8(Q("v_max_u32", "")) ::: CLOB);
    if (V == 3) asm volatile(REP8(Q("v_max_f32", "")) ::: CLOB);
    if (V == 4) asm volatile(REP8(Q("v_max_i32_dpp", QP)) ::: CLOB);
    if (V == 5) asm volatile(REP8(Q("v_max_f32_dpp", QP)) ::: CLOB);
    if (V == 6) asm volatile(REP8(Q("v_add_u32_dpp", QP)) ::: CLOB);
    if (V == 7) asm volatile(REP8(M("v_mov_b32_dpp", QP)) ::: CLOB);
    if (V == 8) asm volatile(REP8(Q("v_max_f32_dpp", RR)) ::: CLOB);
    if (V == 9) asm volatile(REP8(Q("v_pk_max_i16", "")) ::: CLOB);
    if (V == 10) asm volatile(REP8(Q("v_pk_add_i16", "")) ::: CLOB);
    if (V == 11) asm volatile(REP8(Q3("v_max3_i32")) ::: CLOB);
    if (V == 12) asm volatile(REP8(Q("v_max_i16", "")) ::: CLOB);
    if (V == 13) asm volatile(REP8(Q("v_min_f32", "")) ::: CLOB);
    if (V == 14) asm volatile(REP8(Q("v_sub_f32", "")) ::: CLOB);
    if (V == 15) asm volatile(REP8("v_pk_add_f32 v[10:11], v[12:13], v[14:15]\nv_pk_add_f32 v[12:13], v[14:15], v[16:17]\nv_pk_add_f32 v[14:15], v[16:17], v[10:11]\nv_pk_add_f32 v[16:17], v[10:11], v[12:13]\n") ::: "v10","v11","v12","v13","v14","v15","v16","v17");
    if (V == 16) asm volatile(REP8(Q3("v_add3_u32")) ::: CLOB);
    if (V == 17) asm volatile(REP8(Q("v_and_b32", "")) ::: CLOB);
    if (V == 18) asm volatile(REP8(Q3("v_alignbit_b32")) ::: CLOB);
    if (V == 19) asm volatile(REP8(Q3("v_perm_b32")) ::: CLOB);
    if (V == 20) asm volatile(REP8(Q3("v_bfi_b32")) ::: CLOB);
    if (V == 21) asm volatile(REP8(Q3("v_med3_i32")) ::: CLOB);
    if (V == 22) asm volatile(REP8(Q("v_sub_u32_dpp", RR)) ::: CLOB);
    if (V == 23) asm volatile(REP8(Q("v_max_i32_sdwa", " dst_sel:DWORD src0_sel:DWORD src1_sel:DWORD")) ::: CLOB);
    if (V == 24) asm volatile(REP8(Q("v_pk_max_f16", "")) ::: CLOB);
    if (V == 25) asm volatile(REP8(Q3("v_max3_f32")) ::: CLOB);
    if (V == 26) asm volatile(REP8(Q3("v_fma_f32")) ::: CLOB);
    if (V == 27) asm volatile(REP8(Q("v_xor_b32", "")) ::: CLOB);
    if (V == 28) asm volatile(REP8(Q("v_lshlrev_b32", "")) ::: CLOB);
    if (V == 29) asm volatile(REP8(Q("v_max_u16", "")) ::: CLOB);
  }
  out[blockIdx.x * 256 + threadIdx.x] = n;
}
__global__ void clk(unsigned long long* o) {
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = 0; i < 4000; ++i) asm volatile(REP8(Q("v_add_u32", "")) ::: CLOB);
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = t1 - t0; o[1] = r1 - r0; }
}
template <int V> void run(const char* name, int* d) {
  const int wps = 8; int n = 1024, grid = 256 * wps;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.2f ns per instruction per SIMD = %.2f cycles@2.4GHz\n", name, ms * 1e6 / ((double)wps * n * 32), ms * 1e-3 * 2.4e9 / ((double)wps * n * 32));
}
int main() { int* d; hipMalloc(&d, 2048 * 256 * 4);
  unsigned long long* o; hipMalloc(&o, 16); hipLaunchKernelGGL(clk, dim3(2048), dim3(256), 0, 0, o); unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("cycle counter %llu ticks, wall clock %llu ticks (100 MHz): counter rate %.1f MHz\n", h[0], h[1], h[0] * 100.0 / h[1]);
  run<0>("v_add_u32", d); run<1>("v_max_i32", d); run<2>("v_max_u32", d); run<3>("v_max_f32", d); run<4>("v_max_i32_dpp quad", d); run<5>("v_max_f32_dpp quad", d);
  run<6>("v_add_u32_dpp quad", d); run<7>("v_mov_b32_dpp quad", d); run<8>("v_max_f32_dpp row_ror", d); run<9>("v_pk_max_i16", d); run<10>("v_pk_add_i16", d);
  run<11>("v_max3_i32", d); run<12>("v_max_i16", d); run<13>("v_min_f32", d); run<14>("v_sub_f32", d); run<15>("v_pk_add_f32", d); run<16>("v_add3_u32", d);
  run<17>("v_and_b32", d); run<18>("v_alignbit_b32", d); run<19>("v_perm_b32", d); run<20>("v_bfi_b32", d); run<21>("v_med3_i32", d); run<22>("v_sub_u32_dpp row_ror", d);
  run<23>("v_max_i32_sdwa", d); run<24>("v_pk_max_f16", d); run<25>("v_max3_f32", d); run<26>("v_fma_f32", d); run<27>("v_xor_b32", d); run<28>("v_lshlrev_b32", d); run<29>("v_max_u16", d);
  return 0; }
