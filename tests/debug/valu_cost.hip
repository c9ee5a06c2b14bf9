// microbenchmark: issue cost of the ACS step's instructions on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define BODY_FULL  "v_dot4_i32_i8 %[S], %[nsig], %[xs], %[pm]\n\t" "v_dot4_i32_i8 %[K], %[sig], %[xs], %[pm]\n\t" "v_readlane_b32 %[xn], %[xv], 7\n\t" "v_sub_u32 %[D], %[Kp], %[pm]\n\t" "v_alignbit_b32 %[bits], %[bits], %[D], 31\n\t" "v_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define BODY_NODOT "v_sub_u32 %[S], %[pm], %[nsig]\n\t" "v_add_u32 %[K], %[pm], %[sig]\n\t" "v_readlane_b32 %[xn], %[xv], 7\n\t" "v_sub_u32 %[D], %[Kp], %[pm]\n\t" "v_alignbit_b32 %[bits], %[bits], %[D], 31\n\t" "v_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define BODY_NORL  "v_dot4_i32_i8 %[S], %[nsig], %[xs], %[pm]\n\t" "v_dot4_i32_i8 %[K], %[sig], %[xs], %[pm]\n\t" "s_mov_b32 %[xn], %[xs]\n\t" "v_sub_u32 %[D], %[Kp], %[pm]\n\t" "v_alignbit_b32 %[bits], %[bits], %[D], 31\n\t" "v_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define BODY_NODPP "v_dot4_i32_i8 %[S], %[nsig], %[xs], %[pm]\n\t" "v_dot4_i32_i8 %[K], %[sig], %[xs], %[pm]\n\t" "v_readlane_b32 %[xn], %[xv], 7\n\t" "v_sub_u32 %[D], %[Kp], %[pm]\n\t" "v_alignbit_b32 %[bits], %[bits], %[D], 31\n\t" "v_max_i32 %[pm], %[S], %[K]"
#define BODY_ADD6  "v_add_u32 %[S], %[pm], %[nsig]\n\t" "v_add_u32 %[K], %[pm], %[sig]\n\t" "v_add_u32 %[D], %[Kp], %[pm]\n\t" "v_add_u32 %[bits], %[bits], %[D]\n\t" "v_add_u32 %[D], %[D], %[S]\n\t" "v_max_i32 %[pm], %[S], %[K]\n\ts_mov_b32 %[xn], %[xs]"
#define BODY_VX    "v_dot4_i32_i8 %[S], %[nsig], %[xv], %[pm]\n\t" "v_dot4_i32_i8 %[K], %[sig], %[xv], %[pm]\n\t" "v_sub_u32 %[D], %[Kp], %[pm]\n\t" "v_alignbit_b32 %[bits], %[bits], %[D], 31\n\t" "s_nop 0\n\t" "v_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_mov_b32 %[xn], %[xs]"
#define BODY_VX2   "v_dot4_i32_i8 %[S], %[nsig], %[xv], %[pm]\n\t" "v_dot4_i32_i8 %[K], %[sig], %[xv], %[pm]\n\t" "v_sub_u32 %[D], %[Kp], %[pm]\n\t" "v_alignbit_b32 %[bits], %[bits], %[D], 31\n\t" "v_nop\n\t" "v_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_mov_b32 %[xn], %[xs]"
#define BODY_IND6  "v_add_u32 %[S], %[sig], %[nsig]\n\t" "v_add_u32 %[K], %[xv], %[sig]\n\t" "v_add_u32 %[D], %[Kp], %[sig]\n\t" "v_add_u32 %[bits], %[xv], %[nsig]\n\t" "v_add_u32 %[D], %[sig], %[xv]\n\t" "v_max_i32 %[pm], %[sig], %[nsig]\n\ts_mov_b32 %[xn], %[xs]"
template <int V> __global__ __launch_bounds__(256) void k(int* out, int n, int lds_touch) {
  extern __shared__ int lds[];
  int lane = threadIdx.x;
  int pm = lane, Kp = lane * 3, sig = 0x01ff01ff ^ lane, nsig = ~sig, xv = lane * 0x01010101, xs = 0x11223344; unsigned bits = 0;
  if (lds_touch) lds[lane] = lane;
  for (int i = 0; i < n; ++i) {
    int S, K, D, xn;
#define STEP(B) asm volatile(B : [pm] "+v"(pm), [bits] "+v"(bits), [S] "=&v"(S), [K] "=&v"(K), [D] "=&v"(D), [xn] "=&s"(xn) : [sig] "v"(sig), [nsig] "v"(nsig), [xs] "s"(xs), [xv] "v"(xv), [Kp] "v"(Kp)); Kp = K; xs = xn;
#define STEP8(B) STEP(B) STEP(B) STEP(B) STEP(B) STEP(B) STEP(B) STEP(B) STEP(B)
    if (V == 0) { STEP8(BODY_FULL) } else if (V == 1) { STEP8(BODY_NODOT) } else if (V == 2) { STEP8(BODY_NORL) }
    else if (V == 3) { STEP8(BODY_NODPP) } else if (V == 4) { STEP8(BODY_ADD6) } else if (V == 5) { STEP8(BODY_IND6) } else if (V == 6) { STEP8(BODY_VX) } else { STEP8(BODY_VX2) }
  }
  out[blockIdx.x * 256 + lane] = pm + bits + Kp + xs;
}
template <int V> void run(const char* name, int* d) {
  int n = 2048;   // x8 steps
  for (int wps : {2, 3, 4}) {
    // waves per SIMD controlled by LDS: 160 KB per CU; blocks of 4 waves (1 per SIMD)
    size_t lds = 160 * 1024 / wps - 64;
    if (lds > 65536) hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int grid = 256 * wps * 4;    // 4 rounds
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, 16, 1); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), lds, 0, d, n, 1); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double steps_per_simd = (double)grid * 4 / 1024 * n * 8;    // waves per SIMD in total x steps
    printf("%-8s waves/SIMD %d: %.3f ms, %.2f ns per wave-step per SIMD = %.1f cycles @2.4GHz (6 instr/step)\n", name, wps, ms, ms * 1e6 / steps_per_simd, ms * 1e6 / steps_per_simd * 2.4);
  }
}
int main() { int* d; hipMalloc(&d, 256 * 8192 * 4 * 4);
  run<0>("full", d); run<6>("vx_snop", d); run<7>("vx_vnop", d); return 0; }
