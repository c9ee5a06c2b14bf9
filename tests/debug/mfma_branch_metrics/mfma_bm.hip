// Layout check and issue cost of v_mfma_i32_4x4x4_16b_i8 as a branch-metric generator: D_r[lane] should be
// sum_k A[lane (lane & ~3) + r][k] * B[lane][k]  (A = soft values of step r, B = the lane's +-1 signs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k_layout(const int* a, const int* b, int* d) {
  const int l = threadIdx.x;
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_4x4x4i8(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[4 * l + r] = acc[r];
}
#define REP4(x) x x x x
__global__ __launch_bounds__(256) void k_cost(int* out, int n, int mode) {
  int pm = threadIdx.x, K, S, D, bits = 0, xa = threadIdx.x * 0x01020304, sg = 0x01ff01ff;
  v4i bm = {1, 2, 3, 4};
  for (int i = 0; i < n; ++i) {
    if (mode == 0)        // 4 steps: mfma + 4 x (add, sub, sub, alignbit, max_dpp)
      asm volatile("v_mfma_i32_4x4x4_16b_i8 %[bm], %[xa], %[sg], 0\n\ts_nop 7\n\t"
                   "v_sub_u32 %[S], %[pm], %[bm0]\n\tv_add_u32 %[K], %[pm], %[bm0]\n\tv_sub_u32 %[D], %[K], %[pm]\n\tv_alignbit_b32 %[bits], %[bits], %[D], 31\n\tv_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_sub_u32 %[S], %[pm], %[bm1]\n\tv_add_u32 %[K], %[pm], %[bm1]\n\tv_sub_u32 %[D], %[K], %[pm]\n\tv_alignbit_b32 %[bits], %[bits], %[D], 31\n\tv_max_i32_dpp %[pm], %[S], %[K] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_sub_u32 %[S], %[pm], %[bm2]\n\tv_add_u32 %[K], %[pm], %[bm2]\n\tv_sub_u32 %[D], %[K], %[pm]\n\tv_alignbit_b32 %[bits], %[bits], %[D], 31\n\tv_max_i32_dpp %[pm], %[S], %[K] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_sub_u32 %[S], %[pm], %[bm3]\n\tv_add_u32 %[K], %[pm], %[bm3]\n\tv_sub_u32 %[D], %[K], %[pm]\n\tv_alignbit_b32 %[bits], %[bits], %[D], 31\n\tv_max_i32_dpp %[pm], %[S], %[K] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                   : [pm] "+v"(pm), [bits] "+v"(bits), [K] "=&v"(K), [S] "=&v"(S), [D] "=&v"(D), [bm] "+v"(bm)
                   : [xa] "v"(xa), [sg] "v"(sg), [bm0] "v"(bm[0]), [bm1] "v"(bm[1]), [bm2] "v"(bm[2]), [bm3] "v"(bm[3]));
    else                  // 4 steps the current way: dot4, dot4, sub, alignbit, max_dpp
      asm volatile(REP4("v_dot4_i32_i8 %[S], %[sg], %[xa], %[pm]\n\tv_dot4_i32_i8 %[K], %[xa], %[sg], %[pm]\n\ts_nop 0\n\tv_sub_u32 %[D], %[K], %[pm]\n\tv_alignbit_b32 %[bits], %[bits], %[D], 31\n\tv_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t")
                   : [pm] "+v"(pm), [bits] "+v"(bits), [K] "=&v"(K), [S] "=&v"(S), [D] "=&v"(D) : [xa] "v"(xa), [sg] "v"(sg));
  }
  out[blockIdx.x * 256 + threadIdx.x] = pm + bits + bm[0];
}
int main() {
  int ha[64], hb[64], hd[256], *a, *b, *d;
  srand(1);
  for (int i = 0; i < 64; ++i) { ha[i] = rand(); hb[i] = rand(); }
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d); hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int ref = 0, la = (l & ~3) + r;
    for (int k = 0; k < 4; ++k) ref += (int)(int8_t)(ha[la] >> (8 * k)) * (int)(int8_t)(hb[l] >> (8 * k));
    if (ref != hd[4 * l + r]) ++bad;
  }
  printf("layout D_r[lane] = sum_k A[block row r][k] * B[lane][k]: %d mismatches of 256\n", bad);
  int* o; hipMalloc(&o, 2048 * 256 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    const int grid = 256 * 8, n = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_cost, dim3(grid), dim3(256), 0, 0, o, 16, mode); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_cost, dim3(grid), dim3(256), 0, 0, o, n, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.2f cycles@2.4GHz per trellis step per SIMD\n", mode == 0 ? "mfma branch metrics + add/sub" : "dot4 pair (current)", ms * 1e-3 * 2.4e9 / (8.0 * n * 4));
  }
  return 0; }
