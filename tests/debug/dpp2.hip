#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  int K = 1000 + lane, S = 10 * lane;
  int a, b, c, d;
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(a) : "v"(S), "v"(K));
  asm volatile("s_nop 1\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(b) : "v"(S), "v"(K));
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(c) : "v"(S), "v"(K));
  asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(S), "v"(K));
  out[lane*4+0]=a; out[lane*4+1]=b; out[lane*4+2]=c; out[lane*4+3]=d;
}
int main() {
  int *o; int ho[256];
  hipMalloc(&o, sizeof ho);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  for (int l = 0; l < 4; l++) printf("lane %d: K=%d S=%d S_nbr=%d | subrev_dpp=%d sub_dpp=%d subrev_bc=%d max_dpp=%d\n", l, 1000+l, 10*l, 10*(l^1), ho[l*4], ho[l*4+1], ho[l*4+2], ho[l*4+3]);
  return 0;
}
