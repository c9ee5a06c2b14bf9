// debug harness: compare the asm ACS run against a builtin-only reference (forward pass only)
#include "../../abracadabra_amd/csrc/dabx_kernels.hip"
#include <cstdio>
#include <cstdlib>
template <int PH> __device__ int exch_ref(int v, int lane) { return exchange<PH>(v, lane); }
template <int PH> __device__ void acs_ref(int& pm, int sig, int nsig, int xs, int lane, uint32_t& bits) {
  int keep = __builtin_amdgcn_sdot4(sig, xs, pm, false), send = __builtin_amdgcn_sdot4(nsig, xs, pm, false);
  int recv = exch_ref<PH>(send, lane);
  bits = __builtin_amdgcn_alignbit(bits, (uint32_t)(keep - recv), 31);
  pm = max(keep, recv);
}
__global__ void k(const int* in, int* out, int cnt, int s0) {
  int lane = threadIdx.x;
  int sig[6], nsig[6];
  for (int ph = 0; ph < 6; ++ph) { sig[ph] = in[64 * ph + lane]; nsig[ph] = in[64 * (6 + ph) + lane]; }
  int xv = in[64 * 12 + lane];
  int pm0 = in[64 * 13 + lane];
  for (int variant = 0; variant < 3; ++variant) {
    int pm = pm0; uint32_t bits = 0;
    int lane_x32 = (lane ^ 32) << 2;
    if (variant == 0) acs_run<0>(pm, sig, nsig, xv, s0, cnt, lane_x32, bits);
    if (variant == 1) acs_run<2>(pm, sig, nsig, xv, s0, cnt, lane_x32, bits);
    if (variant == 2) acs_run<4>(pm, sig, nsig, xv, s0, cnt, lane_x32, bits);
    int pr = pm0; uint32_t br = 0;
    int ph = variant * 2;
    for (int s = 0; s < cnt; ++s) {
      int xs = __builtin_amdgcn_readlane(xv, s0 + s);
      switch (ph) {
        case 0: acs_ref<0>(pr, sig[0], nsig[0], xs, lane, br); break;
        case 1: acs_ref<1>(pr, sig[1], nsig[1], xs, lane, br); break;
        case 2: acs_ref<2>(pr, sig[2], nsig[2], xs, lane, br); break;
        case 3: acs_ref<3>(pr, sig[3], nsig[3], xs, lane, br); break;
        case 4: acs_ref<4>(pr, sig[4], nsig[4], xs, lane, br); break;
        default: acs_ref<5>(pr, sig[5], nsig[5], xs, lane, br); break;
      }
      ph = (ph + 1) % 6;
    }
    out[(variant * 64 + lane) * 4 + 0] = pm; out[(variant * 64 + lane) * 4 + 1] = pr;
    out[(variant * 64 + lane) * 4 + 2] = (int)bits; out[(variant * 64 + lane) * 4 + 3] = (int)br;
  }
}
int main() {
  static int h[64 * 14], ho[3 * 64 * 4]; int *d, *o;
  srand(3);
  for (int i = 0; i < 64 * 12; i++) { int sg = 0; for (int j = 0; j < 4; j++) sg |= ((rand() & 1) ? 0xff : 0x01) << (8 * j); h[i] = sg; }
  for (int ph = 0; ph < 6; ph++) for (int l = 0; l < 64; l++) { int sg = h[64 * ph + l], ng = 0; for (int j = 0; j < 4; j++) ng |= ((((sg >> (8 * j)) & 0xff) == 0xff) ? 0x01 : 0xff) << (8 * j); h[64 * (6 + ph) + l] = ng; }
  for (int l = 0; l < 64; l++) { int x = 0; for (int j = 0; j < 4; j++) x |= ((rand() % 255 - 127) & 0xff) << (8 * j); h[64 * 12 + l] = x; h[64 * 13 + l] = rand() % 2000 - 1000; }
  (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, sizeof ho); (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int cnt : {1, 5, 6, 7, 32}) for (int s0 : {0, 32}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, cnt, s0); (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int v = 0; v < 3; v++) { int bp = 0, bb = 0; for (int l = 0; l < 64; l++) { bp += ho[(v * 64 + l) * 4] != ho[(v * 64 + l) * 4 + 1]; bb += ho[(v * 64 + l) * 4 + 2] != ho[(v * 64 + l) * 4 + 3]; }
      printf("cnt %2d s0 %2d PH0 %d: pm mismatches %d, bits mismatches %d\n", cnt, s0, 2 * v, bp, bb); }
  }
  return 0;
}
