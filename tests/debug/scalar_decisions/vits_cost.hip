// microbenchmark: the generated forward (32-step) and traceback (96-step) texts of k_viterbi_s in isolation
// and mixed across waves, 8 waves/SIMD.  cycles@2.4GHz per trellis step per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../abracadabra_amd/csrc/dabx_vits.inc"
#include "vits_debug.inc"
#define CLOB_FWD "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","scc","memory"
#define CLOB_TB CLOB_FWD,"s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99"
#define SIGS [s0] "v"(sig[0]), [s1] "v"(sig[1]), [s2] "v"(sig[2]), [s3] "v"(sig[3]), [s4] "v"(sig[4]), [s5] "v"(sig[5]), \
             [n0] "v"(nsig[0]), [n1] "v"(nsig[1]), [n2] "v"(nsig[2]), [n3] "v"(nsig[3]), [n4] "v"(nsig[4]), [n5] "v"(nsig[5])
__device__ __forceinline__ void fwd96(int& pm, const int* sig, const int* nsig, int xv, int ad, uint64_t* dp) {
  int S, Ka, Kb, D, xa, xb;
#define OPS : [pm] "+v"(pm), [S] "=&v"(S), [Ka] "=&v"(Ka), [Kb] "=&v"(Kb), [D] "=&v"(D), [xa] "=&s"(xa), [xb] "=&s"(xb) : [xv] "v"(xv), [ad] "v"(ad), [dpf] "s"(dp), SIGS : CLOB_FWD
  asm volatile(DABX_FWD32_TEXT_0 OPS); dp += 32;
  asm volatile(DABX_FWD32_TEXT_2 OPS); dp += 32;
  asm volatile(DABX_FWD32_TEXT_4 OPS);
}
// mode 0: all waves forward; 1: all traceback; 2: waves alternate (even slot forward, odd traceback); 3: each wave does fwd then tb
__global__ __launch_bounds__(256) void k(uint64_t* scratch, int nblk96, int mode, int* sink) {
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int odd = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1;          // HW wave slot parity: mixes roles inside each SIMD
  uint64_t* slot = scratch + (size_t)wave * 1536;
  int sig[6], nsig[6];
  for (int i = 0; i < 6; ++i) { sig[i] = 0x01ff01ff ^ (lane * 0x01010101 * (i + 1) & 0xfefefefe & 0); nsig[i] = 0xff01ff01; }
  int pm = lane, xv = lane * 0x01020304, ad = (lane ^ 32) << 2;
  uint32_t L = 0, acc = 0;
  const bool do_f = mode == 0 || mode == 3 || mode == 4 || ((mode == 2 || mode == 5 || mode == 6) && !odd), do_t = mode == 1 || mode == 3 || (mode == 2 && odd);
  if ((mode == 5 || mode == 6) && odd) {          // pure SALU spin (5: dependent chain, 6: independent ops), no memory
    uint32_t a = wave, b = 1;
    for (int i = 0; i < nblk96 * 4 * 96 / 8; ++i) {
      if (mode == 5) asm volatile("s_bitcmp1_b32 %0, 3\ns_cselect_b32 %1, 7, 0\ns_xor_b32 %0, %0, %1\ns_bitcmp1_b32 %0, 5\ns_cselect_b32 %1, 8, 0\ns_xor_b32 %0, %0, %1\n"
                   "s_bitcmp1_b32 %0, 3\ns_cselect_b32 %1, 7, 0\ns_xor_b32 %0, %0, %1\ns_bitcmp1_b32 %0, 5\ns_cselect_b32 %1, 8, 0\ns_xor_b32 %0, %0, %1\n"
                   "s_bitcmp1_b32 %0, 3\ns_cselect_b32 %1, 7, 0\ns_xor_b32 %0, %0, %1\ns_bitcmp1_b32 %0, 5\ns_cselect_b32 %1, 8, 0\ns_xor_b32 %0, %0, %1\n"
                   "s_bitcmp1_b32 %0, 3\ns_cselect_b32 %1, 7, 0\ns_xor_b32 %0, %0, %1\ns_bitcmp1_b32 %0, 5\ns_cselect_b32 %1, 8, 0\ns_xor_b32 %0, %0, %1\n" : "+s"(a), "+s"(b) :: "scc");
      else asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n" ::: "memory");
    }
    if (a == 0x12345) sink[1] = a;
  }
  for (int rep = 0; rep < 4; ++rep) {
    if (do_f) for (int b = 0; b < nblk96; ++b) fwd96(pm, sig, nsig, xv, ad, slot + 96 * b);
    if (do_t) for (int b = nblk96 - 1; b >= 0; --b) {
      uint32_t o0, o1, o2;
      asm volatile(DABX_TB96_TEXT : [L] "+s"(L), [o0] "=&s"(o0), [o1] "=&s"(o1), [o2] "=&s"(o2) : [dpt] "s"(slot + 96 * b) : CLOB_TB);
      acc ^= o0 ^ o1 ^ o2;
    }
  }
#define WOVEN(TEXT) for (int rep = 0; rep < 4; ++rep) for (int b = 0; b < nblk96; ++b) { \
    int S, Ka, Kb, D; uint32_t o0, o1, o2; \
    asm volatile(TEXT : [pm] "+v"(pm), [S] "=&v"(S), [Ka] "=&v"(Ka), [Kb] "=&v"(Kb), [D] "=&v"(D), [L] "+s"(L), [o0] "=&s"(o0), [o1] "=&s"(o1), [o2] "=&s"(o2) \
                 : [xv0] "v"(xv), [xv1] "v"(xv), [xv2] "v"(xv), [ad] "v"(ad), [dpf] "s"(slot + 96 * b), [dpt] "s"(slot + 1536 - 96 - 96 * b), SIGS : CLOB_TB); \
    acc ^= o0 ^ o1 ^ o2; }
  if (mode == 8) WOVEN(DABX_FWDTB96_NOLOAD_TEXT)
  if (mode == 9) WOVEN(DABX_FWDTB96_NOSTORE_TEXT)
  if (mode == 10) WOVEN(DABX_FWDTB96_NOMEM_TEXT)
  if (mode == 11) WOVEN(DABX_FWD96_NOSTORE_TEXT)
  if (mode == 12) for (int rep = 0; rep < 4; ++rep) for (int b = nblk96 - 1; b >= 0; --b) {
      uint32_t o0, o1, o2;
      asm volatile(DABX_TB96_NOLOAD_TEXT : [L] "+s"(L), [o0] "=&s"(o0), [o1] "=&s"(o1), [o2] "=&s"(o2) : [dpt] "s"(slot + 96 * b) : CLOB_TB);
      acc ^= o0 ^ o1 ^ o2; }
  if (mode == 7) for (int rep = 0; rep < 4; ++rep) for (int b = 0; b < nblk96; ++b) {
    int S, Ka, Kb, D; uint32_t o0, o1, o2;
    asm volatile(DABX_FWDTB96_TEXT : [pm] "+v"(pm), [S] "=&v"(S), [Ka] "=&v"(Ka), [Kb] "=&v"(Kb), [D] "=&v"(D), [L] "+s"(L), [o0] "=&s"(o0), [o1] "=&s"(o1), [o2] "=&s"(o2)
                 : [xv0] "v"(xv), [xv1] "v"(xv), [xv2] "v"(xv), [ad] "v"(ad), [dpf] "s"(slot + 96 * b), [dpt] "s"(slot + 1536 - 96 - 96 * b), SIGS : CLOB_TB);
    acc ^= o0 ^ o1 ^ o2;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
  if (pm == 0x7fffffff || acc == 0x12345) sink[0] = pm + L;
  if (lane == 0) atomicAdd(&sink[2 + odd], 1);
}
int main() {
  const int wps = 8, nblk = 16;
  uint64_t* scr; int* sink; hipMalloc(&scr, (size_t)256 * wps * 4 * 1536 * 8); hipMalloc(&sink, 16); hipMemset(sink, 0, 16);
  hipMemset(scr, 0x5a, (size_t)256 * wps * 4 * 1536 * 8);
  const char* names[13] = {"forward only", "traceback only", "half the blocks fwd, half tb", "every wave fwd then tb", "forward only, 4 waves/SIMD", "half fwd, half SALU chain (no mem)", "half fwd, half s_nop", "woven fwd+tb in every wave", "woven, tb without loads", "woven, fwd without stores", "woven, no loads no stores", "fwd 96 without stores", "tb without loads"};
  for (int mode = 0; mode < 13; ++mode) {
    const int grid = mode == 4 ? 256 * 4 : 256 * wps;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, scr, nblk, mode, sink); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, scr, nblk, mode, sink); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double steps_per_simd = (double)wps * 4 * nblk * 96 * ((mode == 2 || (mode >= 4 && mode <= 6)) ? 0.5 : 1.0);   // codeword-steps (fwd+tb of a step count once in mode 3)
    printf("%-32s %.3f ms  %.1f cycles@2.4GHz per step per SIMD\n", names[mode], ms, ms * 1e-3 * 2.4e9 / steps_per_simd);
  }
  for (int w : {1, 2, 4, 8}) for (int mode : {1, 12}) {
    const int grid = 256 * w;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, scr, nblk, mode, sink); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, scr, nblk, mode, sink); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-20s %d waves/SIMD: %.3f ms = %.1f cycles per step per WAVE\n", mode == 1 ? "traceback" : "traceback, no loads", w, ms, ms * 1e-3 * 2.4e9 / (4.0 * nblk * 96));
  }
  int h[4]; hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost); printf("even-slot waves %d, odd-slot waves %d (all launches)\n", h[2], h[3]);
  return 0; }
