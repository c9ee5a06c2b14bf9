// microbenchmark 2: issue cost of the candidate ACS/decision opcodes on gfx950 (4 waves/SIMD), and a
// functional check of scalar stores (s_store_dwordx2 + s_dcache_wb) as a decision sink.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define CLOB "v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27","s4","s5","s6","s7","s8","s9","vcc","scc"
#define DPPQ " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
template <int V> __global__ __launch_bounds__(256) void k(int* out, int n) {
  for (int i = 0; i < n; ++i) {
    if (V == 0) asm volatile(REP8("v_add_u32 v10, v20, v25\nv_add_u32 v11, v21, v26\nv_add_u32 v12, v22, v27\nv_add_u32 v13, v23, v24\n") ::: CLOB);
    if (V == 1) asm volatile(REP8("v_cmp_lt_i32 vcc, v20, v25\nv_cmp_lt_i32 vcc, v21, v26\nv_cmp_lt_i32 vcc, v22, v27\nv_cmp_lt_i32 vcc, v23, v24\n") ::: CLOB);
    if (V == 2) asm volatile(REP8("v_cmp_lt_i16 vcc, v20, v25\nv_cmp_lt_i16 vcc, v21, v26\nv_cmp_lt_i16 vcc, v22, v27\nv_cmp_lt_i16 vcc, v23, v24\n") ::: CLOB);
    if (V == 3) asm volatile(REP8("v_cmp_lt_i32 s[4:5], v20, v25\nv_cmp_lt_i32 s[6:7], v21, v26\nv_cmp_lt_i32 s[4:5], v22, v27\nv_cmp_lt_i32 s[6:7], v23, v24\n") ::: CLOB);
    if (V == 4) asm volatile(REP8("v_addc_co_u32 v10, vcc, v20, v25, vcc\nv_addc_co_u32 v11, vcc, v21, v26, vcc\nv_addc_co_u32 v12, vcc, v22, v27, vcc\nv_addc_co_u32 v13, vcc, v23, v24, vcc\n") ::: CLOB);
    if (V == 5) asm volatile(REP8("v_max_i32_dpp v10, v20, v25" DPPQ "v_max_i32_dpp v11, v21, v26" DPPQ "v_max_i32_dpp v12, v22, v27" DPPQ "v_max_i32_dpp v13, v23, v24" DPPQ) ::: CLOB);
    if (V == 6) asm volatile(REP8("v_max_i16_dpp v10, v20, v25" DPPQ "v_max_i16_dpp v11, v21, v26" DPPQ "v_max_i16_dpp v12, v22, v27" DPPQ "v_max_i16_dpp v13, v23, v24" DPPQ) ::: CLOB);
    if (V == 7) asm volatile(REP8("v_readlane_b32 s4, v20, 3\nv_readlane_b32 s5, v21, 5\nv_readlane_b32 s6, v22, 7\nv_readlane_b32 s7, v23, 9\n") ::: CLOB);
    if (V == 8) asm volatile(REP8("v_cndmask_b32 v10, v20, v25, vcc\nv_cndmask_b32 v11, v21, v26, vcc\nv_cndmask_b32 v12, v22, v27, vcc\nv_cndmask_b32 v13, v23, v24, vcc\n") ::: CLOB);
    if (V == 9) asm volatile(REP8("v_sub_u16 v10, v20, v25\nv_sub_u16 v11, v21, v26\nv_sub_u16 v12, v22, v27\nv_sub_u16 v13, v23, v24\n") ::: CLOB);
    if (V == 10) asm volatile(REP8("v_add_u32_dpp v10, v20, v25" DPPQ "v_add_u32_dpp v11, v21, v26" DPPQ "v_add_u32_dpp v12, v22, v27" DPPQ "v_add_u32_dpp v13, v23, v24" DPPQ) ::: CLOB);
    if (V == 11) asm volatile(REP8("v_dot4_i32_i8 v10, v20, s4, v25\nv_dot4_i32_i8 v11, v21, s4, v26\nv_dot4_i32_i8 v12, v22, s4, v27\nv_dot4_i32_i8 v13, v23, s4, v24\n") ::: CLOB);
    if (V == 12) asm volatile(REP8("v_dot2_i32_i16 v10, v20, v21, v25\nv_dot2_i32_i16 v11, v21, v22, v26\nv_dot2_i32_i16 v12, v22, v23, v27\nv_dot2_i32_i16 v13, v23, v20, v24\n") ::: CLOB);
    if (V == 13) asm volatile(REP8("v_mad_i32_i24 v10, v20, v21, v25\nv_mad_i32_i24 v11, v21, v22, v26\nv_mad_i32_i24 v12, v22, v23, v27\nv_mad_i32_i24 v13, v23, v20, v24\n") ::: CLOB);
    if (V == 14) asm volatile(REP8("v_add_u16 v10, v20, v25\nv_add_u16 v11, v21, v26\nv_add_u16 v12, v22, v27\nv_add_u16 v13, v23, v24\n") ::: CLOB);
    if (V == 15) asm volatile(REP8("v_mov_b32_dpp v10, v20" DPPQ "v_mov_b32_dpp v11, v21" DPPQ "v_mov_b32_dpp v12, v22" DPPQ "v_mov_b32_dpp v13, v23" DPPQ) ::: CLOB);
    if (V == 16) asm volatile(REP8("v_cmp_lt_i32 vcc, v20, v25\ns_mov_b64 s[4:5], vcc\nv_cmp_lt_i32 vcc, v21, v26\ns_mov_b64 s[6:7], vcc\nv_cmp_lt_i32 vcc, v22, v27\ns_mov_b64 s[8:9], vcc\nv_cmp_lt_i32 vcc, v23, v24\ns_mov_b64 s[4:5], vcc\n") ::: CLOB);
    if (V == 17) asm volatile(REP8("v_sub_co_u32 v10, vcc, v20, v25\nv_sub_co_u32 v11, vcc, v21, v26\nv_sub_co_u32 v12, vcc, v22, v27\nv_sub_co_u32 v13, vcc, v23, v24\n") ::: CLOB);
    if (V == 18) asm volatile(REP8("v_sub_u32_sdwa v10, v20, v25 dst_sel:DWORD src0_sel:WORD_0 src1_sel:WORD_0\nv_sub_u32_sdwa v11, v21, v26 dst_sel:DWORD src0_sel:WORD_0 src1_sel:WORD_0\nv_sub_u32_sdwa v12, v22, v27 dst_sel:DWORD src0_sel:WORD_0 src1_sel:WORD_0\nv_sub_u32_sdwa v13, v23, v24 dst_sel:DWORD src0_sel:WORD_0 src1_sel:WORD_0\n") ::: CLOB);
  }
  out[blockIdx.x * 256 + threadIdx.x] = n;
}
template <int V> void run(const char* name, int* d) {
  int n = 4096, grid = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  int per = (V == 16) ? 32 : 32;
  printf("%-28s %.2f cycles@2.4GHz per instr per SIMD\n", name, ms * 1e-3 * 2.4e9 / ((double)grid * 4 / 1024 * n * per));
}

// scalar store as decision sink: each wave writes vcc of (lane < i) for i = 0..63 to its slot, flushes,
// reads it back through the scalar cache, and the host checks the memory afterwards
__global__ __launch_bounds__(256) void k_sstore(unsigned long long* buf, unsigned long long* readback) {
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  unsigned long long* slot = buf + (size_t)wave * 64;
  for (int i = 0; i < 64; ++i) {
    unsigned long long* p = slot + i;
    asm volatile("v_cmp_lt_i32 vcc, %[lane], %[i]\n\t"
                 "s_nop 1\n\t"
                 "s_store_dwordx2 vcc, %[p], 0x0\n\t" :: [lane] "v"(lane), [i] "v"(i), [p] "s"(p) : "vcc", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long acc = 0;
  for (int i = 0; i < 64; ++i) {
    unsigned long long v;
    const unsigned long long* p = slot + i;
    asm volatile("s_load_dwordx2 %[v], %[p], 0x0\n\ts_waitcnt lgkmcnt(0)" : [v] "=s"(v) : [p] "s"(p) : "memory");
    acc += v;
  }
  if (lane == 0) readback[wave] = acc;
}

// throughput: VALU compare + scalar store per "step", 32 steps per base-pointer bump, 8 waves/SIMD
template <int MODE> __global__ __launch_bounds__(256) void k_sstore_rate(unsigned long long* buf, int n) {
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  unsigned long long* p = buf + (size_t)wave * 2048;          // 16 KB slot per wave, reused
  int a = lane, b = 17;
  for (int i = 0; i < n; ++i) {
    unsigned long long* q = p + (i & 63) * 32;
#define ST(off) "v_cmp_lt_i32 s[4:5], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t" \
                "v_cmp_lt_i32 s[6:7], %[b], %[a]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t" \
                "s_store_dwordx2 s[4:5], %[q], " #off "\n\ts_store_dwordx2 s[6:7], %[q], " #off "+8\n\t"
#define NST(off) "v_cmp_lt_i32 s[4:5], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t" \
                "v_cmp_lt_i32 s[6:7], %[b], %[a]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\tv_add_u32 %[a], %[a], %[b]\n\t"
    if (MODE == 1) asm volatile(ST(0) ST(16) ST(32) ST(48) ST(64) ST(80) ST(96) ST(112) ST(128) ST(144) ST(160) ST(176) ST(192) ST(208) ST(224) ST(240)
                 : [a] "+v"(a) : [b] "v"(b), [q] "s"(q) : "s4", "s5", "s6", "s7", "memory");
    else asm volatile(NST(0) NST(16) NST(32) NST(48) NST(64) NST(80) NST(96) NST(112) NST(128) NST(144) NST(160) NST(176) NST(192) NST(208) NST(224) NST(240)
                 : [a] "+v"(a) : [b] "v"(b), [q] "s"(q) : "s4", "s5", "s6", "s7", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  if (a == 12345) buf[0] = a;
}
template <int MODE> void run_rate(const char* name, unsigned long long* buf) {
  int n = 2048, grid = 256 * 8;                                  // 8 waves per SIMD
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_sstore_rate<MODE>, dim3(grid), dim3(256), 0, 0, buf, 16); hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k_sstore_rate<MODE>, dim3(grid), dim3(256), 0, 0, buf, n); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.2f cycles@2.4GHz per step (4 VALU + 1 store) per SIMD, %.1f GB/s of scalar stores\n", name,
         ms * 1e-3 * 2.4e9 / ((double)grid * 4 / 1024 * n * 32), MODE ? (double)grid * 4 * n * 32 * 8 / (ms * 1e-3) / 1e9 : 0.0);
}

int main() { int* d; hipMalloc(&d, 1024 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_cmp_lt_i32 vcc", d); run<2>("v_cmp_lt_i16 vcc", d); run<3>("v_cmp_lt_i32 sgpr(e64)", d);
  run<4>("v_addc_co_u32", d); run<5>("v_max_i32_dpp", d); run<6>("v_max_i16_dpp", d); run<7>("v_readlane_b32", d); run<8>("v_cndmask_b32", d);
  run<9>("v_sub_u16", d); run<10>("v_add_u32_dpp", d); run<11>("v_dot4 (sgpr src)", d); run<12>("v_dot2_i32_i16", d); run<13>("v_mad_i32_i24", d);
  run<14>("v_add_u16", d); run<15>("v_mov_b32_dpp", d); run<16>("cmp+s_mov_b64 (per pair)", d); run<17>("v_sub_co_u32", d); run<18>("v_sub_u32_sdwa", d);
  { unsigned long long* big; hipMalloc(&big, (size_t)8192 * 2048 * 8);
    run_rate<0>("4 VALU, no store", big); run_rate<1>("4 VALU + s_store_dwordx2", big); hipFree(big); }
  const int waves = 4096;
  unsigned long long *buf, *rb; hipMalloc(&buf, (size_t)waves * 64 * 8); hipMalloc(&rb, waves * 8);
  hipMemset(buf, 0xff, (size_t)waves * 64 * 8);
  hipLaunchKernelGGL(k_sstore, dim3(waves / 4), dim3(256), 0, 0, buf, rb);
  hipError_t e = hipDeviceSynchronize();
  printf("k_sstore: %s\n", hipGetErrorString(e));
  std::vector<unsigned long long> h((size_t)waves * 64), r(waves);
  hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), rb, waves * 8, hipMemcpyDeviceToHost);
  long bad = 0, badr = 0; unsigned long long expect_sum = 0;
  for (int i = 0; i < 64; ++i) expect_sum += (i == 0) ? 0ull : (i == 64 ? ~0ull : ((1ull << i) - 1));
  for (int w = 0; w < waves; ++w) { for (int i = 0; i < 64; ++i) if (h[(size_t)w * 64 + i] != ((1ull << i) - 1)) ++bad; if (r[w] != expect_sum) ++badr; }
  printf("scalar store: %ld bad words of %zu in memory, %ld bad scalar read-backs of %d\n", bad, h.size(), badr, waves);
  return 0; }
