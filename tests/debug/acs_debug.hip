#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(const int* in, int* out) {
  int lane = threadIdx.x;
  int pm = in[lane], sig = in[64+lane], nsig = in[128+lane], xv = in[192+lane];
  int xs = __builtin_amdgcn_readlane(xv, 5);
  // reference
  int keep = __builtin_amdgcn_sdot4(sig, xs, pm, false);
  int send = __builtin_amdgcn_sdot4(nsig, xs, pm, false);
  int recv = __builtin_amdgcn_mov_dpp(send, 0xB1, 0xf, 0xf, false);
  int refD = keep - recv, refP = max(keep, recv);
  int recv2 = __builtin_amdgcn_mov_dpp(send, 0x141, 0xf, 0xf, false);
  int recv3 = __builtin_amdgcn_mov_dpp(send, 0x128, 0xf, 0xf, false);
  int recv4 = __builtin_amdgcn_ds_swizzle(send, 0x401F);
  int recv5 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, send);
  // asm
  int S, K, D, xn, P = pm; unsigned bits = 0; int idx = 7;
  asm volatile("v_dot4_i32_i8 %[S], %[nsig], %[xs], %[pm]\n\t"
               "v_dot4_i32_i8 %[K], %[sig], %[xs], %[pm]\n\t"
               "v_readlane_b32 %[xn], %[xv], %[idx]\n\t"
               "v_subrev_u32_dpp %[D], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "v_max_i32_dpp %[pm], %[S], %[K] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "v_alignbit_b32 %[bits], %[bits], %[D], 31"
               : [pm] "+v"(P), [bits] "+v"(bits), [S] "=&v"(S), [K] "=&v"(K), [D] "=&v"(D), [xn] "=&s"(xn)
               : [sig] "v"(sig), [nsig] "v"(nsig), [xs] "s"(xs), [xv] "v"(xv), [idx] "s"(idx));
  int D2, P2, D3, P3;
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %2, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "v_max_i32_dpp %1, %2, %3 row_half_mirror row_mask:0xf bank_mask:0xf" : "=&v"(D2), "=&v"(P2) : "v"(send), "v"(keep));
  asm volatile("s_nop 1\n\tv_subrev_u32_dpp %0, %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_max_i32_dpp %1, %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(D3), "=&v"(P3) : "v"(send), "v"(keep));
  int R4, R5; int ad = (lane ^ 32) << 2;
  asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,16)\n\ts_waitcnt lgkmcnt(0)" : "=&v"(R4) : "v"(send));
  asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(R5) : "v"(ad), "v"(send));
  int* o = out + lane * 24;
  o[0]=keep; o[1]=K; o[2]=send; o[3]=S; o[4]=refD; o[5]=D; o[6]=refP; o[7]=P; o[8]=xn; o[9]=__builtin_amdgcn_readlane(xv,7);
  o[10]=keep-recv2; o[11]=D2; o[12]=max(keep,recv2); o[13]=P2; o[14]=keep-recv3; o[15]=D3; o[16]=max(keep,recv3); o[17]=P3;
  o[18]=recv4; o[19]=R4; o[20]=recv5; o[21]=R5; o[22]=(int)bits; o[23]=(int)((unsigned)refD>>31);
}
int main() {
  int h[256], *d, *o; int ho[64*24];
  srand(1);
  for (int i=0;i<64;i++){ h[i]=rand()%100000-50000; int sg=0,ng=0; for(int j=0;j<4;j++){int n=rand()&1; sg|=(n?0xff:0x01)<<(8*j); ng|=(n?0x01:0xff)<<(8*j);} h[64+i]=sg; h[128+i]=ng;
    int x=0; for(int j=0;j<4;j++) x|=((rand()%255-127)&0xff)<<(8*j); h[192+i]=x; }
  hipMalloc(&d,sizeof h); hipMalloc(&o,sizeof ho); hipMemcpy(d,h,sizeof h,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,d,o); hipMemcpy(ho,o,sizeof ho,hipMemcpyDeviceToHost);
  const char* names[12]={"keep","send","D0","P0","xn","D2","P2","D3","P3","swz","bperm","bit"};
  for (int p=0;p<12;p++){ int bad=0; for(int l=0;l<64;l++) if(ho[l*24+2*p]!=ho[l*24+2*p+1]) bad++; printf("%s mismatches %d (lane0: %d vs %d)\n",names[p],bad,ho[2*p],ho[2*p+1]); }
  return 0;
}
