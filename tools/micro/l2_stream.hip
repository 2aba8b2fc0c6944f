// How fast can ONE wave per SIMD stream a 700 KB, L2-resident weight stream (1 KB per element) while it issues 4 dependent-free MFMAs per
// element? (a) buffer_load_dwordx4 into a register ring of depth D, (b) LDS-DMA (buffer_load_dwordx4 ... lds) into a per-wave LDS FIFO of
// F slots, read back with ds_read_b128 through a 3-element register stage. Prints cycles per element (128 = the MFMA bound).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
#define NEL 700

static __device__ __forceinline__ void mfma4(const f32x4 w, f32x4& a0, f32x4& a1) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %2, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %3, %1\n\t"
               "v_mfma_f32_16x16x4_f32 %0, %4, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %5, %5, %1"
               : "+v"(a0), "+v"(a1) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
}

template <int D>
__global__ void __launch_bounds__(256) k_reg(const float* __restrict__ stream, float* __restrict__ out, long long* __restrict__ cyc, int reps) {
  const int lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stream), 0, (NEL + 64) * 1024, 0x00020000);
  f32x4 r[D], a0 = {0, 0, 0, 0}, a1 = a0;
  const long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    int so = 0;
#pragma unroll
    for (int s = 0; s < D; ++s) { r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so, 0)); so += 1024; }
#pragma unroll 1
    for (int e = 0; e < NEL; e += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const f32x4 w = r[s];
        r[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so, 0)); so += 1024;
        __builtin_amdgcn_sched_barrier(0);
        mfma4(w, a0, a1);
      }
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int F>
__global__ void __launch_bounds__(256) k_lds(const float* __restrict__ stream, float* __restrict__ out, long long* __restrict__ cyc, int reps) {
  __shared__ __attribute__((aligned(16))) float fifo[4][F * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long a = (unsigned long long)stream;
  u32x4 rs = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, (unsigned)((NEL + 64) * 1024), 0x00020000u};
#pragma unroll
  for (int j = 0; j < 4; ++j) rs[j] = __builtin_amdgcn_readfirstlane(rs[j]);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(&fifo[wave][0]));
  const int loff = lane * 16;
  f32x4 r[3], a0 = {0, 0, 0, 0}, a1 = a0;
  auto dma = [&](unsigned slot_off, int so) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(loff), "s"(rs), "s"(base + slot_off), "s"(so) : "memory");
  };
  const long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    int so = 0;
    unsigned wr = 0, rd = 0;                                   // byte offsets of the next slot to fill / to read
    for (int s = 0; s < F - 2; ++s) { dma(wr, so); so += 1024; wr = wr + 1024 == F * 1024 ? 0 : wr + 1024; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      r[s] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(&fifo[wave][0]) + rd + loff); rd = rd + 1024 == F * 1024 ? 0 : rd + 1024;
      dma(wr, so); so += 1024; wr = wr + 1024 == F * 1024 ? 0 : wr + 1024;
    }
#pragma unroll 1
    for (int e = 0; e < NEL; e += 3) {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const f32x4 w = r[s];
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(F - 3) : "memory");
        r[s] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(&fifo[wave][0]) + rd + loff); rd = rd + 1024 == F * 1024 ? 0 : rd + 1024;
        dma(wr, so); so += 1024; wr = wr + 1024 == F * 1024 ? 0 : wr + 1024;
        __builtin_amdgcn_sched_barrier(0);
        mfma4(w, a0, a1);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float *stream, *out; long long* cyc;
  hipMalloc(&stream, (NEL + 64) * 1024); hipMalloc(&out, 4 * 256 * 2048); hipMalloc(&cyc, 8);
  hipMemset(stream, 0, (NEL + 64) * 1024);
  const int reps = 20;
  auto report = [&](const char* name, int blocks) {
    long long c; hipDeviceSynchronize(); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s blocks %4d: %.1f cycles per 1-KB element (4 MFMAs = 128)\n", name, blocks, (double)c / (reps * NEL));
  };
  for (int blocks : {256, 512}) {        // one / two 4-wave workgroups per CU = 1 / 2 waves per SIMD
    hipLaunchKernelGGL(k_reg<9>, dim3(blocks), dim3(256), 0, 0, stream, out, cyc, reps); report("register ring 9", blocks);
    hipLaunchKernelGGL(k_reg<17>, dim3(blocks), dim3(256), 0, 0, stream, out, cyc, reps); report("register ring 17", blocks);
    hipLaunchKernelGGL(k_reg<33>, dim3(blocks), dim3(256), 0, 0, stream, out, cyc, reps); report("register ring 33", blocks);
    hipLaunchKernelGGL(k_lds<8>, dim3(blocks), dim3(256), 0, 0, stream, out, cyc, reps); report("LDS-DMA FIFO 8 + 3 regs", blocks);
    hipLaunchKernelGGL(k_lds<12>, dim3(blocks), dim3(256), 0, 0, stream, out, cyc, reps); report("LDS-DMA FIFO 12 + 3 regs", blocks);
    hipLaunchKernelGGL(k_lds<16>, dim3(blocks), dim3(256), 0, 0, stream, out, cyc, reps); report("LDS-DMA FIFO 16 + 3 regs", blocks);
  }
  return 0;
}
