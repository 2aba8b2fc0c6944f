// bf16x6_gemm.hip -- prototype behind DESIGN.md section 7 "lever 2": ONE 128 x 128 layer of the PPO update over 40 960 minibatch rows
// (forward Y = X W^T, input gradient dX = dY W, weight gradient dW = dY^T X) computed two ways on gfx950:
//   (a) fp32-input MFMA (v_mfma_f32_16x16x4_f32: the vector rate, what the product's update kernels use);
//   (b) every fp32 operand split into three bf16 pieces x = x1 + x2 + x3 (8 mantissa bits each) and the product formed from the six
//       piece products with i + j <= 4 on the bf16 pipe (v_mfma_f32_16x16x32_bf16, fp32 accumulation): 6 / 16 of the matrix cycles,
//       ~2^-24 relative accuracy per product; (b3) the three products with i + j <= 3 for comparison (~2^-16).
// Both variants share one tiling (a workgroup of 8 waves takes 128 rows x 128 columns x a 128-deep K chunk, the shared operand staged
// through LDS, the per-wave operand straight from memory), so the difference is the matrix pipe and the splitting work, not the data
// path. Reports time per GEMM and max / rms error against an fp64 reference. One such GEMM over 40 960 x 128 activations is bound by
// reading and writing them (42 MB), which the product's update kernels do not do per layer (a unit's activations stay on chip across
// its layers): the matrix-side cost per layer is therefore measured as well, as the SLOPE of the time over `reps` repetitions of the
// K loop on resident operands (for bf16 x 6 each repetition splits its activations again, as every layer would). Not product code.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x6_gemm bf16x6_gemm.hip && ./bf16x6_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 128;            // columns of C = rows of Bt
constexpr int KC = 128;           // K chunk a workgroup handles
constexpr int MB = 128;           // rows of A per workgroup (8 waves x 16)
constexpr int LDB32 = KC + 4;     // padded LDS row pitch (floats) of the fp32 B chunk
constexpr int LDB16 = KC + 8;     // padded LDS row pitch (bf16) of a bf16 B piece

__device__ __forceinline__ unsigned short bf16_rn(float x) {          // round to nearest even
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }
// x = p1 + p2 + p3 + O(2^-24 x)
__device__ __forceinline__ void split3(float x, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
  p1 = bf16_rn(x);
  const float r1 = x - bf16_f(p1);
  p2 = bf16_rn(r1);
  const float r2 = r1 - bf16_f(p2);
  p3 = bf16_rn(r2);
}

// C[M x N] (+)= A[M x K] Bt[N x K]^T over the K chunk of this workgroup. grid = (M / MB, K / KC); partial results go to
// Cpart[kchunk][M][N] (the caller reduces them when there is more than one chunk). A, Bt row-major fp32.
__global__ void __launch_bounds__(512) gemm_f32(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ Cpart, int M, int K, int reps) {
  __shared__ float sB[N * LDB32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * MB + wave * 16, k0 = blockIdx.y * KC;
  for (int e = tid; e < N * KC / 4; e += 512) {                    // stage the B chunk (coalesced float4 reads)
    const int n = e / (KC / 4), kq = e % (KC / 4);
    const float4 v = *(const float4*)&Bt[(size_t)n * K + k0 + 4 * kq];
    *(float4*)&sB[n * LDB32 + 4 * kq] = v;
  }
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lk = lane >> 4;
  // (the sum over k does not care about its order: step s gives lane group lk the index k = 32 lk + s, on both operands, so that a
  // lane's 32 values of A are contiguous)
  const float* Arow = A + (size_t)(m0 + li) * K + k0 + 32 * lk;
  float a[32];
#pragma unroll
  for (int q = 0; q < 8; ++q) { const float4 v = *(const float4*)&Arow[4 * q]; a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], sB[(16 * c + li) * LDB32 + 32 * lk + s], acc[c], 0, 0, 0);
    }
  }
  float* C = Cpart + (size_t)blockIdx.y * M * N;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(size_t)(m0 + 4 * lk + r) * N + 16 * c + li] = acc[c][r];
}

// The same GEMM on the bf16 pipe. NPROD = 6: pieces (1,1) (1,2) (2,1) (1,3) (3,1) (2,2); NPROD = 3: the first three.
// PRESPLIT: Bt arrives as three bf16 matrices Bp[3][N][K] (weights: split once per minibatch); otherwise it is split while it is staged.
template <int NPROD, bool PRESPLIT>
__global__ void __launch_bounds__(512) gemm_bf16x(const float* __restrict__ A, const float* __restrict__ Bt, const unsigned short* __restrict__ Bp,
                                                  float* __restrict__ Cpart, int M, int K, int reps) {
  __shared__ unsigned short sB[3][N * LDB16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * MB + wave * 16, k0 = blockIdx.y * KC;
  if (PRESPLIT) {
    for (int e = tid; e < 3 * N * KC / 8; e += 512) {
      const int p = e / (N * KC / 8), r = e % (N * KC / 8), n = r / (KC / 8), kq = r % (KC / 8);
      *(uint4*)&sB[p][n * LDB16 + 8 * kq] = *(const uint4*)&Bp[((size_t)p * N + n) * K + k0 + 8 * kq];
    }
  } else {
    for (int e = tid; e < N * KC / 4; e += 512) {
      const int n = e / (KC / 4), kq = e % (KC / 4);
      const float4 v = *(const float4*)&Bt[(size_t)n * K + k0 + 4 * kq];
      const float x[4] = {v.x, v.y, v.z, v.w};
      unsigned short p1[4], p2[4], p3[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(x[j], p1[j], p2[j], p3[j]);
      *(uint2*)&sB[0][n * LDB16 + 4 * kq] = make_uint2(p1[0] | (uint32_t)p1[1] << 16, p1[2] | (uint32_t)p1[3] << 16);
      *(uint2*)&sB[1][n * LDB16 + 4 * kq] = make_uint2(p2[0] | (uint32_t)p2[1] << 16, p2[2] | (uint32_t)p2[3] << 16);
      *(uint2*)&sB[2][n * LDB16 + 4 * kq] = make_uint2(p3[0] | (uint32_t)p3[1] << 16, p3[2] | (uint32_t)p3[3] << 16);
    }
  }
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, lk = lane >> 4;
  const float* Arow = A + (size_t)(m0 + li) * K + k0;
  float x[KC / 32][8];                           // this lane's 8 consecutive k of its row in each of the four 32-deep steps
#pragma unroll
  for (int s = 0; s < KC / 32; ++s) {
    const float4 v0 = *(const float4*)&Arow[32 * s + 8 * lk], v1 = *(const float4*)&Arow[32 * s + 8 * lk + 4];
    x[s][0] = v0.x; x[s][1] = v0.y; x[s][2] = v0.z; x[s][3] = v0.w; x[s][4] = v1.x; x[s][5] = v1.y; x[s][6] = v1.z; x[s][7] = v1.w;
  }
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int s = 0; s < KC / 32; ++s) {
      bf16x8 a1, a2, a3;                          // split again in every repetition: a layer's input is new every time
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned short p1, p2, p3;
        split3(x[s][j], p1, p2, p3);
        a1[j] = (short)p1; a2[j] = (short)p2; a3[j] = (short)p3;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int off = (16 * c + li) * LDB16 + 32 * s + 8 * lk;
        const bf16x8 b1 = *(const bf16x8*)&sB[0][off], b2 = *(const bf16x8*)&sB[1][off];
        // smallest products first
        if (NPROD == 6) {
          const bf16x8 b3 = *(const bf16x8*)&sB[2][off];
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc[c], 0, 0, 0);
        }
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < KC / 32; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[s][j]));      // (opaque: the split stays inside the repetition loop)
  }
  float* C = Cpart + (size_t)blockIdx.y * M * N;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(size_t)(m0 + 4 * lk + r) * N + 16 * c + li] = acc[c][r];
}

__global__ void reduce_parts(const float* __restrict__ part, float* __restrict__ C, int elems, int nparts) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float acc = 0.f;
  for (int p = 0; p < nparts; ++p) acc += part[(size_t)p * elems + e];
  C[e] = acc;
}

__global__ void presplit(const float* __restrict__ B, unsigned short* __restrict__ Bp, int elems) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  unsigned short p1, p2, p3;
  split3(B[e], p1, p2, p3);
  Bp[e] = p1; Bp[elems + e] = p2; Bp[2 * (size_t)elems + e] = p3;
}

struct Err { double max_abs, rms, ref_rms; };
static Err compare(const std::vector<float>& got, const std::vector<double>& ref) {
  double mx = 0, s2 = 0, r2 = 0;
  for (size_t i = 0; i < ref.size(); ++i) { const double d = got[i] - ref[i]; mx = fmax(mx, fabs(d)); s2 += d * d; r2 += ref[i] * ref[i]; }
  return {mx, sqrt(s2 / ref.size()), sqrt(r2 / ref.size())};
}
static double randn(uint64_t& st) {
  auto u = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return ((st >> 11) + 0.5) / 9007199254740992.0; };
  return sqrt(-2.0 * log(u())) * cos(6.283185307179586 * u());
}

int main() {
  const int R = 40960, D = 128;
  uint64_t st = 12345;
  std::vector<float> X((size_t)R * D), dY((size_t)R * D), W((size_t)D * D), Wt((size_t)D * D), Xt((size_t)D * R), dYt((size_t)D * R);
  for (auto& v : X) v = (float)randn(st);                       // post-ELU-like activations: unit scale
  for (auto& v : dY) v = (float)(1e-3 * randn(st));
  for (auto& v : W) v = (float)(0.09 * randn(st));              // ~ 1 / sqrt(fan_in)
  for (int o = 0; o < D; ++o) for (int k = 0; k < D; ++k) Wt[(size_t)k * D + o] = W[(size_t)o * D + k];
  for (int r = 0; r < R; ++r) for (int k = 0; k < D; ++k) { Xt[(size_t)k * R + r] = X[(size_t)r * D + k]; dYt[(size_t)k * R + r] = dY[(size_t)r * D + k]; }
  // fp64 references
  std::vector<double> Yref((size_t)R * D), dXref((size_t)R * D), dWref((size_t)D * D, 0.0);
  for (int r = 0; r < R; ++r)
    for (int o = 0; o < D; ++o) {
      double a = 0, b = 0;
      for (int k = 0; k < D; ++k) { a += (double)X[(size_t)r * D + k] * W[(size_t)o * D + k]; b += (double)dY[(size_t)r * D + k] * W[(size_t)k * D + o]; }
      Yref[(size_t)r * D + o] = a; dXref[(size_t)r * D + o] = b;
    }
  for (int r = 0; r < R; ++r)
    for (int o = 0; o < D; ++o) { const double g = dY[(size_t)r * D + o]; for (int k = 0; k < D; ++k) dWref[(size_t)o * D + k] += g * X[(size_t)r * D + k]; }
  float *dX_, *ddY, *dW_, *dWt, *dXt, *ddYt, *dC, *dPart;
  unsigned short *dWp, *dWtp;
  const size_t act = (size_t)R * D * 4, wsz = (size_t)D * D * 4;
  HIP_OK(hipMalloc(&dX_, act)); HIP_OK(hipMalloc(&ddY, act)); HIP_OK(hipMalloc(&dXt, act)); HIP_OK(hipMalloc(&ddYt, act));
  HIP_OK(hipMalloc(&dW_, wsz)); HIP_OK(hipMalloc(&dWt, wsz)); HIP_OK(hipMalloc(&dC, act));
  HIP_OK(hipMalloc(&dPart, (size_t)(R / KC) * D * D * 4)); HIP_OK(hipMalloc(&dWp, 3 * wsz / 2)); HIP_OK(hipMalloc(&dWtp, 3 * wsz / 2));
  HIP_OK(hipMemcpy(dX_, X.data(), act, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(ddY, dY.data(), act, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dXt, Xt.data(), act, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(ddYt, dYt.data(), act, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dW_, W.data(), wsz, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dWt, Wt.data(), wsz, hipMemcpyHostToDevice));
  presplit<<<(D * D + 255) / 256, 256>>>(dW_, dWp, D * D);
  presplit<<<(D * D + 255) / 256, 256>>>(dWt, dWtp, D * D);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> out((size_t)R * D), outw((size_t)D * D);
  auto timeit = [&](auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / 50;
  };
  const double flops = 2.0 * R * D * D;
  printf("one 128 x 128 layer, %d rows; errors against fp64; 'us per pass' = slope of the time over repetitions of the K loop on resident operands\n", R);
  printf("%-34s %10s %10s %12s %12s %12s %10s\n", "GEMM / variant", "us", "TFLOP/s", "max |err|", "rms err / rms", "us per pass", "TFLOP/s");
  struct Case { const char* name; const float* A; const float* Bt; const unsigned short* Bp; int M, K; const std::vector<double>* ref; };
  const Case cases[3] = {{"forward  Y = X W^T", dX_, dW_, dWp, R, D, &Yref}, {"dgrad   dX = dY W", ddY, dWt, dWtp, R, D, &dXref},
                         {"wgrad   dW = dY^T X", ddYt, dXt, nullptr, D, R, &dWref}};
  for (const Case& c : cases) {
    const dim3 grid(c.M / MB, c.K / KC);
    const bool splitk = c.K > KC;
    float* target = splitk ? dPart : dC;
    auto finish = [&]() { if (splitk) reduce_parts<<<(D * D + 255) / 256, 256>>>(dPart, dC, D * D, c.K / KC); };
    auto fetch = [&](std::vector<float>& o) { HIP_OK(hipMemcpy(o.data(), dC, o.size() * 4, hipMemcpyDeviceToHost)); };
    std::vector<float>& o = splitk ? outw : out;
    for (int variant = 0; variant < 3; ++variant) {
      auto run = [&](int reps) {
        if (variant == 0) gemm_f32<<<grid, 512>>>(c.A, c.Bt, target, c.M, c.K, reps);
        else if (c.Bp) { if (variant == 1) gemm_bf16x<6, true><<<grid, 512>>>(c.A, c.Bt, c.Bp, target, c.M, c.K, reps); else gemm_bf16x<3, true><<<grid, 512>>>(c.A, c.Bt, c.Bp, target, c.M, c.K, reps); }
        else { if (variant == 1) gemm_bf16x<6, false><<<grid, 512>>>(c.A, c.Bt, nullptr, target, c.M, c.K, reps); else gemm_bf16x<3, false><<<grid, 512>>>(c.A, c.Bt, nullptr, target, c.M, c.K, reps); }
        finish();
      };
      const double us = timeit([&]() { run(1); });
      HIP_OK(hipDeviceSynchronize());
      fetch(o);
      const Err e = compare(o, *c.ref);
      const double us17 = timeit([&]() { run(17); });
      const double slope = (us17 - us) / 16.0;               // matrix-side time of one more pass over resident operands
      char nm[96];
      snprintf(nm, sizeof nm, "%s  %s", c.name, variant == 0 ? "fp32 MFMA" : (variant == 1 ? "bf16 x 6" : "bf16 x 3"));
      printf("%-34s %10.1f %10.1f %12.3e %12.3e %12.1f %10.1f\n", nm, us, flops / us / 1e6, e.max_abs, e.rms / e.ref_rms, slope, flops / slope / 1e6);
    }
  }
  return 0;
}
