// Where do the four wavefronts of a 256-thread workgroup land (gfx950)?  1024 workgroups x 256 threads with 40 KB of LDS each (four per
// CU, as four robots of wbc_step_kernel would need), after a perturbing launch of another shape -- as the step kernel follows the
// policy kernel in the rollout. Prints, per launch: do a workgroup's waves sit on four different SIMDs of one CU; is wave i always on
// the same SIMD; which workgroups share a CU (their in-XCD index differences); how many distinct rotations (g / 32) mod 4 meet on a CU.
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip && ./wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <set>
#include <algorithm>

__global__ void __launch_bounds__(256) probe(unsigned* out, int spin) {
  __shared__ float lds[10240];          // 40 KB
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = (float)spin;
  float x = lds[(threadIdx.x * 7) & 255];
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;       // keep the waves resident for a while (everything is resident at once anyway)
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) & 15u;
    out[blockIdx.x * 4 + wave] = ((hw >> 4) & 3u) | (((hw >> 8) & 0xFFu) << 2) | (xcc << 10) | (x == 123.f ? 1u << 31 : 0u);
  }
}
__global__ void perturb(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
  const int G = 1024;
  unsigned* d; float* junk;
  hipMalloc(&d, G * 4 * sizeof(unsigned)); hipMalloc(&junk, 1 << 24);
  std::vector<unsigned> h(G * 4);
  srand(1);
  for (int rep = 0; rep < 6; ++rep) {
    const int pg = 100 + rand() % 900, pb = 64 << (rand() % 3);
    hipLaunchKernelGGL(perturb, dim3(pg), dim3(pb), 0, 0, junk, 1 << 22);
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d, 20000);
    hipMemcpy(h.data(), d, G * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    int same_cu = 0, distinct_simd = 0, wave_eq_simd = 0;
    std::map<unsigned, std::vector<int>> cu_wgs;
    std::map<int, int> rot_hist;
    for (int g = 0; g < G; ++g) {
      std::set<unsigned> cus, simds;
      for (int w = 0; w < 4; ++w) { cus.insert(h[g * 4 + w] >> 2); simds.insert(h[g * 4 + w] & 3u); }
      same_cu += cus.size() == 1; distinct_simd += simds.size() == 4;
      int eq = 0; for (int w = 0; w < 4; ++w) eq += (int)(h[g * 4 + w] & 3u) == w;
      wave_eq_simd += eq == 4;
      rot_hist[(int)(h[g * 4] & 3u)]++;
      cu_wgs[h[g * 4] >> 2].push_back(g);
    }
    printf("launch %d (after a %d x %d launch): workgroups on one CU %d / %d, on four different SIMDs %d, wave i on SIMD i %d; SIMD of wave 0: ", rep, pg, pb, same_cu, G, distinct_simd, wave_eq_simd);
    for (auto& kv : rot_hist) printf("%d:%d ", kv.first, kv.second);
    printf("\n");
    // per CU: number of workgroups, their in-XCD index j = g >> 3 differences, distinct values of (j / 32) & 3, and of the SIMD of wave 0
    std::map<int, int> nper, nrot, nsimd0; std::map<int,int> diffs;
    for (auto& kv : cu_wgs) {
      auto v = kv.second; std::sort(v.begin(), v.end());
      nper[(int)v.size()]++;
      std::set<int> r, s0;
      for (int g : v) { r.insert(((g >> 3) / 32) & 3); s0.insert((int)(h[g * 4] & 3u)); }
      nrot[(int)r.size()]++; nsimd0[(int)s0.size()]++;
      for (size_t i = 1; i < v.size(); ++i) diffs[(v[i] >> 3) - (v[i - 1] >> 3)]++;
    }
    printf("   CUs by number of workgroups: "); for (auto& kv : nper) printf("%d:%d ", kv.first, kv.second);
    printf("| by distinct (j/32)&3 among their workgroups: "); for (auto& kv : nrot) printf("%d:%d ", kv.first, kv.second);
    printf("| by distinct SIMD-of-wave-0: "); for (auto& kv : nsimd0) printf("%d:%d ", kv.first, kv.second);
    printf("\n   in-XCD index differences between consecutive workgroups of a CU: "); int c = 0; for (auto& kv : diffs) if (c++ < 12) printf("%d:%d ", kv.first, kv.second);
    printf("\n");
  }
  return 0;
}
