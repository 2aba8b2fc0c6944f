// wbc_terrain_kernel.hip -- LeggedRobot._get_heights (reference legged_gym/envs/base/legged_robot.py:793-829) for gfx950:
// the terrain height under a grid of points around every robot. Integer / index work: the result is BIT-EXACT against
// oracle/terrain_oracle.py, which restates the reference's tensor expressions one rounded fp32 operation at a time:
//
//   quat_yaw = normalize((0, 0, qz, qw))                                  legged_gym/utils/math.py:38-42 (quat_apply_yaw)
//   p = quat_apply(quat_yaw, height_points) + root_pos                    isaacgym.torch_utils.quat_apply: b + w t + xyz x t, t = 2 xyz x b
//   p += border_size;  idx = (p / horizontal_scale).long()                 LR:816-817 (truncation toward zero)
//   px = clip(idx_x, 0, rows - 2);  py = clip(idx_y, 0, cols - 2)          LR:820-821
//   h = min(H[px, py], H[px + 1, py], H[px, py + 1]) * vertical_scale      LR:823-829
//
// `p / horizontal_scale` with a Python-float divisor is evaluated by PyTorch's CUDA/HIP kernel as p * (1 / horizontal_scale)
// (aten/src/ATen/native/cuda/BinaryDivTrueKernel.cu: reciprocal of a CPU-scalar divisor, computed in fp32), which is what
// decides the index of points that sit on a cell boundary; the kernel and the oracle do the same. No fused multiply-adds:
// every product and sum below is rounded separately, as the reference's chain of element-wise kernels does.
#include <hip/hip_runtime.h>
#include "wbc_stream_guard.h"
#include <stdint.h>

extern "C" __global__ void __launch_bounds__(256) wbc_get_heights_kernel(const float* __restrict__ base_quat, int quat_stride,
                                                                       const float* __restrict__ root_pos, int pos_stride,
                                                                       const float* __restrict__ height_points, const int16_t* __restrict__ H,
                                                                       int rows, int cols, float border, float inv_hs, float vs,
                                                                       float* __restrict__ out, int num_envs, int num_points) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= num_envs * num_points) return;
  const int env = gid / num_points;
  const float* q = base_quat + (size_t)env * quat_stride;
  const float z = q[2], w = q[3];
  float n = __fsqrt_rn(__fadd_rn(__fmul_rn(z, z), __fmul_rn(w, w)));       // ||(0, 0, z, w)||
  n = fmaxf(n, 1e-9f);                                                      // normalize(): .clamp(min=eps)
  const float qz = __fdiv_rn(z, n), qw = __fdiv_rn(w, n);
  const float bx = height_points[(size_t)gid * 3], by = height_points[(size_t)gid * 3 + 1];
  // t = 2 (xyz x b) with xyz = (0, 0, qz); r = b + qw t + xyz x t  (x and y components; z is not used)
  const float tx = __fmul_rn(-__fmul_rn(qz, by), 2.f), ty = __fmul_rn(__fmul_rn(qz, bx), 2.f);
  const float rx = __fadd_rn(__fadd_rn(bx, __fmul_rn(qw, tx)), -__fmul_rn(qz, ty));
  const float ry = __fadd_rn(__fadd_rn(by, __fmul_rn(qw, ty)), __fmul_rn(qz, tx));
  const float* rp = root_pos + (size_t)env * pos_stride;
  const float px = __fadd_rn(__fadd_rn(rx, rp[0]), border), py = __fadd_rn(__fadd_rn(ry, rp[1]), border);
  long long ix = (long long)__fmul_rn(px, inv_hs), iy = (long long)__fmul_rn(py, inv_hs);        // .long(): toward zero
  ix = ix < 0 ? 0 : (ix > rows - 2 ? rows - 2 : ix);
  iy = iy < 0 ? 0 : (iy > cols - 2 ? cols - 2 : iy);
  const int16_t h1 = H[ix * cols + iy], h2 = H[(ix + 1) * cols + iy], h3 = H[ix * cols + iy + 1];
  int16_t h = h1 < h2 ? h1 : h2;
  h = h < h3 ? h : h3;
  out[gid] = __fmul_rn((float)h, vs);
}

// base_quat: device f32, row stride quat_stride floats, (x, y, z, w); root_pos: device f32, row stride pos_stride floats
// (x, y first); height_points: device f32 [N, P, 3]; height_samples: device int16 [rows, cols]; out: device f32 [N, P].
extern "C" int wbc_get_heights(const float* base_quat, int quat_stride, const float* root_pos, int pos_stride, const float* height_points,
                               const int16_t* height_samples, int rows, int cols, float border_size, float horizontal_scale,
                               float vertical_scale, float* out, int num_envs, int num_points, void* stream) {
  StreamDeviceGuard sdg(stream);
  if (!base_quat || !root_pos || !height_points || !height_samples || !out || rows < 2 || cols < 2 || num_envs <= 0 || num_points <= 0 ||
      quat_stride < 4 || pos_stride < 2 || !(horizontal_scale > 0.f))
    return -1;
  const int total = num_envs * num_points;
  const float inv_hs = 1.0f / horizontal_scale;                             // fp32 reciprocal of the scalar divisor (see the header)
  hipLaunchKernelGGL(wbc_get_heights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, base_quat, quat_stride, root_pos,
                     pos_stride, height_points, height_samples, rows, cols, border_size, inv_hs, vertical_scale, out, num_envs, num_points);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
