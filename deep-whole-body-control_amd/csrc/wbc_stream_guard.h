// wbc_stream_guard.h -- the launch entry points that take only a stream (no wbc_sim) run with that stream's device current and
// restore the caller's current device afterwards: a learner on cuda:1 in a process whose current device is still 0 would
// otherwise launch into a stream of another device. (The default stream belongs to whatever device is current: nothing to do.)
#pragma once
#include <hip/hip_runtime.h>
struct StreamDeviceGuard {
  int prev = -1;
  explicit StreamDeviceGuard(void* stream) {
    hipDevice_t dev = -1;
    int cur = -1;
    if (stream && hipStreamGetDevice((hipStream_t)stream, &dev) == hipSuccess && hipGetDevice(&cur) == hipSuccess && cur != (int)dev) {
      prev = cur;
      (void)hipSetDevice((int)dev);
    }
  }
  ~StreamDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  StreamDeviceGuard(const StreamDeviceGuard&) = delete;
  StreamDeviceGuard& operator=(const StreamDeviceGuard&) = delete;
};
