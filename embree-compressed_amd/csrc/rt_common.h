// Host-side basics: error funnel type, intrusive refcount, HIP error mapping, tiny vector math.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/embree3/rtcore.h"

namespace rtamd {

// Thrown anywhere below the C API; the API wrapper turns it into the device error state
// (reference: rtcore_error + RTC_CATCH_BEGIN/END, kernels/common/rtcore.h:40-67).
struct rtc_error
{
  RTCError code;
  std::string msg;
  rtc_error(RTCError c, std::string m) : code(c), msg(std::move(m)) {}
};

#define RT_THROW(code, msg) throw ::rtamd::rtc_error((code), (msg))

// hipError_t -> RTC error (OUT_OF_MEMORY for allocation failures, UNKNOWN otherwise).
inline void hip_check(hipError_t e, const char* what)
{
  if (e == hipSuccess) return;
  RTCError code = (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) ? RTC_ERROR_OUT_OF_MEMORY : RTC_ERROR_UNKNOWN;
  throw rtc_error(code, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_CHECK(expr) ::rtamd::hip_check((expr), #expr)

// Intrusive reference counting like the reference's RefCount (common/sys/ref.h): objects start at 1.
struct RefCounted
{
  std::atomic<int> refs{1};
  virtual ~RefCounted() {}
  void retain() { refs.fetch_add(1); }
  void release()
  {
    if (refs.fetch_sub(1) == 1) delete this;
  }
};

struct V3
{
  float x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit V3(float a) : x(a), y(a), z(a) {}
  float operator[](int i) const { return (&x)[i]; }
  float& operator[](int i) { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 vmin(V3 a, V3 b) { return V3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return V3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }

struct Box3
{
  V3 lo, hi;
  Box3() : lo(std::numeric_limits<float>::infinity()), hi(-std::numeric_limits<float>::infinity()) {}
  Box3(V3 l, V3 h) : lo(l), hi(h) {}
  void extend(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
  void extend(const Box3& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
  bool empty() const { return lo.x > hi.x || lo.y > hi.y || lo.z > hi.z; }
  V3 size() const { return hi - lo; }
  V3 center2() const { return lo + hi; }
  float half_area() const
  {
    if (empty()) return 0.f;
    V3 d = size();
    return d.x * (d.y + d.z) + d.y * d.z;
  }
};

} // namespace rtamd
