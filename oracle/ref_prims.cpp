// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// oracle/_ref/libref_prims.so: the reference's OWN arithmetic primitives, compiled from the sources where they
// lie under /root/reference (common/math, common/simd, common/sys headers only — these need no generated
// code and no external library).  kernels/* cannot be built this way: every file there includes
// kernels/common/default.h -> kernels/config.h, which only the reference's CMake run generates, and the
// fork's files need Eigen3, which is absent ("Oracle").  So this library pins the building
// blocks the restatement in embree_oracle.c relies on — rcp (rcpps + Newton step), dot / cross /
// stable_triangle_normal with their AVX2 FMA placement, select_min's tie rule — and tests/test_oracle_ref.py
// checks the restatement against them bit for bit on random inputs.
//
// Only the extern "C" shims below are ours; everything they call is the reference's code.
#include "common/math/vec3.h"
#include "common/math/vec3fa.h"
#include "common/simd/simd.h"

using namespace embree;

static inline Vec3<vfloat4> load3(const float* p) { return Vec3<vfloat4>(vfloat4::loadu(p), vfloat4::loadu(p + 4), vfloat4::loadu(p + 8)); }
static inline void store3(float* p, const Vec3<vfloat4>& v)
{
  vfloat4::storeu(p, v.x);
  vfloat4::storeu(p + 4, v.y);
  vfloat4::storeu(p + 8, v.z);
}

extern "C" {

// common/math/math.h:60-75
float ref_rcp(float x) { return rcp(x); }
// common/simd/vfloat4_sse2.h:244-258 on 4 lanes
void ref_rcp4(const float* x, float* o) { vfloat4::storeu(o, rcp(vfloat4::loadu(x))); }
// common/math/vec3fa.h:163-168
void ref_rcp_safe3(const float* x, float* o)
{
  const Vec3fa r = rcp_safe(Vec3fa(x[0], x[1], x[2]));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void ref_zero_fix3(const float* x, float* o)
{
  const Vec3fa r = zero_fix(Vec3fa(x[0], x[1], x[2]));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
// common/math/vec3.h:193,198,200-212 on SoA operands a = x[4] y[4] z[4]
void ref_dot4(const float* a, const float* b, float* o) { vfloat4::storeu(o, dot(load3(a), load3(b))); }
void ref_cross4(const float* a, const float* b, float* o) { store3(o, cross(load3(a), load3(b))); }
void ref_stable_triangle_normal4(const float* a, const float* b, const float* c, float* o)
{
  store3(o, stable_triangle_normal(load3(a), load3(b), load3(c)));
}
// common/simd/vfloat4_sse2.h:654-659
int ref_select_min4(unsigned validMask, const float* v)
{
  const vboolf4 valid((validMask & 1) != 0, (validMask & 2) != 0, (validMask & 4) != 0, (validMask & 8) != 0);
  return (int)select_min(valid, vfloat4::loadu(v));
}
// max(a,b,c,d) / min(a,b,c,d) as used by the node tests (common/simd/vfloat4_sse2.h)
void ref_max4of(const float* a, const float* b, const float* c, const float* d, float* o)
{
  vfloat4::storeu(o, max(vfloat4::loadu(a), vfloat4::loadu(b), vfloat4::loadu(c), vfloat4::loadu(d)));
}
void ref_min4of(const float* a, const float* b, const float* c, const float* d, float* o)
{
  vfloat4::storeu(o, min(vfloat4::loadu(a), vfloat4::loadu(b), vfloat4::loadu(c), vfloat4::loadu(d)));
}
// common/math/math.h:86-97
float ref_rsqrt(float x) { return rsqrt(x); }
// Vec3fa dot / length (vec3fa.h:289-310): the fork's traversal measures its projected ray with these (compressed.h:498,587)
float ref_dot3fa(const float* a, const float* b) { return dot(Vec3fa(a[0], a[1], a[2]), Vec3fa(b[0], b[1], b[2])); }
float ref_length3(const float* x) { return length(Vec3fa(x[0], x[1], x[2])); }
// Vec3<float> rcp_safe (vec3.h:74-80), the form intersect_frustum uses (compressed_help.h:111)
void ref_rcp_safe3f(const float* x, float* o)
{
  const Vec3f r = rcp_safe(Vec3f(x[0], x[1], x[2]));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
// Vec3fa normalize (rsqrt based; used by the tutorials' camera, the fork's frame construction and its traversal, compressed.h:499,505)
void ref_normalize3(const float* x, float* o)
{
  const Vec3fa r = normalize(Vec3fa(x[0], x[1], x[2]));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
}
