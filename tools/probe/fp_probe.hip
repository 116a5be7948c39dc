// Probe: are fp32 divide / sqrt correctly rounded in device code built with the library's flags?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
__global__ void k(const float* a, const float* b, float* q, float* s, float* rs, float* r, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { q[i] = a[i] / b[i]; s[i] = sqrtf(a[i]); rs[i] = 1.0f / sqrtf(a[i]); r[i] = 1.0f / b[i]; }
}
int main() {
  const int n = 1 << 20;
  std::vector<float> a(n), b(n), q(n), s(n), rs(n), r(n);
  srand(1);
  for (int i = 0; i < n; i++) { a[i] = (float)rand() / RAND_MAX * 100.f + 1e-3f; b[i] = ((float)rand() / RAND_MAX - 0.5f) * 50.f + 1e-4f; }
  float *da, *db, *dq, *ds, *drs, *dr;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dq, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&drs, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dq, ds, drs, dr, n);
  hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(rs.data(), drs, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
  int eq = 0, es = 0, ers = 0, er = 0;
  for (int i = 0; i < n; i++) {
    if (q[i] != a[i] / b[i]) eq++;
    if (s[i] != sqrtf(a[i])) es++;
    if (rs[i] != 1.0f / sqrtf(a[i])) ers++;
    if (r[i] != 1.0f / b[i]) er++;
  }
  printf("mismatches of %d: div %d sqrt %d rsqrt %d rcp %d\n", n, eq, es, ers, er);
  return 0;
}
