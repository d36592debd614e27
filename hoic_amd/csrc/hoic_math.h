// hoic_math.h — small float math helpers for the device code.
#pragma once
#include <hip/hip_runtime.h>

#define HD __device__ __forceinline__
// pointer to global memory as such (address space 1), and the cast that tells the compiler a pointer it cannot trace
// (loaded from a table in memory) is one: accesses become global_load/store instead of flat ones
#define GPTR(T) T __attribute__((address_space(1)))*
template <class T> HD GPTR(T) as_global(T* p) { return (GPTR(T))p; }
// pointer into the workgroup's LDS as such (address space 3, 32 bits)
#define LPTR(T) T __attribute__((address_space(3)))*
// Ordering point between lanes of ONE wavefront that talk through LDS (every workgroup of the simulator kernels is a
// single wavefront).  The LDS pipeline executes a wave's instructions in order, so a ds_read issued after a ds_write
// sees it without any wait; all that is needed is that the compiler keeps the order.  __syncthreads() drains every
// outstanding LDS / global access (s_waitcnt 0) at each of the ~60 hand-over points of a pass.
// -DHOIC_FULL_BARRIER restores it (development aid).
#ifdef HOIC_FULL_BARRIER
#define wsync() __syncthreads()
#else
#define wsync() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#define MINVALF 1e-15f

// hides a lane-varying loop-invariant value from the optimiser so that the lane masks derived from it are
// recomputed on the spot (one v_cmp) instead of being hoisted out of the substep loop into (spilling) SGPR pairs
HD int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
HD float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
HD void cross3(const float* a, const float* b, float* o) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
// 1-ulp hardware reciprocal / square root / reciprocal square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32) where the simulator's
// float32 arithmetic does not need the correctly rounded forms: an IEEE division is ~10 instructions (v_div_scale x2, v_rcp,
// four fmas, v_div_fmas, v_div_fixup), an IEEE sqrtf ~14; the parity bounds against the float64 oracle are 1e-4 .. 1e-6.
// (The hardware forms flush denormal arguments: the guards below test against 1e-36, not 1e-40.)
// -DHOIC_IEEE_DIV (development build ../libhoic_ieee.so): the correctly rounded forms, for the attribution of long-horizon
// deviations (tools/attribute_bottle_outliers.py).
#ifdef HOIC_IEEE_DIV
HD float frcp(float x) { return 1.f / x; }
HD float fdiv(float a, float b) { return a / b; }
HD float fsqrt(float x) { return sqrtf(x); }
HD float frsq(float x) { return 1.f / sqrtf(x); }
#else
HD float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
HD float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
HD float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
HD float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
#endif
HD float normalize3(float* a) {
  const float n2 = dot3(a, a);
  if (n2 < 1e-36f) { a[0] = 1.f; a[1] = 0.f; a[2] = 0.f; return 0.f; }
  const float inv = frsq(n2);
  a[0] *= inv; a[1] *= inv; a[2] *= inv;
  return n2 * inv;
}
HD void quat2mat(const float* q, float* R) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}
HD void mulquat(const float* a, const float* b, float* o) {
  float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
HD void normquat(float* q) {
  const float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n2 < 1e-36f) { q[0] = 1.f; q[1] = q[2] = q[3] = 0.f; return; }
  const float inv = frsq(n2);
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
// rotate v by the unit quaternion q:  v + 2 w (u x v) + 2 u x (u x v)
HD void qrot(const float* q, const float* v, float* o) {
  float t[3], c[3];
  cross3(q + 1, v, t);
  t[0] *= 2.f; t[1] *= 2.f; t[2] *= 2.f;
  cross3(q + 1, t, c);
  const float x = v[0] + q[0] * t[0] + c[0], y = v[1] + q[0] * t[1] + c[1], z = v[2] + q[0] * t[2] + c[2];
  o[0] = x; o[1] = y; o[2] = z;
}
// sin / cos with a three-term Cody-Waite reduction by pi/2 and the cephes minimax polynomials (about 1 ulp for
// |x| < 100; joint half-angles stay below pi).  No slow path, no scratch: unlike sincosf this is a fixed
// sequence of 25 VALU operations.
HD void sincos_pi(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(k, -1.5703125f, x);
  r = fmaf(k, -4.837512969970703125e-4f, r);
  r = fmaf(k, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float s = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(z, -0.5f, 1.f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}
HD float dot6(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
HD void matvec(const float* R, const float* v, float* o) {
  float x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
        z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
HD void mattvec(const float* R, const float* v, float* o) {
  float x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
        z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
HD void matcol(const float* R, int k, float* o) { o[0] = R[k]; o[1] = R[3 + k]; o[2] = R[6 + k]; }

// reference-compatible helpers (uhc/utils/transformation.py quaternion_inverse / quaternion_matrix)
HD void quat_inv(const float* q, float* o) {
  float n = 1.f / (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  o[0] = q[0] * n; o[1] = -q[1] * n; o[2] = -q[2] * n; o[3] = -q[3] * n;
}
HD void quat_matrix_ref(const float* qin, float* R) {
  float n = qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3];
  if (n < 1e-12f) { R[0] = R[4] = R[8] = 1.f; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0.f; return; }
  float s = sqrtf(2.f / n), q[4] = {qin[0] * s, qin[1] * s, qin[2] * s, qin[3] * s};
  R[0] = 1.f - q[2] * q[2] - q[3] * q[3]; R[1] = q[1] * q[2] - q[3] * q[0]; R[2] = q[1] * q[3] + q[2] * q[0];
  R[3] = q[1] * q[2] + q[3] * q[0]; R[4] = 1.f - q[1] * q[1] - q[3] * q[3]; R[5] = q[2] * q[3] - q[1] * q[0];
  R[6] = q[1] * q[3] - q[2] * q[0]; R[7] = q[2] * q[3] + q[1] * q[0]; R[8] = 1.f - q[1] * q[1] - q[2] * q[2];
}

HD float rl(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// wavefront (64-lane) reductions on the DPP crossbar (no LDS round trips): quad butterflies, half-row and row
// mirrors give every lane of a 16-lane row the row total, four readlanes combine the rows.
template <int CTRL> HD float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
HD float wave_sum(float v) {
  v += dpp_mov<0xb1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4e>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
HD float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xb1>(v));
  v = fmaxf(v, dpp_mov<0x4e>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
}
HD float wave_min(float v) { return -wave_max(-v); }
HD unsigned wave_or(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);
  return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16)) |
         (unsigned)(__builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}
HD float sel3(float v0, float v1, float v2, int i) { return i == 0 ? v0 : (i == 1 ? v1 : v2); }
HD void sel3v(const float (*A)[3], int i, float* o) { for (int k = 0; k < 3; k++) o[k] = sel3(A[0][k], A[1][k], A[2][k], i); }
// float64 wave sum on the DPP crossbar (two 32-bit moves per step) instead of six ds_bpermute round trips
template <int CTRL> HD double dpp_mov_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
HD double rl_d(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
HD double wave_sum_d(double v) {
  v += dpp_mov_d<0xb1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_d<0x4e>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_d<0x141>(v);   // row_half_mirror
  v += dpp_mov_d<0x140>(v);   // row_mirror
  return (rl_d(v, 0) + rl_d(v, 16)) + (rl_d(v, 32) + rl_d(v, 48));
}
HD double wave_max_d(double v) {
  v = fmax(v, dpp_mov_d<0xb1>(v));
  v = fmax(v, dpp_mov_d<0x4e>(v));
  v = fmax(v, dpp_mov_d<0x141>(v));
  v = fmax(v, dpp_mov_d<0x140>(v));
  return fmax(fmax(rl_d(v, 0), rl_d(v, 16)), fmax(rl_d(v, 32), rl_d(v, 48)));
}
// inclusive prefix sum of a small non-negative integer over the wave (ballot-free, DPP row shifts + readlanes)
HD int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
  int x = v;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    int t;
    if (o == 1) t = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);        // row_shr:1
    else if (o == 2) t = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    else if (o == 4) t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    else t = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);               // row_shr:8
    x += t;
  }
  const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
  const int row = lane >> 4;
  return x + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}
