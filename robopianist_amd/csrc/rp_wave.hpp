// rp_wave.hpp — wave-level building blocks of the step kernels: lane broadcasts, DPP
// reductions, LDS atomics, small fixed-size linear algebra, MuJoCo's impedance function.
#pragma once
#include "rp_model.hpp"

namespace rpk {

template <typename T> struct Num;
template <> struct Num<float> {
  static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
  static __device__ __forceinline__ float pow(float x, float y) { return powf(x, y); }
  static __device__ __forceinline__ void sincos(float x, float* s, float* c) { sincosf(x, s, c); }
  static __device__ __forceinline__ float eps() { return 1.1920929e-7f; }
};
template <> struct Num<double> {
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double abs(double x) { return fabs(x); }
  static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
  static __device__ __forceinline__ void sincos(double x, double* s, double* c) { ::sincos(x, s, c); }
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
};
#define RPK_MINVAL ((T)1e-15)

__device__ __forceinline__ float bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v),
                                                             __builtin_amdgcn_readfirstlane(l)));
}
__device__ __forceinline__ int bcast(int v, int l) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}
__device__ __forceinline__ double bcast(double v, int l) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  int sl = __builtin_amdgcn_readfirstlane(l);
  lo = __builtin_amdgcn_readlane(lo, sl);
  hi = __builtin_amdgcn_readlane(hi, sl);
  long long r = ((long long)hi << 32) | (unsigned int)lo;
  return __builtin_bit_cast(double, r);
}
// Wave-wide sum, result uniform in every lane.  Four DPP butterfly steps (xor 1, xor 2,
// half-mirror, mirror) leave each 16-lane row holding its row sum without touching the
// LDS crossbar; the four row sums are then read with v_readlane and added as scalars.
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_move(double v) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// (round 6: the four row sums are combined by two more DPP steps -- row_bcast:15 adds row r - 1's sum into row r, then
// row_bcast:31 adds lane 31's (r0 + r1) into rows 2 and 3 -- and the total is read from lane 63: (r3 + r2) + (r1 + r0),
// bit for bit the old (r0 + r1) + (r2 + r3) since every IEEE addition commutes; 20 instructions for a double instead of
// 29 (eight v_readlane, the moves of the scalar operands back into vector registers and three adds): the lean solver stage
// takes ~120 wave sums per mj_step and is instruction-issue bound.)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  v += dpp_move<0x142>(v);  // row_bcast:15 (row 0 receives 0: bound_ctrl)
  v += dpp_move<0x143>(v);  // row_bcast:31 (rows 0, 1 receive 0)
  return bcast(v, 63);
}
template <int CTRL> __device__ __forceinline__ int dpp_move(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// integer wave reductions on the same DPP butterfly (uniform result)
__device__ __forceinline__ int wave_or(int v) {
  v |= dpp_move<0xB1>(v); v |= dpp_move<0x4E>(v); v |= dpp_move<0x141>(v); v |= dpp_move<0x140>(v);
  return (bcast(v, 0) | bcast(v, 16)) | (bcast(v, 32) | bcast(v, 48));
}
__device__ __forceinline__ int wave_max(int v) {
  v = max(v, dpp_move<0xB1>(v)); v = max(v, dpp_move<0x4E>(v));
  v = max(v, dpp_move<0x141>(v)); v = max(v, dpp_move<0x140>(v));
  return max(max(bcast(v, 0), bcast(v, 16)), max(bcast(v, 32), bcast(v, 48)));
}
__device__ __forceinline__ unsigned long long wave_or(unsigned long long v) {
  unsigned lo = (unsigned)wave_or((int)(v & 0xffffffffull)), hi = (unsigned)wave_or((int)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// (RPK_SREG_CONSTRAINT: "+s" = keep the pointer in SGPRs; the CPU wave emulator of tests/wavesim
// compiles this file for the host and overrides it)
#ifndef RPK_SREG_CONSTRAINT
#define RPK_SREG_CONSTRAINT "+s"
#endif
template <typename P>
__device__ __forceinline__ const P* fresh(const P* p) {
  asm volatile("" : RPK_SREG_CONSTRAINT(p));
  return p;
}
// fire-and-forget LDS accumulate (ds_add_f32 / ds_add_f64): no read-modify-write round trip
template <typename T> __device__ __forceinline__ void lds_add(T* p, T v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// A wave-uniform floating-point value, moved to scalar registers: the compiler then keeps it (and, under
// pressure, spills it to a LANE of a vector register) instead of occupying a full vector register pair.
__device__ __forceinline__ float uni(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double uni(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// A wave-uniform pointer to memory no kernel writes (model tables), in the constant address space: loads through
// it are scalar loads (one fetch for the wave, scalar cache) instead of 64-lane vector loads.  (RPK_CONST_AS: the
// CPU wave emulator of tests/wavesim compiles this file for the host and defines it away.)
#ifndef RPK_CONST_AS
#define RPK_CONST_AS __attribute__((address_space(4)))
#endif
template <typename P>
__device__ __forceinline__ const P RPK_CONST_AS* uniform_const(const P* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32));
  return (const P RPK_CONST_AS*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) {
  return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ int tri(int i, int j) { return (i * (i + 1) >> 1) + j; }

template <typename T> __device__ __forceinline__ T dot3(const T* a, const T* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
template <typename T> __device__ __forceinline__ void cross3(T* r, const T* a, const T* b) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> __device__ __forceinline__ void mat_vec(T* r, const T* m, const T* v) {
  T x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  T y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  T z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> __device__ __forceinline__ void matT_vec(T* r, const T* m, const T* v) {
  T x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  T y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  T z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> __device__ __forceinline__ void mat_mul(T* r, const T* a, const T* b) {
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = t[i];
}
// spatial inertia (about the tree reference point) times motion vector
// I = [Ixx Iyy Izz Ixy Ixz Iyz mdx mdy mdz m]
template <typename T> __device__ __forceinline__ void mul_inert(T* res, const T* I, const T* v) {
  T t[3];
  res[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
  res[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
  res[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
  cross3(t, I + 6, v + 3);
  res[0] += t[0]; res[1] += t[1]; res[2] += t[2];
  cross3(t, I + 6, v);
  res[3] = I[9] * v[3] - t[0]; res[4] = I[9] * v[4] - t[1]; res[5] = I[9] * v[5] - t[2];
}
template <typename T> __device__ __forceinline__ T dot6(const T* a, const T* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
template <typename T> __device__ __forceinline__ void make_frame(const T* n, T* t1, T* t2) {
  if (Num<T>::abs(n[1]) < (T)0.5) { t1[0] = 0; t1[1] = 1; t1[2] = 0; }
  else { t1[0] = 0; t1[1] = 0; t1[2] = 1; }
  T dp = dot3(n, t1);
  t1[0] -= dp * n[0]; t1[1] -= dp * n[1]; t1[2] -= dp * n[2];
  T inv = (T)1 / Num<T>::sqrt(dot3(t1, t1));
  t1[0] *= inv; t1[1] *= inv; t1[2] *= inv;
  cross3(t2, n, t1);
}
template <typename T>
__device__ __forceinline__ T impedance(const T* solimp, T pos) {  // margin == 0
  T dmin = fmin((T)0.9999, fmax((T)0.0001, solimp[0]));
  T dmax = fmin((T)0.9999, fmax((T)0.0001, solimp[1]));
  T width = fmax(RPK_MINVAL, solimp[2]);
  T mid = fmin((T)0.9999, fmax((T)0.0001, solimp[3]));
  T power = fmax((T)1, solimp[4]);
  if (dmin == dmax || width <= RPK_MINVAL) return (T)0.5 * (dmin + dmax);
  T x = Num<T>::abs(pos) / width;
  if (x >= (T)1) return dmax;
  if (x == (T)0) return dmin;
  T y;
  if (power == (T)2) {  // the MuJoCo default; pow(x, 2) is exactly x * x
    if (x <= mid) y = x * x / mid;
    else y = (T)1 - ((T)1 - x) * ((T)1 - x) / ((T)1 - mid);
  } else if (x <= mid) y = Num<T>::pow(x, power) / Num<T>::pow(mid, power - 1);
  else y = (T)1 - Num<T>::pow((T)1 - x, power) / Num<T>::pow((T)1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}
}  // namespace rpk
