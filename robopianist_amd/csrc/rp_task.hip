// Fused task-layer kernels (see include/rp_task.h).  gfx950 only.
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/rp_task.h"

namespace {
thread_local std::string g_task_err;

template <typename T> __device__ __forceinline__ T wsum(T v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// dm_control.utils.rewards.tolerance, gaussian sigmoid, value_at_margin = 0.1, bounds (0, b)
template <typename T> __device__ __forceinline__ T tol_gauss(T x, T upper, T margin) {
  if (x >= (T)0 && x <= upper) return (T)1;
  const T d = (x < (T)0 ? -x : x - upper) / margin;
  const T s = (T)2.145966026289347;  // sqrt(-2 ln 0.1)
  const T z = d * s;
  return exp((T)-0.5 * z * z);
}

// one wavefront per env; lane k owns keys k and k + 64
template <typename T>
__global__ __launch_bounds__(64) void rp_task_reward_kernel(rp_task_reward_args a) {
  const int env = blockIdx.x, lane = threadIdx.x;
  const size_t E = (size_t)a.n_envs;
  const T* goal = (const T*)a.goal_current + (size_t)env * 89;
  const T* nstate = (const T*)a.key_norm_state + (size_t)env * 88;
  const unsigned char* act = a.key_activation + (size_t)env * 88;
  const T* qpos = (const T*)a.qpos + (size_t)env * a.nv;
  const T* sites = (const T*)a.site_xpos + (size_t)env * a.n_sites * 3;
  const T kclose = (T)a.key_close, fclose = (T)a.finger_close;

  T kp_sum = 0, fg_sum = 0;
  int n_on = 0, false_pos = 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int k = lane + 64 * s;
    if (k < RP_TASK_N_KEYS) {
      const T g = goal[k];
      const bool on = g > (T)0;
      const bool pressed = act[k] != 0;
      if (on) {
        n_on++;
        kp_sum += tol_gauss(g - nstate[k], kclose, kclose * (T)10);
        if (a.use_fingering) {
          long long f = a.finger_current[(size_t)env * 88 + k];
          const int fid = f < 0 ? 4 : (int)f;  // a note without fingering counts as finger 4 (:401-412)
          const T* tip = sites + (size_t)a.tip_site[fid] * 3;
          const T* an = (const T*)a.key_anchor + 3 * k;
          const T* hf = (const T*)a.key_half + 3 * k;
          const T q = qpos[a.key_qadr[k]];
          const T hx = hf[0];
          // key geom centre + (0.35 size_x, 0, 0.5 size_z)  (:311-313)
          const T tx = an[0] + hx * cos(q) + (T)0.35 * hx;
          const T ty = an[1];
          const T tz = an[2] - hx * sin(q) + (T)0.5 * hf[2];
          const T dx = tx - tip[0], dy = ty - tip[1], dz = tz - tip[2];
          fg_sum += tol_gauss(sqrt(dx * dx + dy * dy + dz * dz), fclose, fclose * (T)10);
        }
      } else if (pressed) false_pos = 1;
    }
  }
  // energy: |actuatorfrc| * |actuatorvel| over the hand actuators
  T en = 0;
  for (int i = lane; i < a.n_hand_act; i += 64) {
    const int j = a.hand_act[i];
    en += fabs(((const T*)a.act_force)[(size_t)env * a.nu + j]) * fabs(((const T*)a.act_vel)[(size_t)env * a.nu + j]);
  }
  // forearm: any contact between a right-forearm geom and a left-forearm geom
  int hit = 0;
  if (a.use_forearm) {
    for (int c = lane; c < a.n_contacts; c += 64) {
      const int ga = a.contact_geoms[((size_t)env * a.n_contacts + c) * 2];
      const int gb = a.contact_geoms[((size_t)env * a.n_contacts + c) * 2 + 1];
      bool ar = false, al = false, br = false, bl = false;
      for (int i = 0; i < a.n_rfa; i++) { ar |= ga == a.rfa[i]; br |= gb == a.rfa[i]; }
      for (int i = 0; i < a.n_lfa; i++) { al |= ga == a.lfa[i]; bl |= gb == a.lfa[i]; }
      if ((ar && bl) || (al && br)) hit = 1;
    }
  }
  kp_sum = wsum(kp_sum); fg_sum = wsum(fg_sum); en = wsum(en);
  const int non = (int)wsum((float)n_on);
  const bool fpos = __ballot(false_pos) != 0ull, fhit = __ballot(hit) != 0ull;
  if (lane == 0) {
    const T key_press = (non > 0 ? (T)0.5 * kp_sum / (T)non : (T)0) + (T)0.5 * (fpos ? (T)0 : (T)1);
    const T sustain = tol_gauss(goal[88] - (a.sustain_activation[env] ? (T)1 : (T)0), kclose, kclose * (T)10);
    const T energy = -(T)a.energy_coef * en;
    const T fingering = a.use_fingering ? (non > 0 ? fg_sum / (T)non : (T)0) : (T)0;
    const T forearm = a.use_forearm ? (fhit ? (T)0 : (T)0.5) : (T)0;
    T* t = (T*)a.terms;
    t[0 * E + env] = key_press; t[1 * E + env] = sustain; t[2 * E + env] = energy;
    t[3 * E + env] = fingering; t[4 * E + env] = forearm;
    T tot = (T)0 + key_press;
    tot += sustain; tot += energy;
    if (a.use_fingering) tot += fingering;
    if (a.use_forearm) tot += forearm;
    ((T*)a.total)[env] = tot;
  }
}
}  // namespace

extern "C" {

const char* rp_task_last_error(void) { return g_task_err.c_str(); }

int rp_task_rewards(const rp_task_reward_args* a, void* hip_stream) {
  if (!a) { g_task_err = "rp_task_rewards: null args"; return -1; }
  if (a->precision != 32 && a->precision != 64) { g_task_err = "rp_task_rewards: precision must be 32 or 64"; return -1; }
  if (a->n_envs <= 0) { g_task_err = "rp_task_rewards: n_envs must be positive"; return -1; }
  if (!a->qpos || !a->act_force || !a->act_vel || !a->site_xpos || !a->contact_geoms || !a->goal_current ||
      !a->key_norm_state || !a->key_activation || !a->sustain_activation || !a->finger_current || !a->key_qadr ||
      !a->key_anchor || !a->key_half || !a->hand_act || !a->tip_site || !a->terms || !a->total) {
    g_task_err = "rp_task_rewards: null array pointer";
    return -1;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  if (a->precision == 32) hipLaunchKernelGGL(rp_task_reward_kernel<float>, dim3(a->n_envs), dim3(64), 0, s, *a);
  else hipLaunchKernelGGL(rp_task_reward_kernel<double>, dim3(a->n_envs), dim3(64), 0, s, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_task_err = std::string("rp_task_rewards: ") + hipGetErrorString(e); return -2; }
  return 0;
}

}  // extern "C"
