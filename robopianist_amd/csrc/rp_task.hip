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

// precision / recall / F1 with sklearn's average="binary", zero_division=1 conventions
__device__ __forceinline__ void prf(int tp, int fp, int fn, double* out) {
  const double p = tp + fp > 0 ? (double)tp / (double)(tp + fp) : 1.0;
  const double r = tp + fn > 0 ? (double)tp / (double)(tp + fn) : 1.0;
  double f = p + r > 0.0 ? 2.0 * p * r / (p + r) : 0.0;
  if (tp + fp + fn == 0) f = 1.0;
  out[0] = p; out[1] = r; out[2] = f;
}

// Optimal-transport fingering term (piano_with_shadow_hands.py:333-369): minimum-cost assignment
// between the 10 fingertips and the keys to press (scipy.optimize.linear_sum_assignment in the
// reference), mean tolerance of the assigned distances.  One wavefront; shortest augmenting paths
// with potentials (Kuhn-Munkres), rows = the smaller side (<= 10), columns over lanes (two per
// lane).  Always in double.  cost[r * m + c]; returns the mean tolerance in every lane.
struct OtScratch {
  double cost[10 * RP_TASK_N_KEYS];
  double u[11];
  int p[RP_TASK_N_KEYS + 1], way[RP_TASK_N_KEYS + 1];   // 1-based columns, 0 = the virtual start column
};
__device__ __forceinline__ double ot_assign_mean_tolerance(OtScratch& S, int n, int m, int lane, double fclose) {
  const double INF = 1e300;
  double v[2] = {0, 0}, minv[2];
  bool used[2];
  for (int j = lane; j <= m; j += 64) S.p[j] = 0;
  if (lane <= n) S.u[lane] = 0;
  __syncthreads();
  for (int i = 1; i <= n; i++) {
    if (lane == 0) S.p[0] = i;
    minv[0] = minv[1] = INF; used[0] = used[1] = false;
    int j0 = 0;
    __syncthreads();
    while (true) {
#pragma unroll
      for (int s = 0; s < 2; s++) if (j0 == lane + 64 * s + 1) used[s] = true;
      const int i0 = S.p[j0];
      const double ui0 = S.u[i0];
      double best = INF;
      int bj = 0x7fffffff;
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int j = lane + 64 * s + 1;
        if (j <= m && !used[s]) {
          const double cur = S.cost[(size_t)(i0 - 1) * m + (j - 1)] - ui0 - v[s];
          if (cur < minv[s]) { minv[s] = cur; S.way[j] = j0; }
          if (minv[s] < best) { best = minv[s]; bj = j; }
        }
      }
      // wave argmin (ties: smallest column, as a sequential scan would pick)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oj = __shfl_xor(bj, off, 64);
        if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
      }
      const double delta = best;
      const int j1 = bj;
      __syncthreads();  // every lane has read u[i0] / p[j0] before the potentials move
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int j = lane + 64 * s + 1;
        if (j <= m) {
          if (used[s]) { S.u[S.p[j]] += delta; v[s] -= delta; }   // distinct rows: no conflict
          else minv[s] -= delta;
        }
      }
      if (lane == 0) S.u[i] += delta;   // the virtual column holds row i
      j0 = j1;
      __syncthreads();
      if (S.p[j0] == 0) break;
    }
    // augment along the alternating path (uniform; lane 0 writes)
    __syncthreads();
    if (lane == 0) {
      int j = j0;
      while (j) { const int jn = S.way[j]; S.p[j] = S.p[jn]; j = jn; }
    }
    __syncthreads();
  }
  double sum = 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int j = lane + 64 * s + 1;
    if (j <= m && S.p[j] != 0) sum += tol_gauss(S.cost[(size_t)(S.p[j] - 1) * m + (j - 1)], fclose, fclose * 10.0);
  }
  sum = wsum(sum);
  __syncthreads();
  return sum / (double)n;
}

// Per-lane view of the task state the reward terms read: lane k owns keys k and k + 64.
template <typename T> struct KeyView { T goal[2], nstate[2]; bool pressed[2]; long long finger[2]; T goal_sustain; bool sustain_on; };

// All reward terms of one env (one wavefront); the result is valid in lane 0.
template <typename T>
__device__ __forceinline__ T reward_env(const rp_task_reward_args& a, int env, int lane, const KeyView<T>& kv) {
  const size_t E = (size_t)a.n_envs;
  const T* qpos = (const T*)a.qpos + (size_t)env * a.nv;
  const T* sites = (const T*)a.site_xpos + (size_t)env * a.n_sites * 3;
  const T kclose = (T)a.key_close, fclose = (T)a.finger_close;
  T kp_sum = 0, fg_sum = 0;
  int n_on = 0, n_fg = 0, false_pos = 0;
  const bool ot = a.use_fingering == 2;
  T tgt[2][3] = {{0, 0, 0}, {0, 0, 0}};   // fingertip target of this lane's goal keys
  bool on[2] = {false, false};
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int k = lane + 64 * s;
    if (k < RP_TASK_N_KEYS) {
      const T g = kv.goal[s];
      if (g > (T)0) {
        n_on++;
        on[s] = true;
        kp_sum += tol_gauss(g - kv.nstate[s], kclose, kclose * (T)10);
        if (a.use_fingering) {
          const T* an = (const T*)a.key_anchor + 3 * k;
          const T* hf = (const T*)a.key_half + 3 * k;
          const T q = qpos[a.key_qadr[k]];
          const T hx = hf[0];
          // key geom centre + (0.35 size_x, 0, 0.5 size_z)  (:311-313)
          tgt[s][0] = an[0] + hx * cos(q) + (T)0.35 * hx;
          tgt[s][1] = an[1];
          tgt[s][2] = an[2] - hx * sin(q) + (T)0.5 * hf[2];
        }
        // two hands: a note without fingering counts as finger 4 (:401-412); one hand: only the
        // notes fingered by this hand take part (finger holds the local index or -1)
        const long long f = kv.finger[s];
        if (a.use_fingering == 1 && (a.hand_filter == 0 || f >= 0)) {
          const int fid = f < 0 ? 4 : (int)f;
          n_fg++;
          const T* tip = sites + (size_t)a.tip_site[fid] * 3;
          const T dx = tgt[s][0] - tip[0], dy = tgt[s][1] - tip[1], dz = tgt[s][2] - tip[2];
          fg_sum += tol_gauss(sqrt(dx * dx + dy * dy + dz * dz), fclose, fclose * (T)10);
        }
      } else if (kv.pressed[s]) false_pos = 1;
    }
  }
  // OT fingering (:333-369): assignment between the fingertips and the keys to press
  T ot_rew = (T)1;  // no key to press
  if (ot) {
    __shared__ OtScratch ots;
    const unsigned long long m0 = __ballot(on[0]), m1 = __ballot(on[1]);
    const int k0 = __popcll(m0), nk = k0 + __popcll(m1);   // uniform
    if (nk > 0) {
      const int ntip = a.hand_filter ? 5 : 10;
      const bool tips_are_rows = nk >= ntip;   // rows = the smaller side
      const int n = tips_are_rows ? ntip : nk, m = tips_are_rows ? nk : ntip;
      const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
      for (int s = 0; s < 2; s++) {
        if (on[s]) {
          const int kc = (s == 0 ? 0 : k0) + __popcll((s == 0 ? m0 : m1) & below);  // compact key index
          for (int f = 0; f < ntip; f++) {
            const T* tip = sites + (size_t)a.tip_site[f] * 3;
            const double dx = (double)tgt[s][0] - (double)tip[0], dy = (double)tgt[s][1] - (double)tip[1],
                         dz = (double)tgt[s][2] - (double)tip[2];
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            if (tips_are_rows) ots.cost[(size_t)f * m + kc] = d; else ots.cost[(size_t)kc * m + f] = d;
          }
        }
      }
      __syncthreads();
      ot_rew = (T)ot_assign_mean_tolerance(ots, n, m, lane, (double)a.finger_close);
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
  const int non = (int)wsum((float)n_on), nfg = (int)wsum((float)n_fg);
  const bool fpos = __ballot(false_pos) != 0ull, fhit = __ballot(hit) != 0ull;
  T tot = 0;
  if (lane == 0) {
    const T key_press = (non > 0 ? (T)0.5 * kp_sum / (T)non : (T)0) + (T)0.5 * (fpos ? (T)0 : (T)1);
    const T sustain = tol_gauss(kv.goal_sustain - (kv.sustain_on ? (T)1 : (T)0), kclose, kclose * (T)10);
    const T energy = -(T)a.energy_coef * en;
    const T fingering = ot ? ot_rew : (a.use_fingering ? (nfg > 0 ? fg_sum / (T)nfg : (T)0) : (T)0);
    const T forearm = a.use_forearm ? (fhit ? (T)0 : (T)0.5) : (T)0;
    T* t = (T*)a.terms;
    t[0 * E + env] = key_press; t[1 * E + env] = sustain; t[2 * E + env] = energy;
    t[3 * E + env] = fingering; t[4 * E + env] = forearm;
    tot = (T)0 + key_press;
    tot += sustain; tot += energy;
    if (a.use_fingering) tot += fingering;
    if (a.use_forearm) tot += forearm;
  }
  return tot;
}

// one wavefront per env
template <typename T>
__global__ __launch_bounds__(64) void rp_task_reward_kernel(rp_task_reward_args a) {
  const int env = blockIdx.x, lane = threadIdx.x;
  KeyView<T> kv;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int k = lane + 64 * s, kk = k < RP_TASK_N_KEYS ? k : 0;
    kv.goal[s] = ((const T*)a.goal_current)[(size_t)env * 89 + kk];
    kv.nstate[s] = ((const T*)a.key_norm_state)[(size_t)env * 88 + kk];
    kv.pressed[s] = a.key_activation[(size_t)env * 88 + kk] != 0;
    kv.finger[s] = a.finger_current[(size_t)env * 88 + kk];
  }
  kv.goal_sustain = ((const T*)a.goal_current)[(size_t)env * 89 + 88];
  kv.sustain_on = a.sustain_activation[env] != 0;
  const T tot = reward_env<T>(a, env, lane, kv);
  if (lane == 0) ((T*)a.total)[env] = tot;
}

template <typename T>
__global__ __launch_bounds__(64) void rp_task_advance_kernel(rp_task_advance_args p) {
  const rp_task_reward_args& a = p.rw;
  const int env = blockIdx.x, lane = threadIdx.x;
  const bool resetting = p.needs_reset[env] != 0, active = !resetting;
  long long t = p.t_idx[env];
  T dstate = ((T*)p.discount_state)[env];
  if (resetting) { t = 0; dstate = (T)1; }  // _reset_quantities_at_episode_init (:146-149)
  long long song = p.song_id[env];
  if (resetting && p.next_ready && p.next_ready[env]) {  // prefetch mode: switch to the prepared slot
    song ^= 1;
    if (lane == 0) { p.song_id[env] = song; p.next_ready[env] = 0; p.consumed[env] = 1; }
  }
  const long long slen = p.song_len[song];
  const T* qpos = (const T*)a.qpos + (size_t)env * a.nv;
  T* gstate = (T*)p.goal_state + (size_t)env * (p.n_lookahead + 1) * 89;

  // ---- Piano._update_key_state + after_step
  KeyView<T> kv;
  int fail = 0;
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const int k = lane + 64 * s;
    kv.goal[s] = 0; kv.nstate[s] = 0; kv.pressed[s] = false; kv.finger[s] = -1;
    if (k < RP_TASK_N_KEYS) {
      const T lo = ((const T*)p.key_qrange)[2 * k], hi = ((const T*)p.key_qrange)[2 * k + 1];
      const T q = qpos[a.key_qadr[k]];
      const T st = q < lo ? lo : (q > hi ? hi : q);
      const T ns = st / hi;
      const bool pressed = fabs(st - hi) <= (T)p.key_threshold;
      ((T*)p.key_state)[(size_t)env * 88 + k] = st;
      ((T*)a.key_norm_state)[(size_t)env * 88 + k] = ns;
      ((unsigned char*)a.key_activation)[(size_t)env * 88 + k] = pressed;
      const T g = gstate[k];                                   // goal_state[:, 0] of the last observation
      const long long f = p.finger_next[(size_t)env * 88 + k];
      ((T*)a.goal_current)[(size_t)env * 89 + k] = g;
      ((long long*)a.finger_current)[(size_t)env * 88 + k] = f;
      kv.goal[s] = g; kv.nstate[s] = ns; kv.pressed[s] = pressed; kv.finger[s] = f;
      if (pressed && g == (T)0) fail = 1;
    }
  }
  if (p.traj_record) {
    T* rec = (T*)p.traj_record + (size_t)env * (a.nv + 3 + (sizeof(T) == 8 ? 2 : 3));
    for (int i = lane; i < a.nv; i += 64) rec[i] = qpos[i];
    const unsigned long long b0 = __ballot(kv.pressed[0]), b1 = __ballot(kv.pressed[1]);
    if (lane == 0) {
      unsigned* w = (unsigned*)(rec + a.nv + 3);
      w[0] = (unsigned)b0; w[1] = (unsigned)(b0 >> 32); w[2] = (unsigned)b1;
      if (sizeof(T) == 8) w[3] = 0u;
    }
  }
  const T gsus = gstate[88];
  const bool sus_on = ((const T*)p.sustain_state)[env] >= (T)p.sustain_threshold;
  kv.goal_sustain = gsus; kv.sustain_on = sus_on;
  const bool failure = __ballot(fail) != 0ull;
  // MidiEvaluationWrapper counts (goal row of the simulated step vs the new activations)
  int e_tp = 0, e_fp = 0, e_fn = 0;
  if (p.eval_sums) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const bool on = kv.goal[s] > (T)0, pr = kv.pressed[s];   // (lanes beyond key 87 hold goal 0 / not pressed)
      e_tp += __popcll(__ballot(on && pr)); e_fp += __popcll(__ballot(!on && pr)); e_fn += __popcll(__ballot(on && !pr));
    }
  }
  t += active ? 1 : 0;
  bool term = t == slen;                                       // (t_idx - 1) == len - 1
  if (lane == 0) {
    ((T*)a.goal_current)[(size_t)env * 89 + 88] = gsus;
    ((unsigned char*)a.sustain_activation)[env] = sus_on;
  }

  // ---- observables for the next step: goal look-ahead and fingering (:371-412)
  const bool live = t < slen;
  if (live) {
    const int n = (p.n_lookahead + 1) * 89;
    for (int i = lane; i < n; i += 64) {
      const int j = i / 89, c = i - 89 * j;
      const long long stp = t + j;
      const long long bi = stp < p.bank_len - 1 ? stp : p.bank_len - 1;
      const T g = stp < slen ? ((const T*)p.goal_bank)[((size_t)song * p.bank_len + bi) * 89 + c] : (T)0;
      gstate[i] = g;
    }
    const long long bi = t < p.bank_len - 1 ? t : p.bank_len - 1;
    unsigned fbits = 0;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int k = lane + 64 * s;
      if (k < RP_TASK_N_KEYS) {
        const bool gn = ((const T*)p.goal_bank)[((size_t)song * p.bank_len + bi) * 89 + k] > (T)0;
        long long f = gn ? p.finger_bank[((size_t)song * p.bank_len + bi) * 88 + k] : -1;
        bool mine = gn;
        if (a.hand_filter == 1) { mine = gn && f < 5; f = mine ? (f < 0 ? 4 : f) : -1; }        // :299-303
        else if (a.hand_filter == 2) { mine = gn && f >= 5; f = mine ? f - 5 : -1; }           // :304-306
        p.finger_next[(size_t)env * 88 + k] = f;
        if (mine) fbits |= 1u << (f < 0 ? 4 : (int)f);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) fbits |= __shfl_xor(fbits, off, 64);
    const int nfingers = a.hand_filter ? 5 : 10;
    if (lane < nfingers) ((T*)p.fingering_state)[(size_t)env * nfingers + lane] = (fbits >> lane) & 1u ? (T)1 : (T)0;
  }

  // ---- rewards, termination, discount, step type
  T reward = reward_env<T>(a, env, lane, kv);
  if (lane == 0) {
    // composer.Environment.step reads get_discount BEFORE should_terminate_episode (:213-220 then
    // sets task._discount = 0): the LAST TimeStep of a wrong press still carries the old discount
    T disc = dstate;
    if (p.wrong_press_termination) {
      if (failure && !term) dstate = (T)0;
      term = term || failure;
    }
    bool terminate = term && active;
    // physics divergence ends the episode (PhysicsError semantics); so do the engine's capacity
    // overflows selected by warn_fatal_mask (dropped contacts / cross terms: wrong physics from here on)
    const int fatal = p.warn_fatal_mask ? p.warn_fatal_mask : 1;
    const bool bad = (p.warn[env] & fatal) != 0 && active;
    if (bad && p.fatal_count) p.fatal_count[env] += 1;
    terminate = terminate || bad;
    if (terminate && p.warn_count && (p.warn[env] & p.warn_count_mask) != 0) p.warn_count[env] += 1;
    if (bad) { reward = (T)0; disc = (T)0; }
    int st = terminate ? 2 : 1;
    if (resetting) { st = 0; reward = (T)0; disc = (T)1; }
    ((T*)a.total)[env] = reward;
    ((T*)p.discount)[env] = disc;
    p.step_type[env] = st;
    if (p.traj_record) {   // (the multi-GPU gather's record: reward | discount | step_type; qpos and the bits below)
      T* rec = (T*)p.traj_record + (size_t)env * (a.nv + 3 + (sizeof(T) == 8 ? 2 : 3)) + a.nv;
      rec[0] = reward; rec[1] = disc; rec[2] = (T)st;
    }
    p.t_idx[env] = t;
    p.should_terminate[env] = t == slen;
    p.failure_termination[env] = failure;
    ((T*)p.discount_state)[env] = dstate;
    p.needs_reset[env] = terminate;
    if (p.eval_sums) {
      double* sums = p.eval_sums + (size_t)env * 6;
      double cnt = p.eval_count[env];
      if (active) {
        double v[6];
        prf(e_tp, e_fp, e_fn, v);
        const bool st_ = gsus > (T)0;
        prf(st_ && sus_on, !st_ && sus_on, st_ && !sus_on, v + 3);
        for (int i = 0; i < 6; i++) sums[i] += v[i];
        cnt += 1.0;
      }
      if (terminate) {
        const long long nf = p.eval_nfinished[env];
        double* h = p.eval_hist + ((size_t)env * p.eval_deque + (size_t)(nf % p.eval_deque)) * 6;
        const double den = cnt > 1.0 ? cnt : 1.0;
        for (int i = 0; i < 6; i++) { h[i] = sums[i] / den; sums[i] = 0.0; }
        p.eval_nfinished[env] = nf + 1;
        cnt = 0.0;
      }
      p.eval_count[env] = cnt;
    }
  }
}
// ---- goal tables of augmented songs -----------------------------------------------------------
// thread k < 88 owns piano key k of the job's slot, thread 88 the sustain column
#define RP_ONSET_BIT (1ll << 40)
template <typename T>
__global__ __launch_bounds__(128) void rp_task_raster_kernel(rp_task_raster_args a) {
  const int job = blockIdx.x, tid = threadIdx.x;
  const int song = a.job_song[job];
  const long long slot = a.job_slot[job];
  const int* kinds = a.op_kind + (size_t)job * a.max_ops;
  const double* vals = a.op_value + (size_t)job * a.max_ops;
  double total = a.total_time[song];
  for (int o = 0; o < a.max_ops; o++)
    if (kinds[o] == 1 && vals[o] != 1.0) total *= vals[o];
  const long long Tn = (long long)(total * a.fps + 1.0), nb = a.n_buffer;
  if (Tn + nb > a.bank_len) {  // uniform: the host grows the bank and retries
    if (tid == 0) a.status[job] = 1;
    return;
  }
  T* goal = (T*)a.goal_bank + (size_t)slot * a.bank_len * 89;
  long long* fing = a.finger_bank + (size_t)slot * a.bank_len * 88;
  for (long long i = tid; i < (long long)a.bank_len * 89; i += blockDim.x) goal[i] = (T)0;
  for (long long i = tid; i < (long long)a.bank_len * 88; i += blockDim.x) fing[i] = -1;
  __syncthreads();
  if (tid < 88) {
    const int k = tid;
    // rows of this song hold (part + 1) | onset flag while the notes are laid down
    for (long long t = 0; t < Tn; t++) fing[(nb + t) * 88 + k] = 0;
    for (long long i = a.note_ofs[song]; i < a.note_ofs[song + 1]; i++) {
      double st = a.note_start[i], en = a.note_end[i];
      long long pitch = a.note_pitch[i];
      bool kept = true;
      for (int o = 0; o < a.max_ops && kept; o++) {
        if (kinds[o] == 1) { if (vals[o] != 1.0) { st *= vals[o]; en *= vals[o]; } }
        else if (kinds[o] == 2) { pitch += (long long)vals[o]; kept = pitch >= 21 && pitch <= 108; }  // out of range: deleted
      }
      if (!kept || pitch - 21 != k) continue;
      const long long s0 = (long long)(st * a.fps);
      long long e0 = (long long)ceil(en * a.fps);
      if (e0 < s0 + 1) e0 = s0 + 1;                      // every note fills at least one frame
      const T on = a.note_velocity[i] != 0 ? (T)1 : (T)0;
      const long long code = (long long)a.note_part[i] + 1;
      for (long long t = s0; t < e0 && t < Tn; t++) {
        goal[(nb + t) * 89 + k] = on;                     // later notes overwrite earlier ones
        long long* f = &fing[(nb + t) * 88 + k];
        *f = code | (*f & RP_ONSET_BIT) | (t == s0 ? RP_ONSET_BIT : 0);
      }
    }
    // repeated note: a key that is held and struck again in the same frame is released for that frame
    for (long long t = Tn - 1; t >= 0; t--) {
      const long long c = fing[(nb + t) * 88 + k];
      const bool act = goal[(nb + t) * 89 + k] != (T)0;
      const bool rep = t > 0 && act && goal[(nb + t - 1) * 89 + k] != (T)0 && (c & RP_ONSET_BIT);
      const bool on = act && !rep;
      goal[(nb + t) * 89 + k] = on ? (T)1 : (T)0;
      fing[(nb + t) * 88 + k] = on ? (c & 0xffffffffll) - 1 : -1;
    }
  } else if (tid == 88) {
    for (long long i = a.cc_ofs[song]; i < a.cc_ofs[song + 1]; i++) {
      double tm = a.cc_time[i];
      for (int o = 0; o < a.max_ops; o++)
        if (kinds[o] == 1 && vals[o] != 1.0) tm *= vals[o];
      const long long fr = (long long)(tm * a.fps);
      if (fr < Tn) goal[(nb + fr) * 89 + 88] = (T)(a.cc_value[i] + 1);   // last event of a frame decides
    }
    T prev = 0;
    for (long long t = 0; t < Tn; t++) {
      const int ev = (int)goal[(nb + t) * 89 + 88];
      T sus = prev;
      if (ev >= 1 && ev <= 64) sus = 0;
      else if (ev >= 65 && ev <= 128) sus = 1;
      goal[(nb + t) * 89 + 88] = sus;
      prev = sus;
    }
  }
  if (tid == 0) { a.song_len[slot] = Tn + nb; a.status[job] = 0; }
}
// rp_task_prestep (include/rp_task.h): 16 lanes per env walk its action row.
template <typename T>
__global__ __launch_bounds__(64) void rp_task_prestep_kernel(rp_task_prestep_args a) {
  const int env = blockIdx.x * 4 + (int)(threadIdx.x >> 4), l = (int)(threadIdx.x & 15);
  if (env >= a.n_envs) return;
  const bool resetting = a.needs_reset[env] != 0;
  if (l == 0) { a.active[env] = resetting ? 0 : 1; a.reset_mask[env] = resetting ? 1 : 0; }
  // (scripted replay: the env's own row of the action table; every lane of the env reads the index before lane 0
  // stores the next one -- same wave, same instruction)
  const long long ti = a.action_table ? a.action_index[env] : 0;
  const T* act = a.action_table ? (const T*)a.action_table + (size_t)(ti < 0 ? 0 : (ti >= a.action_table_len ? a.action_table_len - 1 : ti)) * a.n_action
                                : (const T*)a.action + (size_t)env * a.n_action;
  const T *lo = (const T*)a.act_lo, *rng = (const T*)a.act_range;
  for (int i = l; i < a.n_action; i += 16) {
#pragma clang fp contract(off)   // (the wrapper's torch expression rounds after every operation)
    T v = act[i];
    if (lo) {
      if (a.clip) v = (v != v) ? v : fmin((T)1, fmax((T)-1, v));   // (torch.clamp propagates NaN: a diverged policy's action reaches ctrl and raises RP_WARN_BADSTATE, as on the torch path)
      v = lo[i] + (v + (T)1) * (T)0.5 * rng[i];   // (the wrapper's expression, term by term)
    }
    if (i == a.n_action - 1) ((T*)a.sustain_state)[env] = resetting ? (T)0 : v;
    else if (!resetting) ((T*)a.ctrl)[(size_t)env * a.nu + a.hand_act[i]] = v;
  }
  if (a.action_table && l == 0) a.action_index[env] = resetting ? 0 : (ti + 1 < a.action_table_len ? ti + 1 : a.action_table_len - 1);
}

}  // namespace

extern "C" {

const char* rp_task_last_error(void) { return g_task_err.c_str(); }

int rp_task_prestep(const rp_task_prestep_args* a, void* hip_stream) {
  if (!a) { g_task_err = "rp_task_prestep: null args"; return -1; }
  if (a->precision != 32 && a->precision != 64) { g_task_err = "rp_task_prestep: precision must be 32 or 64"; return -1; }
  if (a->n_envs <= 0 || a->n_action < 1 || a->nu < a->n_action - 1) { g_task_err = "rp_task_prestep: bad sizes"; return -1; }
  if ((!a->action && !a->action_table) || (a->action_table && (!a->action_index || a->action_table_len <= 0)) || !a->needs_reset ||
      !a->hand_act || !a->ctrl || !a->sustain_state || !a->active || !a->reset_mask || (a->act_lo && !a->act_range)) {
    g_task_err = "rp_task_prestep: null array pointer";
    return -1;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  const int nb = (a->n_envs + 3) / 4;   // (64 threads = 16 lanes per env x 4 envs)
  if (a->precision == 32) hipLaunchKernelGGL(rp_task_prestep_kernel<float>, dim3(nb), dim3(64), 0, s, *a);
  else hipLaunchKernelGGL(rp_task_prestep_kernel<double>, dim3(nb), dim3(64), 0, s, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_task_err = std::string("rp_task_prestep: ") + hipGetErrorString(e); return -2; }
  return 0;
}

int rp_task_rewards(const rp_task_reward_args* a, void* hip_stream) {
  if (!a) { g_task_err = "rp_task_rewards: null args"; return -1; }
  if (a->precision != 32 && a->precision != 64) { g_task_err = "rp_task_rewards: precision must be 32 or 64"; return -1; }
  if (a->n_envs <= 0) { g_task_err = "rp_task_rewards: n_envs must be positive"; return -1; }
  if (a->hand_filter < 0 || a->hand_filter > 2) { g_task_err = "rp_task_rewards: hand_filter must be 0, 1 or 2"; return -1; }
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

int rp_task_advance(const rp_task_advance_args* p, void* hip_stream) {
  if (!p) { g_task_err = "rp_task_advance: null args"; return -1; }
  const rp_task_reward_args* a = &p->rw;
  if (a->precision != 32 && a->precision != 64) { g_task_err = "rp_task_advance: precision must be 32 or 64"; return -1; }
  if (a->n_envs <= 0 || p->n_lookahead < 0 || p->bank_len <= 0) { g_task_err = "rp_task_advance: bad sizes"; return -1; }
  if (a->hand_filter < 0 || a->hand_filter > 2) { g_task_err = "rp_task_advance: hand_filter must be 0, 1 or 2"; return -1; }
  if (p->next_ready && !p->consumed) { g_task_err = "rp_task_advance: next_ready without consumed"; return -1; }
  if (p->eval_sums && (!p->eval_count || !p->eval_hist || !p->eval_nfinished || p->eval_deque <= 0)) {
    g_task_err = "rp_task_advance: incomplete evaluation buffers";
    return -1;
  }
  if (!a->qpos || !a->act_force || !a->act_vel || !a->site_xpos || !a->contact_geoms || !a->goal_current ||
      !a->key_norm_state || !a->key_activation || !a->sustain_activation || !a->finger_current || !a->key_qadr ||
      !a->key_anchor || !a->key_half || !a->hand_act || !a->tip_site || !a->terms || !a->total || !p->warn ||
      !p->key_qrange || !p->goal_bank || !p->finger_bank || !p->song_len || !p->song_id || !p->key_state ||
      !p->sustain_state || !p->t_idx || !p->should_terminate || !p->failure_termination || !p->discount_state ||
      !p->goal_state || !p->finger_next || !p->fingering_state || !p->needs_reset || !p->discount || !p->step_type) {
    g_task_err = "rp_task_advance: null array pointer";
    return -1;
  }
  hipStream_t s = (hipStream_t)hip_stream;
  if (a->precision == 32) hipLaunchKernelGGL(rp_task_advance_kernel<float>, dim3(a->n_envs), dim3(64), 0, s, *p);
  else hipLaunchKernelGGL(rp_task_advance_kernel<double>, dim3(a->n_envs), dim3(64), 0, s, *p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_task_err = std::string("rp_task_advance: ") + hipGetErrorString(e); return -2; }
  return 0;
}

int rp_task_rasterize(const rp_task_raster_args* a, void* hip_stream) {
  if (!a) { g_task_err = "rp_task_rasterize: null args"; return -1; }
  if (a->precision != 32 && a->precision != 64) { g_task_err = "rp_task_rasterize: precision must be 32 or 64"; return -1; }
  if (a->n_jobs < 0 || a->n_songs <= 0 || a->bank_len <= 0 || a->max_ops < 0 || a->n_buffer < 0 || !(a->fps > 0)) {
    g_task_err = "rp_task_rasterize: bad sizes";
    return -1;
  }
  if (!a->note_ofs || !a->note_start || !a->note_end || !a->note_pitch || !a->note_velocity || !a->note_part ||
      !a->cc_ofs || !a->total_time || !a->job_slot || !a->job_song || !a->goal_bank || !a->finger_bank ||
      !a->song_len || !a->status || (a->max_ops > 0 && (!a->op_kind || !a->op_value))) {
    g_task_err = "rp_task_rasterize: null array pointer";
    return -1;
  }
  if (a->n_jobs == 0) return 0;
  hipStream_t s = (hipStream_t)hip_stream;
  if (a->precision == 32) hipLaunchKernelGGL(rp_task_raster_kernel<float>, dim3(a->n_jobs), dim3(128), 0, s, *a);
  else hipLaunchKernelGGL(rp_task_raster_kernel<double>, dim3(a->n_jobs), dim3(128), 0, s, *a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_task_err = std::string("rp_task_rasterize: ") + hipGetErrorString(e); return -2; }
  return 0;
}

}  // extern "C"
