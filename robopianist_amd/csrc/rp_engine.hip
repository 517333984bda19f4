// rp_engine.hip — host side of the C ABI declared in include/rp_engine.h.
// Parses the model blob (robopianist_amd/model/compile.py:to_blob + engine
// tables), uploads the fixed-topology tables, owns the env-major state arrays and
// launches the stage kernels rp_stage_kernel<T, MODE> (rp_kernels.hpp) on the engine's HIP stream.
#include "rp_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rp_engine.h"

namespace {

// Bookkeeping of the lazy position stage (rp_set_lazy_position_stage): which envs' hand-over
// (RpStage) still matches their state.
// (lazy: skip the envs whose hand-over is still the one of their state; also: the envs rp_step_masked has just
// reset, stepped or not -- their position / velocity stage is the physics.forward() of the new episode)
__global__ void rp_lead_mask_kernel(int* lead, const int* active, const unsigned char* valid, int lazy, const unsigned char* also, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) lead[e] = (((active ? active[e] != 0 : true) && !(lazy && valid[e])) || (also && also[e])) ? 1 : 0;
}
__global__ void rp_mark_valid_kernel(unsigned char* valid, const int* active, const unsigned char* also, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n && (!active || active[e] != 0 || (also && also[e]))) valid[e] = 1;
}

// Launch order of the envs for the solver stage that follows a position stage: descending predicted
// cost (counting sort on a small integer key), one workgroup per residue class.  With one wave per env
// and every SIMD running its envs one after the other, a heavy env that starts late is the tail of
// the whole launch; the hardware dispatches workgroups in index order to whichever slot frees up, so
// sorting the heaviest first is longest-processing-time-first list scheduling.
//   * predictor: the structure of the system the solver is about to solve, read from the hand-over
//     the position stage just wrote -- contacts, touched keys, rows of the dense (cross-chain) block --
//     with a typical Newton iteration count.  Measured (profiles/r02_cost_order.md): the per-env solver
//     time is 26 k + it * (30 k + 1.2 k nd + 5 nd^2 + 2 k nk) + 2.7 k ncon cycles (R^2 0.99), and the
//     iteration count of the next solve is nearly unpredictable from the last ones (correlation 0.37),
//     so a constant does better than the previous count.
//   * XCD affinity: workgroup b runs on XCD b mod 8, and an env's state / hand-over should stay in
//     that XCD's L2, so the sort is done within each residue class: order[8 r + x] = the r-th
//     heaviest env among those with index = x (mod 8).  Inactive envs go last in their class.
#define RP_ORDER_BUCKETS 256
#define RP_ORDER_CLASSES 8
__global__ __launch_bounds__(512) void rp_order_kernel(int* order_all, const int* hdr, const int* active, int base, int n,
                                                       int* heavy_list, int* heavy_cnt, unsigned char* listed) {
  // (sorts the envs base .. base + n - 1 into order_all[base .. base + n - 1]; base is a multiple of 8.
  // One workgroup per residue class: eight short kernels side by side instead of one 1024-thread block.)
  const int x = blockIdx.x;
  // the envs outside the light capacity class, compacted for the full-capacity solver stage (any order: envs
  // are independent); the counter is cleared by that stage's last workgroup
  if (heavy_list) {
    for (int e = x + RP_ORDER_CLASSES * (int)threadIdx.x; e < n; e += RP_ORDER_CLASSES * blockDim.x) {
      const bool on = hdr[(base + e) * 8 + 6] != 1 && !(active && active[base + e] == 0);
      if (on) heavy_list[base + atomicAdd(heavy_cnt, 1)] = base + e;
      if (listed) listed[base + e] = on ? 1 : 0;   // (the snapshot the split position launches test: RpState::listed)
    }
  }
  if (!order_all) return;
  int* order = order_all + base;
  __shared__ int hist[RP_ORDER_BUCKETS];
  for (int i = threadIdx.x; i < RP_ORDER_BUCKETS; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  auto key_of = [&](int e) -> int {
    if (active && active[e] == 0) return 0;
    const int nc = hdr[e * 8], nk = hdr[e * 8 + 1];
    const int nd = __popc((unsigned)hdr[e * 8 + 2]) + __popc((unsigned)hdr[e * 8 + 3]);
    if (nc < 0 || nk < 0) return 1;   // (hand-over not written yet)
    // k cycles / 4, five Newton iterations
    const int k = 1 + (26 + 5 * (30 + nd + (nd >> 2) + nd * nd / 200 + 2 * nk) + 3 * nc) / 4;
    return k < RP_ORDER_BUCKETS ? k : RP_ORDER_BUCKETS - 1;
  };
  // (a thread's first env keeps its key in a register between the two passes)
  const int e_first = x + RP_ORDER_CLASSES * (int)threadIdx.x;
  const int k_first = e_first < n ? key_of(base + e_first) : 0;
  for (int e = e_first; e < n; e += RP_ORDER_CLASSES * blockDim.x) atomicAdd(&hist[e == e_first ? k_first : key_of(base + e)], 1);
  __syncthreads();
  // descending exclusive prefix: lane l of the first wave owns the four buckets 255 - 4 l ... 252 - 4 l
  static_assert(RP_ORDER_BUCKETS == 256, "scan layout");
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    int v[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = hist[RP_ORDER_BUCKETS - 1 - 4 * l - k]; tot += v[k]; }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (l >= d) incl += t; }
    int acc = incl - tot;
#pragma unroll
    for (int k = 0; k < 4; k++) { hist[RP_ORDER_BUCKETS - 1 - 4 * l - k] = acc; acc += v[k]; }
  }
  __syncthreads();
  for (int e = e_first; e < n; e += RP_ORDER_CLASSES * blockDim.x) {
    const int r = atomicAdd(&hist[e == e_first ? k_first : key_of(base + e)], 1);
    order[RP_ORDER_CLASSES * r + x] = base + e;
  }
}

thread_local std::string g_err;
int fail(const std::string& s) { g_err = s; return -1; }
#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess)                                                          \
      return fail(std::string(#x) + ": " + hipGetErrorString(e_));                 \
  } while (0)

struct BlobEntry { char name[40]; int32_t dtype, ndim; int64_t count, offset; };

struct Blob {
  std::vector<unsigned char> data;
  int n = 0;
  const BlobEntry* ent = nullptr;
  bool init(const void* p, size_t nb) {
    data.assign((const unsigned char*)p, (const unsigned char*)p + nb);
    if (nb < 12) return false;
    const uint32_t* h = (const uint32_t*)data.data();
    if (h[0] != 0x52504D42u) return false;
    n = (int)h[2];
    ent = (const BlobEntry*)(data.data() + 12);
    return true;
  }
  const BlobEntry* find(const char* name) const {
    for (int i = 0; i < n; i++) if (!strncmp(ent[i].name, name, 40)) return &ent[i];
    return nullptr;
  }
  bool has(const char* name) const { return find(name) != nullptr; }
  std::vector<double> f(const char* name) const {
    const BlobEntry* e = find(name);
    if (!e) throw std::string("blob entry missing: ") + name;
    std::vector<double> v((size_t)e->count);
    if (e->dtype == 0) memcpy(v.data(), data.data() + e->offset, sizeof(double) * e->count);
    else { const int32_t* s = (const int32_t*)(data.data() + e->offset); for (int64_t i = 0; i < e->count; i++) v[i] = s[i]; }
    return v;
  }
  std::vector<int> i(const char* name) const {
    const BlobEntry* e = find(name);
    if (!e) throw std::string("blob entry missing: ") + name;
    if (e->dtype != 1) throw std::string("blob entry not int: ") + name;
    std::vector<int> v((size_t)e->count);
    memcpy(v.data(), data.data() + e->offset, sizeof(int32_t) * e->count);
    return v;
  }
  int i1(const char* name) const { return i(name).at(0); }
  double f1(const char* name) const { return f(name).at(0); }
};

// The internal streams (slices 1.., companion streams) are PROCESS-WIDE, one set per device, shared by every engine and
// never destroyed.  HIP maps a stream to one of the device's hardware queues (four) when it is created; streams that
// share a queue serialise.  With a set of streams per ENGINE, the second engine of a process -- bench.py's auxiliary
// legs, a training script's evaluation env -- got streams that shared queues with each other: its two- and three-slice
// schedules measured 7.2-9.1 ms per step where the first engine's took 5.9-6.1 (round 5).  Engines that share streams
// only order their launches among each other, which independent engines never relied on.
struct StreamPool {
  static const int kSlots = 8;   // [0 .. 3] slice streams (slot 0 unused), [4 .. 7] companion streams
  hipStream_t s[kSlots] = {};
};
static std::mutex g_pool_mutex;
static std::map<int, StreamPool> g_pools;
static hipStream_t pooled_stream(int device, int slot, bool high_priority) {
  // Four hardware queues, one of them the caller's stream: the schedules the engine picks from use {caller, slice 1,
  // companion 0, companion 1} (two slices) or {caller, slice 1, slice 2} (three slices, no companions) -- never both
  // sets at once.  Slice 2 therefore IS companion 0's stream and slice 3 companion 1's: four streams in all, so that the
  // five a process would otherwise own after the schedule trials do not share a queue.
  if (slot == 2) slot = 4;
  if (slot == 3) slot = 5;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  StreamPool& p = g_pools[device];
  if (!p.s[slot]) {
    int least = 0, greatest = 0;
    hipError_t e;
    if (high_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
      e = hipStreamCreateWithPriority(&p.s[slot], hipStreamNonBlocking, greatest);
    else { (void)hipGetLastError(); e = hipStreamCreateWithFlags(&p.s[slot], hipStreamNonBlocking); }
    if (e != hipSuccess) { (void)hipGetLastError(); p.s[slot] = nullptr; }
  }
  return p.s[slot];
}

struct EngineBase {
  virtual ~EngineBase() {}
  int nenv = 0, device = 0, precision = 32;
  int rm_rows = 0, rm_cols = 0;   // shape of RP_DEBUG_MASS_ROWS per env (the kernel build in use)
  int nv = 0, nu = 0, nsite = 0, ntree = 0, nkey = 0, nlink = 0, maxdepth = 0;
  hipStream_t stream = nullptr;
  // ring of HIP event pairs bracketing every step-kernel launch on `stream`
  static const int kRing = 128;
  hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};
  bool ev_pending[kRing] = {};
  int ev_next = 0;
  double kernel_ms = 0; int kernel_launches = 0;
  // second ring: one solver-kernel (mj_step2) launch per rp_step call; the probed substep rotates
  // with the call count, so the running average is the mean over all substeps by construction
  hipEvent_t sv0[kRing] = {}, sv1[kRing] = {};
  unsigned step_calls = 0;
  // (kept per schedule kind -- [0] one launch per stage, [1] fused substeps -- and reported for the kind with more samples)
  double solver_ms_k[2] = {0, 0}, solver_envs_k[2] = {0, 0}; int solver_launches_k[2] = {0, 0};
  int sv_kind[kRing] = {};
  int sv_envs[kRing] = {};            // envs the probed solver launch of that slot covered (a slice or the batch)
  double last_solver_envs = 0;
  int last_solver_kind = 0;
  // schedule of the last rp_step when the engine chooses (n_slices == 0): see step().  Rule-based since round 6.
  int auto_mode = 1;
  bool many_heavy = false;   // the lists of envs outside the light class have been long lately (hysteresis)
  int last_nsl = 1;          // slices of the last rp_step
  void harvest(int i, bool wait) {
    if (!ev_pending[i]) return;
    if (wait) hipEventSynchronize(ev1[i]);
    else if (hipEventQuery(ev1[i]) != hipSuccess) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev0[i], ev1[i]) == hipSuccess) {
      kernel_ms += ms; kernel_launches++;
    }
    if (sv_envs[i] > 0 && hipEventElapsedTime(&ms, sv0[i], sv1[i]) == hipSuccess) {
      const int kd = sv_kind[i] & 1;
      solver_ms_k[kd] += ms; solver_launches_k[kd]++; solver_envs_k[kd] += sv_envs[i];
    }
    sv_envs[i] = 0;   // (a step without a probe leaves 0)
    ev_pending[i] = false;
  }
  virtual int reset(const uint8_t* mask) = 0;
  virtual int set(rp_field f, const void* src) = 0;
  virtual int get(rp_field f, void* dst) = 0;
  virtual int step(int nsub, uint32_t* trace, int mode, const uint8_t* reset_mask = nullptr) = 0;
  virtual void limits(int newton, int ls) = 0;
  virtual void tolerances(double tol, double ls_tol) = 0;
  virtual void mpr_tolerances(double tol, double poly_tol) = 0;
  bool lazy_position = false;
  bool legacy_step = true;   // rp_set_legacy_step: false = dm_control's legacy_step=False output semantics
  bool cost_order = false;   // rp_set_cost_ordered_launch
  int n_slices = 1;          // rp_set_stream_slices (0 = automatic)
  static const int kMaxSlices = 4;
  hipStream_t xstream[kMaxSlices] = {};   // [0] unused: slice 0 runs on the caller's stream
  hipEvent_t ev_fork = nullptr, ev_join[kMaxSlices] = {};
  // capacity classes: the full-capacity solver stage (few envs, one wave per SIMD, long waves) runs beside the
  // lean one on a companion stream of its slice, forked / joined with events every substep -- in one stream the
  // two launches serialise and every substep pays the slowest heavy env on an otherwise idle GPU
  hipStream_t hstream[kMaxSlices] = {};
  hipEvent_t ev_hfork[kMaxSlices] = {}, ev_hjoin[kMaxSlices] = {};
  virtual int acc_sensors(int on) = 0;
  virtual int lean_solver(int on) = 0;
  virtual int fused_substeps(int on) = 0;
  virtual int fused_substeps_on() const = 0;
  virtual int split_position(int on) = 0;
  virtual int split_position_on() const = 0;
  virtual int field_ptr(rp_field f, void** p, size_t* bytes) = 0;
  bool own_stream = true;
  virtual int profile(long long* out, int n, int enable) = 0;
};

template <typename T>
struct Engine : EngineBase {
  RpModel<T> M{};
  RpState<T> S{};
  RpStage<T> B{};
  std::vector<void*> allocs;
  std::vector<T> qpos0;
  T* d_qpos0 = nullptr;
  uint32_t* d_trace = nullptr; size_t trace_cap = 0;
  uint8_t* d_mask = nullptr;
  bool trunk4 = false;  // every tree has a 4-link trunk: launch the specialised solver build
  bool mesh = false;    // the model has convex-hull geoms: position / sensor stages with MPR
  bool graph = false;   // ... some of them with a vertex graph (more than 32 vertices): the MESH = 2 builds
  bool deep = false;    // a trunk of 5..8 links (more than two forearm dofs): the RPK_MAXD_DEEP builds
  bool lean = false;    // light envs are stepped by rp_lean_solver_kernel (rp_solver2.hpp), the others by the full build
  // the position stage of the substeps as three launches: front part, pooled narrow phase (rp_collide.hpp), back part
  // split_mode: 0 = never, 1 = in every per-stage schedule, 2 = automatic: the schedule "three slices, split stage, no
  // companion streams" is a candidate of the engine's own choice (with rp_set_stream_slices(e, 0))
  bool split_capable = false;
  int split_mode = 0;
  bool split_now = false;   // (this rp_step)
  bool split_dropped = false;   // the split stage has overflowed its candidate / record lists once: the rule no longer picks it
  int split_position(int on) override {
    if (on && !split_capable) return fail("rp_set_split_position_stage: the split stage exists for the fp64 default-depth builds only");
    split_mode = on < 0 ? 2 : (on > 2 ? 2 : on);
    return 0;
  }
  // (0 = off, 1 = in use, 2 = the rule may pick it: batches of >= 3072 envs that have never overflowed its lists)
  int split_position_on() const override {
    return split_mode == 1 ? 1 : (split_mode == 2 && n_slices == 0 && nenv >= 3072 && !split_dropped ? (auto_mode == 4 ? 1 : 2) : 0);
  }
  int lean_solver(int on) override {
    if (on && (deep || sizeof(T) != 8)) return fail("rp_set_lean_solver: the lean solver stage exists for the fp64 default builds only");
    lean = on != 0; S.lean = on > 0 ? on : 0;   // (on > 1: the light class capped at that many Jacobian entries)
    return 0;
  }
  int md() const { return deep ? RPK_MAXD_DEEP : RPK_MAXD; }

  ~Engine() override {
    hipSetDevice(device);
    for (void* p : allocs) hipFree(p);
    if (d_trace) hipFree(d_trace);
    if (d_mask) hipFree(d_mask);
    if (h_heavy_peak) hipHostFree(h_heavy_peak);
    for (int i = 0; i < kRing; i++) {
      if (ev0[i]) hipEventDestroy(ev0[i]);
      if (ev1[i]) hipEventDestroy(ev1[i]);
      if (sv0[i]) hipEventDestroy(sv0[i]);
      if (sv1[i]) hipEventDestroy(sv1[i]);
    }
    if (stream && own_stream) hipStreamDestroy(stream);
    for (int i = 1; i < kMaxSlices; i++) {
      // (xstream / hstream: the process-wide pool's, never destroyed)
      if (ev_join[i]) hipEventDestroy(ev_join[i]);
    }
    for (int i = 0; i < kMaxSlices; i++) {
      if (ev_hfork[i]) hipEventDestroy(ev_hfork[i]);
      if (ev_hjoin[i]) hipEventDestroy(ev_hjoin[i]);
    }
    if (ev_fork) hipEventDestroy(ev_fork);
  }
  template <typename U> U* dalloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(U)) != hipSuccess) throw std::string("hipMalloc failed");
    hipMemset(p, 0, (n ? n : 1) * sizeof(U));
    allocs.push_back(p);
    return (U*)p;
  }
  const T* upF(const std::vector<double>& v) {
    std::vector<T> t(v.begin(), v.end());
    T* d = dalloc<T>(t.size());
    if (!t.empty()) hipMemcpy(d, t.data(), t.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
  }
  const int* upI(const std::vector<int>& v) {
    int* d = dalloc<int>(v.size());
    if (!v.empty()) hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    return d;
  }

  void build(const Blob& b, int n_envs) {
    nenv = n_envs;
    M.nlink = nlink = b.i1("eng_nlink"); M.ntree = ntree = b.i1("eng_ntree");
    M.maxdepth = maxdepth = b.i1("eng_maxdepth"); M.nkey = nkey = b.i1("eng_nkey");
    M.ngeom = b.i1("eng_ngeom");
    M.nu = nu = b.i1("eng_nu"); M.nsite = nsite = b.i1("eng_nsite"); M.nv = nv = b.i1("nv");
    if (M.nlink > RPK_NL_DEEP) throw std::string("too many hand dofs for the engine (max 60)");
    if (M.nkey > RPK_NKEYS) throw std::string("too many keys (max 128)");
    if (M.ngeom > RPK_WAVE) throw std::string("too many collision geoms (max 64)");
    if (M.nu > RPK_MAXACT) throw std::string("too many actuators");
    if (M.nsite > RPK_WAVE) throw std::string("too many sites (max 64)");
    if (M.maxdepth > RPK_MAXD_DEEP) throw std::string("tree too deep (max 13 levels: 8 trunk links + 5 finger links)");
    {
      std::vector<int> kind = b.i("eng_act_kind");
      for (int a = 0; a < nu; a++)
        if (kind[a] == 0 && a >= RPK_WAVE) throw std::string("hand actuator index >= 64");
    }
    M.iterations = b.i1("opt_iterations"); M.ls_iterations = b.i1("opt_ls_iterations");
    M.timestep = (T)b.f1("opt_timestep");
    auto g = b.f("opt_gravity");
    M.gx = (T)g[0]; M.gy = (T)g[1]; M.gz = (T)g[2];
    M.tolerance = (T)b.f1("opt_tolerance"); M.ls_tolerance = (T)b.f1("opt_ls_tolerance");
    // single precision cannot resolve MuJoCo's default 1e-8: iterating below 1e-6 only chases
    // rounding noise (measured: same accuracy against the fp64 engine, 5 % fewer iterations);
    // rp_set_solver_tolerance overrides this
    if (sizeof(T) == 4 && M.tolerance < (T)1e-6) M.tolerance = (T)1e-6;
    M.meaninertia = (T)b.f1("stat_meaninertia");
    // opt.impratio [MJ: mj_makeImpedance]: friction dimensions regularised by R / impratio
    {
      const double ir = b.has("opt_impratio") ? b.f1("opt_impratio") : 1.0;
      M.mu_scale = (T)std::sqrt(1.0 / (ir > 1e-15 ? ir : 1e-15));
    }
    M.mpr_tol = (T)1e-6; M.mpr_tol_poly = (T)1e-6;   // MuJoCo's uniform rule (rp_set_mpr_tolerance)
    // ---- pack every model table into the two device arrays (RpLayout offsets)
    std::vector<double> ft((size_t)RpLayout::F_TOTAL, 0.0);
    std::vector<int> it((size_t)RpLayout::I_TOTAL, 0);
    auto putF = [&](int off, int cap, const std::vector<double>& v, const char* name) {
      if ((int)v.size() > cap) throw std::string("model table too large: ") + name;
      for (size_t i = 0; i < v.size(); i++) ft[(size_t)off + i] = v[i];
    };
    auto putI = [&](int off, int cap, const std::vector<int>& v, const char* name) {
      if ((int)v.size() > cap) throw std::string("model table too large: ") + name;
      for (size_t i = 0; i < v.size(); i++) it[(size_t)off + i] = v[i];
    };
#define PF(name, blobname) putF(RpLayout::F_##name, RpLayout::F_##name##_end - RpLayout::F_##name + 1, b.f(blobname), blobname)
#define PI(name, blobname) putI(RpLayout::I_##name, RpLayout::I_##name##_end - RpLayout::I_##name + 1, b.i(blobname), blobname)
    PI(lane_topo, "eng_lane_topo");
    PI(link_parent, "eng_link_parent"); PI(link_depth, "eng_link_depth"); PI(link_tree, "eng_link_tree");
    PI(link_jtype, "eng_link_jtype"); PI(link_dof, "eng_link_dof"); PI(link_sibrank, "eng_link_sibrank");
    PI(level_maxrank, "eng_level_maxrank"); PI(link_anc, "eng_link_anc"); PI(link_limited, "eng_link_limited");
    PI(link_act, "eng_link_act"); PI(link_desc, "eng_link_desc"); PI(link_ndesc, "eng_link_ndesc");
    PI(tree_base, "eng_tree_base"); PI(tree_trunk, "eng_tree_trunk"); PI(chain_first, "eng_chain_first");
    PI(chain_len, "eng_chain_len"); PI(link_ancmask, "eng_link_ancmask");
    {
      auto tt = b.i("eng_tree_trunk");
      trunk4 = M.ntree > 0;
      deep = M.maxdepth > RPK_MAXD || M.nlink > RPK_NL;
      for (int t = 0; t < M.ntree && t < (int)tt.size(); t++) {
        if (tt[t] > 8 || tt[t] < 1) throw std::string("the solver needs a trunk chain of 1..8 links per articulated tree");
        if (tt[t] != 4) trunk4 = false;
        if (tt[t] > 4) deep = true;   // (the register-blocked trunk of the default build holds 4 links)
      }
      if (getenv("RP_FORCE_DEEP")) { deep = true; trunk4 = false; }   // test hook: deep builds on any scene
    }
    for (int v : b.i("eng_chain_len")) if (v > 5) throw std::string("finger chain longer than 5 links is not supported by the solver");
    PF(link_lpos, "eng_link_lpos");
    {
      auto q = b.f("eng_link_lquat");
      std::vector<double> m(9 * (size_t)M.nlink);
      for (int i = 0; i < M.nlink; i++) {
        double w = q[4 * i], x = q[4 * i + 1], y = q[4 * i + 2], z = q[4 * i + 3];
        double* o = &m[9 * i];
        o[0] = 1 - 2 * (y * y + z * z); o[1] = 2 * (x * y - w * z); o[2] = 2 * (x * z + w * y);
        o[3] = 2 * (x * y + w * z); o[4] = 1 - 2 * (x * x + z * z); o[5] = 2 * (y * z - w * x);
        o[6] = 2 * (x * z - w * y); o[7] = 2 * (y * z + w * x); o[8] = 1 - 2 * (x * x + y * y);
      }
      putF(RpLayout::F_link_lmat, RPK_NL_DEEP * 9, m, "link_lmat");
    }
    PF(link_axis, "eng_link_axis"); PF(link_anchor, "eng_link_anchor"); PF(link_mass, "eng_link_mass");
    PF(link_ipos, "eng_link_ipos"); PF(link_inertia, "eng_link_inertia"); PF(link_invw_body, "eng_link_invw_body");
    PF(link_armature, "eng_link_armature"); PF(link_damping, "eng_link_damping");
    PF(link_stiffness, "eng_link_stiffness"); PF(link_springref, "eng_link_springref");
    PF(link_floss, "eng_link_floss"); PF(link_fl_R, "eng_link_fl_R"); PF(link_fl_B, "eng_link_fl_B");
    PF(link_range, "eng_link_range"); PF(link_lim_K, "eng_link_lim_K"); PF(link_lim_B, "eng_link_lim_B");
    PF(link_lim_solimp, "eng_link_lim_solimp"); PF(link_invw_dof, "eng_link_invw_dof");
    PF(link_act_coef, "eng_link_act_coef"); PF(link_gscale, "eng_link_gscale"); PF(tree_gscale, "eng_tree_gscale"); PF(tree_ref, "eng_tree_ref");
    PI(key_dof, "eng_key_dof"); PI(key_act, "eng_key_act"); PI(key_geomid, "eng_key_geomid");
    auto kpos = b.f("eng_key_pos"), khalf = b.f("eng_key_half");
    PF(key_pos, "eng_key_pos"); PF(key_half, "eng_key_half");
    {
      std::vector<double> rb((size_t)M.nkey);
      double zmax = -1e30;
      for (int k = 0; k < M.nkey; k++) {
        rb[k] = std::sqrt(khalf[3 * k] * khalf[3 * k] + khalf[3 * k + 1] * khalf[3 * k + 1] +
                          khalf[3 * k + 2] * khalf[3 * k + 2]);
        // the key box centre moves on a circle of radius hx about the hinge
        zmax = std::max(zmax, kpos[3 * k + 2] + khalf[3 * k] + rb[k]);
      }
      putF(RpLayout::F_key_rbound, RPK_NKEYS, rb, "key_rbound");
      M.key_zmax = (T)zmax;
    }
    PF(key_mass, "eng_key_mass"); PF(key_M, "eng_key_M"); PF(key_stiffness, "eng_key_stiffness");
    PF(key_springref, "eng_key_springref"); PF(key_damping, "eng_key_damping"); PF(key_range, "eng_key_range");
    PF(key_lim_K, "eng_key_lim_K"); PF(key_lim_B, "eng_key_lim_B"); PF(key_lim_solimp, "eng_key_lim_solimp");
    PF(key_invw_dof, "eng_key_invw_dof"); PF(key_invw_body, "eng_key_invw_body"); PF(key_cparam, "eng_key_cparam");
    PI(geom_link, "eng_geom_link"); PI(geom_type, "eng_geom_type"); PI(geom_modelid, "eng_geom_modelid");
    PI(geom_pairmask, "eng_geom_pairmask"); PI(geom_iskeycap, "eng_geom_iskeycap");
    PF(geom_size, "eng_geom_size"); PF(geom_pos, "eng_geom_pos"); PF(geom_mat, "eng_geom_mat");
    PF(geom_rbound, "eng_geom_rbound"); PF(geom_invw, "eng_geom_invw"); PF(geom_cparam, "eng_geom_cparam");
    PF(geom_bcap, "eng_geom_bcap");
    PI(act_kind, "eng_act_kind"); PI(act_lane, "eng_act_lane"); PI(act_ctrllimited, "eng_act_ctrllimited");
    PI(act_forcelimited, "eng_act_forcelimited");
    PF(act_coef, "eng_act_coef"); PF(act_gain, "eng_act_gain"); PF(act_bias, "eng_act_bias");
    PF(act_ctrlrange, "eng_act_ctrlrange"); PF(act_forcerange, "eng_act_forcerange");
    PI(site_link, "eng_site_link"); PF(site_pos, "eng_site_pos");
    if (b.has("eng_site_touch_radius")) PF(site_touch_radius, "eng_site_touch_radius");
    if (b.has("eng_link_bodylink")) PI(link_bodylink, "eng_link_bodylink");
    if (b.has("eng_mesh_vert")) {
      PF(mesh_vert, "eng_mesh_vert"); PI(geom_vertadr, "eng_geom_vertadr"); PI(geom_vertnum, "eng_geom_vertnum");
      if (b.has("eng_geom_vertflip")) PI(geom_vertflip, "eng_geom_vertflip");
      if (b.has("eng_geom_vertgraph")) PI(geom_vertgraph, "eng_geom_vertgraph");
      for (int t : b.i("eng_geom_type")) if (t == GEOM_MESH_) mesh = true;
    }
    // cylinders collide through the portal refinement like hulls; their support function is compiled into the MESH = 2
    // builds only (the builds of scenes with the real hand's colliders: graph hulls, cylinders), so that the benchmark's
    // kernels stay what they were
    for (int t : b.i("eng_geom_type")) if (t == GEOM_CYL_) { mesh = true; graph = true; }
#undef PF
#undef PI
    M.ft = upF(ft);
    M.it = upI(it);
    if (b.has("eng_hull_vert") && !b.f("eng_hull_vert").empty()) {
      if (b.i("eng_hull_graph").size() != b.f("eng_hull_vert").size() / 3 * RPK_HULL_GRAPH_ROW) throw std::string("eng_hull_graph does not match eng_hull_vert");
      M.hull_vert = upF(b.f("eng_hull_vert"));
      M.hull_graph = upI(b.i("eng_hull_graph"));
      graph = true;
    }
    {
      auto q0 = b.f("qpos0");
      qpos0.assign(q0.begin(), q0.end());
      d_qpos0 = const_cast<T*>(upF(q0));
    }
    size_t E = (size_t)nenv;
    S.nenv = nenv;
    S.qpos = dalloc<T>(E * nv); S.qvel = dalloc<T>(E * nv); S.warm = dalloc<T>(E * nv);
    S.ctrl = dalloc<T>(E * nu); S.qfrc_applied = dalloc<T>(E * nv); S.time = dalloc<T>(E);
    S.tree_offset = dalloc<T>(E * (ntree ? ntree : 1) * 3);
    S.act_force = dalloc<T>(E * nu); S.act_vel = dalloc<T>(E * nu);
    S.site_xpos = dalloc<T>(E * (nsite ? nsite : 1) * 3);
    S.contact_dist = dalloc<T>(E * RPK_NCOUT);
    S.ncon = dalloc<int>(E); S.contact_geoms = dalloc<int>(E * RPK_NCOUT * 2);
    S.warn = dalloc<int>(E); S.solver_iter = dalloc<int>(E);
    B.RM = dalloc<T>(E * RPK_NLX(md()) * (md() + 1));
    rm_rows = RPK_NLX(md()); rm_cols = md() + 1;
    B.lanef = dalloc<T>(E * RPK_NLF * 64);
    B.lanei = dalloc<int>(E * RPK_NLI * 64);
    B.hdr = dalloc<int>(E * 8);
    B.entJ = dalloc<T>(E * RpCaps<T>::NE * 3);
    B.entM = dalloc<int>(E * RpCaps<T>::NE * 2);
    B.slots = dalloc<int>(E * 64);
    B.keyslot = dalloc<int>(E * (RPK_NKEYS / 4));
    B.covf = dalloc<T>(E * (RPK_NC - RPK_NCL) * 12);
    B.covi = dalloc<int>(E * (RPK_NC - RPK_NCL) * 4);
    // split position stage (front part -> pooled narrow phase -> back part): the fp64 default-depth builds
    {
      const char* sp = getenv("RP_SPLIT_POS");
      split_capable = sizeof(T) == 8 && !deep;
      split_mode = !split_capable ? 0 : (sp ? (sp[0] == '0' ? 0 : (sp[0] == '1' ? 1 : 2)) : 2);
      // (the buffers -- 78 KB per env: 320 MB at 4096 envs -- are allocated by the first rp_step that runs the split stage:
      // ensure_split_buffers; an engine that never does, small batches and RP_SPLIT_POS=0 among them, never pays for them)
    }
    // hand-over buffers start as NaN / -1 patterns: a read of anything the position kernel
    // did not write this substep shows up as a bad state instead of silently reusing old data
    hipMemset(B.RM, 0xFF, sizeof(T) * E * RPK_NLX(md()) * (md() + 1));
    hipMemset(B.lanef, 0xFF, sizeof(T) * E * RPK_NLF * 64);
    hipMemset(B.lanei, 0xFF, sizeof(int) * E * RPK_NLI * 64);
    hipMemset(B.hdr, 0xFF, sizeof(int) * E * 8);
    hipMemset(B.entJ, 0xFF, sizeof(T) * E * RpCaps<T>::NE * 3);
    hipMemset(B.entM, 0xFF, sizeof(int) * E * RpCaps<T>::NE * 2);
    hipMemset(B.slots, 0xFF, sizeof(int) * E * 64);
    hipMemset(B.keyslot, 0xFF, sizeof(int) * E * (RPK_NKEYS / 4));
    S.key_trace = nullptr;
    S.prof = nullptr;
    d_active = dalloc<int>(E);
    S.active = nullptr;
    d_lead = dalloc<int>(E);
    d_order = dalloc<int>(E);
    {
      std::vector<int> id(E);
      for (size_t i = 0; i < E; i++) id[i] = (int)i;
      hipMemcpy(d_order, id.data(), E * sizeof(int), hipMemcpyHostToDevice);
    }
    S.cost_pos = dalloc<int>(E); S.cost_sol = dalloc<int>(E);
    d_heavy = dalloc<int>(E); d_heavy_cnt = dalloc<int>(2 * kMaxSlices);   // (zero-filled)
    d_listed = dalloc<unsigned char>(E); S.listed = d_listed;
    // ([kMaxSlices]: candidates the split position stage dropped this step, RP_WARN_SPLIT_FULL -- read back with the peaks)
    d_heavy_peak = dalloc<int>(kMaxSlices + 1);
    if (hipHostMalloc((void**)&h_heavy_peak, sizeof(int) * (kMaxSlices + 1)) != hipSuccess) { (void)hipGetLastError(); h_heavy_peak = nullptr; }
    else { for (int i = 0; i < kMaxSlices; i++) h_heavy_peak[i] = -1; h_heavy_peak[kMaxSlices] = 0; }
    S.heavy_list = nullptr; S.heavy_cnt = nullptr; S.heavy_done = nullptr;
    S.qpos_prev = nullptr; S.qvel_prev = nullptr;
    d_valid = dalloc<unsigned char>(E);  // (zero-filled: nothing is valid yet)
    S.max_newton = M.iterations; S.max_ls = M.ls_iterations;
    {
      const char* le = getenv("RP_LEAN");
      lean = sizeof(T) == 8 && !deep && !(le && le[0] == '0');
      S.lean = lean ? 1 : 0;
      if (lean && le && atoi(le) > 1) S.lean = atoi(le);   // (RP_LEAN=n > 1: the light class capped at n Jacobian entries, as rp_set_lean_solver(e, n))
    }
    // The fills and uploads above went through the null stream, which is NOT ordered with the
    // engine's non-blocking stream: everything must have landed before the first kernel.
    if (hipDeviceSynchronize() != hipSuccess) throw std::string("hipDeviceSynchronize failed after model upload");
  }

  // The split position stage's buffers, on first use.  False (and the split stage off for good) when the device has no
  // room for them: the one-kernel stage computes the same bits.
  bool split_alloc_failed = false;
  bool ensure_split_buffers() {
    if (B.frames) return true;
    if (split_alloc_failed) return false;
    const size_t E = (size_t)nenv;
    const size_t mark = allocs.size();
    try {
      T* frames = dalloc<T>(E * RPK_NFRAME * 64);
      B.cand = dalloc<int>(E * RPK_NCAND * 2);
      B.ncand = dalloc<int>(E);
      B.cres = dalloc<T>(E * RPK_NRES * 12);
      B.cres_n = dalloc<int>(E * RPK_NCAND);
      B.tstride = ((E + RPK_NSTRIPE - 1) / RPK_NSTRIPE + 1) * RPK_NCAND;   // (+1: a slice's stripe may hold one env more than cnt / 8)
      B.tlist = dalloc<int>((size_t)RPK_NTYPE * RPK_NSTRIPE * B.tstride * 4);
      B.tcount = dalloc<int>(kMaxSlices * RPK_NSTRIPE * RPK_NTYPE_PAD);   // (zero-filled)
      B.tcount_off = 0;
      hipMemset(B.ncand, 0xFF, sizeof(int) * E);        // -1: no front part has run
      hipMemset(frames, 0xFF, sizeof(T) * E * RPK_NFRAME * 64);
      B.frames = frames;
      // (null-stream fills vs the engine's non-blocking streams, as at the end of build())
      if (hipDeviceSynchronize() != hipSuccess) throw std::string("hipDeviceSynchronize failed");
    } catch (const std::string&) {
      (void)hipGetLastError();
      while (allocs.size() > mark) { hipFree(allocs.back()); allocs.pop_back(); }
      B.frames = nullptr; B.cand = nullptr; B.ncand = nullptr; B.cres = nullptr; B.cres_n = nullptr; B.tlist = nullptr; B.tcount = nullptr;
      split_alloc_failed = true; split_mode = 0;
      return false;
    }
    return true;
  }
  long long* d_prof = nullptr;
  int* d_active = nullptr;
  int* d_order = nullptr;           // cost-ordered launch: workgroup -> env
  // the envs outside the light capacity class, compacted per slice (entries base .. of slice sl; d_heavy_cnt[2 sl]
  // = entries, [2 sl + 1] = finished workgroups of the stage that walks them)
  int *d_heavy = nullptr, *d_heavy_cnt = nullptr;
  unsigned char* d_listed = nullptr;   // 1 = on the list of the substep just solved (rp_order_kernel's snapshot)
  // Grid of the full-capacity solver stage.  Each of its workgroups needs a whole idle SIMD (512 registers) and 55 KB of
  // LDS even to find the list empty, and the slice's join waits for the last of them: the grid follows the longest
  // list the stage saw recently -- read back one step late, never waited for (a list longer than the grid is walked
  // in rounds: slower for a step, never wrong).  Measured with the 55 KB stage, config 2 hull: fixed 128: 618 k
  // env-steps/s, following the list: 651 k (the 40 KB stage of round 3 at 128: 650 k).
  int *d_heavy_peak = nullptr, *h_heavy_peak = nullptr;
  double heavy_est[kMaxSlices] = {8, 8, 8, 8};
  int heavy_grid_for(int sl, int cnt) {
    int g = kHeavyGrid;
    if (!heavy_grid_fixed && h_heavy_peak) {
      const int seen = *(volatile int*)&h_heavy_peak[sl];
      if (seen >= 0) { heavy_est[sl] = seen > heavy_est[sl] ? seen : 0.9 * heavy_est[sl] + 0.1 * seen; *(volatile int*)&h_heavy_peak[sl] = -1; }
      // (config 3, 4096 envs: fixed grids of 24 / 48 / 128: 409 / 436 / 445 k env-steps/s.  Round 6: the floor is 2
      // workgroups, not 8 -- each needs a whole idle SIMD, and on the replay, whose lists are empty most of the time, a
      // grid of one measured +0.7 %: 636.6 against 632.2 k, two runs each inside one call)
      g = (int)(2.0 * heavy_est[sl]) + 2;
      g = g < 2 ? 2 : (g > kHeavyGrid ? kHeavyGrid : g);
    }
    return cnt < g ? cnt : g;
  }
  // fused substeps (rp_fused_steps_kernel): one launch takes every light env through all substeps of an rp_step
  // 0 = off, 1 = on, 2 = automatic (a candidate of the schedule choice when the slice count is automatic too)
  int fused = getenv("RP_FUSED") ? atoi(getenv("RP_FUSED")) : 2;
  const bool fused_split = !(getenv("RP_FUSED_SPLIT") && getenv("RP_FUSED_SPLIT")[0] == '0');
  int fused_substeps(int on) override { fused = on < 0 ? 2 : (on > 2 ? 2 : on); return 0; }
  int fused_substeps_on() const override {
    const bool capable = lean && !deep && !graph && sizeof(T) == 8;
    return !capable ? 0 : (fused == 1 ? 1 : (fused == 2 && n_slices == 0 ? (auto_mode == 3 ? 1 : 2) : 0));
  }
  // MEASUREMENT-ONLY switches (they skip or repeat work: wrong physics / wasted time).  Compiled in only with
  // -DRP_EXPERIMENTS, and loud when set: a stray environment variable must not silently change a production step.
#ifdef RP_EXPERIMENTS
  static bool x_switch(const char* name) {
    const char* v = getenv(name);
    const bool on = v && v[0] == '1';
    if (on) fprintf(stderr, "librp_engine: MEASUREMENT-ONLY switch %s=1 is active -- results are not valid physics\n", name);
    return on;
  }
  const bool x_no_heavy = x_switch("RP_X_NO_HEAVY");
  const bool x_order_twice = x_switch("RP_X_ORDER_TWICE");
#else
  static constexpr bool x_no_heavy = false, x_order_twice = false;
#endif
  // (Measured and not kept, rounds 4-5: the lean launch in FRONT of the full-capacity one where there is no companion stream
  // (660 against 668 k env-steps/s); a high-priority companion stream (no effect on the dispatch order); the full-capacity
  // launch on the slice's own stream with two slices (-2.5 %).)
  // Threads per residue class of rp_order_kernel: two envs per thread.  (A 512-thread workgroup needs a whole
  // idle CU -- with both stage kernels at two waves per SIMD it waited ~60 us for one on every substep of a
  // 2048-env slice -- while a single wave takes too long over 512 envs.  Measured: 2048-env slices 64 / 128 /
  // 256 threads: 681 / 678 / 647 k env-steps/s; 4096-env slices 64 / 128 / 256 / 512: 540 / 547 / 561 / 562 k.)
  int order_threads_for(int cnt) const {
    // (the kernel's scan needs a full first wave and its launch bound is 512: multiples of 64 in 64 .. 512)
    const int t = ((cnt / 16 + 63) / 64) * 64;
    return t < 64 ? 64 : (t > 512 ? 512 : t);
  }
  // (each of these workgroups needs a whole idle SIMD, also just to find the list empty: 512 of them delayed the slice's join;
  // measured 64 ... 128 best on configs 2-4, 16 starves config 3)
  const int kHeavyGrid = getenv("RP_HEAVY_GRID") ? (atoi(getenv("RP_HEAVY_GRID")) > 0 ? atoi(getenv("RP_HEAVY_GRID")) : 1) : 128;   // (one wave of that stage owns a SIMD: half the chip at most)
  const bool heavy_grid_fixed = getenv("RP_HEAVY_GRID") != nullptr;   // (experiment: RP_HEAVY_GRID pins the grid)
  // acceleration-stage sensors (rp_set_acc_sensors): state before the last Euler step, outputs
  bool sensors_on = false;
  T *d_qpos_prev = nullptr, *d_qvel_prev = nullptr, *d_con_force = nullptr, *d_sens_torque = nullptr,
    *d_sens_touch = nullptr;
  int acc_sensors(int on) override {
    if (on && !d_qpos_prev) {
      HIP_OK(hipSetDevice(device));
      try {
        size_t E = (size_t)nenv;
        d_qpos_prev = dalloc<T>(E * nv); d_qvel_prev = dalloc<T>(E * nv);
        d_con_force = dalloc<T>(E * RPK_NC * 4);
        d_sens_torque = dalloc<T>(E * nv); d_sens_touch = dalloc<T>(E * (nsite ? nsite : 1));
      } catch (const std::string& s) { return fail("rp_set_acc_sensors: " + s); }
      HIP_OK(hipDeviceSynchronize());  // (null-stream fills vs the engine's stream)
    }
    sensors_on = on != 0;
    S.con_force = sensors_on ? d_con_force : nullptr;
    return 0;
  }
  int* d_lead = nullptr;            // lazy position stage: active && !valid
  unsigned char* d_valid = nullptr; // hand-over of env e matches its state
  int profile(long long* out, int n, int enable) override {
    HIP_OK(hipSetDevice(device));
    if (!d_prof) { d_prof = dalloc<long long>(RPK_NPROF_ALL); }
    HIP_OK(hipStreamSynchronize(stream));
    if (out) {
      long long h[RPK_NPROF_ALL];
      HIP_OK(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
      for (int i = 0; i < n && i < RPK_NPROF_ALL; i++) out[i] = h[i];
    }
    HIP_OK(hipStreamSynchronize(stream));  // counters of launches still in flight
    HIP_OK(hipMemset(d_prof, 0, sizeof(long long) * RPK_NPROF_ALL));
    HIP_OK(hipDeviceSynchronize());        // (null-stream fill vs the engine's non-blocking stream)
    S.prof = enable ? d_prof : nullptr;
    return 0;
  }
  void mpr_tolerances(double tol, double poly_tol) override {
    const double floor_ = sizeof(T) == 4 ? 1e-6 : 0.0;   // (single precision cannot resolve a tighter portal distance)
    if (tol > 0) M.mpr_tol = (T)(tol > floor_ ? tol : floor_);
    if (poly_tol > 0) M.mpr_tol_poly = (T)(poly_tol > floor_ ? poly_tol : floor_);
    else if (tol > 0) M.mpr_tol_poly = M.mpr_tol;
  }
  void tolerances(double tol, double ls_tol) override {
    if (tol > 0) M.tolerance = (T)tol;
    if (ls_tol > 0) M.ls_tolerance = (T)ls_tol;
  }
  void limits(int newton, int ls) override {
    if (newton > 0) S.max_newton = newton;
    if (ls > 0) S.max_ls = ls;
  }

  int reset(const uint8_t* mask) override {
    HIP_OK(hipSetDevice(device));
    const unsigned char* dmask = nullptr;
    if (mask) {
      hipPointerAttribute_t attr;
      bool on_device = hipPointerGetAttributes(&attr, mask) == hipSuccess &&
                       attr.type == hipMemoryTypeDevice;
      (void)hipGetLastError();
      if (on_device) dmask = mask;
      else {
        if (!d_mask) HIP_OK(hipMalloc((void**)&d_mask, (size_t)nenv));
        HIP_OK(hipMemcpyAsync(d_mask, mask, (size_t)nenv, hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));  // host source may go away
        dmask = d_mask;
      }
    }
    RpState<T> sr = S;
    sr.sens_torque = sensors_on ? d_sens_torque : nullptr; sr.sens_touch = sensors_on ? d_sens_touch : nullptr;
    hipLaunchKernelGGL(rp_reset_kernel<T>, dim3(nenv), dim3(64), 0, stream, sr, d_qpos0, dmask, nv, nu, nsite, d_valid);
    HIP_OK(hipGetLastError());
    return 0;
  }

  bool field(rp_field f, void** p, size_t* bytes, bool* writable) {
    size_t E = (size_t)nenv;
    *writable = false;
    switch (f) {
      case RP_QPOS: *p = S.qpos; *bytes = sizeof(T) * E * nv; *writable = true; return true;
      case RP_QVEL: *p = S.qvel; *bytes = sizeof(T) * E * nv; *writable = true; return true;
      case RP_QACC_WARMSTART: *p = S.warm; *bytes = sizeof(T) * E * nv; *writable = true; return true;
      case RP_CTRL: *p = S.ctrl; *bytes = sizeof(T) * E * nu; *writable = true; return true;
      case RP_QFRC_APPLIED: *p = S.qfrc_applied; *bytes = sizeof(T) * E * nv; *writable = true; return true;
      case RP_ACT_FORCE: *p = S.act_force; *bytes = sizeof(T) * E * nu; return true;
      case RP_ACT_VELOCITY: *p = S.act_vel; *bytes = sizeof(T) * E * nu; return true;
      case RP_SITE_XPOS: *p = S.site_xpos; *bytes = sizeof(T) * E * nsite * 3; return true;
      case RP_TIME: *p = S.time; *bytes = sizeof(T) * E; *writable = true; return true;
      case RP_NCON: *p = S.ncon; *bytes = sizeof(int) * E; return true;
      case RP_CONTACT_GEOMS: *p = S.contact_geoms; *bytes = sizeof(int) * E * RPK_NCOUT * 2; return true;
      case RP_WARN_FLAGS: *p = S.warn; *bytes = sizeof(int) * E; *writable = true; return true;
      case RP_SOLVER_ITER: *p = S.solver_iter; *bytes = sizeof(int) * E; return true;
      case RP_CONTACT_DIST: *p = S.contact_dist; *bytes = sizeof(T) * E * RPK_NCOUT; return true;
      case RP_ACTIVE: *p = d_active; *bytes = sizeof(int) * E; *writable = true; return true;
      case RP_TREE_OFFSET: *p = S.tree_offset; *bytes = sizeof(T) * E * ntree * 3; *writable = true; return true;
      case RP_ENV_COST: *p = S.cost_sol; *bytes = sizeof(int) * E; return true;
      case RP_DEBUG_MASS_ROWS: *p = B.RM; *bytes = sizeof(T) * E * RPK_NLX(md()) * (md() + 1); return true;
      case RP_DEBUG_HANDOVER_HDR: *p = B.hdr; *bytes = sizeof(int) * E * 8; return true;
      case RP_SENSOR_TORQUE: if (!d_sens_torque) return false; *p = d_sens_torque; *bytes = sizeof(T) * E * nv; return true;
      case RP_SENSOR_TOUCH: if (!d_sens_touch) return false; *p = d_sens_touch; *bytes = sizeof(T) * E * nsite; return true;
    }
    return false;
  }
  int field_ptr(rp_field f, void** p, size_t* bytes) override {
    bool w;
    if (!field(f, p, bytes, &w)) return fail("rp_field_ptr: unknown field (sensor fields need rp_set_acc_sensors first)");
    if (f == RP_ACTIVE) S.active = d_active;  // a caller that maps the mask uses it
    return 0;
  }
  int set(rp_field f, const void* src) override {
    void* p; size_t nb; bool w;
    if (!field(f, &p, &nb, &w)) return fail("rp_set: unknown field");
    if (!w) return fail("rp_set: field is read-only");
    if (!src) { if (f == RP_ACTIVE) { S.active = nullptr; return 0; } return fail("rp_set: null source"); }
    HIP_OK(hipSetDevice(device));
    if (nb) {
      // Host sources: drain the stream first.  Stream order alone should put this copy after
      // the kernels already enqueued, but under rocprofv3 --pmc (dispatch interception) a
      // pending kernel was observed to run after a later host-to-device copy.
      hipPointerAttribute_t attr;
      const bool src_on_device = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeDevice;
      (void)hipGetLastError();
      if (!src_on_device) HIP_OK(hipStreamSynchronize(stream));
      HIP_OK(hipMemcpyAsync(p, src, nb, hipMemcpyDefault, stream));
    }
    if (f == RP_ACTIVE) S.active = d_active;
    // state the position stage depends on changed: every env's hand-over is stale
    if (f == RP_QPOS || f == RP_QVEL || f == RP_TREE_OFFSET) HIP_OK(hipMemsetAsync(d_valid, 0, (size_t)nenv, stream));
    return 0;
  }
  int get(rp_field f, void* dst) override {
    void* p; size_t nb; bool w;
    if (!field(f, &p, &nb, &w)) return fail("rp_get: unknown field");
    if (!dst) return fail("rp_get: null destination");
    HIP_OK(hipSetDevice(device));
    if (nb) HIP_OK(hipMemcpyAsync(dst, p, nb, hipMemcpyDefault, stream));
    hipPointerAttribute_t attr;
    bool on_device = hipPointerGetAttributes(&attr, dst) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (!on_device) HIP_OK(hipStreamSynchronize(stream));  // device destinations stay stream-ordered
    return 0;
  }
  int step(int nsub, uint32_t* trace, int mode, const uint8_t* reset_mask = nullptr) override {
    HIP_OK(hipSetDevice(device));
    if (mode == 0 && nsub <= 0) return fail("rp_step: n_substeps must be positive");
    if (reset_mask) {
      // rp_step_masked: physics.reset() of the flagged envs first (same launch as rp_reset with a device mask); their
      // physics.forward() is the leading position / velocity stage below
      hipPointerAttribute_t attr;
      const bool on_device = hipPointerGetAttributes(&attr, reset_mask) == hipSuccess && attr.type == hipMemoryTypeDevice;
      (void)hipGetLastError();
      if (!on_device) return fail("rp_step_masked: the reset mask must be device memory");
      RpState<T> sr = S;
      sr.sens_torque = sensors_on ? d_sens_torque : nullptr; sr.sens_touch = sensors_on ? d_sens_touch : nullptr;
      hipLaunchKernelGGL(rp_reset_kernel<T>, dim3(nenv), dim3(64), 0, stream, sr, d_qpos0, reset_mask, nv, nu, nsite, d_valid);
    }
    RpState<T> s = S;
    size_t need = (size_t)nenv * (nsub > 0 ? nsub : 1) * 4;
    if (trace && mode == 0) {
      if (need > trace_cap) {
        if (d_trace) hipFree(d_trace);
        HIP_OK(hipMalloc((void**)&d_trace, need * sizeof(uint32_t)));
        trace_cap = need;
      }
      s.key_trace = d_trace;
    }
    // inside a stream capture (the caller is recording a hipGraph of its whole step) no
    // event may be synchronised and per-launch events are meaningless: skip the timers
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    const int slot = ev_next;
    if (!capturing) {
      ev_next = (ev_next + 1) % kRing;
      harvest(slot, true);
    }
    const bool timeit = (mode == 0) && !capturing;
    if (timeit) HIP_OK(hipEventRecord(ev0[slot], stream));
    // mj_step1 for the current state, then n_sub x (mj_step2; mj_step1): dm_control's legacy
    // order.  Two small kernels per substep instead of one fused launch: each half fits in
    // registers, and the hand-over (RpStage) stays in L2 / Infinity Cache.
    const int hb = (nenv + 255) / 256;
    // Slices: the batch may be stepped as two halves on two streams.  Within a half the kernels are
    // ordered (solver -> position -> solver ...), between the halves they are not, so the tail of one
    // half's launch (a few heavy envs still running, most SIMDs idle) is filled by the other half's
    // next kernel instead of waiting for a launch boundary.
    int want = n_slices;
    const bool fused_capable = lean && !deep && !graph && sizeof(T) == 8;   // (no fused builds for scenes with graph hulls)
    bool fused_now = fused == 1 && fused_capable && mode == 0;
    bool sched4 = false;
    if (n_slices == 0 && mode == 0 && !fused_now) {
      // The schedule of an rp_step, chosen by RULE (round 6; rounds 3-5 ran timing trials of up to four candidates on
      // some steps of every 128 .. 1024: 1 % of a run, and a choice that differed from run to run).  All schedules give
      // bit-identical results (tests/test_gpu_parity.py), so the rule only has to be good, not exact:
      //   1 = one launch per stage        2 = the same as two slices on two streams, full-capacity launches on companion streams
      //   3 = fused substeps (one launch takes a light env through all substeps; the rest in a clean-up launch)
      //   4 = three slices, the position stage split (front part / pooled narrow phase / back part), no companion streams
      // Measured on MI355X: batches under 3072 envs fill at most one round and a half of the chip's 2048 wave
      // slots -- every launch boundary there is a tail, and the fused schedule (no boundaries) wins by 6-16 %; from 3072
      // envs on, three slices with the split stage win on the replay (lists of envs outside the light class short or
      // empty), two slices with companion streams when those lists are long (random policies: the full-capacity
      // launches then need the chip to themselves beside the lean ones); 6144 envs and more are three rounds per launch
      // and need no slices at all.  `many_heavy` follows the list lengths the device reports one step late (with
      // hysteresis), never a timer.
      double hl = 0;
      for (int i = 0; i < last_nsl && i < kMaxSlices; i++) hl = heavy_est[i] > hl ? heavy_est[i] : hl;   // (the slices the last step used: the others' estimates are stale)
      many_heavy = many_heavy ? hl >= 2.0 : hl >= 4.0;
      const bool fused_ok = fused == 2 && fused_capable;
      // (the split stage keeps at most 256 candidates / 384 result records per env where the one-kernel stage never
      // overflows: once it has dropped any -- RP_WARN_SPLIT_FULL, counted on the device and read back with the list
      // lengths -- the rule stops choosing it for this engine: identical inputs must not give different physics
      // depending on the schedule.  ADVICE round 5.)
      if (h_heavy_peak && *(volatile int*)&h_heavy_peak[kMaxSlices] > 0) split_dropped = true;
      const bool split_ok = split_mode == 2 && split_capable && !capturing && !split_dropped;
      int sched;
      if (capturing) sched = fused_ok ? 3 : 1;
      else if (nenv < 3072) sched = (fused_ok && !many_heavy) ? 3 : (nenv >= 1024 ? 2 : 1);   // (config 5, 2048 envs, long lists: fused 245 k, two slices 286 k)
      else if (split_ok && !many_heavy) sched = 4;
      else sched = nenv >= 6144 ? 1 : 2;
      if (!capturing) auto_mode = sched;
      fused_now = sched == 3;
      want = sched == 2 ? 2 : (sched == 4 ? 3 : 1);
      sched4 = sched == 4;
    }
    split_now = split_capable && mode == 0 && (split_mode == 1 || sched4);
    int nsl = (mode == 0 && want > 1 && nenv >= 1024 && !capturing && !fused_now) ? (want >= 4 ? 4 : want) : 1;
    // (slices 2 / 3 run ON the companion streams of slices 0 / 1 -- pooled_stream: four hardware queues -- so with more
    // than two slices there are no companion streams, whoever asked for the slices: the full-capacity launch then goes in
    // front of the lean one on the slice's own stream.  ADVICE round 5: only the engine's own three-slice schedule turned
    // them off; a forced rp_set_stream_slices(e, 3 | 4) queued slice 2's whole chain behind slice 0's heavy solves.)
    if (split_now && ((capturing && !B.frames) || !ensure_split_buffers())) split_now = false;   // (no allocation inside a stream capture)
    if (mode == 0 && !capturing) last_nsl = nsl;
    const bool companion_now = nsl <= 2;
    if (nsl > 1 && !ev_fork && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev_fork = nullptr; nsl = 1; }
    for (int i = 1; i < nsl; i++) {
      if (xstream[i]) continue;
      xstream[i] = pooled_stream(device, i, false);
      if (!xstream[i] || hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); xstream[i] = nullptr; nsl = 1; break; }
    }
    // slice sl covers the envs [bound(sl), bound(sl + 1)); bounds are multiples of 8 (XCD classes of the order)
    auto bound = [&](int sl) { return sl >= nsl ? nenv : (int)(((long long)nenv * sl / nsl + 7) / 8 * 8); };
    if (cost_order && mode == 0) s.order = d_order;   // (initialised to the identity; refreshed below)
    // (legacy_step = False: the leading stage always runs -- it is what publishes the outputs of the incoming state,
    // which the previous step's last position stage computed but kept to itself)
    const bool lazy_now = lazy_position && legacy_step;
    s.stale_outputs = (!legacy_step && mode == 0) ? 1 : 0;
    const bool lead_masked = (lazy_now || reset_mask) && mode == 0;
    if (lead_masked)
      hipLaunchKernelGGL(rp_lead_mask_kernel, dim3(hb), dim3(256), 0, stream, d_lead, s.active, d_valid, lazy_now ? 1 : 0, reset_mask, nenv);
    if (nsl > 1) {
      HIP_OK(hipEventRecord(ev_fork, stream));
      for (int i = 1; i < nsl; i++) HIP_OK(hipStreamWaitEvent(xstream[i], ev_fork, 0));
    }
    for (int sl = 0; sl < nsl; sl++) {
      hipStream_t st = sl == 0 ? stream : xstream[sl];
      const int base = bound(sl), cnt = bound(sl + 1) - base;
      RpState<T> ss = s;
      ss.env_base = base;
      auto launch_pos_listed = [&](const RpState<T>& q, int k, int grid, hipStream_t str) {
        if constexpr (sizeof(T) == 8) {   // (the light class exists in the fp64 default builds only)
          if (mesh) hipLaunchKernelGGL((rp_pos_list_kernel<T, 1>), dim3(grid), dim3(64), 0, str, M, q, B, k, nsub);
          else hipLaunchKernelGGL((rp_pos_list_kernel<T, 0>), dim3(grid), dim3(64), 0, str, M, q, B, k, nsub);
        }
      };
      // the position / velocity stage of substep k as front part, pooled narrow phase, back part (same results, bit for bit)
      auto launch_pos_split = [&](const RpState<T>& q, int k) {
        if constexpr (sizeof(T) == 8) {
          RpStage<T> Bs = B;
          Bs.tcount_off = sl * RPK_NSTRIPE * RPK_NTYPE_PAD;
          Bs.split_dropped = capturing ? nullptr : d_heavy_peak + kMaxSlices;
          int ng = cnt / 2;
          ng = ng < 64 ? 64 : (ng > 2048 ? 2048 : ng);
#define RP_SPLIT_LAUNCH(MESH_)                                                                                                     \
          {                                                                                                                        \
            hipLaunchKernelGGL((rp_pos_front_kernel<T, MESH_>), dim3(cnt), dim3(64), 0, st, M, q, Bs, k, nsub);                     \
            hipLaunchKernelGGL((rp_narrow_kernel<T, MESH_>), dim3(ng), dim3(64), 0, st, M, q, Bs);                                  \
            hipLaunchKernelGGL((rp_pos_back_kernel<T, MESH_>), dim3(cnt), dim3(64), 0, st, M, q, Bs, k, nsub);                      \
          }
          if (mesh && graph) RP_SPLIT_LAUNCH(2) else if (mesh) RP_SPLIT_LAUNCH(1) else RP_SPLIT_LAUNCH(0)
#undef RP_SPLIT_LAUNCH
        }
      };
      auto launch_pos_on = [&](const RpState<T>& q, int k) {
        if (split_now && !deep && sizeof(T) == 8) { launch_pos_split(q, k); return; }
        if (deep && mesh && graph) hipLaunchKernelGGL((rp_stage_kernel<T, 0, 0, RPK_MAXD_DEEP, 2>), dim3(cnt), dim3(64), 0, st, M, q, B, k, nsub);
        else if (mesh && graph) hipLaunchKernelGGL((rp_stage_kernel<T, 0, 0, RPK_MAXD, 2>), dim3(cnt), dim3(64), 0, st, M, q, B, k, nsub);
        else if (deep && mesh) hipLaunchKernelGGL((rp_stage_kernel<T, 0, 0, RPK_MAXD_DEEP, 1>), dim3(cnt), dim3(64), 0, st, M, q, B, k, nsub);
        else if (deep) hipLaunchKernelGGL((rp_stage_kernel<T, 0, 0, RPK_MAXD_DEEP>), dim3(cnt), dim3(64), 0, st, M, q, B, k, nsub);
        else if (mesh) hipLaunchKernelGGL((rp_stage_kernel<T, 0, 0, RPK_MAXD, 1>), dim3(cnt), dim3(64), 0, st, M, q, B, k, nsub);
        else hipLaunchKernelGGL((rp_stage_kernel<T, 0>), dim3(cnt), dim3(64), 0, st, M, q, B, k, nsub);
      };
      // mj_step1 for the current state, in index order: d_order may date from a step with another slice
      // count (a permutation of other ranges), and every slice must write the hand-over of exactly ITS envs
      // before its solver stage reads it (the solver stage consumes the hand-over: it parks values in it)
      RpState<T> lead = ss;
      lead.order = nullptr;
      if (lead_masked) lead.active = d_lead;   // skipped for envs whose hand-over is still the one of their state
      launch_pos_on(lead, -1);
      if (mode != 0) continue;
      if constexpr (sizeof(T) == 8) {
        if (fused_now) {
          // heaviest envs first (4096 envs are two rounds of resident waves), from the hand-over just written
          if (cost_order)
            hipLaunchKernelGGL(rp_order_kernel, dim3(RP_ORDER_CLASSES), dim3(order_threads_for(cnt)), 0, st, d_order, B.hdr, s.active, base, cnt, (int*)nullptr, (int*)nullptr, (unsigned char*)nullptr);
          RpState<T> sf = ss;
          sf.heavy_list = d_heavy + base; sf.heavy_cnt = d_heavy_cnt + 2 * sl; sf.heavy_done = d_heavy_cnt + 2 * sl + 1;
          sf.heavy_peak = capturing ? nullptr : d_heavy_peak + sl;
          if (sensors_on) { sf.qpos_prev = d_qpos_prev; sf.qvel_prev = d_qvel_prev; }
          const bool probe = timeit && sl == 0;
          if (probe) { HIP_OK(hipEventRecord(sv0[slot], st)); sv_envs[slot] = cnt * nsub; sv_kind[slot] = 1; }
          const int cgrid = capturing ? (cnt < kHeavyGrid ? cnt : kHeavyGrid) : heavy_grid_for(sl, cnt);
          // (round 6: with the split stage's bodies where they exist -- the one-kernel position body spills 276 registers
          // in the hull builds; RP_FUSED_SPLIT=0: the round-3 kernel)
          const bool fsplit = fused_split && split_capable && !deep && (capturing ? B.frames != nullptr : ensure_split_buffers());
          RpStage<T> Bf = B;
          Bf.tlist = nullptr;   // (no pooled lists: every wave runs its own env's narrow phase)
          if (fsplit && mesh) {
            hipLaunchKernelGGL((rp_fused_split_kernel<T, 1>), dim3(cnt), dim3(64), 0, st, M, sf, Bf, nsub);
            if (trunk4) hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 1, 4>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
            else hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 1, 0>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
          } else if (fsplit) {
            hipLaunchKernelGGL((rp_fused_split_kernel<T, 0>), dim3(cnt), dim3(64), 0, st, M, sf, Bf, nsub);
            if (trunk4) hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 0, 4>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
            else hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 0, 0>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
          } else if (mesh) {
            hipLaunchKernelGGL((rp_fused_steps_kernel<T, 1>), dim3(cnt), dim3(64), 0, st, M, sf, B, nsub);
            if (trunk4) hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 1, 4>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
            else hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 1, 0>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
          } else {
            hipLaunchKernelGGL((rp_fused_steps_kernel<T, 0>), dim3(cnt), dim3(64), 0, st, M, sf, B, nsub);
            if (trunk4) hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 0, 4>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
            else hipLaunchKernelGGL((rp_cleanup_steps_kernel<T, 0, 0>), dim3(cgrid), dim3(64), 0, st, M, sf, B, nsub);
          }
          if (probe) HIP_OK(hipEventRecord(sv1[slot], st));
          if (sensors_on) {
            // sensor stage: position / velocity stage of the state before the last substep + mj_rnePostConstraint
            // with the constrained qacc (S.warm) and the contact row forces of that substep's solver stage
            RpState<T> sq = ss;
            sq.qpos = d_qpos_prev; sq.qvel = d_qvel_prev;
            sq.sens_torque = d_sens_torque; sq.sens_touch = d_sens_touch;
            sq.key_trace = nullptr; sq.prof = nullptr;
            if (mesh) hipLaunchKernelGGL((rp_stage_kernel<T, 2, 0, RPK_MAXD, 1>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
            else hipLaunchKernelGGL((rp_stage_kernel<T, 2>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
          }
          continue;
        }
      }
      // ... then n_sub x (mj_step2; mj_step1): dm_control's legacy order.  Two kernels per substep instead
      // of one fused launch: each half fits in registers, the hand-over (RpStage) stays in L2 / Infinity Cache.
      int hgrid_step = 0;   // (the full-capacity stage's grid: one choice per step and slice)
      bool split_step = false;
      for (int k = 0; k < nsub; k++) {
        const bool probe = timeit && sl == 0 && k == (int)(step_calls % (unsigned)nsub);
        const bool sense = sensors_on && k == nsub - 1;
        // cost-ordered launch: heaviest envs first, from the hand-over the position stage just wrote
        // (the same pass compacts the envs outside the light class for the full-capacity solver stage)
        const bool listed = lean && d_heavy != nullptr;
        if (cost_order || listed)
          hipLaunchKernelGGL(rp_order_kernel, dim3(RP_ORDER_CLASSES), dim3(order_threads_for(cnt)), 0, st, cost_order ? d_order : nullptr, B.hdr, s.active, base, cnt,
                             listed ? d_heavy : nullptr, listed ? d_heavy_cnt + 2 * sl : nullptr, listed ? d_listed : nullptr);
        // (RP_X_ORDER_TWICE=1: MEASUREMENT ONLY -- the order pass a second time (no list): what the pass costs the step)
        if (x_order_twice && cost_order)
          hipLaunchKernelGGL(rp_order_kernel, dim3(RP_ORDER_CLASSES), dim3(order_threads_for(cnt)), 0, st, d_order, B.hdr, s.active, base, cnt, (int*)nullptr, (int*)nullptr, (unsigned char*)nullptr);
        if (sense) {  // the state this substep's forces belong to (the solver stage integrates in place)
          HIP_OK(hipMemcpyAsync(d_qpos_prev + (size_t)base * nv, S.qpos + (size_t)base * nv, sizeof(T) * (size_t)cnt * nv, hipMemcpyDeviceToDevice, st));
          HIP_OK(hipMemcpyAsync(d_qvel_prev + (size_t)base * nv, S.qvel + (size_t)base * nv, sizeof(T) * (size_t)cnt * nv, hipMemcpyDeviceToDevice, st));
        }
        if (probe) { HIP_OK(hipEventRecord(sv0[slot], st)); sv_envs[slot] = cnt; sv_kind[slot] = 0; }
        // solver stage; the build specialised for "every tree has a 4-link trunk" when it applies
        // light envs on the lean build (two waves per SIMD), the others on the full-capacity build (it skips
        // the light ones) -- side by side: the full-capacity launch goes to the slice's companion stream
        hipStream_t hs = st;
        if (lean && !capturing && companion_now) {
          if (!hstream[sl]) hstream[sl] = pooled_stream(device, 4 + sl, false);
          if (hstream[sl] && !ev_hfork[sl] && (hipEventCreateWithFlags(&ev_hfork[sl], hipEventDisableTiming) != hipSuccess ||
                                               hipEventCreateWithFlags(&ev_hjoin[sl], hipEventDisableTiming) != hipSuccess)) {
            (void)hipGetLastError(); hstream[sl] = nullptr;
          }
          if (hstream[sl]) {
            hs = hstream[sl];
            HIP_OK(hipEventRecord(ev_hfork[sl], st));
            HIP_OK(hipStreamWaitEvent(hs, ev_hfork[sl], 0));
          }
        }
        RpState<T> sh = ss;
        int hgrid = cnt;
        if (listed) {
          sh.heavy_list = d_heavy + base; sh.heavy_cnt = d_heavy_cnt + 2 * sl; sh.heavy_done = d_heavy_cnt + 2 * sl + 1;
          sh.heavy_peak = capturing ? nullptr : d_heavy_peak + sl;
          // (the split pays when the list is long: config 3 446 -> 455 k; on a batch whose lists are empty the extra
          // launch and the later join cost 1-4 %: config 2 657 -> 632 ... 651 k -- so it follows the same lagged estimate)
          hgrid = capturing ? (cnt < kHeavyGrid ? cnt : kHeavyGrid) : (k == 0 ? (hgrid_step = heavy_grid_for(sl, cnt)) : hgrid_step);
          split_step = hs != st && !deep && !graph && sizeof(T) == 8 && heavy_est[sl] >= 4.0;
          sh.heavy_keep = (split_step && !sense) ? 1 : 0;
        }
        // (RP_X_NO_HEAVY=1: MEASUREMENT ONLY -- the full-capacity launch is suppressed, envs outside the light class are
        // not stepped at all: what the launch costs a batch whose lists are empty)
        if (x_no_heavy && listed) { /* nothing */ }
        else if (deep) hipLaunchKernelGGL((rp_stage_kernel<T, 1, 0, RPK_MAXD_DEEP>), dim3(hgrid), dim3(64), 0, hs, M, sh, B, k, nsub);
        else if (trunk4) hipLaunchKernelGGL((rp_stage_kernel<T, 1, 4>), dim3(hgrid), dim3(64), 0, hs, M, sh, B, k, nsub);
        else hipLaunchKernelGGL((rp_stage_kernel<T, 1>), dim3(hgrid), dim3(64), 0, hs, M, sh, B, k, nsub);
        // ... and, except at a substep the sensor stage follows, the heavy envs' position / velocity stage goes with
        // them: the slice's own position launch then skips them and no longer waits for the slowest heavy solve
        // (config 3: 0.27 ms of every 0.81 ms substep); the streams join after it, in front of the next order pass
        const bool split_pos = split_step && listed && !sense;
        if (hs != st && !split_pos) HIP_OK(hipEventRecord(ev_hjoin[sl], hs));
        if (lean) hipLaunchKernelGGL((rp_lean_solver_kernel<T>), dim3(cnt), dim3(64), 0, st, M, ss, B);
        if (hs != st && !split_pos) HIP_OK(hipStreamWaitEvent(st, ev_hjoin[sl], 0));
        if (probe) HIP_OK(hipEventRecord(sv1[slot], st));
        if (sense) {
          // sensor stage: position / velocity stage of the saved state + mj_rnePostConstraint with the
          // constrained qacc (S.warm) and the contact row forces the solver stage just stored
          RpState<T> sq = ss;
          sq.qpos = d_qpos_prev; sq.qvel = d_qvel_prev;
          sq.sens_torque = d_sens_torque; sq.sens_touch = d_sens_touch;
          sq.key_trace = nullptr; sq.prof = nullptr;
          if (deep && mesh && graph) hipLaunchKernelGGL((rp_stage_kernel<T, 2, 0, RPK_MAXD_DEEP, 2>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
          else if (mesh && graph) hipLaunchKernelGGL((rp_stage_kernel<T, 2, 0, RPK_MAXD, 2>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
          else if (deep && mesh) hipLaunchKernelGGL((rp_stage_kernel<T, 2, 0, RPK_MAXD_DEEP, 1>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
          else if (deep) hipLaunchKernelGGL((rp_stage_kernel<T, 2, 0, RPK_MAXD_DEEP>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
          else if (mesh) hipLaunchKernelGGL((rp_stage_kernel<T, 2, 0, RPK_MAXD, 1>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
          else hipLaunchKernelGGL((rp_stage_kernel<T, 2>), dim3(cnt), dim3(64), 0, st, M, sq, B, -1, nsub);
        }
        if (split_pos) {
          RpState<T> sp = sh;              // (the list; walked in list order)
          sp.order = nullptr; sp.heavy_keep = 0;
          launch_pos_listed(sp, k, hgrid, hs);
          HIP_OK(hipEventRecord(ev_hjoin[sl], hs));
          RpState<T> sm_ = ss;
          sm_.skip_heavy = 1;
          launch_pos_on(sm_, k);
          HIP_OK(hipStreamWaitEvent(st, ev_hjoin[sl], 0));
        } else {
          launch_pos_on(ss, k);
        }
      }
    }
    for (int i = 1; i < nsl; i++) { HIP_OK(hipEventRecord(ev_join[i], xstream[i])); HIP_OK(hipStreamWaitEvent(stream, ev_join[i], 0)); }
    hipLaunchKernelGGL(rp_mark_valid_kernel, dim3(hb), dim3(256), 0, stream, d_valid, s.active, reset_mask, nenv);
    if (mode == 0 && lean && !capturing && h_heavy_peak && !heavy_grid_fixed) {
      // the longest lists of this step, for the grids of a later one (the host never waits for the copy)
      HIP_OK(hipMemcpyAsync(h_heavy_peak, d_heavy_peak, sizeof(int) * (kMaxSlices + 1), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipMemsetAsync(d_heavy_peak, 0, sizeof(int) * (kMaxSlices + 1), stream));
    }
    HIP_OK(hipGetLastError());
    if (mode == 0) step_calls++;
    if (timeit) { HIP_OK(hipEventRecord(ev1[slot], stream)); ev_pending[slot] = true; }
    if (trace && mode == 0)
      HIP_OK(hipMemcpyAsync(trace, d_trace, need * sizeof(uint32_t), hipMemcpyDefault, stream));
    return 0;
  }
};

EngineBase* E(rp_engine* e) { return reinterpret_cast<EngineBase*>(e); }

}  // namespace

extern "C" {

const char* rp_last_error(void) { return g_err.c_str(); }

int rp_create(const void* model_blob, size_t blob_bytes, int n_envs, int device_id, int precision,
              rp_engine** out) {
  if (!out) return fail("rp_create: out is null");
  *out = nullptr;
  if (!model_blob || n_envs <= 0) return fail("rp_create: bad arguments");
  if (precision != 32 && precision != 64) return fail("rp_create: precision must be 32 or 64");
  Blob b;
  if (!b.init(model_blob, blob_bytes)) return fail("rp_create: not a model blob (bad magic)");
  if (!b.has("eng_nlink")) return fail("rp_create: blob lacks engine tables (eng_*)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail("rp_create: no HIP device available (the engine has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail("rp_create: bad device id");
  HIP_OK(hipSetDevice(device_id));
  EngineBase* e = nullptr;
  try {
    if (precision == 32) { auto* p = new Engine<float>(); e = p; p->device = device_id; p->precision = 32;
      HIP_OK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)); p->build(b, n_envs); }
    else { auto* p = new Engine<double>(); e = p; p->device = device_id; p->precision = 64;
      HIP_OK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)); p->build(b, n_envs); }
  } catch (const std::string& s) {
    delete e;
    return fail("rp_create: " + s);
  }
  for (int i = 0; i < EngineBase::kRing; i++) {
    HIP_OK(hipEventCreate(&e->ev0[i]));
    HIP_OK(hipEventCreate(&e->ev1[i]));
    HIP_OK(hipEventCreate(&e->sv0[i]));
    HIP_OK(hipEventCreate(&e->sv1[i]));
  }
  int r = e->reset(nullptr);
  if (r) { delete e; return r; }
  *out = reinterpret_cast<rp_engine*>(e);
  return 0;
}

int rp_destroy(rp_engine* e) {
  if (!e) return fail("rp_destroy: null engine");
  delete E(e);
  return 0;
}
int rp_reset(rp_engine* e, const uint8_t* mask) { return e ? E(e)->reset(mask) : fail("null engine"); }
int rp_set(rp_engine* e, rp_field f, const void* src) { return e ? E(e)->set(f, src) : fail("null engine"); }
int rp_get(rp_engine* e, rp_field f, void* dst) { return e ? E(e)->get(f, dst) : fail("null engine"); }
int rp_step(rp_engine* e, int n_substeps, uint32_t* key_trace) {
  return e ? E(e)->step(n_substeps, key_trace, 0) : fail("null engine");
}
int rp_forward(rp_engine* e) { return e ? E(e)->step(0, nullptr, 1) : fail("null engine"); }
int rp_step_masked(rp_engine* e, int n_substeps, uint32_t* key_trace, const uint8_t* reset_mask) {
  return e ? E(e)->step(n_substeps, key_trace, 0, reset_mask) : fail("null engine");
}
int rp_set_solver_limits(rp_engine* e, int max_newton_iter, int max_ls_iter) {
  if (!e) return fail("null engine");
  E(e)->limits(max_newton_iter, max_ls_iter);
  return 0;
}
int rp_profile(rp_engine* e, long long* out, int n, int enable) {
  return e ? E(e)->profile(out, n, enable) : fail("null engine");
}
int rp_set_mpr_tolerance(rp_engine* e, double tolerance, double polytope_tolerance) {
  if (!e) return fail("null engine");
  E(e)->mpr_tolerances(tolerance, polytope_tolerance);
  return 0;
}
int rp_set_solver_tolerance(rp_engine* e, double tolerance, double ls_tolerance) {
  if (!e) return fail("null engine");
  E(e)->tolerances(tolerance, ls_tolerance);
  return 0;
}
int rp_set_lazy_position_stage(rp_engine* e, int on) {
  if (!e) return fail("null engine");
  E(e)->lazy_position = on != 0;
  return 0;
}
int rp_set_legacy_step(rp_engine* e, int on) {
  if (!e) return fail("null engine");
  E(e)->legacy_step = on != 0;
  return 0;
}
int rp_set_acc_sensors(rp_engine* e, int on) { return e ? E(e)->acc_sensors(on) : fail("null engine"); }
int rp_set_lean_solver(rp_engine* e, int on) { return e ? E(e)->lean_solver(on) : fail("null engine"); }
int rp_set_fused_substeps(rp_engine* e, int on) { return e ? E(e)->fused_substeps(on) : fail("null engine"); }
int rp_get_fused_substeps(rp_engine* e) { return e ? E(e)->fused_substeps_on() : fail("null engine"); }
int rp_set_split_position_stage(rp_engine* e, int on) { return e ? E(e)->split_position(on) : fail("null engine"); }
int rp_get_split_position_stage(rp_engine* e) { return e ? E(e)->split_position_on() : fail("null engine"); }
int rp_set_stream_slices(rp_engine* e, int n) {
  if (!e) return fail("null engine");
  if (n != 0 && n != 1 && n != 2 && n != 3 && n != 4) return fail("rp_set_stream_slices: 0 (automatic), 1, 2, 3 or 4");
  E(e)->n_slices = n;
  return 0;
}
int rp_set_cost_ordered_launch(rp_engine* e, int on) {
  if (!e) return fail("null engine");
  E(e)->cost_order = on != 0;
  return 0;
}
int rp_sync(rp_engine* e) {
  if (!e) return fail("null engine");
  HIP_OK(hipSetDevice(E(e)->device));
  HIP_OK(hipStreamSynchronize(E(e)->stream));
  return 0;
}
int rp_set_stream(rp_engine* e, void* hip_stream) {
  if (!e) return fail("null engine");
  EngineBase* b = E(e);
  HIP_OK(hipSetDevice(b->device));
  HIP_OK(hipStreamSynchronize(b->stream));
  for (int i = 0; i < EngineBase::kRing; i++) b->harvest(i, true);
  if (b->own_stream && b->stream) HIP_OK(hipStreamDestroy(b->stream));
  b->stream = (hipStream_t)hip_stream;
  b->own_stream = false;
  return 0;
}
int rp_field_ptr(rp_engine* e, rp_field f, void** ptr, size_t* bytes) {
  if (!e || !ptr || !bytes) return fail("null argument");
  return E(e)->field_ptr(f, ptr, bytes);
}
int rp_get_stream(rp_engine* e, void** hip_stream) {
  if (!e || !hip_stream) return fail("null argument");
  *hip_stream = (void*)E(e)->stream;
  return 0;
}
int rp_n_envs(const rp_engine* e) { return e ? reinterpret_cast<const EngineBase*>(e)->nenv : -1; }
int rp_dim(const rp_engine* e, const char* name) {
  if (!e || !name) return -1;
  const EngineBase* b = reinterpret_cast<const EngineBase*>(e);
  if (!strcmp(name, "nv")) return b->nv;
  if (!strcmp(name, "nu")) return b->nu;
  if (!strcmp(name, "nsite")) return b->nsite;
  if (!strcmp(name, "ntree")) return b->ntree;
  if (!strcmp(name, "nkey")) return b->nkey;
  if (!strcmp(name, "nlink")) return b->nlink;
  if (!strcmp(name, "maxdepth")) return b->maxdepth;
  if (!strcmp(name, "precision")) return b->precision;
  if (!strcmp(name, "rm_rows")) return b->rm_rows;
  if (!strcmp(name, "rm_cols")) return b->rm_cols;
  return -1;
}
int rp_kernel_time(rp_engine* e, double* avg_ms, int* n_launches) {
  if (!e) return fail("null engine");
  EngineBase* b = E(e);
  HIP_OK(hipSetDevice(b->device));
  for (int i = 0; i < EngineBase::kRing; i++) b->harvest(i, true);
  if (avg_ms) *avg_ms = b->kernel_launches ? b->kernel_ms / b->kernel_launches : 0.0;
  if (n_launches) *n_launches = b->kernel_launches;
  b->kernel_ms = 0; b->kernel_launches = 0;
  return 0;
}
int rp_solver_kernel_time(rp_engine* e, double* avg_ms, int* n_launches) {
  if (!e) return fail("null engine");
  EngineBase* b = E(e);
  HIP_OK(hipSetDevice(b->device));
  // harvest without clearing the step-sequence statistics
  double km = b->kernel_ms; int kl = b->kernel_launches;
  for (int i = 0; i < EngineBase::kRing; i++) b->harvest(i, true);
  const int kd = b->solver_launches_k[1] > b->solver_launches_k[0] ? 1 : 0;
  if (avg_ms) *avg_ms = b->solver_launches_k[kd] ? b->solver_ms_k[kd] / b->solver_launches_k[kd] : 0.0;
  if (n_launches) *n_launches = b->solver_launches_k[kd];
  b->last_solver_envs = b->solver_launches_k[kd] ? b->solver_envs_k[kd] / b->solver_launches_k[kd] : 0.0;
  b->last_solver_kind = kd;
  for (int k = 0; k < 2; k++) { b->solver_ms_k[k] = 0; b->solver_launches_k[k] = 0; b->solver_envs_k[k] = 0; }
  (void)km; (void)kl;
  return 0;
}
int rp_solver_kernel_fused(rp_engine* e) { return e ? E(e)->last_solver_kind : fail("null engine"); }
int rp_solver_kernel_envs(rp_engine* e, double* avg_envs) {
  if (!e) return fail("null engine");
  if (avg_envs) *avg_envs = E(e)->last_solver_envs;
  return 0;
}

}  // extern "C"
