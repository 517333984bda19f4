// wavesim.cpp -- TEST INFRASTRUCTURE ONLY (see wavesim.hpp).
#include "wavesim.hpp"

#include <dlfcn.h>
#include <sys/mman.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" void ws_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl ws_switch
.type ws_switch,@function
ws_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size ws_switch, .-ws_switch
)");

namespace wavesim {
namespace {

constexpr size_t kStack = 512 * 1024;
enum { RUNNABLE = 0, WAITING = 1, DONE = 2 };

struct Fiber {
  void* sp = nullptr;
  int state = DONE;
  Kind kind = K_BARRIER;
  const void* site = nullptr;
  uint32_t val = 0, arg = 0, val2 = 0;
  uint64_t result = 0;
};

struct Group {
  std::vector<Fiber> f;
  char* stacks = nullptr;
  size_t nstacks = 0;
  int n = 0, cur = -1;
  void* main_sp = nullptr;
  const std::function<void()>* fn = nullptr;
  Ctx ctx;
  ~Group() { if (stacks) munmap(stacks, nstacks * kStack); }
};

thread_local Group g;

[[noreturn]] void die(const char* msg, const Fiber* a = nullptr, int lane = -1) {
  fprintf(stderr, "wavesim: %s", msg);
  if (a) {
    fprintf(stderr, " (kind %d, site %p, lane %d, arg %u)", (int)a->kind, a->site, lane, a->arg);
    Dl_info info;
    if (dladdr(a->site, &info) && info.dli_fbase)
      fprintf(stderr, "\n  addr2line -Cfie %s 0x%lx", info.dli_fname, (unsigned long)((const char*)a->site - (const char*)info.dli_fbase));
  }
  fprintf(stderr, "\n");
  abort();
}

// source lane of a DPP control word for lane l (-1: no valid source)
int dpp_source(int ctrl, int l) {
  const int row = l & ~15, r = l & 15;
  if (ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; return r + n < 16 ? l + n : -1; }
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; return r - n >= 0 ? l - n : -1; }
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl - 0x120; return row | ((r - n) & 15); }
  switch (ctrl) {
    case 0x130: return l + 1 < 64 ? l + 1 : -1;
    case 0x134: return (l + 1) & 63;
    case 0x138: return l - 1 >= 0 ? l - 1 : -1;
    case 0x13C: return (l - 1) & 63;
    case 0x140: return row | (15 - r);
    case 0x141: return (l & ~7) | (7 - (l & 7));
    case 0x142: return row >= 16 ? row - 1 : -1;
    case 0x143: return l >= 32 ? 31 : -1;
  }
  die("unsupported DPP control");
}

void resolve_wave_group(int base, int nw, const std::vector<int>& lanes) {
  // lanes: indices (within the workgroup) of the fibers of one wave waiting at one (kind, site)
  Fiber* f = g.f.data();
  const Fiber& first = f[lanes[0]];
  bool in[64] = {false};
  for (int i : lanes) in[i - base] = true;
  switch (first.kind) {
    case K_BARRIER: break;
    case K_READLANE:
      for (int i : lanes) {
        const int s = (int)(f[i].arg & 63);
        if (s >= nw || !in[s]) {
          for (int q = 0; q < nw; q++)
            fprintf(stderr, "  lane %2d state %d kind %d site %p arg %u\n", q, f[base + q].state, (int)f[base + q].kind, f[base + q].site, f[base + q].arg);
          die("readlane from a lane that is not at this call site", &f[i], i - base);
        }
        f[i].result = f[base + s].val;
      }
      break;
    case K_READFIRSTLANE:
      for (int i : lanes) f[i].result = first.val;
      break;
    case K_BALLOT: {
      uint64_t m = 0;
      for (int i : lanes) if (f[i].val) m |= 1ull << (i - base);
      for (int i : lanes) f[i].result = m;
      break;
    }
    case K_DPP:
      for (int i : lanes) {
        const int l = i - base;
        const int ctrl = (int)(f[i].arg & 0xFFFF), rmask = (f[i].arg >> 16) & 15, bmask = (f[i].arg >> 20) & 15;
        const bool bound = (f[i].arg >> 24) & 1;
        if (!((rmask >> (l >> 4)) & 1) || !((bmask >> ((l >> 2) & 3)) & 1)) { f[i].result = f[i].val2; continue; }
        const int s = dpp_source(ctrl, l);
        if (s >= 0 && s < nw && in[s]) f[i].result = f[base + s].val;
        else f[i].result = bound ? 0u : f[i].val2;
      }
      break;
    case K_BPERMUTE:
      for (int i : lanes) {
        const int s = (int)(f[i].arg & 63);
        f[i].result = (s < nw && in[s]) ? f[base + s].val : 0u;
      }
      break;
    case K_PERMUTE: {
      uint32_t out[64] = {0};
      for (int i : lanes) out[f[i].arg & 63] = f[i].val;   // (highest lane wins, as on the hardware)
      for (int i : lanes) f[i].result = out[i - base];
      break;
    }
    default: die("bad collective kind");
  }
  for (int i : lanes) f[i].state = RUNNABLE;
}

// every live fiber is waiting: release what can be released
void resolve() {
  Fiber* f = g.f.data();
  bool any = false;
  // WAVESIM_SITE=0: lanes are matched by the kind of operation only (optimised builds may duplicate a call site)
  static const bool use_site = !(getenv("WAVESIM_SITE") && atoi(getenv("WAVESIM_SITE")) == 0);
  std::vector<int> lanes;
  for (int base = 0; base < g.n; base += 64) {
    const int nw = g.n - base < 64 ? g.n - base : 64;
    bool taken[64] = {false};
    int ngroups = 0;
    for (int a = 0; a < nw; a++) {
      const Fiber& fa = f[base + a];
      if (taken[a] || fa.state != WAITING || fa.kind == K_SYNCTHREADS) continue;
      lanes.clear();
      for (int b = a; b < nw; b++) {
        const Fiber& fb = f[base + b];
        if (!taken[b] && fb.state == WAITING && fb.kind == fa.kind && (fb.site == fa.site || !use_site)) { taken[b] = true; lanes.push_back(base + b); }
      }
      if (++ngroups > 1 && !getenv("WAVESIM_ALLOW_DIVERGENT")) {
        // The emulator cannot know where diverged lanes reconverge: kernels keep their cross-lane
        // operations in wave-uniform control flow (ballots inside a branch are written as a ballot of
        // "in the branch && predicate" in front of it).
        for (int q = 0; q < nw; q++) {
          Dl_info info; unsigned long off = 0;
          if (dladdr(f[base + q].site, &info) && info.dli_fbase) off = (unsigned long)((const char*)f[base + q].site - (const char*)info.dli_fbase);
          fprintf(stderr, "  lane %2d state %d kind %d site +0x%lx\n", q, f[base + q].state, (int)f[base + q].kind, off);
        }
        die("cross-lane operation in divergent control flow (lanes of one wave wait at different call sites)", &fa, a);
      }
      resolve_wave_group(base, nw, lanes);
      any = true;
    }
  }
  if (any) return;
  // only workgroup barriers are pending
  for (int i = 0; i < g.n; i++) if (f[i].state == WAITING) { f[i].state = RUNNABLE; any = true; }
  if (!any) die("deadlock: nothing to release");
}

// hands the OS thread to the next runnable fiber; `self` < 0: called from the launcher
void schedule(int self) {
  Fiber* f = g.f.data();
  for (;;) {
    int j = -1;
    for (int k = 1; k <= g.n; k++) {
      const int c = (g.cur + k) % g.n;
      if (f[c].state == RUNNABLE) { j = c; break; }
    }
    if (j < 0) {
      bool waiting = false;
      for (int i = 0; i < g.n; i++) if (f[i].state == WAITING) { waiting = true; break; }
      if (!waiting) {  // all done
        if (self < 0) return;
        g.cur = -1;
        void* dummy;
        ws_switch(self >= 0 ? &f[self].sp : &dummy, g.main_sp);
        die("resumed a finished fiber");
      }
      resolve();
      continue;
    }
    g.cur = j;
    g.ctx.threadIdx = Dim3((unsigned)j, 0, 0);
    if (j == self) return;
    ws_switch(self >= 0 ? &f[self].sp : &g.main_sp, f[j].sp);
    if (self >= 0) g.ctx.threadIdx = Dim3((unsigned)self, 0, 0);
    return;
  }
}

void fiber_entry() {
  const int me = g.cur;
  (*g.fn)();
  g.f[me].state = DONE;
  schedule(me);
  die("fiber_entry fell through");
}

void run_block(unsigned bx, Dim3 grid, Dim3 block, const std::function<void()>& fn) {
  const int n = (int)(block.x * block.y * block.z);
  if ((size_t)n > g.nstacks) {
    if (g.stacks) munmap(g.stacks, g.nstacks * kStack);
    g.nstacks = (size_t)n;
    g.stacks = (char*)mmap(nullptr, g.nstacks * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g.stacks == (char*)MAP_FAILED) die("mmap of fiber stacks failed");
  }
  g.f.assign((size_t)n, Fiber());
  g.n = n; g.cur = -1; g.fn = &fn;
  g.ctx.blockIdx = Dim3(bx, 0, 0); g.ctx.blockDim = block; g.ctx.gridDim = grid;
  for (int i = 0; i < n; i++) {
    uintptr_t top = ((uintptr_t)(g.stacks + (size_t)(i + 1) * kStack)) & ~(uintptr_t)15;
    void** s = (void**)top;
    s[-1] = nullptr;                 // fake return address of fiber_entry
    s[-2] = (void*)&fiber_entry;     // `ret` target of the first switch
    for (int k = 3; k <= 8; k++) s[-k] = nullptr;  // rbp rbx r12 r13 r14 r15
    g.f[i].sp = (void*)(s - 8);
    g.f[i].state = RUNNABLE;
  }
  schedule(-1);
}

}  // namespace

Ctx& ctx() { return g.ctx; }

uint64_t collective(Kind kind, const void* site, uint32_t val, uint32_t arg, uint32_t val2) {
  const int me = g.cur;
  if (me < 0) die("collective outside a kernel");
  Fiber& f = g.f[me];
  f.kind = kind; f.site = site; f.val = val; f.arg = arg; f.val2 = val2;
  f.state = WAITING;
  schedule(me);
  return g.f[me].result;
}

void launch(Dim3 grid, Dim3 block, const std::function<void()>& fn) {
  const unsigned nb = grid.x * grid.y * grid.z;
  unsigned nt = std::thread::hardware_concurrency();
  if (const char* e = getenv("WAVESIM_THREADS")) nt = (unsigned)atoi(e);
  if (nt < 1) nt = 1;
  if (nt > nb) nt = nb;
  if (nt <= 1) {
    for (unsigned b = 0; b < nb; b++) run_block(b, grid, block, fn);
    return;
  }
  std::atomic<unsigned> next{0};
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; t++)
    pool.emplace_back([&]() {
      for (;;) {
        const unsigned b = next.fetch_add(1);
        if (b >= nb) break;
        run_block(b, grid, block, fn);
      }
    });
  for (auto& t : pool) t.join();
}

}  // namespace wavesim
