// wavesim.hpp -- TEST INFRASTRUCTURE ONLY.  A CPU emulation of the wave-level execution model the
// engine's kernels are written against, so that the product's OWN kernel sources (robopianist_amd/csrc)
// can be exercised against the oracle on a machine without a GPU.  Nothing in the product path loads
// what is built from this directory: engine.load_library() only ever opens csrc/librp_engine.so unless a
// TEST points RP_ENGINE_LIB here.
//
// Model: one workgroup = N fibers (one per work-item) on one OS thread, switched cooperatively.  A
// work-item runs until it reaches a cross-lane operation (readlane, DPP, ballot, bpermute, wave barrier,
// __syncthreads); when every live work-item of the group has arrived, the operation is resolved for the
// lanes that sit at the same call site and they continue.  That is stricter than the hardware in one
// way (LDS write -> read hand-overs between lanes need a wave barrier / WSYNC in between, which the
// kernels have anyway) and laxer in none that the kernels rely on, with one caveat: a read-then-write
// hazard between lanes that the hardware resolves by lockstep execution needs a barrier here.
#pragma once
#include <cstdint>
#include <cstddef>
#include <functional>

namespace wavesim {

struct Dim3 {
  unsigned x, y, z;
  Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct Ctx {           // per OS thread: the work-item currently running
  Dim3 threadIdx, blockIdx, blockDim, gridDim;
};
Ctx& ctx();

enum Kind : int {
  K_BARRIER = 1,     // wave barrier (WSYNC)
  K_SYNCTHREADS,     // workgroup barrier
  K_READLANE,        // arg = source lane
  K_READFIRSTLANE,
  K_BALLOT,          // val = predicate
  K_DPP,             // arg = ctrl | row_mask << 16 | bank_mask << 20 | bound_ctrl << 24 ; old in val2
  K_BPERMUTE,        // arg = source lane (per lane)
  K_PERMUTE,         // arg = destination lane (per lane)
};

// Blocks until every live lane of the wave (workgroup for K_SYNCTHREADS) has arrived, returns this
// lane's result.
uint64_t collective(Kind kind, const void* site, uint32_t val, uint32_t arg, uint32_t val2 = 0);

// Runs kernel body `fn` for grid x block work-items (blocks are spread over a few OS threads).
void launch(Dim3 grid, Dim3 block, const std::function<void()>& fn);

}  // namespace wavesim
