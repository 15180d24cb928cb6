// LAKO_DETERMINISTIC=1 (round 6): bit-reproducible accumulation at the sites where workgroups (or the waves of one) add floats into a shared
// location in whatever order they arrive.  Float addition is not associative, 64-bit integer addition is: in this mode such a site adds
// its contribution as a FIXED-POINT integer (value · 2^40, rounded once; exact for |value| < 2^23 down to 2^-40) by an integer atomic into
// a small static table of its code object, and the workgroup that arrives last (a ticket) converts every slot back and adds it to the
// float target with ONE rounding — the result does not depend on the arrival order.  The default build path is untouched (float
// atomics: fire-and-forget, nobody waits); the price of the mode is in DESIGN.md §4.
// The table is per translation unit and shared by every launch of that unit's kernels: launches of ONE stream run one after the other,
// which is all the reader does; two engines training on two streams of one process must not both use the mode.
#pragma once
#include <stdlib.h>

namespace lako_det {

constexpr int SLOTS = 16384;
constexpr double SCALE = 1099511627776.0;      // 2^40

inline bool on() {
  static const bool v = [] {
    const char* e = getenv("LAKO_DETERMINISTIC");
    return e && e[0] == '1';
  }();
  return v;
}
__device__ __forceinline__ unsigned long long to_fixed(float v) { return (unsigned long long)__double2ll_rn((double)v * SCALE); }
__device__ __forceinline__ float from_fixed(unsigned long long s) { return (float)((double)(long long)s * (1.0 / SCALE)); }
__device__ __forceinline__ void add(unsigned long long* slot, float v) { atomicAdd(slot, to_fixed(v)); }

// Call with EVERY thread of the workgroup, after its adds: true in all threads of the workgroup that arrived last (every add of the
// launch is then visible to its atomic reads), which must call `take` for each slot it owns and nothing else afterwards.
__device__ __forceinline__ bool last_block(unsigned int* ticket, unsigned int nblocks, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's atomics have been performed (at the memory side: one point of coherence)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_flag = t == nblocks - 1;
    if (t == nblocks - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
  __syncthreads();
  return *lds_flag != 0;
}
__device__ __forceinline__ float take(unsigned long long* slot) {      // the slot's sum, leaving it zero for the next launch
  return from_fixed(__hip_atomic_exchange(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

}  // namespace lako_det

#define LAKO_DET_STORAGE                                                   \
  static __device__ unsigned long long lako_det_slots[lako_det::SLOTS];    \
  static __device__ unsigned int lako_det_ticket;
