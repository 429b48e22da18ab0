// LDS-DMA (global_load_lds_dwordx4) and explicit vmcnt waits, shared by the kernels that stream operands global -> LDS
// without touching VGPRs (kernels_conv3x3_x3.hip, kernels_convt_rs.hip).
#pragma once

// Every lane supplies a global address, the wave's 64 x 16 bytes land at (wave-uniform LDS address) + lane * 16.
#ifndef VP_GLOBAL_LOAD_LDS16  // the CPU emulation shim (tests/emul) provides its own
#define VP_GLOBAL_LOAD_LDS16(G, L)                                                                                          \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(G), (__attribute__((address_space(3))) void*)(L), 16, 0, 0)
// s_waitcnt vmcnt(N) only (lgkmcnt / expcnt left open); N is an immediate, 0..63.  vmcnt retires in issue order.
#define VP_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (15 << 8))
// Workgroup barrier that publishes LDS only.  __syncthreads() is fence(release, workgroup) + s_barrier + fence(acquire), and in a
// kernel that uses LDS-DMA the compiler implements that release as s_waitcnt vmcnt(0): EVERY barrier then drains every
// outstanding global load, store and DMA of the wave, whatever VP_WAIT_VMCNT asked for a few instructions earlier (seen in the
// ISA of kernels_conv3x3_x3.hip: 10 of 12 barriers).  Here the caller states what must have landed (VP_WAIT_VMCNT for the DMA
// tile the other waves are about to read); ds_write / ds_read are drained by lgkmcnt(0); the "memory" clobber keeps the
// compiler from moving memory operations across.
#define VP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// ---- hand-off between workgroups (stream-K slabs, split-K "last arriver"): MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility".  Per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores.
//   producer: plain stores; VP_DRAIN_VMEM() in every thread; __syncthreads(); ONE lane: VP_FENCE_RELEASE_AGENT(); VP_DRAIN_VMEM();
//             VP_FLAG_STORE(flag, v)
//   consumer: ONE lane polls VP_FLAG_LOAD(flag); VP_FENCE_ACQUIRE_AGENT(); __syncthreads(); plain loads
// VP_DRAIN_VMEM is INLINE ASM on purpose: a __builtin_amdgcn_s_waitcnt in front of the release lets the compiler prove the wave's vmcnt
// scoreboard empty and drop the wait behind buffer_wbl2, so the flag can overtake the write-back (ROCm 7.2 hazard, same section).
#define VP_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define VP_FENCE_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define VP_FENCE_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define VP_FLAG_STORE(P, V) __hip_atomic_store((P), (V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VP_FLAG_LOAD(P) __hip_atomic_load((P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define VP_FLAG_ADD(P, V) __hip_atomic_fetch_add((P), (V), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
