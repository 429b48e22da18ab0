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
#endif
