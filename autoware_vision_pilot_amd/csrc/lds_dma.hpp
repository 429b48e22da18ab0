// LDS-DMA (global_load_lds_dwordx4) and explicit vmcnt waits, shared by the kernels that stream operands global -> LDS
// without touching VGPRs (kernels_conv3x3_x3.hip, kernels_convt_rs.hip).
#pragma once

// Every lane supplies a global address, the wave's 64 x 16 bytes land at (wave-uniform LDS address) + lane * 16.
#ifndef VP_GLOBAL_LOAD_LDS16  // the CPU emulation shim (tests/emul) provides its own
#define VP_GLOBAL_LOAD_LDS16(G, L)                                                                                          \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(G), (__attribute__((address_space(3))) void*)(L), 16, 0, 0)
// s_waitcnt vmcnt(N) only (lgkmcnt / expcnt left open); N is an immediate, 0..63.  vmcnt retires in issue order.
#define VP_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (((N) >> 4) << 14) | (7 << 4) | (15 << 8))
// s_waitcnt lgkmcnt(N) only (vmcnt / expcnt left open); N = how many of the NEWEST LDS operations may stay outstanding (they return in issue
// order), 0..15.  Why it exists (round 4, seen in the ISA): with fragments prefetched one tap ahead the compiler places `s_waitcnt lgkmcnt(0)`
// in front of the current tap's MFMAs, i.e. it drains the prefetch it has just issued -- stated through the builtin (which its scoreboard
// models), "all but the newest N" replaces that drain.
// The s_nop behind it is what makes the compiler COMMIT the wait to its scoreboard: a pre-existing s_waitcnt is merged with whatever the NEXT
// instruction requires, and with an LDS-DMA in flight ("pending flat") that requirement is always lgkmcnt(0) -- min(N, 0) = 0 (measured in the
// ISA: the explicit lgkmcnt(8) in front of an MFMA group came out as lgkmcnt(0) on every other tap).  The s_nop requires nothing, so the wait
// is applied as written, and the MFMAs behind it find their operands complete in the model.
#define VP_WAIT_LGKMCNT(N)                                                                  \
  do {                                                                                      \
    __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (((N) & 15) << 8));               \
    asm volatile("s_nop 0");                                                                \
  } while (0)
// Workgroup barrier that publishes LDS only.  __syncthreads() is fence(release, workgroup) + s_barrier + fence(acquire), and in a
// kernel that uses LDS-DMA the compiler implements that release as s_waitcnt vmcnt(0): EVERY barrier then drains every
// outstanding global load, store and DMA of the wave, whatever VP_WAIT_VMCNT asked for a few instructions earlier (seen in the
// ISA of kernels_conv3x3_x3.hip: 10 of 12 barriers).  Here the caller states what must have landed (VP_WAIT_VMCNT for the DMA
// tile the other waves are about to read); ds_write / ds_read are drained by lgkmcnt(0); the "memory" clobber keeps the
// compiler from moving memory operations across.
// Round 4: the wait is ALSO stated through the builtin, which the compiler's scoreboard models (the asm is opaque to it).  Without that it
// believed the reads drained here were still pending, and -- an LDS-DMA in flight makes every lgkm wait it derives a full lgkmcnt(0) (the
// instruction is FLAT-encoded and touches two address spaces: LLVM's "pending flat" rule) -- it drained the NEXT prefetch, issued right
// behind the barrier, in front of MFMAs that did not need it.
#define VP_LDS_BARRIER()                                 \
  do {                                                   \
    VP_WAIT_LGKMCNT(0);                                  \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
  } while (0)
#endif
