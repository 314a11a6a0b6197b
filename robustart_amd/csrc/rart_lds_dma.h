// LDS-DMA through a buffer resource (gfx950), shared by the split-bf16 GEMM kernels (csrc/gemm_pair.hip, gemm_pair_pp.hip, conv_tail_pair.hip).
//
// Round 6 finding (profiles/r06_pp_lab.json, scratch/r6/pp_lab.py): a `global_load_lds_dwordx4` whose 64-bit per-lane address is formed with
// vector instructions (tap offset add, in-image select against a zero page, plane delta) costs the issuing wave 170-230 cycles when another
// wave of its SIMD is issuing MFMAs -- the address VALU competes with the matrix instructions for the SIMD's vector issue -- against ~100
// alone: the stage loads of a K step, not the matrix work, paced k_gemm_pair.  `buffer_load_dwordx4 ... offen lds` takes the address as a
// wave-uniform resource + ONE per-lane 32-bit offset + a scalar offset, so a K step (and a tap) only move scalars; and the resource's range
// check ZERO-FILLS a lane whose offset lies beyond num_records, which replaces the zero page (rows past M, pixels outside the image).
// scratch/r6/bl_test.hip printed both rules on the MI355X: LDS address = M0 + 16 * lane; voffset + soffset >= num_records -> zeros.
#pragma once
#include <stdint.h>

typedef unsigned int rart_srd_t __attribute__((ext_vector_type(4)));   // base lo, base hi (16 bits; stride 0), num_records, flags

// a byte offset no operand plane reaches: the host entries refuse planes of 2 GiB and more
#define RART_DMA_OOR 0x80000000u

// The resource of a plane starting at `base`.  readfirstlane: the words must be PROVABLY wave-uniform to be allocated to SGPRs.
__device__ __forceinline__ rart_srd_t rart_dma_srd(const void* base) {
  const unsigned long long b = (unsigned long long)base;
  rart_srd_t r = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu)),
                  0x7FFFFFFFu, 0x00020000u};
  return r;
}
// One 1 KiB piece: lane l's 16 bytes at srd.base + voff + soff land at lds_addr + 16 l.  Inline asm: hipcc neither counts the load (no
// vmcnt(0) of its own at barriers or LDS reads) nor moves it; M0 (the DMA's LDS base) is written in the same statement.  The caller owns
// the `s_waitcnt vmcnt(N)` + barrier that make the data visible.
__device__ __forceinline__ void rart_dma_load16(uint32_t voff, rart_srd_t srd, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}
// the same with the non-temporal cache policy: for operands no other workgroup reads (ViT's per-(image, head) K / V)
__device__ __forceinline__ void rart_dma_load16_nt(uint32_t voff, rart_srd_t srd, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds" : : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void rart_dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
