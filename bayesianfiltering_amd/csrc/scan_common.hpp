// Building blocks shared by the scan kernels (Kalman / Gaussian-sum): per-lane constant
// selection, wave-level LDS ordering, LDS-DMA loads with counted waits, and the LDS tile that
// transposes time for the reference output layout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "kf_math.hpp"

namespace bf {

enum { EMIT_SCALAR = 0, EMIT_NONE = 1, EMIT_STAGED = 2 };  // EMIT_NONE: no per-component streams compiled in (collapsed-only runs)

// table[idx * STRIDE + off] for a lane-dependent idx < CNT, as an unrolled select chain (the
// table is a kernel argument: a runtime index would copy it to scratch)
template <int CNT>
__device__ __forceinline__ float pick(const float* table, int limit, int idx, int stride, int off) {
  float r = (off < limit) ? table[off] : 0.f;
  BF_UNROLL for (int q = 1; q < CNT; ++q)
      if (q * stride + off < limit) r = (idx == q) ? table[q * stride + off] : r;
  return r;
}

// Order LDS traffic between the lanes of ONE wave: the hardware executes a wave's DS
// instructions in order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. it waits
// (s_waitcnt vmcnt(0)) until every output store the wave has in flight is acknowledged by HBM -- a microsecond
// per barrier in kernels whose phases talk through LDS and merely stream their results out.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ unsigned lds_byte_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

// One LDS-DMA dword per lane: LDS[lds_base + 4*lane] <- *src.  No VGPR destination, so the
// compiler neither tracks nor waits for it (cdna_hip_programming.md 5.7): completion is
// awaited with wait_vm(n) below.  M0 carries the wave-uniform LDS base.
__device__ __forceinline__ void lds_dma_dword(const float* src, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_base)
      : "memory");
}

// s_waitcnt vmcnt(n): wait until at most n of this wave's vector-memory operations (loads,
// stores and LDS-DMA count together, in issue order) are outstanding.  n is wave-uniform.
__device__ __forceinline__ void wait_vm(int n) {
#define BF_VMCASE(K_) case K_: asm volatile("s_waitcnt vmcnt(" #K_ ")" ::: "memory"); break;
  switch (n) {
    BF_VMCASE(1) BF_VMCASE(2) BF_VMCASE(3) BF_VMCASE(4) BF_VMCASE(5) BF_VMCASE(6) BF_VMCASE(7) BF_VMCASE(8)
    BF_VMCASE(9) BF_VMCASE(10) BF_VMCASE(11) BF_VMCASE(12) BF_VMCASE(13) BF_VMCASE(14) BF_VMCASE(15)
    BF_VMCASE(16) BF_VMCASE(17) BF_VMCASE(18) BF_VMCASE(19) BF_VMCASE(20) BF_VMCASE(21) BF_VMCASE(22)
    BF_VMCASE(23) BF_VMCASE(24) BF_VMCASE(25) BF_VMCASE(26) BF_VMCASE(27) BF_VMCASE(28) BF_VMCASE(29)
    BF_VMCASE(30) BF_VMCASE(31) BF_VMCASE(32) BF_VMCASE(33) BF_VMCASE(34) BF_VMCASE(35) BF_VMCASE(36)
    BF_VMCASE(37) BF_VMCASE(38) BF_VMCASE(39) BF_VMCASE(40)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef BF_VMCASE
}

// ---------------------------------------------------------------------------------------
// Per-wave LDS tile that transposes time for one output stream.
//   rows = the CPW trajectories of the wave; a row holds W = TS*E floats (TS consecutive steps)
//   at a pitch of W + PAD floats.  PAD = 4 keeps the per-step writes of the lanes of a
//   half-wave on different banks while rows stay 16-byte aligned for the ds_read_b128 of the
//   flush.  A flush writes every row's W*4 contiguous bytes with dwordx4 stores.
template <int E, int W, int CPW, int PAD>
struct Tile {
  static constexpr int TS = W / E;
  static constexpr int CH = W / 4;                     // 16-byte chunks per row
  static constexpr int PITCH = W + PAD;
  static constexpr int FLOATS = CPW * PITCH;
  static constexpr int NCHUNK = CPW * CH;              // chunks per tile
  static constexpr int ITER = (NCHUNK + 63) / 64;      // dwordx4 store instructions per flush
  // POW2: chunks per row a power of two and whole store instructions per flush -- lane -> (row, chunk) by shifts and
  // one precomputed lane offset (the layout of n in {1, 2, 4, 8}).  Otherwise (n = 3, 5, 6, 7: rows of 36, 100, 196 ...
  // floats) the same tile with chunk q = lane + 64 i -> row q / CH, chunk q % CH (constant divisions) and a masked tail.
  static constexpr bool POW2 = ((CH & (CH - 1)) == 0) && (NCHUNK >= 64) && (NCHUNK % 64 == 0);
  static constexpr bool GOK = (W % E == 0) && (W % 4 == 0) && (PAD % 4 == 0);
  static constexpr bool OK = GOK && POW2;

  // lane-dependent byte offset of this lane's first chunk relative to the wave's base (the host
  // guarantees that the rows of one wave span less than 4 GiB, so 32 bits are enough)
  static __device__ __forceinline__ unsigned lane_off(int lane, long long sB) {
    if constexpr (POW2) return (unsigned)((lane / CH) * sB * 4 + (lane % CH) * 16);
    else return 0u;
  }

  static __device__ __forceinline__ void read(const float* tile, int lane, float4* v) {
    if constexpr (POW2) {
      constexpr int RPI = 64 / CH;
      const int c = lane % CH;
      const int r0 = lane / CH;
      BF_UNROLL for (int i = 0; i < ITER; ++i)
          v[i] = *reinterpret_cast<const float4*>(tile + (r0 + i * RPI) * PITCH + c * 4);
    } else {
      BF_UNROLL for (int i = 0; i < ITER; ++i) {
        const int q = lane + 64 * i;
        const int qq = q < NCHUNK ? q : NCHUNK - 1;   // the tail lanes re-read the last chunk (never stored)
        const int r = qq / CH, c = qq - r * CH;
        v[i] = *reinterpret_cast<const float4*>(tile + r * PITCH + c * 4);
      }
    }
  }

  // dst_wave (wave-uniform): address of element (first trajectory of the wave, first step of the
  // row, e = 0).  Only chunks below chunk_limit are written (CH for a complete row).
  // Streaming output that this kernel never reads back: non-temporal stores (global_store_dwordx4 ... nt).  Measured on
  // the headline launch (107 GB per launch, same box, bench.py): 19.65 ms with plain stores, 18.77 ms with nt.
  static __device__ __forceinline__ void store16(char* p, const float4& v) {
    typedef float nt_v4f __attribute__((ext_vector_type(4)));
    const nt_v4f q = {v.x, v.y, v.z, v.w};
#ifdef BF_KF_PLAIN_STORES
    *reinterpret_cast<nt_v4f*>(p) = q;
#else
    __builtin_nontemporal_store(q, reinterpret_cast<nt_v4f*>(p));
#endif
  }
  static __device__ __forceinline__ void write(const float4* v, int lane, char* dst_wave, unsigned lane_byte_off,
                                               long long sB, int chunk_limit) {
    if constexpr (POW2) {
      constexpr int RPI = 64 / CH;
      BF_UNROLL for (int i = 0; i < ITER; ++i) {
        char* base_i = dst_wave + (size_t)i * (size_t)RPI * (size_t)sB * 4;  // uniform
        if (chunk_limit >= CH || (lane % CH) < chunk_limit) store16(base_i + lane_byte_off, v[i]);
      }
    } else {
      BF_UNROLL for (int i = 0; i < ITER; ++i) {
        const int q = lane + 64 * i;
        const int r = q / CH, c = q - r * CH;
        const unsigned off = (unsigned)(r * sB * 4 + c * 16);     // < 4 GiB: see lane_off
        if (q < NCHUNK && c < chunk_limit) store16(dst_wave + off, v[i]);
      }
    }
  }
};

}  // namespace bf
