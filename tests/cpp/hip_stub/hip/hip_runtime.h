// Test-only stand-in for <hip/hip_runtime.h>: just enough of the HIP device
// vocabulary for g++ to compile pico_tree_amd/csrc/ptk_kernels.hpp as ordinary
// host functions, so that tests/cpp/emulate_kernels.cpp can run the REAL kernel
// source lane by lane on a CPU-only machine (`-m "not gpu"` tests).  It is never
// on the include path of the product build.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct uint2 { uint32_t x, y; };
struct uint3 { uint32_t x, y, z; };
struct uint4 { uint32_t x, y, z, w; };
struct float2 { float x, y; };
struct double2 { double x, y; };
struct double4 { double x, y, z, w; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct dim3 { uint32_t x = 1, y = 1, z = 1; };

inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline uint3 make_uint3(uint32_t x, uint32_t y, uint32_t z) { return uint3{x, y, z}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }

// One rounding per operation; the emulator is built with -ffp-contract=off.
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dsub_rn(double a, double b) { return a - b; }
inline double __dmul_rn(double a, double b) { return a * b; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline int32_t __float_as_int(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }
inline double __longlong_as_double(long long i) { double f; std::memcpy(&f, &i, 8); return f; }

// Wave-level collectives.  The emulator runs the 64 lanes of a wavefront as 64
// host threads; __ballot is the rendezvous (see emulate_kernels.cpp).
unsigned long long emu_ballot(bool pred);
inline unsigned long long __ballot(bool pred) { return emu_ballot(pred); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
int emu_shfl(int v, int src_lane);
inline int __shfl(int v, int src_lane) { return emu_shfl(v, src_lane); }
inline float __shfl(float v, int src_lane) {
  int i;
  std::memcpy(&i, &v, 4);
  i = emu_shfl(i, src_lane);
  std::memcpy(&v, &i, 4);
  return v;
}
int emu_lane();
inline int __shfl_up(int v, int delta) { const int l = emu_lane(); return emu_shfl(v, l >= delta ? l - delta : l); }
inline float __shfl_up(float v, int delta) { const int l = emu_lane(); return __shfl(v, l >= delta ? l - delta : l); }
inline int __shfl_xor(int v, int mask) { return emu_shfl(v, emu_lane() ^ mask); }
inline float __shfl_xor(float v, int mask) { return __shfl(v, emu_lane() ^ mask); }
inline double __shfl(double v, int src_lane) {  // (two rendezvous: every lane of the wavefront calls both)
  int w[2];
  std::memcpy(w, &v, 8);
  w[0] = emu_shfl(w[0], src_lane);
  w[1] = emu_shfl(w[1], src_lane);
  std::memcpy(&v, w, 8);
  return v;
}
inline double __shfl_xor(double v, int mask) { return __shfl(v, emu_lane() ^ mask); }
inline long long __double_as_longlong(double f) { long long i; std::memcpy(&i, &f, 8); return i; }
inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_shfl(v, lane); }
inline int __builtin_amdgcn_readfirstlane(int v) { return emu_shfl(v, 0); }  // all lanes alive where it is used
void emu_barrier();
inline void __syncthreads() { emu_barrier(); }  // a rendezvous of the wavefront (for_each_wave) or of the block (for_each_block)

// Lanes run one after the other (or as cooperative fibers), so a plain read-modify-write is atomic.
template <typename T>
inline T atomicAdd(T* p, T v) {
  T old = *p;
  *p = old + v;
  return old;
}

template <typename T>
inline T atomicMax(T* p, T v) {
  T old = *p;
  if (v > old) *p = v;
  return old;
}

extern thread_local dim3 threadIdx;
extern thread_local dim3 blockIdx;
extern thread_local dim3 gridDim;
extern thread_local dim3 blockDim;
