// bf16 STORAGE of activation / gradient rows (training step, BASELINE config 5): tensors rest in HBM as bfloat16,
// every kernel widens to fp32 on load and rounds to nearest-even on store (v_cvt_pk_bf16_f32); all arithmetic and
// every accumulation stay fp32.  Accesses are 4, 8 or 16 bytes wide: sub-dword global accesses run at about half
// the rate of their dword forms on this target.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace vp {

typedef uint16_t bf16_t;
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
typedef float f32x2_native __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }        // element 0 of a pair
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }  // element 1
__device__ __forceinline__ unsigned pack_bf16x2(float e0, float e1) {
  const f32x2_native v = {e0, e1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_native));
}
__device__ __forceinline__ bf16_t to_bf16(float v) { return (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float from_bf16(bf16_t h) { return __uint_as_float((unsigned)h << 16); }

// Element access by storage type.  load8 / store8: eight consecutive elements at an 8-element-aligned index
// (16 bytes of bf16, two 16-byte accesses of fp32); load2 / store2: an even-aligned pair.
template <class T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void load2(const float* p, float (&v)[2]) {
    const float2 r = *reinterpret_cast<const float2*>(p);
    v[0] = r.x, v[1] = r.y;
  }
  static __device__ __forceinline__ void store2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
  static __device__ __forceinline__ void load8(const float* p, float* v) {
    const float4 r0 = *reinterpret_cast<const float4*>(p), r1 = *reinterpret_cast<const float4*>(p + 4);
    v[0] = r0.x, v[1] = r0.y, v[2] = r0.z, v[3] = r0.w, v[4] = r1.x, v[5] = r1.y, v[6] = r1.z, v[7] = r1.w;
  }
  static __device__ __forceinline__ void store8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return from_bf16(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = to_bf16(v); }
  static __device__ __forceinline__ void load2(const bf16_t* p, float (&v)[2]) {
    const unsigned r = *reinterpret_cast<const unsigned*>(p);
    v[0] = bf16_lo(r), v[1] = bf16_hi(r);
  }
  static __device__ __forceinline__ void store2(bf16_t* p, float a, float b) { *reinterpret_cast<unsigned*>(p) = pack_bf16x2(a, b); }
  static __device__ __forceinline__ void load8(const bf16_t* p, float* v) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = bf16_lo(r.x), v[1] = bf16_hi(r.x), v[2] = bf16_lo(r.y), v[3] = bf16_hi(r.y);
    v[4] = bf16_lo(r.z), v[5] = bf16_hi(r.z), v[6] = bf16_lo(r.w), v[7] = bf16_hi(r.w);
  }
  static __device__ __forceinline__ void store8(bf16_t* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                              pack_bf16x2(v[6], v[7]));
  }
};

}  // namespace vp
