// Shared host/device declarations of libvolpick_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/volpick_hip.h"

namespace vp {

// Every activation row in device memory is [HALO zeros][L samples][zeros up to ls]:
// conv tile loaders read raw float4s with no bounds checks and the zero margins ARE the
// convolution padding.  Producers only ever write logical samples [0, L).
constexpr int HALO = 8;

typedef float f32x4 __attribute__((ext_vector_type(4)));

void set_error(const char* fmt, ...);

#define VP_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      vp::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VP_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define VP_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      vp::set_error(__VA_ARGS__); \
      return VP_ERR_INVALID;      \
    }                             \
  } while (0)

struct ParamDesc {
  const char* name;
  int shape[3];
  int ndim;
  size_t size;
};

// Flat fp32 blob in canonical order + name lookup.
struct ParamView {
  std::map<std::string, std::pair<const float*, const ParamDesc*>> by_name;
  const float* get(const std::string& name, const ParamDesc** d = nullptr) const;
};
int param_table(int model_kind, const ParamDesc** table);
bool build_param_view(int model_kind, const float* blob, size_t n_floats, ParamView* out);

// Activation tensor [capacity windows][C][ls] fp32 with zero halos.
struct Tensor {
  std::string name;
  float* p = nullptr;
  int C = 0;   // channels
  int L = 0;   // logical length
  int ls = 0;  // row stride in floats (multiple of 4), >= HALO + L + right margin
  int need = 0;  // max physical index any consumer reads + 1 (plan-time bookkeeping)
  size_t win_stride() const { return (size_t)C * ls; }
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int floor4(int x) { return (x >= 0) ? (x / 4) * 4 : -(((-x) + 3) / 4) * 4; }

}  // namespace vp
