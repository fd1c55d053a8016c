// Host-visible plans / argument blocks shared by the translation units of libbanet_hip.so.
#pragma once
#include "common.hpp"

namespace banet {

// ---- assemble.hip --------------------------------------------------------------------
struct AsmPlan {
  int G, tiles, tiles_x, tiles_y, P, pstride, nb;
  size_t partial_bytes;
};
int plan_assemble(const banet_level_t* lv, AsmPlan* pl);
int launch_assemble(const banet_level_t* lv, const AsmPlan& pl, const float* R, const float* T, const float* Wc,
                    const int32_t* active, int active_stride, float* partials, float* AtA, float* Atb, float* absres,
                    float* nvalid, hipStream_t s);
void launch_reduce(const float* partials, const int32_t* active, int active_stride, int B, int G, int pstride, int P,
                   int C, float* AtA, float* Atb, float* absres, float* nvalid, hipStream_t s);

// optional launch timing (banet_profile_begin/_end)
int profile_begin(int max_launches);
int profile_end(int max_tags, int32_t* tag_points, int32_t* tag_launches, double* tag_ms, int32_t* ntags);

// ---- eqcon.hip -----------------------------------------------------------------------
struct EqPlan {
  int Gr, tiles, pstride, nb;
  size_t partial_bytes;
};
int plan_eq(int B, int N, int C, int P, EqPlan* pl);
int launch_eq(const float* J, const float* G, const float* d, float* AtA, float* Atb, int B, int N, int C, int P,
              const EqPlan& pl, float* partials, hipStream_t s);
int launch_eq_grad(const float* J, const float* G, const float* d, const float* g0, const float* g1, float* gJ,
                   float* gG, float* gd, int B, int N, int C, int P, hipStream_t s);

// ---- solve.hip -----------------------------------------------------------------------
struct LmCtl {  // per-window loop state of the legacy early-termination LM (device memory)
  int32_t active;
  int32_t pending;
  float avg_prev;
  float uw, ut;
  float Rprev[9];
  float Tprev[3];
  int32_t pad_[3];
};
static_assert(sizeof(LmCtl) == 80, "LmCtl layout");

struct SolveArgs {
  int B, N, C, K, P, variant;
  float l2_base;
  int max_iters;  // only used with ctl
  banet_mlp_t mlp;
  int use_mlp;
  const float* AtA;
  const float* Atb;
  const float* absres;
  const float* nvalid;
  banet_state_t st;
  LmCtl* ctl;  // nullptr: fixed-count mode (every call performs one update)
};
int launch_solve(const SolveArgs& a, hipStream_t s);
void launch_ctl_init(LmCtl* ctl, int32_t* iters, int B, hipStream_t s);
void launch_zero_iters(int32_t* iters, int B, hipStream_t s);

}  // namespace banet
