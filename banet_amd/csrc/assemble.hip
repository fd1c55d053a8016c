// One BA assembly pass = a gather kernel (gather*.hip; HBM-bound: warp, sample, residual, 2x2 channel
// reductions, pose block) + ba_fold_kernel + a SYRK kernel (syrk*.hip; depth-basis blocks on the matrix cores) +
// ba_reduce2_kernel (fixed-order sum of the per-tile / per-workgroup partials).  Also hosts the optional
// launch timer behind banet_profile_begin/_end.
#include <atomic>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

namespace banet {

namespace {
struct Profiler {
  bool on = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> tag;
  int used = 0;
} g_prof;

struct Timed {  // brackets one kernel launch with two events when profiling is enabled
  hipStream_t s;
  int slot;
  Timed(hipStream_t st, int tag) : s(st), slot(-1) {
    if (g_prof.on && (size_t)(2 * g_prof.used + 1) < g_prof.ev.size()) {
      slot = g_prof.used++;
      g_prof.tag[slot] = tag;
      (void)hipEventRecord(g_prof.ev[2 * slot], s);
    }
  }
  ~Timed() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
  }
};
}  // namespace

// ---- roctx ranges (banet_profile_ranges / BANET_ROCTX=1): the marker library is loaded on first use, never linked ----
namespace {
struct Roctx {
  int state = -1;                       // -1: not probed, 0: off / unavailable, 1: on
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool load() {
    if (push) return true;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return true;
      push = nullptr;
    }
    return false;
  }
} g_roctx;
}  // namespace

int profile_ranges(int enable) {
  g_roctx.state = (enable && g_roctx.load()) ? 1 : 0;
  return g_roctx.state;
}

RangeScope::RangeScope(const char* role, int n) : on(false) {
  if (g_roctx.state < 0) {
    const char* e = getenv("BANET_ROCTX");
    profile_ranges(e && e[0] == '1');
  }
  if (g_roctx.state != 1) return;
  char buf[64];
  if (n >= 0) snprintf(buf, sizeof buf, "banet.%s N=%d", role, n);
  else snprintf(buf, sizeof buf, "banet.%s", role);
  g_roctx.push(buf);
  on = true;
}
RangeScope::~RangeScope() {
  if (on) g_roctx.pop();
}

int profile_begin(int max_launches) {
  if (g_prof.on || max_launches <= 0) return BANET_ERR_INVALID_ARG;
  g_prof.ev.resize((size_t)2 * max_launches);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) return BANET_ERR_LAUNCH;
  g_prof.tag.assign(max_launches, 0);
  g_prof.used = 0;
  g_prof.on = true;
  return BANET_OK;
}

int profile_end(int max_tags, int32_t* tag_points, int32_t* tag_launches, double* tag_ms, int32_t* ntags) {
  if (!g_prof.on || !tag_points || !tag_launches || !tag_ms || !ntags) return BANET_ERR_INVALID_ARG;
  int nt = 0;
  for (int i = 0; i < g_prof.used; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return BANET_ERR_LAUNCH;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    int k = 0;
    while (k < nt && tag_points[k] != g_prof.tag[i]) ++k;
    if (k == nt) {
      if (nt == max_tags) continue;
      tag_points[nt] = g_prof.tag[i];
      tag_launches[nt] = 0;
      tag_ms[nt] = 0.0;
      ++nt;
    }
    tag_launches[k] += 1;
    tag_ms[k] += ms;
  }
  *ntags = nt;
  for (auto& e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.on = false;
  return BANET_OK;
}

int num_cus() {
  constexpr int kMaxDev = 64, kDefault = 256;
  static std::atomic<int> cached[kMaxDev];        // 0 = not queried yet
  // BANET_NUM_CUS: pin the CU count the plans are made for (the selection-table test pins 256 so that it does not depend on the
  // host's GPU; also a way to reproduce another part's plans).  Read once.
  static const int pinned = [] {
    const char* e = std::getenv("BANET_NUM_CUS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 && v <= 4096 ? v : 0;
  }();
  if (pinned) return pinned;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
    (void)hipGetLastError();
    return kDefault;
  }
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = kDefault;
    }
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

int plan_assemble(const banet_level_t* lv, AsmPlan* pl) {
  int rc = plan_gather(lv, &pl->g);
  if (rc != BANET_OK) return rc;
  rc = plan_syrk(lv->B, selection_batch(lv), lv->N, lv->K, npairs(lv), lv->flags, &pl->s);
  if (rc != BANET_OK) return rc;
  pl->P = 6 * npairs(lv) + lv->K;
  pl->off_rec = pl->g.partial_bytes;
  pl->off_spart = pl->off_rec + pl->g.rec_bytes;
  pl->ws_bytes = pl->off_spart + pl->s.partial_bytes;
  return BANET_OK;
}

int* assemble_queue(const AsmPlan& pl, void* ws) {
  return pl.g.c128 ? reinterpret_cast<int*>(static_cast<char*>(ws) + pl.g.off_queue) : nullptr;
}

// tags for the profiler: +N = gather kernel at a level with N points, -N = syrk kernel
int launch_assemble(const banet_level_t* lv, const AsmPlan& pl, const float* R, const float* T, const float* Wc,
                    const int32_t* active, int active_stride, void* ws, float* AtA, float* Atb, float* absres,
                    float* nvalid, hipStream_t s, bool reset_queue, const banet_mlp_t* role_mlp, float* role_y,
                    unsigned char* mask_out, int f16_stats) {
  char* base = static_cast<char*>(ws);
  float* gpart = reinterpret_cast<float*>(base);
  float* rec = lv->K > 0 ? reinterpret_cast<float*>(base + pl.off_rec) : nullptr;
  float* spart = lv->K > 0 ? reinterpret_cast<float*>(base + pl.off_spart) : nullptr;
  int rc;
  if (reset_queue) prepare_gather(lv, pl.g, gpart, s);
  {
    RangeScope r("gather", lv->N);
    Timed t(s, lv->N);
    rc = launch_gather(lv, pl.g, R, T, Wc, active, active_stride, rec, gpart, s, mask_out);
  }
  if (rc != BANET_OK) return rc;
  const float* gred;
  {
    RangeScope r("fold", lv->N);
    gred = finish_gather(lv, pl.g, active, active_stride, gpart, s);
  }
  if (lv->K > 0) {
    RangeScope r("syrk", lv->N);
    Timed t(s, -lv->N);
    MlpRole mr{};
    if (role_mlp != nullptr && role_y != nullptr && syrk_runs_mlp_role(pl.s)) {
      mr.gpart = gred;
      mr.grows = pl.g.frows;
      mr.gstride = pl.g.pstride;
      mr.mlp = *role_mlp;
      mr.C = lv->C;
      mr.pairs = npairs(lv);
      mr.Nf = (float)lv->N * (float)npairs(lv);
      mr.y = role_y;
    }
    rc = launch_syrk(lv->basis, rec, lv->B, lv->N, lv->K, npairs(lv), pl.s, active, active_stride, spart, s,
                     mr.y != nullptr ? &mr : nullptr, f16_stats);
    if (rc != BANET_OK) return rc;
  }
  RangeScope r("reduce", lv->N);
  launch_reduce2(gred, pl.g.frows, pl.g.pstride, spart, pl.s.Gs, pl.s.pstride, active, active_stride, lv->B, lv->K, lv->C,
                 npairs(lv), AtA, Atb, absres, nvalid, s);
  return hipGetLastError() == hipSuccess ? BANET_OK : BANET_ERR_LAUNCH;
}

}  // namespace banet
