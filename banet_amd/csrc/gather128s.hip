// ba_gather128s_kernel -- the strip gather: the gather pass for C = 128 on large dense levels with the target map read
// (almost) once.  The patch kernel (gather128p.hip) runs at the fabric's practical rate but fetches the target map 2.15 x:
// a wave's 8x8 tile touches (8 + 3)^2 texels and nothing of that halo survives in the L2 until a neighbouring tile wants it.
// Here a wave owns a STRIP SEGMENT of 16 x 16 source pixels (16 x 32 behind a switch: fewer halo rows, but half as many work
// items per wave and a longer tail -- measured slower, profiles/r03_run11_*) and shares the halo with itself:
//
//   * the channel loop is outermost: four passes over the segment, one per 32-channel slice (a texel's slice is one full
//     128-byte line), each with a ROLLING WINDOW of the last 7 texel rows x 21 texels of the target map in the wave's own LDS
//     (18.4 KB), filled by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) as far ahead as the ring allows and
//     consumed behind COUNTED s_waitcnt vmcnt(n) -- the per-step row counts and wait counts depend only on the segment's
//     geometry, so they are planned once per segment (strip_plan.hpp, unit-tested on the host) and replayed by the 4 slices;
//     target fetch = 20/16 x 19/16 = 1.48 texels per pixel instead of 2.15 (counters: the launch moves 1.12-1.19 x its
//     algorithmic bytes, the patch kernel 1.43-1.47 x: profiles/pmc_traffic.json);
//   * what has to live across the slices is per-pixel state of the whole segment (tap parameters, depth, the five channel
//     sums): 9 registers x 4 (8) chunks of 64 pixels, held as lane = pixel in a TRANSPOSED order (lane 4 p + k <-> pixel (row k,
//     column p) of a chunk of 4 rows x 16), so that the tap phase -- one pixel ROW per step, 4 lanes per pixel, 8 channels per
//     lane -- finds a pixel's parameters inside its own quad (one DPP quad_perm broadcast, no LDS) and leaves the pixel's
//     sums, after a 2-step DPP reduction over the quad, on exactly the lane that owns the pixel: no staging tables, no
//     transposition through LDS.  (A first version ran 8 lanes per pixel, 4 channels per lane, two instruction groups per
//     step: parity-green, 1.12 x the algorithmic bytes by the counters -- and VALU-bound at 433 instructions per step, because
//     the per-group overhead (weights, addresses, reductions) was paid per 4 channels: profiles/r03_run3_*.)
//   * still no workgroup barrier (a workgroup is 2 independent waves), still one partial row per work item in a fixed place
//     and fixed summation orders: bit-reproducible run to run;
//   * pixel rows whose footprint does not fit the window (local scale > ~1.12, a depth discontinuity) read their taps
//     directly; pixels on the image rim take the generic slow routine (as in the other C = 128 kernels).
// Same arithmetic per pixel as ba_gather128_kernel; the channel sums are added slice by slice (different rounding order).
#include <cstdlib>

#include "quad_common.hpp"
#include "strip_plan.hpp"

#ifndef BANET_STRIP_OPT_DEFAULT
#define BANET_STRIP_OPT_DEFAULT 0
#endif

namespace banet {

constexpr int kC128s = 128;
constexpr int kSBlock = 128;                       // 2 waves per workgroup, 4 workgroups per CU (LDS: 2 x 19.9 KB)
constexpr int kSWaves = kSBlock / kWave;
constexpr int kWinFloats = kWinRows * kWinTex * 32;
constexpr int kWinPitchF = kWinTex * 32;           // floats per window row

// lane `l` of `old` replaced by the wave-uniform value `v` (v_writelane_b32 cannot take two different SGPR operands:
// constant-bus limit; a compare + select does it)
__device__ __forceinline__ int write_lane(int v, int l, int old) { return lane_id() == l ? v : old; }

// the rolling-window plan with one pixel row per lane (lanes 0..31) and the replay loop on the scalar unit
struct StripLaneEnv {
  int modev, ytv, ybv;     // static mode, lowest / highest texel row read (lane r = pixel row r)
  int ctlv, yfv;           // result: StripStep.ctl / .yfirst of row r on lane r
  int ringv;               // ring slot i -> sequence number of its last load, on lane i
  __device__ __forceinline__ int mode(int r) const { return __builtin_amdgcn_readlane(modev, r); }
  __device__ __forceinline__ int yt(int r) const { return __builtin_amdgcn_readlane(ytv, r); }
  __device__ __forceinline__ int yb(int r) const { return __builtin_amdgcn_readlane(ybv, r); }
  __device__ __forceinline__ void set_step(int r, int ctl, int yfirst) {
    ctlv = write_lane(ctl, r, ctlv);
    yfv = write_lane(yfirst, r, yfv);
  }
  __device__ __forceinline__ int ring_get(int i) const { return __builtin_amdgcn_readlane(ringv, i); }
  __device__ __forceinline__ void ring_set(int i, int v) { ringv = write_lane(v, i, ringv); }
};

#ifndef BANET_STRIP_NT     // development: 1 = source features with the nt hint, 2 = window rows too (A/B)
#define BANET_STRIP_NT 0
#endif
#if BANET_STRIP_NT >= 2
#define BANET_WIN_NT_ " nt"
#else
#define BANET_WIN_NT_ ""
#endif
// ---- asynchronous loads the compiler does not see (its s_waitcnt bookkeeping would drain them) ----------------------
// 1 KB from global memory straight into LDS: lane i's 16 bytes go to lds_dst + 16 i (wave-uniform M0 base + lane x 16)
__device__ __forceinline__ void glds16(const float* gbase, unsigned voff_bytes, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" BANET_WIN_NT_ "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff_bytes), "s"(gbase), "s"(lds_dst)
               : "memory");
}
// One window row = three LDS-DMA instructions with ONE M0 write: the instruction offset is added to the LDS address AND to the
// global address, so instruction k (LDS dst + 1024 k, global row + 4096 k: the next 8 texels of 512 B) takes inst_offset 1024 k and
// a lane offset of its own, voff + 3072 k.  The third instruction runs under `mask3` (only the texel columns a row reads).
// (development variant OPT bit 1; the product path issues three glds16, 5 scalar instructions each)
__device__ __forceinline__ void glds_row(const float* gbase, unsigned v0, unsigned v1, unsigned v2, unsigned lds_dst,
                                         unsigned long long mask3) {
  unsigned keep;
  unsigned long long keepx;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5" BANET_WIN_NT_ "\n\t"
      "global_load_lds_dwordx4 %3, %5 offset:1024" BANET_WIN_NT_ "\n\t"
      "s_mov_b64 %1, exec\n\ts_mov_b64 exec, %7\n\t"
      "global_load_lds_dwordx4 %4, %5 offset:2048" BANET_WIN_NT_ "\n\t"
      "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(keepx)
      : "v"(v0), "v"(v1), "v"(v2), "s"(gbase), "s"(lds_dst), "s"(mask3)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt(int n) {   // n wave-uniform, 0 .. kMaxWait; rounds DOWN to an even count (a smaller
  if (n >= 8) {                                        // count is always safe): 3 scalar branches instead of a 16-way switch
    if (n >= 12) {
      if (n >= 14) wait_vm<14>();
      else wait_vm<12>();
    } else {
      if (n >= 10) wait_vm<10>();
      else wait_vm<8>();
    }
  } else {
    if (n >= 4) {
      if (n >= 6) wait_vm<6>();
      else wait_vm<4>();
    } else {
      if (n >= 2) wait_vm<2>();
      else wait_vm<0>();
    }
  }
}
static_assert(kMaxWait == 15, "wait_vmcnt implements counts up to 15");

// The source features of the next steps travel in VGPRs v224 .. v255 that the compiler never allocates (the kernel is limited
// to 224 registers with amdgpu_num_vgpr; these asm statements name the other 32 in their text and clobber lists, which also
// makes them part of the wave's allocation): a load whose destination the compiler knew about could be copied or spilled by
// it while the data is still in flight (its loop-carried v_mov copies did exactly that to a first version).  Slot j
// (0..3 = step mod 4), group t: v[224 + 8 j + 4 t .. + 3]; issued by buffer_load, copied out with v_mov after the counted wait.
constexpr int kStripVgprs = 224;   // the attribute counts in units of (1 VGPR + 1 AGPR) on the unified register file: 112
#if BANET_STRIP_NT >= 1
#define BANET_SRC_NT_ " nt"
#else
#define BANET_SRC_NT_ ""
#endif
#define BANET_SRC_ISSUE_(A0, A1, A2, A3, voff, rs, soff)                                                            \
  asm volatile("buffer_load_dwordx4 v[" #A0 ":" #A3 "], %0, %1, %2 offen" BANET_SRC_NT_ ::"v"(voff), "s"(rs), "s"(soff) \
               : "memory", "v" #A0, "v" #A1, "v" #A2, "v" #A3)
#define BANET_SRC_READ_(A0, A1, A2, A3, d)                                                                          \
  asm volatile("v_mov_b32 %0, v" #A0 "\n\tv_mov_b32 %1, v" #A1 "\n\tv_mov_b32 %2, v" #A2 "\n\tv_mov_b32 %3, v" #A3 \
               : "=v"(d.x), "=v"(d.y), "=v"(d.z), "=v"(d.w))
template <int SLOT, int T>
__device__ __forceinline__ void src_issue(unsigned voff, __amdgpu_buffer_rsrc_t rs, unsigned soff) {
  if constexpr (SLOT == 0 && T == 0) BANET_SRC_ISSUE_(224, 225, 226, 227, voff, rs, soff);
  if constexpr (SLOT == 0 && T == 1) BANET_SRC_ISSUE_(228, 229, 230, 231, voff, rs, soff);
  if constexpr (SLOT == 1 && T == 0) BANET_SRC_ISSUE_(232, 233, 234, 235, voff, rs, soff);
  if constexpr (SLOT == 1 && T == 1) BANET_SRC_ISSUE_(236, 237, 238, 239, voff, rs, soff);
  if constexpr (SLOT == 2 && T == 0) BANET_SRC_ISSUE_(240, 241, 242, 243, voff, rs, soff);
  if constexpr (SLOT == 2 && T == 1) BANET_SRC_ISSUE_(244, 245, 246, 247, voff, rs, soff);
  if constexpr (SLOT == 3 && T == 0) BANET_SRC_ISSUE_(248, 249, 250, 251, voff, rs, soff);
  if constexpr (SLOT == 3 && T == 1) BANET_SRC_ISSUE_(252, 253, 254, 255, voff, rs, soff);
}
template <int SLOT, int T>
__device__ __forceinline__ float4 src_read() {
  float4 d;
  if constexpr (SLOT == 0 && T == 0) BANET_SRC_READ_(224, 225, 226, 227, d);
  if constexpr (SLOT == 0 && T == 1) BANET_SRC_READ_(228, 229, 230, 231, d);
  if constexpr (SLOT == 1 && T == 0) BANET_SRC_READ_(232, 233, 234, 235, d);
  if constexpr (SLOT == 1 && T == 1) BANET_SRC_READ_(236, 237, 238, 239, d);
  if constexpr (SLOT == 2 && T == 0) BANET_SRC_READ_(240, 241, 242, 243, d);
  if constexpr (SLOT == 2 && T == 1) BANET_SRC_READ_(244, 245, 246, 247, d);
  if constexpr (SLOT == 3 && T == 0) BANET_SRC_READ_(248, 249, 250, 251, d);
  if constexpr (SLOT == 3 && T == 1) BANET_SRC_READ_(252, 253, 254, 255, d);
  return d;
}
template <int V>
struct IC {
  static constexpr int value = V;
};

// KV4 = number of 128-coefficient chunks of a basis row (0: pose only; K % 4 == 0, K <= 128 KV4)
// NCH = chunks per segment: 2 (16 x 8 pixels, 11/8 x: mid-size launches that have too few 16-row segments per wave), 4 (the default),
// 8 (16 x 32 pixels, target rows fetched 35/32 x) where a launch has enough of them to fill the
// chip evenly, 4 (16 x 16 pixels, 19/16 x) on launches with fewer, coarser items (gather.hip::plan_gather)
// FP ("frame-parallel", multi-frame windows): a workgroup is `pairs` waves that process the SAME segment at the same time, wave p
// against target frame p.  The key frame's data is then fetched from HBM once per window instead of once per target frame: the
// depth dot (the basis rows) is divided among the waves and shared through LDS, and the source rows, which every wave still
// loads into its own registers, are requested by all `pairs` waves of a CU within the same few microseconds -- one fabric read,
// the others hit the L2.  (Looping over the frames inside one wave -- the other mode -- re-reads them ~150 us apart, by which
// time the chip has moved hundreds of MB: 29 % of a 5-frame window's traffic.)  Two workgroup barriers per segment (item
// broadcast, depth hand-over), none inside the counted section.
extern __shared__ __attribute__((aligned(16))) float sDynS[];
constexpr int kSWaveLdsFloats = kWinFloats + kC128s;     // per wave: the rolling window + row statistics / sum|d|
inline size_t strip_fp_lds_bytes(int pairs, int nch) { return ((size_t)pairs * kSWaveLdsFloats + (size_t)nch * 64 + 4) * sizeof(float); }

// OPT (development A/B, BANET_STRIP_OPT): bit 0 = both 16-byte pieces' window reads of a step issued before the first piece's
// channel maths (the second LDS round trip hidden behind it); bit 1 = one M0 write per window row (glds_row)
template <int KV4, int NCH, bool FP, int OPT = 0>
__global__ __launch_bounds__(FP ? 512 : kSBlock, 2) __attribute__((amdgpu_num_vgpr(kStripVgprs / 2))) void ba_gather128s_kernel(const GatherArgs a) {
  constexpr bool kPipe = (OPT & 1) != 0, kM0Row = (OPT & 2) != 0;
  constexpr int SEGH = 4 * NCH;          // pixel rows per segment
  constexpr int NH = (NCH + 3) / 4;      // groups of four chunks
  constexpr int CPG = NCH < 4 ? NCH : 4; // chunks per group (NCH = 2: 8-row segments for mid-size launches, one half-filled group)
  __shared__ __attribute__((aligned(16))) float sWinS[FP ? 1 : kSWaves][FP ? 4 : kWinFloats];    // the rolling window: [slot][texel][32 channels]
  __shared__ __attribute__((aligned(16))) float sScrS[FP ? 1 : kSWaves][FP ? 4 : kC128s];        // row statistics (plan input), later C x sum|d|
  const banet_level_t& lv = a.lv;
  const int b = blockIdx.y;
  if (a.active != nullptr && a.active[(size_t)b * a.active_stride] == 0) return;
  const int lane = threadIdx.x & 63;
  const int w = wave_id();
  [[maybe_unused]] const int nw = FP ? a.pairs : kSWaves;                           // waves per workgroup
  float* const sWinW = FP ? sDynS + (size_t)w * kSWaveLdsFloats : &sWinS[FP ? 0 : w][0];
  float* const sScrW = FP ? sWinW + kWinFloats : &sScrS[FP ? 0 : w][0];
  [[maybe_unused]] float* const sDep = sDynS + (size_t)nw * kSWaveLdsFloats;        // FP: [NCH][64] depths of the segment
  [[maybe_unused]] int* const sItem = reinterpret_cast<int*>(sDep + NCH * 64);      // FP: [2] the segment index, ping-pong
  const int N = lv.N, K = lv.K, H = lv.H, W = lv.W;
  constexpr int C = kC128s;
  const float* __restrict__ src_b = lv.src + (size_t)b * N * C;
  const float* __restrict__ dep_b = lv.depth + (size_t)b * N;
  const float* __restrict__ bas_b = KV4 ? lv.basis + (size_t)b * N * K : nullptr;
  const int nitems = a.tiles, segs_y = a.tiles_y;
  // tap phase: a step is one pixel row; lane = (pixel column pc, channel octet oc): 4 lanes per pixel, 8 channels per lane
  // as two 16-byte pieces of the texel's 128-byte slice -- pieces 4 hA + oc and 4 (1 - hA) + oc with hA = bit 1 of pc, which
  // keeps the ds_read_b128 lane groups (4 pixels x 4 lanes) on 64 distinct banks when neighbouring pixels hit neighbouring
  // texels (a texel is 32 banks wide)
  const int pc = lane >> 2, oc = lane & 3, hA = (lane >> 3) & 1;
  const int jA = 4 * hA + oc, jB = 4 * (1 - hA) + oc;
  const int half = lane >> 5, li = lane & 31;     // depth dot: one basis row per half wave
  // lane = pixel phases: this lane owns pixel (row oc, column pc) of a chunk (4 pixel rows x 16)
  const unsigned win_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)sWinW;
  const unsigned dma_off = (unsigned)((lane >> 3) * 512 + (lane & 7) * 16);   // LDS-DMA: (texel lane >> 3 of an 8-texel run, piece lane & 7)
  const unsigned srcA_off = (unsigned)(pc * 512 + jA * 16), srcB_off = (unsigned)(pc * 512 + jB * 16);
  StripRowStat* sStat = reinterpret_cast<StripRowStat*>(sScrW);

  float wreg[KV4 ? KV4 : 1][4];  // this lane's slice of the depth coefficients
  if constexpr (KV4 > 0) {
#pragma unroll
    for (int kc = 0; kc < KV4; ++kc)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kc * 128 + li * 4 + e;
        wreg[kc][e] = (k < K) ? a.Wc[(size_t)b * K + k] : 0.f;
      }
  }
  [[maybe_unused]] const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_b), 0, N * C * 4, 0x00020000);

  int* __restrict__ queue = a.queue + blockIdx.y * 8;
  auto pop_raw = [&]() {
    int v = 0;
    if (lane == 0) v = atomicAdd(&queue[0], 1);
    return v;
  };
  int raw_next = (!FP || w == 0) ? pop_raw() : 0;

  for (int it = 0;; ++it) {
    int wi;
#if defined(BANET_TIMING) && BANET_TIMING == 3
    BANET_TICK(tb0);
#endif
    if constexpr (FP) {     // wave 0 pops for the workgroup
      if (w == 0 && lane == 0) sItem[it & 1] = raw_next;
      __syncthreads();
      wi = rfl(sItem[it & 1]);
    } else {
      wi = rfl(raw_next);
    }
    if (wi >= nitems) return;
    if (!FP || w == 0) raw_next = pop_raw();  // issued now, read at the top of the next item
    const int sx = wi / segs_y, sy = wi - sx * segs_y;
    const int px = sx * kStripW + pc;                         // this lane's pixel column (all chunks)
    const int py0 = sy * SEGH + oc;                        // its pixel row in chunk 0 (+ 4 per chunk)

    BANET_TICK(ts0);
#ifdef BANET_TIMING
    float ts_wait = 0.f;
    [[maybe_unused]] float ts_issue = 0.f, ts_ldsA = 0.f, ts_mathA = 0.f, ts_ldsB = 0.f, ts_mathB = 0.f, ts_bar = 0.f;
#if BANET_TIMING == 3
    ts_bar = (float)(ts0 - tb0);     // the item barrier (FP) + the queue pop
#endif
#endif
    // ---- 1. depth of every pixel of the segment: D = D0 + b . W, chunk by chunk (once per window) ----------------------
    fvec4 Dv[NH];
    if constexpr (KV4 > 0) {
      // batches of RB basis rows (16 at K <= 128, 8 at K = 256: 64 registers per buffer): batch n + 1 is in flight while
      // batch n is reduced (two register buffers, sched_barrier keeps the issue order), so the depth dot exposes one memory
      // latency per segment instead of one per batch
      constexpr int RB = FP ? (KV4 == 2 ? 4 : 8) : (KV4 == 1 ? 16 : 8), NB = 32 / RB;   // (FP: a wave does one chunk of four; 8-row batches keep the depth dot out of scratch)
      f32x4 bv[2][RB][KV4];
      auto issue_batch = [&](auto bufc, int c, int hb) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          // leaf i of half h ends on lane 32 h + brev5(i), which owns pixel (row ld & 3, column ld >> 2) of the chunk
          const int ld = half * 32 + brev5s(RB * hb + i);
          const int jx = sx * kStripW + (ld >> 2), jy = sy * SEGH + 4 * c + (ld & 3);
          const bool vj = (jx < W) && (jy < H);
#if defined(BANET_ABLATE_STRIP) && BANET_ABLATE_STRIP >= 2
          const float* row = bas_b + (size_t)(vj ? (jy & 3) * W + (jx & 63) : 0) * K;
#else
          const float* row = bas_b + (size_t)(vj ? jy * W + jx : 0) * K;
#endif
#pragma unroll
          for (int kc = 0; kc < KV4; ++kc) {
            const int k = kc * 128 + li * 4;
            bv[BUF][i][kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + (k < K ? k : 0)));
          }
        }
      };
      auto reduce_batch = [&](auto bufc, float (&pend)[6], int hb) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          float acc = 0.f;
#pragma unroll
          for (int kc = 0; kc < KV4; ++kc)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(bv[BUF][i][kc][e], wreg[kc][e], acc);   // k >= K: wreg is 0
          carry_push_s<5, 16>(pend, acc, RB * hb + i);
        }
      };
      if constexpr (FP) {
        // chunk c is computed by wave c mod nw and handed to the others through LDS
        if (w < NCH) issue_batch(IC<0>{}, w, 0);
#pragma unroll 1
        for (int c = w; c < NCH; c += nw) {
          const int py = py0 + 4 * c;
          const bool valid = (px < W) && (py < H);
          float D = valid ? dep_b[py * W + px] : 0.f;
          float pend[6];
#pragma unroll
          for (int hb = 0; hb < NB; ++hb) {
            __builtin_amdgcn_sched_barrier(0);
            if (hb + 1 < NB) {
              if ((hb & 1) == 0) issue_batch(IC<1>{}, c, hb + 1);
              else issue_batch(IC<0>{}, c, hb + 1);
            } else if (c + nw < NCH) {
              issue_batch(IC<0>{}, c + nw, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((hb & 1) == 0) reduce_batch(IC<0>{}, pend, hb);
            else reduce_batch(IC<1>{}, pend, hb);
          }
          sDep[c * 64 + lane] = D + pend[5];
        }
#if defined(BANET_TIMING) && BANET_TIMING == 3
        BANET_TICK(tb2);
#endif
        __syncthreads();
#if defined(BANET_TIMING) && BANET_TIMING == 3
        BANET_TICK(tb3);
        ts_bar += (float)(tb3 - tb2);
#endif
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
          for (int c4 = 0; c4 < CPG; ++c4) Dv[hh][c4] = sDep[(4 * hh + c4) * 64 + lane];
      } else {
      issue_batch(IC<0>{}, 0, 0);
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
#pragma unroll 1
      for (int c4 = 0; c4 < CPG; ++c4) {
        const int c = 4 * hh + c4;
        const int py = py0 + 4 * c;
        const bool valid = (px < W) && (py < H);
        float D = valid ? dep_b[py * W + px] : 0.f;
        float pend[6];
#pragma unroll
        for (int hb = 0; hb < NB; ++hb) {
          __builtin_amdgcn_sched_barrier(0);
          if (hb + 1 < NB) {                       // the next batch of this chunk ...
            if ((hb & 1) == 0) issue_batch(IC<1>{}, c, hb + 1);
            else issue_batch(IC<0>{}, c, hb + 1);
          } else if (c < NCH - 1) {                      // ... or the first one of the next chunk (NB is even: buffer 0)
            issue_batch(IC<0>{}, c + 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if ((hb & 1) == 0) reduce_batch(IC<0>{}, pend, hb);
          else reduce_batch(IC<1>{}, pend, hb);
        }
        D += pend[5];
        Dv[hh][c4] = D;
      }
      }
    } else {
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
#pragma unroll 1
      for (int c4 = 0; c4 < CPG; ++c4) {
        const int c = 4 * hh + c4;
        const int py = py0 + 4 * c;
        const bool valid = (px < W) && (py < H);
        Dv[hh][c4] = valid ? dep_b[py * W + px] : 0.f;
      }
    }

    BANET_TICK(ts1);
#pragma unroll 1
    for (int pr = FP ? w : 0; pr < (FP ? w + 1 : a.pairs); ++pr) {   // target frames of the window: same pixels, depth and source features (FP: this wave's frame)
      const int vb = b * a.pairs + pr;
      const float* __restrict__ tgt_b = lv.tgt + (size_t)vb * H * W * C;
      float* __restrict__ rec_b = KV4 ? a.rec + (size_t)vb * N * 8 : nullptr;
      float* __restrict__ part = a.partials + ((size_t)vb * nitems + wi) * (kGHdr + C);
      const float* Rm = a.R + vb * 9;
      const float* Tv = a.T + vb * 3;
      const PoseIntr pq = load_pose_intr(lv, b, Rm, Tv);     // 16 scalars, once per (window, target frame): quad_common.hpp

      // ---- 2. geometry (lane = pixel): tap parameters of the whole segment + the row statistics the plan needs ----------
      ivec4 P0v[NH];       // x0 | y0 << 12 | ((y0 - 1) mod 7) << 24 | fast << 27
      fvec4 DXv[NH], DYv[NH];
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
#pragma unroll 1
      for (int c4 = 0; c4 < CPG; ++c4) {
        const int c = 4 * hh + c4;
        const int py = py0 + 4 * c;
        const bool valid = (px < W) && (py < H);
        SGeo ge;
        strip_geometry(lv, pq, valid, px, py, Dv[hh][c4], ge);
        const bool fast = (ge.flags & 2) != 0;
        const int m7 = fast ? (ge.y0 - 1) % kWinRows : 0;
        P0v[hh][c4] = (fast ? (ge.x0 | (ge.y0 << 12)) : 0) | (m7 << 24) | (fast ? (1 << 27) : 0);
        DXv[hh][c4] = ge.dx;
        DYv[hh][c4] = ge.dy;
        // min / max of (x0, y0) over the 16 fast pixels of this lane's pixel row (lanes that agree in lane & 3)
        const int big = 0x3fffffff;
        int ymn = fast ? ge.y0 : big, ymx = fast ? ge.y0 : -big, xmn = fast ? ge.x0 : big, xmx = fast ? ge.x0 : -big;
#pragma unroll
        for (int sh = 4; sh < 64; sh <<= 1) {
          ymn = min(ymn, __shfl_xor(ymn, sh, 64));
          ymx = max(ymx, __shfl_xor(ymx, sh, 64));
          xmn = min(xmn, __shfl_xor(xmn, sh, 64));
          xmx = max(xmx, __shfl_xor(xmx, sh, 64));
        }
        if (lane < 4) {
          StripRowStat st;
          st.ymin = ymn;
          st.ymax = ymx;
          st.xmin = xmn;
          st.xmax = xmx;
          sStat[4 * c + lane] = st;
        }
      }
      // the plan (strip_plan.hpp): lane r < 32 owns pixel row r for the static part, the replay loop is scalar
      [[maybe_unused]] unsigned long long mask3 = 0;   // lanes of the third LDS-DMA instruction of a row (kM0Row)
      int xl, ncol3, plan_ctl, plan_yf;       // the plan: StripStep.ctl / .yfirst of pixel row r on lane r (read with v_readlane)
      {
        StripRowStat st = sStat[lane & (SEGH - 1)];
        if (lane >= SEGH) st.ymin = 1, st.ymax = 0;                    // lanes SEGH..63: no row
        int xm = st.ymin <= st.ymax ? st.xmin : 0x3fffffff;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) xm = min(xm, __shfl_xor(xm, sh, 64));
        xl = rfl(strip_window_origin(xm, W));
        StripLaneEnv env;
        env.modev = strip_static_mode(st, xl, W);
        env.ytv = st.ymin - 1;
        env.ybv = st.ymax + 2;
        env.ctlv = env.yfv = env.ringv = 0;
        int ye = env.modev == kStepWindow ? env.ybv : -0x3fffffff;
        int xe = env.modev == kStepWindow ? st.xmax + 2 - xl : 0;      // last window column any row served from the window reads
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
          ye = max(ye, __shfl_xor(ye, sh, 64));
          xe = max(xe, __shfl_xor(xe, sh, 64));
        }
        // lanes of the third LDS-DMA instruction of a window row (texels 16 .. xe, 8 lanes each): the 21-texel pitch is the
        // capacity, a unit-scale segment reads 19 or 20 of them -- the rest is not fetched.  At least one texel: the plan
        // counts kRowOps = 3 operations per row, and an instruction whose lanes are all off is branched around.
        ncol3 = 8 * max(rfl(xe) + 1 - 16, 1);
        mask3 = ncol3 >= 64 ? ~0ull : ((1ull << ncol3) - 1ull);
        strip_plan_dynamic(env, SEGH, rfl(ye));
        if (lv.flags & (1 << 20)) {   // parity tests: every pixel row takes the direct (window-less) path
          if (step_mode(env.ctlv) == kStepWindow) env.ctlv = kStepDirect;
        }
        plan_ctl = env.ctlv;
        plan_yf = env.yfv;
      }

      BANET_TICK(ts2);
      // ---- 3. the four channel slices ---------------------------------------------------------------------------------------
      fvec4 Q0[NH], Q1[NH], Q2[NH], Q3[NH], Q4[NH];   // per chunk: m11 m12 m22 g1 g2
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) Q0[hh] = Q1[hh] = Q2[hh] = Q3[hh] = Q4[hh] = 0.f;
      const int rowC = W * C;
#pragma unroll 1
      for (int s = 0; s < 4; ++s) {
        float absA[4] = {0.f, 0.f, 0.f, 0.f}, absB[4] = {0.f, 0.f, 0.f, 0.f};   // |d| of this lane's two channel pieces
        const float* tgt_s = tgt_b + 32 * s;
        // byte offset of pixel row r of this segment's source rows, slice s (the lane's pixel and piece are added per lane)
#if defined(BANET_ABLATE_STRIP) && BANET_ABLATE_STRIP >= 2
        auto src_soff = [&](int r) { return (unsigned)(((((sy * SEGH + r) & 3) * W + (sx & 3) * kStripW) * C + 32 * s) * 4); };
#else
        auto src_soff = [&](int r) { return (unsigned)((((sy * SEGH + r) * W + sx * kStripW) * C + 32 * s) * 4); };
#endif
        auto issue_row = [&](int Y, int slot) __attribute__((always_inline)) {     // texel row Y, columns xl .. xl + 20 -> ring slot Y mod 7
#ifdef BANET_ABLATE_STRIP   // development: the same instruction stream on a cache-resident footprint (is the kernel paced by memory?)
          const float* gb = tgt_s + ((size_t)(Y & 3) * W + (xl & 15)) * C;
#else
          const float* gb = tgt_s + ((size_t)Y * W + xl) * C;
#endif
          const unsigned dst = win_base + (unsigned)slot * (unsigned)kWinPitchB;
          if constexpr (kM0Row) {
            glds_row(gb, dma_off, dma_off + 3072u, dma_off + 6144u, dst, mask3);
          } else {
            glds16(gb, dma_off, dst);
            glds16(gb + 8 * C, dma_off, dst + 1024u);
            if (lane < ncol3) glds16(gb + 16 * C, dma_off, dst + 2048u);
          }
        };
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the counted section starts with nothing in flight
        {
          const int c0 = __builtin_amdgcn_readlane(plan_ctl, 0), c1 = __builtin_amdgcn_readlane(plan_ctl, 1);
          if (step_src_pre(c0)) {
            src_issue<0, 0>(srcA_off, rs_src, src_soff(0));
            src_issue<0, 1>(srcB_off, rs_src, src_soff(0));
          }
          if (step_src_pre(c1)) {
            src_issue<1, 0>(srcA_off, rs_src, src_soff(1));
            src_issue<1, 1>(srcB_off, rs_src, src_soff(1));
          }
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll 1
        for (int c4 = 0; c4 < CPG; ++c4) {
          const int c = 4 * hh + c4;
          const int p0c = P0v[hh][c4];
          const float dxc = DXv[hh][c4], dyc = DYv[hh][c4];
          float qacc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};    // this lane's pixel of chunk c, slice s
          // step r = 4 c + k: pixel row k of the chunk (k is a compile-time constant: the registers of the source features
          // are named in the instruction text, the parameter broadcast is a DPP quad_perm)
          auto do_step = [&](auto kconst) __attribute__((always_inline)) {
            constexpr int k = decltype(kconst)::value;
            const int r = 4 * c + k;
            const int ctl = __builtin_amdgcn_readlane(plan_ctl, r);
            const int mode = step_mode(ctl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the previous step's window reads are done (WAR)
#if defined(BANET_TIMING) && BANET_TIMING >= 4
            BANET_TICK(tq0);
#endif
            if (step_src_next(ctl)) {
              src_issue<(k + 2) & 3, 0>(srcA_off, rs_src, src_soff(r + kSrcAhead));
              src_issue<(k + 2) & 3, 1>(srcB_off, rs_src, src_soff(r + kSrcAhead));
            }
            if (mode == kStepSkip) return;
            // parameters of pixel (row k, column pc) live on lane 4 pc + k: inside this lane's quad
            const int p0 = quad_bcast<k>(p0c);
            const float dx = quad_bcast<k>(dxc), dy = quad_bcast<k>(dyc);
            const bool fast = (p0 >> 27) & 1;
            const float mk = fast ? 1.f : 0.f;
            const float w00 = mk * ((1.f - dx) * (1.f - dy)), w01 = mk * (dx * (1.f - dy)), w10 = mk * ((1.f - dx) * dy),
                        w11 = mk * (dx * dy);
            float qq[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            if (mode == kStepWindow) {
              const int y0r = __builtin_amdgcn_readlane(plan_yf, r), nr = step_nrows(ctl);
              int slot = y0r % kWinRows;
              for (int i = 0; i < nr; ++i) {
                issue_row(y0r + i, slot);
                slot = slot + 1 == kWinRows ? 0 : slot + 1;
              }
#if defined(BANET_TIMING) && BANET_TIMING == 1
              BANET_TICK(tw0);
#endif
              wait_vmcnt(step_wait(ctl));
#if defined(BANET_TIMING) && BANET_TIMING == 1
              BANET_TICK(tw1);
              BANET_TACC(ts_wait, tw0, tw1);
#endif
              const float4 fA = src_read<k, 0>(), fB = src_read<k, 1>();      // landed: behind the counted wait
              const int mtop = step_mtop(ctl);
              const int xr = fast ? (p0 & 0xfff) - 1 - xl : 0;             // window column of texel x0 - 1
              const int m0 = fast ? (p0 >> 24) & 7 : mtop;                 // ring slot of texel row y0 - 1
              const int m1i = m0 + 1 >= kWinRows ? m0 + 1 - kWinRows : m0 + 1;
              const int m2i = m1i + 1 >= kWinRows ? m1i + 1 - kWinRows : m1i + 1;
              const int m3i = m2i + 1 >= kWinRows ? m2i + 1 - kWinRows : m2i + 1;
              const float* l = sWinW + xr * 32;
              const float* l0 = l + m0 * kWinPitchF;
              const float* l1 = l + m1i * kWinPitchF;
              const float* l2 = l + m2i * kWinPitchF;
              const float* l3 = l + m3i * kWinPitchF;
#if defined(BANET_TIMING) && BANET_TIMING >= 4
              BANET_TICK(tl0);                       // window mode, after the counted wait: address set-up + reads of piece A start here
              [[maybe_unused]] unsigned long long tlm = tl0;
              ts_issue += (float)(tl0 - tq0);        // step start -> here: source issue, parameter broadcast, row issue, counted wait
#endif
              if constexpr (kPipe) {
                // both pieces' 12 window reads are in flight before the first piece's channel maths: the second LDS round trip
                // (12 x 1 KB wave reads through a port shared by 8 waves) is hidden behind ~100 packed instructions
                float4 tA[12], tB[12];
                auto rd = [&](float4 (&t)[12], int po) __attribute__((always_inline)) {
                  t[0] = *reinterpret_cast<const float4*>(l1 + po), t[1] = *reinterpret_cast<const float4*>(l1 + po + 32);
                  t[2] = *reinterpret_cast<const float4*>(l1 + po + 64), t[3] = *reinterpret_cast<const float4*>(l1 + po + 96);
                  t[4] = *reinterpret_cast<const float4*>(l2 + po), t[5] = *reinterpret_cast<const float4*>(l2 + po + 32);
                  t[6] = *reinterpret_cast<const float4*>(l2 + po + 64), t[7] = *reinterpret_cast<const float4*>(l2 + po + 96);
                  t[8] = *reinterpret_cast<const float4*>(l0 + po + 32), t[9] = *reinterpret_cast<const float4*>(l0 + po + 64);
                  t[10] = *reinterpret_cast<const float4*>(l3 + po + 32), t[11] = *reinterpret_cast<const float4*>(l3 + po + 64);
                };
                rd(tA, 4 * jA);
                rd(tB, 4 * jB);
                __builtin_amdgcn_sched_barrier(0);
                tap_math_s(fA, tA[0], tA[1], tA[2], tA[3], tA[4], tA[5], tA[6], tA[7], tA[8], tA[9], tA[10], tA[11], w00, w01, w10, w11, mk, qq, absA);
                __builtin_amdgcn_sched_barrier(0);
                tap_math_s(fB, tB[0], tB[1], tB[2], tB[3], tB[4], tB[5], tB[6], tB[7], tB[8], tB[9], tB[10], tB[11], w00, w01, w10, w11, mk, qq, absB);
              } else {
#pragma unroll
              for (int hp = 0; hp < 2; ++hp) {      // the lane's two 16-byte pieces of every tap
                const int po = 4 * (hp ? jB : jA);
                const float4 f1 = hp ? fB : fA;
                const float4 a0 = *reinterpret_cast<const float4*>(l1 + po), a1 = *reinterpret_cast<const float4*>(l1 + po + 32),
                             a2 = *reinterpret_cast<const float4*>(l1 + po + 64), a3 = *reinterpret_cast<const float4*>(l1 + po + 96);
                const float4 b0 = *reinterpret_cast<const float4*>(l2 + po), b1 = *reinterpret_cast<const float4*>(l2 + po + 32),
                             b2 = *reinterpret_cast<const float4*>(l2 + po + 64), b3 = *reinterpret_cast<const float4*>(l2 + po + 96);
                const float4 m1 = *reinterpret_cast<const float4*>(l0 + po + 32), m2 = *reinterpret_cast<const float4*>(l0 + po + 64);
                const float4 p1 = *reinterpret_cast<const float4*>(l3 + po + 32), p2 = *reinterpret_cast<const float4*>(l3 + po + 64);
#if defined(BANET_TIMING) && BANET_TIMING >= 4
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                BANET_TICK(tl1);
                if (hp) ts_ldsB += (float)(tl1 - tlm); else ts_ldsA += (float)(tl1 - tl0);
#endif
                if (hp)
                  tap_math_s(f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2, w00, w01, w10, w11, mk, qq, absB);
                else
                  tap_math_s(f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2, w00, w01, w10, w11, mk, qq, absA);
#if defined(BANET_TIMING) && BANET_TIMING >= 4
                asm volatile("" ::"v"(qq[0]), "v"(qq[4]) : "memory");
                BANET_TICK(tl2);
                if (hp) ts_mathB += (float)(tl2 - tl1); else ts_mathA += (float)(tl2 - tl1);
                tlm = tl2;
#endif
              }
              }
            } else {   // kStepDirect: the footprint of this pixel row does not fit the window -- taps straight from memory
              const int x0 = fast ? (p0 & 0xfff) : 1, y0 = fast ? (p0 >> 12) & 0xfff : 1;
              const int gpx = sx * kStripW + pc, gpy = sy * SEGH + r;
              const bool gv = (gpx < W) && (gpy < H);
#pragma unroll
              for (int hp = 0; hp < 2; ++hp) {
                const int po = 32 * s + 4 * (hp ? jB : jA);
                const unsigned osrc = (unsigned)((gv ? gpy * W + gpx : 0) * C + po);
                const unsigned oa = (unsigned)((y0 * W + x0) * C + po);
                const float* ra = tgt_b + (size_t)oa;
                const float* rb = ra + rowC;
                const float* rm = ra - rowC;
                const float* rp = rb + rowC;
                const f32x4 f1v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src_b + (size_t)osrc));
                const float4 f1 = make_float4(f1v[0], f1v[1], f1v[2], f1v[3]);
                const float4 a0 = *reinterpret_cast<const float4*>(ra - C), a1 = *reinterpret_cast<const float4*>(ra),
                             a2 = *reinterpret_cast<const float4*>(ra + C), a3 = *reinterpret_cast<const float4*>(ra + 2 * C);
                const float4 b0 = *reinterpret_cast<const float4*>(rb - C), b1 = *reinterpret_cast<const float4*>(rb),
                             b2 = *reinterpret_cast<const float4*>(rb + C), b3 = *reinterpret_cast<const float4*>(rb + 2 * C);
                const float4 m1 = *reinterpret_cast<const float4*>(rm), m2 = *reinterpret_cast<const float4*>(rm + C);
                const float4 p1 = *reinterpret_cast<const float4*>(rp), p2 = *reinterpret_cast<const float4*>(rp + C);
                if (hp)
                  tap_math_s(f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2, w00, w01, w10, w11, mk, qq, absB);
                else
                  tap_math_s(f1, a0, a1, a2, a3, b0, b1, b2, b3, m1, m2, p1, p2, w00, w01, w10, w11, mk, qq, absA);
              }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              const float tot = quad_sum(qq[i]);
              qacc[i] += (oc == k) ? tot : 0.f;       // the lane that owns pixel (row k, column pc) keeps it
            }
          };
          do_step(IC<0>{});
          do_step(IC<1>{});
          do_step(IC<2>{});
          do_step(IC<3>{});
          Q0[hh][c4] += qacc[0];
          Q1[hh][c4] += qacc[1];
          Q2[hh][c4] += qacc[2];
          Q3[hh][c4] += qacc[3];
          Q4[hh][c4] += qacc[4];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the slice's 32 x sum|d|: piece oc (channels 4 oc + e) and piece 4 + oc of every lane, folded over the 16 pixel
        // columns (fixed order); lanes 0..3 publish
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float lo = hA ? absB[e] : absA[e], hi = hA ? absA[e] : absB[e];
#pragma unroll
          for (int sh = 4; sh < 64; sh <<= 1) {
            lo += __shfl_xor(lo, sh, 64);
            hi += __shfl_xor(hi, sh, 64);
          }
          if (lane < 4) {
            sScrW[32 * s + 4 * oc + e] = lo;
            sScrW[32 * s + 16 + 4 * oc + e] = hi;
          }
        }
      }

      BANET_TICK(ts3);
      // ---- 4. per-pixel 6x6 algebra (lane = pixel), records, the segment's 28 pose sums ------------------------------------
      float absd2[1][2] = {{0.f, 0.f}};  // rim pixels (generic routine: channels 2 lane, 2 lane + 1)
      float accp[28];                     // this lane's 8 pixels: upper triangle of Jc^T M Jc (21), Jc^T g (6), valid count
#pragma unroll
      for (int i = 0; i < 28; ++i) accp[i] = 0.f;
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
#pragma unroll 1
      for (int c4 = 0; c4 < CPG; ++c4) {
        const int c = 4 * hh + c4;
        const int py = py0 + 4 * c;
        const bool valid = (px < W) && (py < H);
        const int pt = valid ? py * W + px : 0;
        SGeo ge;
        strip_geometry(lv, pq, valid, px, py, Dv[hh][c4], ge);
        Q5 qv;
        qv.m11 = Q0[hh][c4];
        qv.m12 = Q1[hh][c4];
        qv.m22 = Q2[hh][c4];
        qv.g1 = Q3[hh][c4];
        qv.g2 = Q4[hh][c4];
        {
          // patch the pixels whose stencil touches the image rim (rare): generic slow routine
          unsigned long long slow = __ballot((ge.flags & 4) != 0);
          while (slow) {  // wave-uniform
            const int j = __builtin_ctzll(slow);
            slow &= slow - 1;
            const float jdx = rdl(ge.dx, j), jdy = rdl(ge.dy, j);
            Q5 e = border_pixel_q5<2, 1>(rdl(ge.x0, j), rdl(ge.y0, j), (1.f - jdx) * (1.f - jdy), jdx * (1.f - jdy),
                                         (1.f - jdx) * jdy, jdx * jdy, src_b + (size_t)rdl(pt, j) * C, tgt_b, C, H, W, lane,
                                         absd2);
            e.m11 = wave_sum(e.m11);
            e.m12 = wave_sum(e.m12);
            e.m22 = wave_sum(e.m22);
            e.g1 = wave_sum(e.g1);
            e.g2 = wave_sum(e.g2);
            if (lane == j) {
              qv.m11 += e.m11;
              qv.m12 += e.m12;
              qv.m22 += e.m22;
              qv.g1 += e.g1;
              qv.g2 += e.g2;
            }
          }
        }
        const float* jc = ge.jc;
        float mj[12];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          mj[i] = qv.m11 * jc[i] + qv.m12 * jc[6 + i];
          mj[6 + i] = qv.m12 * jc[i] + qv.m22 * jc[6 + i];
        }
        {   // chunks in order 0..7: fixed summation order
          int o = 0;
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int jj = i; jj < 6; ++jj) {
              accp[o] += jc[i] * mj[jj] + jc[6 + i] * mj[6 + jj];
              ++o;
            }
#pragma unroll
          for (int i = 0; i < 6; ++i) accp[21 + i] += jc[i] * qv.g1 + jc[6 + i] * qv.g2;
          accp[27] += (float)(ge.flags & 1);
          if (a.mask_out != nullptr && valid) a.mask_out[(size_t)vb * N + pt] = (unsigned char)(ge.flags & 1);
        }

        if constexpr (KV4 > 0) {
          if (valid) {
            const float md0 = qv.m11 * ge.jd0 + qv.m12 * ge.jd1, md1 = qv.m12 * ge.jd0 + qv.m22 * ge.jd1;
            float4 ua, ub;
            ua.x = jc[0] * md0 + jc[6] * md1;
            ua.y = jc[1] * md0 + jc[7] * md1;
            ua.z = jc[2] * md0 + jc[8] * md1;
            ua.w = jc[3] * md0 + jc[9] * md1;
            ub.x = jc[4] * md0 + jc[10] * md1;
            ub.y = jc[5] * md0 + jc[11] * md1;
            ub.z = ge.jd0 * md0 + ge.jd1 * md1;      // s_n
            ub.w = ge.jd0 * qv.g1 + ge.jd1 * qv.g2;  // r_n
            float4* rp = reinterpret_cast<float4*>(rec_b + (size_t)pt * 8);
            rp[0] = ua;
            rp[1] = ub;
          }
        }
      }
      {
        // the segment's 28 pose sums: leaves 0..27 = accp, 28..31 zero; 5 levels of the transposing butterfly (lane distance
        // 32..2) + one xor-1 add: lane l ends with leaf brev5(l >> 1)
        float pend[6];
#pragma unroll
        for (int i = 0; i < 28; ++i) carry_push_s<5, 32>(pend, accp[i], i);
#pragma unroll
        for (int i = 28; i < 32; ++i) carry_push_s<5, 32>(pend, 0.f, i);
        float tot = pend[5];
        tot += dpp_mov<kDppXor1>(tot);
        const int leaf = brev5s(lane >> 1);
        if ((lane & 1) == 0 && leaf < 28) part[leaf] = tot;
      }
      // ---- 5. the segment's C x sum|d| ----------------------------------------------------------------------------------------
      sScrW[2 * lane] += absd2[0][0];       // same wave: LDS operations retire in program order
      sScrW[2 * lane + 1] += absd2[0][1];
      part[kGHdr + lane] = sScrW[lane];
      part[kGHdr + 64 + lane] = sScrW[64 + lane];
#ifdef BANET_TIMING   // tools/time_strip.py: cycles of this segment (the last target frame of the window)
      {
        BANET_TICK(ts9);
        if (lane == 0) {
#if BANET_TIMING == 1
          part[28] = ts_wait;                  // parked in the counted waits (window rows / source features not landed)
          part[29] = (float)(ts3 - ts2);       // the four slice passes
#elif BANET_TIMING == 3
          part[28] = ts_bar;                   // FP: the two workgroup barriers of this segment (waiting for the other frames' waves)
          part[29] = (float)(ts2 - ts0) - (ts_bar - (float)(ts0 - tb0));   // depth dot + geometry + plan without the barrier
#elif BANET_TIMING == 4
          part[28] = ts_issue;                 // per step: start -> after the counted wait (source issue, broadcast, row issue, wait)
          part[29] = ts_ldsA + ts_ldsB;        // window reads: issue -> returned, both pieces
#elif BANET_TIMING == 5
          part[28] = ts_mathA + ts_mathB;      // channel maths of both pieces
          part[29] = (float)(ts3 - ts2) - (ts_issue + ts_ldsA + ts_ldsB + ts_mathA + ts_mathB);   // rest of the slice passes (reductions, loop)
#else
          part[28] = (float)(ts1 - ts0);       // depth dot
          part[29] = (float)(ts2 - ts1);       // geometry + plan
#endif
#if BANET_TIMING >= 4
          part[30] = (float)(ts3 - ts2);       // the four slice passes
#else
          part[30] = (float)(ts9 - ts3);       // rim, algebra, records, partial row
#endif
#if BANET_TIMING == 3
          part[31] = (float)(ts9 - tb0);       // whole segment incl. the item barrier
#else
          part[31] = (float)(ts9 - ts0);       // whole segment
#endif
        }
      }
#endif
    }  // pairs
  }  // items
}

// development A/B (builds with -DBANET_STRIP_OPT_VARIANTS only; measured neutral, profiles/r05_run2_*): BANET_STRIP_OPT selects the
// OPT variant of the K <= 128, 16-row-segment kernels (read once)
static int strip_opt() {
  static const int v = [] {
    const char* e = std::getenv("BANET_STRIP_OPT");
    const int x = e ? std::atoi(e) : BANET_STRIP_OPT_DEFAULT;
    return x >= 0 && x <= 3 ? x : 0;
  }();
  return v;
}

int launch_gather128s(const GatherArgs& a, int K, hipStream_t s) {
  dim3 grid(a.G, a.lv.B), block(kSBlock);
  const bool tall = a.seg_h == 32, low = a.seg_h == 8;
  if (a.seg_h != 32 && a.seg_h != 16 && a.seg_h != 8) return BANET_ERR_INVALID_ARG;
  if (a.strip_fp) {    // frame-parallel workgroups: `pairs` waves, dynamic LDS (16-row segments only)
    if (tall || low || a.pairs < 2 || a.pairs > 7) return BANET_ERR_INVALID_ARG;
    const size_t shm = strip_fp_lds_bytes(a.pairs, 4);
    block = dim3(64 * a.pairs);
#define BANET_LAUNCH_FP(KV4)                                                                                              \
  do {                                                                                                                    \
    if (shm > 64 * 1024)                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_gather128s_kernel<KV4, 4, true>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                    \
    hipLaunchKernelGGL((ba_gather128s_kernel<KV4, 4, true>), grid, block, shm, s, a);                                     \
  } while (0)
#define BANET_LAUNCH_FPO(OPT)                                                                                             \
  do {                                                                                                                    \
    if (shm > 64 * 1024)                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_gather128s_kernel<1, 4, true, OPT>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                    \
    hipLaunchKernelGGL((ba_gather128s_kernel<1, 4, true, OPT>), grid, block, shm, s, a);                                  \
  } while (0)
    [[maybe_unused]] const int opt = strip_opt();
    if (K == 0) BANET_LAUNCH_FP(0);
#ifdef BANET_STRIP_OPT_VARIANTS
    else if ((K & 3) == 0 && K <= 128 && opt == 1) BANET_LAUNCH_FPO(1);
    else if ((K & 3) == 0 && K <= 128 && opt == 2) BANET_LAUNCH_FPO(2);
    else if ((K & 3) == 0 && K <= 128 && opt == 3) BANET_LAUNCH_FPO(3);
#endif
    else if ((K & 3) == 0 && K <= 128) BANET_LAUNCH_FP(1);
    else if ((K & 3) == 0 && K <= 256) BANET_LAUNCH_FP(2);
    else return BANET_ERR_UNSUPPORTED;
#undef BANET_LAUNCH_FP
#undef BANET_LAUNCH_FPO
    return BANET_OK;
  }
#define BANET_LAUNCH_S(KV4)                                                                     \
  do {                                                                                          \
    if (tall) hipLaunchKernelGGL((ba_gather128s_kernel<KV4, 8, false>), grid, block, 0, s, a);  \
    else if (low) hipLaunchKernelGGL((ba_gather128s_kernel<KV4, 2, false>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((ba_gather128s_kernel<KV4, 4, false>), grid, block, 0, s, a);       \
  } while (0)
  [[maybe_unused]] const int opt = strip_opt();
  if (K == 0) BANET_LAUNCH_S(0);
#ifdef BANET_STRIP_OPT_VARIANTS
  else if ((K & 3) == 0 && K <= 128 && !tall && !low && opt == 1) hipLaunchKernelGGL((ba_gather128s_kernel<1, 4, false, 1>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 128 && !tall && !low && opt == 2) hipLaunchKernelGGL((ba_gather128s_kernel<1, 4, false, 2>), grid, block, 0, s, a);
  else if ((K & 3) == 0 && K <= 128 && !tall && !low && opt == 3) hipLaunchKernelGGL((ba_gather128s_kernel<1, 4, false, 3>), grid, block, 0, s, a);
#endif
  else if ((K & 3) == 0 && K <= 128) BANET_LAUNCH_S(1);
  else if ((K & 3) == 0 && K <= 256) BANET_LAUNCH_S(2);
  else return BANET_ERR_UNSUPPORTED;
#undef BANET_LAUNCH_S
  return BANET_OK;
}

}  // namespace banet
