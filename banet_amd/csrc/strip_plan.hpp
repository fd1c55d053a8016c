// Rolling-window plan of one strip segment of ba_gather128s_kernel (gather128s.hip).  Plain C++ that compiles for the
// host as well: tests/test_strip_plan_cpu.py builds it with g++ and replays the planned instruction stream against a
// model of the LDS ring and of the in-order VMEM counter.
//
// A segment is kStripW x 16 source pixels (kStripH = 32 at most, behind a switch).  For one 32-channel slice of the target map the wave keeps the last
// kWinRows texel rows of a kWinTex-texel-wide column band in its own LDS ("window"); texel row Y lives in ring slot
// Y mod kWinRows.  The wave walks the segment's pixel rows top to bottom ("steps"); before the taps of step r it (1) issues
// the source-feature loads of step r + kSrcAhead, (2) issues the LDS-DMA loads of the window rows the plan assigns to step r -- as far
// ahead as the ring allows without overwriting a row step r still reads -- and (3) waits with a COUNTED s_waitcnt vmcnt(n)
// until the last window row step r needs and its own source features have landed.  Everything that decides those three
// actions depends only on the geometry of the segment (not on the channel slice), so it is computed once here and replayed
// for the four slices.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BANET_HD __host__ __device__ __forceinline__
#else
#define BANET_HD inline
#endif

namespace banet {

constexpr int kStripW = 16;     // source pixels per pixel row of a segment (= 2 instruction groups of 8 pixels)
constexpr int kStripH = 32;     // most pixel rows per segment the plan is sized for (8 chunks of 4 rows); the default is 16
constexpr int kWinTex = 21;     // texels per window row: kStripW + 3 (stencil) + 2 (slack: local scale up to ~1.12)
constexpr int kWinRows = 7;     // ring slots
constexpr int kWinPitchB = kWinTex * 128;        // bytes per window row: 32 channels x 4 B per texel
constexpr int kRowOps = 3;      // LDS-DMA instructions per window row (1 KB each: 8 + 8 + 5 texels)
constexpr int kSrcOps = 2;      // source-feature loads per step (2 groups of 8 pixels)
constexpr int kSrcAhead = 2;    // steps the source-feature loads run ahead of their use (register buffers: kSrcAhead + 2 ... see kernel)
constexpr int kMaxWait = 15;    // largest counted wait the kernel's switch implements (smaller is always safe)

struct StripRowStat {           // over the fast (in-image, interior) pixels of one pixel row; ymin > ymax: the row has none
  int32_t ymin, ymax;           // floor(py) range
  int32_t xmin, xmax;           // floor(px) range
};

enum { kStepSkip = 0, kStepWindow = 1, kStepDirect = 2 };

struct StripStep {              // 16 bytes
  int32_t ctl;                  // mode | nrows << 4 | wait << 8 | src_next << 16 | src_pre << 17 | mtop << 20  (src_next: issue
                                // the source loads of step r + kSrcAhead; src_pre: step r < kSrcAhead, loads issued before the loop)
  int32_t yfirst;               // first texel row this step issues (nrows consecutive rows)
  int32_t ytop;                 // lowest texel row the step reads (ymin - 1)
  int32_t pad;
};
BANET_HD int step_mode(int ctl) { return ctl & 15; }
BANET_HD int step_nrows(int ctl) { return (ctl >> 4) & 15; }
BANET_HD int step_wait(int ctl) { return (ctl >> 8) & 255; }
BANET_HD int step_src_next(int ctl) { return (ctl >> 16) & 1; }
BANET_HD int step_src_pre(int ctl) { return (ctl >> 17) & 1; }
BANET_HD int step_mtop(int ctl) { return (ctl >> 20) & 15; }

// static part, one row: does the row's footprint fit the window at all?
BANET_HD int strip_static_mode(const StripRowStat& s, int xl, int img_w) {
  if (!(s.ymin <= s.ymax)) return kStepSkip;
  const bool fx = (s.xmin - 1 >= xl) && (s.xmax + 2 - xl <= kWinTex - 1);
  const bool fy = (s.ymax + 2) - (s.ymin - 1) + 1 <= kWinRows;
  return (fx && fy && img_w >= kWinTex) ? kStepWindow : kStepDirect;
}
BANET_HD int strip_window_origin(int xmin_all, int img_w) {   // xmin_all: min x0 over the segment's fast pixels (or huge)
  int xl = xmin_all - 1;
  if (xl > img_w - kWinTex) xl = img_w - kWinTex;
  if (xl < 0) xl = 0;
  return xl;
}
BANET_HD int mod_rows(int y) { return ((y % kWinRows) + kWinRows) % kWinRows; }

// dynamic part: replay the issue order.  seq = VMEM operations issued so far; a wait for "everything up to operation k" is
// s_waitcnt vmcnt(seq - k).  Env supplies the rows' static results and the storage (host: arrays; device: one row per lane,
// read / written with v_readlane / v_writelane so that this whole loop runs on the scalar unit):
//   int mode(r), yt(r), yb(r);  void set_step(r, ctl, yfirst);  int ring_get(slot);  void ring_set(slot, seq)
template <class Env>
BANET_HD void strip_plan_dynamic(Env& e, int n, int yend) {
  int seq = 0, issued = 0, lo = 0;      // resident rows: [max(lo, issued - kWinRows + 1), issued] once have = true
  bool have = false;
  for (int i = 0; i < kWinRows; ++i) e.ring_set(i, 0);
  static_assert(kSrcAhead == 2, "the rotating source sequence numbers below are written for a distance of 2");
  int src0 = 0, src1 = 0, src2 = 0;     // seq after the source loads of steps r, r + 1, r + 2 were issued
  if (n > 0 && e.mode(0) == kStepWindow) {   // the source loads of the first kSrcAhead steps precede the loop
    seq += kSrcOps;
    src0 = seq;
  }
  if (n > 1 && e.mode(1) == kStepWindow) {
    seq += kSrcOps;
    src1 = seq;
  }
  for (int r = 0; r < n; ++r) {
    int mode = e.mode(r);
    const int src_pre = (r < kSrcAhead && mode == kStepWindow) ? 1 : 0;      // its source loads were issued before the loop
    const int src_next = (r + kSrcAhead < n && e.mode(r + kSrcAhead) == kStepWindow) ? 1 : 0;
    src2 = 0;
    if (src_next) {
      seq += kSrcOps;
      src2 = seq;
    }
    int yfirst = 0, nrows = 0, wait = 0, ytop = 0;
    if (mode == kStepWindow) {
      const int yt = e.yt(r), yb = e.yb(r);
      ytop = yt;
      // a row below the ring's oldest resident row (non-monotonic footprint) cannot be served any more
      if (have && yt < (lo > issued - kWinRows + 1 ? lo : issued - kWinRows + 1) && yt <= issued) {
        mode = kStepDirect;
      } else {
        int target = yt + kWinRows - 1;
        if (target > yend) target = yend;
        if (target < yb) target = yb;
        int start = yt;
        if (have && issued + 1 > start) start = issued + 1;
        if (!have || start > issued + 1) lo = start;            // a fresh contiguous range begins here
        nrows = target - start + 1;
        if (nrows < 0) nrows = 0;
        for (int y = start; y <= target; ++y) {
          seq += kRowOps;
          e.ring_set(mod_rows(y), seq);
        }
        if (nrows > 0) {
          issued = target;
          have = true;
        }
        yfirst = start;
        const int rs = e.ring_get(mod_rows(yb));
        const int need = rs > src0 ? rs : src0;
        wait = seq - need;
        if (wait > kMaxWait) wait = kMaxWait;
      }
    }
    const int mtop = mod_rows(ytop);
    e.set_step(r, mode | (nrows << 4) | (wait << 8) | (src_next << 16) | (src_pre << 17) | (mtop << 20), yfirst);
    src0 = src1;
    src1 = src2;
  }
}

// host / reference driver: stat[n] -> step[n]; returns the window's column origin xl (texel column of window column 0).
// img_w: width of the target map (the window never leaves it: 0 <= xl <= img_w - kWinTex).  row_seq: kWinRows ints of scratch.
struct StripArrayEnv {
  const StripRowStat* stat;
  StripStep* step;
  int* row_seq;
  BANET_HD int mode(int r) const { return step_mode(step[r].ctl); }     // static mode until set_step overwrites it
  BANET_HD int yt(int r) const { return stat[r].ymin - 1; }
  BANET_HD int yb(int r) const { return stat[r].ymax + 2; }
  BANET_HD void set_step(int r, int ctl, int yfirst) {
    step[r].ctl = ctl;
    step[r].yfirst = yfirst;
    step[r].ytop = step_mode(ctl) == kStepSkip ? 0 : stat[r].ymin - 1;
    step[r].pad = 0;
  }
  BANET_HD int ring_get(int i) const { return row_seq[i]; }
  BANET_HD void ring_set(int i, int v) { row_seq[i] = v; }
};
BANET_HD int strip_plan(const StripRowStat* stat, int n, int img_w, StripStep* step, int* row_seq) {
  int xmin_all = 0x3fffffff;
  for (int r = 0; r < n; ++r)
    if (stat[r].ymin <= stat[r].ymax && stat[r].xmin < xmin_all) xmin_all = stat[r].xmin;
  const int xl = strip_window_origin(xmin_all, img_w);
  int yend = -0x3fffffff;
  for (int r = 0; r < n; ++r) {
    const int mode = strip_static_mode(stat[r], xl, img_w);
    if (mode == kStepWindow && stat[r].ymax + 2 > yend) yend = stat[r].ymax + 2;
    step[r].ctl = mode;
  }
  StripArrayEnv env{stat, step, row_seq};
  strip_plan_dynamic(env, n, yend);
  return xl;
}

}  // namespace banet
