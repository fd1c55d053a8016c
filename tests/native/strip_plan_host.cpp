// host build of the strip-gather window plan (banet_amd/csrc/strip_plan.hpp) for tests/test_strip_plan_cpu.py
#include "../../banet_amd/csrc/strip_plan.hpp"
extern "C" int banet_test_strip_plan(const int32_t* stat, int n, int img_w, int32_t* steps) {
  int ring[banet::kWinRows];
  return banet::strip_plan(reinterpret_cast<const banet::StripRowStat*>(stat), n, img_w,
                           reinterpret_cast<banet::StripStep*>(steps), ring);
}
extern "C" void banet_test_strip_consts(int32_t* out) {
  out[0] = banet::kStripW; out[1] = banet::kStripH; out[2] = banet::kWinTex; out[3] = banet::kWinRows;
  out[4] = banet::kRowOps; out[5] = banet::kSrcOps; out[6] = banet::kMaxWait; out[7] = banet::kSrcAhead;
}
