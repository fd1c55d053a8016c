"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/banet_hip.h declares, mirrors the struct layouts, and validates arguments without
touching a GPU.  (No compute calls here -- those are the -m gpu tests.)"""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
# the host-side plans below are tabulated for a 256-CU part (MI355X); num_cus() would otherwise follow whatever GPU the host has
os.environ.setdefault("BANET_NUM_CUS", "256")
HEADER = os.path.join(ROOT, "include", "banet_hip.h")


@pytest.fixture(scope="module")
def capi():
    sys.path.insert(0, ROOT)
    from banet_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _capi


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(banet_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(capi):
    L = capi.lib()
    names = declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(L, n), "libbanet_hip.so does not export %s" % n
        assert n in capi.EXPORTS, "ctypes binding lacks %s" % n


def test_version_and_error_strings(capi):
    L = capi.lib()
    assert L.banet_version() == 150
    assert L.banet_error_string(0) == b"ok"
    assert b"workspace" in L.banet_error_string(-2)


def test_struct_layouts_match_the_header(capi, tmp_path):
    """compile a tiny C program against the header and compare sizeof/offsetof with ctypes"""
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "banet_hip.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(banet_level_t), '
                    'offsetof(banet_level_t, scale), offsetof(banet_level_t, src), offsetof(banet_level_t, intr), '
                    'sizeof(banet_mlp_t), sizeof(banet_state_t), offsetof(banet_state_t, iters), '
                    'offsetof(banet_level_t, variant), offsetof(banet_level_t, pairs));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(capi.Level), capi.Level.scale.offset, capi.Level.src.offset, capi.Level.intr.offset,
            ctypes.sizeof(capi.Mlp), ctypes.sizeof(capi.State), capi.State.iters.offset, capi.Level.variant.offset, capi.Level.pairs.offset]
    assert got == want
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "banet_hip.h"\n'
                    'int main(){printf("%zu %zu %zu\\n", sizeof(banet_lm_params_t), offsetof(banet_lm_params_t, residual_ratio), '
                    'offsetof(banet_lm_params_t, solver));return 0;}\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [ctypes.sizeof(capi.LmParams), capi.LmParams.residual_ratio.offset, capi.LmParams.solver.offset]


def test_lm_params_defaults_are_the_reference_globals(capi):
    """banet_lm_params_default == legacy/ba.py:5-9; banet_lm_level_ex_f32 validates the struct before touching the GPU"""
    from banet_amd import ops, legacy
    p = ops.lm_params()
    assert abs(p.angle_change - 0.002 * (3.14 / 180.0)) < 1e-12 and abs(p.translation_change - 0.0002) < 1e-10
    assert p.residual_ratio == 1.0 and p.solver == capi.SOLVER_QR
    assert ops.lm_params(qr=False).solver == capi.SOLVER_INVERSE
    # the module globals of banet_amd.legacy are read at call time (the reference's drivers overwrite them)
    old = legacy.angle_change, legacy.qr
    try:
        legacy.angle_change, legacy.qr = 0.5, False
        q = legacy._params()
        assert q.angle_change == 0.5 and q.solver == capi.SOLVER_INVERSE
    finally:
        legacy.angle_change, legacy.qr = old
    L = capi.lib()
    lv = capi.Level()
    lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = 1, 64, 8, 0, 8, 8
    lv.variant, lv.dense = capi.LEGACY_LM, 0
    assert L.banet_lm_level_ex_f32(ctypes.byref(lv), None, 1.0, 1, 1, ctypes.byref(p), None, None, 0, None) == -1


def test_lambda_mlp_shapes_are_checked_before_launch(capi):
    """the solve kernel reads the MLP with fixed extents C -> 2C -> 4C -> 2C -> C -> 1; weights for another C
    must be refused on the host, not read out of bounds on the device."""
    import torch
    from banet_amd import ops
    from banet_amd.bundlenet import he_normal_lambda_weights
    good = ops.MlpWeights(he_normal_lambda_weights(8, 1), "cpu")
    good.check(8)
    with pytest.raises(capi.BanetError):
        good.check(16)                                      # weights of a C = 8 level offered to a C = 16 level
    bad = he_normal_lambda_weights(8, 1)
    bad[2] = (torch.zeros(32, 8), torch.zeros(8))           # chains (32 -> 8 -> ...) but is not 4C -> 2C
    bad[3] = (torch.zeros(8, 8), torch.zeros(8))
    with pytest.raises(capi.BanetError):
        ops.MlpWeights(bad, "cpu")


def test_mlp_cache_follows_weight_updates():
    """reassigned or in-place-updated lambda weights must not leave a stale device copy behind"""
    import torch
    from banet_amd import ops
    from banet_amd.bundlenet import he_normal_lambda_weights
    lw = {"0": he_normal_lambda_weights(4, 3)}
    cache = ops.MlpCache()
    m1 = cache.get(lw, 0, "cpu")
    assert cache.get(lw, "0", "cpu") is m1                  # unchanged -> cached
    lw["0"][0][0].mul_(2.0)                                 # in-place update (optimizer step)
    m2 = cache.get(lw, 0, "cpu")
    assert m2 is not m1 and torch.equal(m2.w[0], lw["0"][0][0])
    lw["0"] = he_normal_lambda_weights(4, 9)                # checkpoint reload
    m3 = cache.get(lw, 0, "cpu")
    assert m3 is not m2 and torch.equal(m3.w[4], lw["0"][4][0])
    with pytest.raises(KeyError):
        cache.get(lw, 7, "cpu")


def test_argument_validation_without_gpu(capi):
    L = capi.lib()
    assert L.banet_equation_construction_workspace_bytes(1, 4096, 128, 6) > 0
    assert L.banet_equation_construction_workspace_bytes(1, 4096, 128, 134) > 0
    assert L.banet_equation_construction_workspace_bytes(1, 4096, 128, 298) > 0       # cfg-5's P: 19 blocks, the LDS-tiled kernel
    assert L.banet_equation_construction_workspace_bytes(1, 4096, 128, 305) == 0      # > 19 blocks: unsupported
    assert L.banet_equation_construction_workspace_bytes(0, 4096, 128, 6) == 0
    assert L.banet_equation_construction_f32(None, None, None, None, None, 1, 8, 4, 6, None, 0, None) == -1
    assert L.banet_equation_construction_grad_f32(None, None, None, None, None, None, None, None, 1, 8, 4, 6, None, 0,
                                                  None) == -1
    lv = capi.Level()
    lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = 2, 160 * 120, 128, 32, 120, 160
    lv.variant, lv.dense, lv.scale = capi.BUNDLE, 1, 1.0
    nb = L.banet_lm_level_workspace_bytes(ctypes.byref(lv))
    assert nb > 0 and nb % 256 == 0
    assert L.banet_ba_assemble_workspace_bytes(ctypes.byref(lv)) <= nb
    lv.K = 200                                                                           # large bases: the solve's matrix moves
    assert L.banet_lm_level_workspace_bytes(ctypes.byref(lv)) > nb                        # into the workspace
    lv.K = 257                                                                           # beyond the compiled set
    assert L.banet_lm_level_workspace_bytes(ctypes.byref(lv)) == 0
    lv.K = 32
    assert L.banet_ba_assemble_f32(ctypes.byref(lv), None, None, None, None, None, None, None, None, 0, None) == -1


def test_gpu_only_no_fallback(capi):
    """the product path must fail loudly on CPU tensors instead of silently computing elsewhere"""
    import torch
    from banet_amd import ops
    J = torch.zeros(1, 8, 2, 6)
    G = torch.zeros(1, 8, 4, 2)
    d = torch.zeros(1, 8, 4, 1)
    with pytest.raises(capi.BanetError):
        ops.equation_construction(J, G, d)


def test_dispatcher_registration_is_hip_only():
    """torch.ops.banet.equation_construction[_grad] exist, infer shapes on meta tensors, and have no CPU kernel"""
    import torch
    from banet_amd import ops  # noqa: F401  (registers the ops)
    J, G, d = (torch.zeros(s, device="meta") for s in ((2, 8, 2, 38), (2, 8, 4, 2), (2, 8, 4, 1)))
    AtA, Atb = torch.ops.banet.equation_construction(J, G, d)
    assert tuple(AtA.shape) == (2, 38, 38) and tuple(Atb.shape) == (2, 38, 1)
    gJ, gG, gd = torch.ops.banet.equation_construction_grad(J, G, d, AtA, Atb)
    assert gJ.shape == J.shape and gG.shape == G.shape and gd.shape == d.shape
    with pytest.raises(NotImplementedError):
        torch.ops.banet.equation_construction(torch.zeros(1, 8, 2, 6), torch.zeros(1, 8, 4, 2), torch.zeros(1, 8, 4, 1))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "banet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt.replace(
                    "oracle/synth.py", ""), f


def test_plain_c_program_links_against_the_library(tmp_path):
    """the drop-in boundary is a C ABI: a C translation unit (no C++, no torch) includes the header, links
    libbanet_hip.so and calls entry points that need no GPU"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "banet_hip.h"\n'
                   'int main(void){ banet_level_t lv; memset(&lv, 0, sizeof lv);\n'
                   '  if (banet_version() != BANET_VERSION) return 1;\n'
                   '  if (strcmp(banet_error_string(BANET_OK), "ok")) return 2;\n'
                   '  if (banet_ba_assemble_f32(&lv, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) != BANET_ERR_INVALID_ARG) return 3;\n'
                   '  if (banet_resample_f32(0, 0, 0, 1, 1, 1, 1, 1, BANET_RESAMPLE_CLAMP, 0) != BANET_ERR_INVALID_ARG) return 4;\n'
                   '  printf("c-abi ok %d\\n", banet_version()); return 0; }\n')
    libdir = os.path.join(ROOT, "banet_amd", "lib")
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lbanet_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)]).decode()
    assert out.startswith("c-abi ok 150")


def test_backward_entry_points_validate_their_arguments_without_gpu(capi):
    """banet_dense_adjoint_f32 / banet_target_map_adjoint_f32 / banet_sample_stats_grad_det_f32: supported-shape queries and
    argument validation run on the host (no launch)."""
    L = capi.lib()
    lv = capi.Level()
    lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = 2, 48 * 64, 128, 128, 48, 64
    lv.variant, lv.dense, lv.scale, lv.pairs = capi.BUNDLE, 1, 1.0, 1
    nb = L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv))
    assert nb > 0 and nb % 256 == 0
    for field, bad in (("K", 257), ("K", 0), ("C", 257), ("pairs", 2), ("dense", 0), ("tgt_has_grad", 1), ("N", 100)):
        keep = getattr(lv, field)
        setattr(lv, field, bad)
        assert L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv)) == 0, field      # outside the supported set
        setattr(lv, field, keep)
    lv.K = 256
    assert L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv)) > 0                  # round 3: K <= 256
    lv.variant = capi.BUNDLE_CAMERA                                                     # pose only: K = 0 exactly
    assert L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv)) == 0
    lv.K = 0
    assert L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv)) > 0
    lv.variant, lv.K = capi.BUNDLE, 128
    assert L.banet_dense_adjoint_f32(ctypes.byref(lv), *([None] * 11), None, 0, None) == -1   # null pointers
    assert L.banet_dense_adjoint_f32(None, *([None] * 11), None, 0, None) == -1
    assert L.banet_target_map_adjoint_f32(None, None, 1, 4, 4, 8, None) == -1
    assert L.banet_sample_stats_grad_workspace_bytes(2, 1000, 128, 48, 64) > 0
    assert L.banet_sample_stats_grad_workspace_bytes(2, 1000, 257, 48, 64) == 0              # C > 256
    assert L.banet_sample_stats_grad_det_f32(*([None] * 4), 2, 1000, 128, 48, 64, *([None] * 5), None, 0, None) == -1
    assert L.banet_spd_solve_f32(None, None, None, 1, 134, None) == -1
    buf = ctypes.c_void_p(4096)                                                                    # never dereferenced: rejected on the host
    assert L.banet_spd_solve_f32(buf, buf, buf, 1, 8, None) == -3 and L.banet_spd_solve_f32(buf, buf, buf, 1, 400, None) == -3


def test_solve_update_workspace_query_without_gpu(capi):
    """banet_ba_solve_update_workspace_bytes: 0 while the damped system fits the LDS (banet_ba_solve_update_f32 alone is enough),
    the matrix's size beyond that (K = 256 levels: banet_ba_solve_update_ws_f32); argument validation on the host."""
    L = capi.lib()
    lv = capi.Level()
    lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = 8, 1280 * 960, 128, 128, 960, 1280
    lv.variant, lv.dense, lv.scale, lv.pairs = capi.BUNDLE, 1, 1.0, 1
    assert L.banet_ba_solve_update_workspace_bytes(ctypes.byref(lv)) == 0             # P = 134
    lv.K = 256
    nb = L.banet_ba_solve_update_workspace_bytes(ctypes.byref(lv))                        # P = 262
    assert nb >= 8 * 262 * 262 * 4 and nb % 256 == 0
    lv.pairs = 7                                                                         # cfg-5: P = 298
    assert L.banet_ba_solve_update_workspace_bytes(ctypes.byref(lv)) > nb
    assert L.banet_ba_solve_update_workspace_bytes(None) == 0
    assert L.banet_ba_solve_update_ws_f32(ctypes.byref(lv), None, 1000.0, None, None, None, None, None, None, 0, None) == -1


def test_bench_roofline_record_arithmetic():
    """bench.py's roofline object: the HBM side (algorithmic bytes / gather time) and the matrix-core side of the SYRK
    (algorithmic fp32 flops and executed bf16 MFMA flops / SYRK time) from a synthetic launch profile."""
    import types
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    N, K, B = 1024, 128, 2
    lvl = types.SimpleNamespace(N=N, c=types.SimpleNamespace(W=32, H=32, K=K))
    ba = types.SimpleNamespace(problems=[lvl], algorithmic_bytes_per_iteration=lambda li: 4 * N * (2 * 128 + K + 1))
    prob = types.SimpleNamespace(ba=ba, B=B, pairs=1)
    prof = {N: (10, 2.0), -N: (10, 1.0)}            # 10 gather launches in 2 ms, 10 SYRK launches in 1 ms
    r = bench.roofline_record(prob, prof, elapsed_s=0.004, traffic=123)
    by = 4 * N * (2 * 128 + K + 1) * B * 10
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] == 123
    assert abs(r["achieved"] - by / 2.0 / 1e6) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert r["launches"] == 10 and abs(r["kernel_time_share"] - 0.5) < 1e-6
    assert abs(r["traffic_GBps"] - 123 / 0.2e-3 / 1e9) < 0.1 and r["streaming_copy_GBps"] == 6300.0
    m = r["syrk_kernel"]["mfma"]
    flops = N * (K * (K + 1) + 14 * K) * B * 10
    executed = N / 32 * 6 * 16384 * (36 + 8) * B * 10
    assert m["bound"] == "mfma" and m["peak"] == 2500.0 and m["peak_fp32_matrix_TFLOPs"] == 157.3
    assert abs(m["algorithmic_fp32_TFLOPs"] - flops / 1.0 / 1e9) < 0.06
    assert abs(m["achieved"] - executed / 1.0 / 1e9) < 0.06 and abs(m["frac"] - m["achieved"] / 2500.0) < 1e-3


def test_bench_compact_line_stays_under_4_kb(tmp_path, capsys, monkeypatch):
    """bench.py's LAST stdout line is the driver's record: one JSON object < 4 KB with the contract's fields, `roofline` and
    `cpu_baseline` (BENCH_r03's 38 KB line was not parseable); the full record goes to bench_detail.json and '#detail' lines.
    Stubbed run: the full record of a real run (profiles/r03_run28_bench.json), bloated further."""
    import json
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_run28_bench.json")))
    for i in range(12):         # a sweep twice as long as today's
        full["sweep"]["extra_%d" % i] = dict(full["sweep"]["B8_2frame"])
    monkeypatch.setenv("BANET_BENCH_DETAIL_DIR", str(tmp_path))
    bench.emit(full)
    lines = capsys.readouterr().out.strip().split("\n")
    last = lines[-1]
    assert len(last) < 4096 and all(l.startswith("#detail") for l in lines[:-1])
    rec = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "sweep"):
        assert k in rec, k
    assert "workload" in rec["config"] and "model" not in rec["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert rec["value"] == full["value"] and rec["roofline"]["frac"] == full["roofline"]["frac"]
    assert json.load(open(tmp_path / "bench_detail.json"))["sweep"].keys() == full["sweep"].keys()
    # a record too large even in compact form degrades to pointers instead of growing
    for i in range(200):
        full["sweep"]["more_%d" % i] = dict(full["sweep"]["B8_2frame"])
    assert len(bench.compact_record(full)) < 4096


def test_bench_compact_line_of_an_8_rank_run(tmp_path, capsys, monkeypatch):
    """The N-rank line the driver's scaling run parses (bench.py --gpus 8 under torch.distributed.run): < 4 KB, carries
    config.world_size / backend / parallelism, and -- round 5 -- the parity numbers per group (everything the damping conditions vs
    the undamped last coefficient with its float32 yardstick) and the SYRK's arithmetic form per level.  Stubbed from a real
    record (profiles/r04_run12_bench_detail.json) with the round-5 fields filled in the way bench.py fills them."""
    import json
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_run12_bench_detail.json")))
    full["n_gpus"] = 8
    full["config"].update(world_size=8, backend="nccl", parallelism="dp8: 32 windows per GPU, contiguous shards, one all-gather per solve",
                          windows_total=256)
    full["parity"].update(max_pose_depth=3.9e-5, max_last=8.1e-5, max_last_ref32=2.4e-2)
    full["roofline"]["syrk_form"] = "40x30:b16x3 80x60:b16x3 160x120:b16x3 320x240:f16x2 640x480:f16x2"
    for rec in full["sweep"].values():
        if "parity" in rec:
            rec["parity"].update(max_pose_depth=1.8e-6, max_last=2.579e-3, max_last_ref32=1.728e-3)
    # round 6: at N > 1 ranks the sweep holds BASELINE's two 8-GPU configurations, run by all ranks (bench.py main): configs[3] and
    # configs[4], each with the whole job's value, the step time, the gather's roofline fraction and the slowest / fastest rank
    one = dict(full["sweep"]["cfg3_5frame_B32"])
    one.pop("parity", None)
    full["sweep"] = {"cfg4_5frame_B256_dp8": dict(one, value=81234.5, n_gpus=8, windows_total=256, rank_ms_per_step_max=157.3,
                                                  rank_ms_per_step_min=151.9),
                     "cfg5_8frame_B64_dp8": dict(one, value=8650.1, n_gpus=8, windows_total=64, rank_ms_per_step_max=552.0,
                                                 rank_ms_per_step_min=541.2)}
    full.update(value_exact_syrk=190000.0, ms_per_step_exact_syrk=67.1, rank_ms_per_step_max=63.9, rank_ms_per_step_min=62.8,
                build_mode="build_id=0123456789abcdef recompiled=0 of 18 objects (2026-10-01T00:00:00Z)")
    monkeypatch.setenv("BANET_BENCH_DETAIL_DIR", str(tmp_path))
    bench.emit(full)
    last = capsys.readouterr().out.strip().split("\n")[-1]
    assert len(last) < 4096, len(last)
    rec = json.loads(last)
    assert set(rec["sweep"]) == {"cfg4_5frame_B256_dp8", "cfg5_8frame_B64_dp8"}
    for e in rec["sweep"].values():
        assert e["n_gpus"] == 8 and e["rank_ms_per_step_max"] >= e["rank_ms_per_step_min"] > 0 and e["value"] > 0 and "frac" in e
    assert rec["value_exact_syrk"] == 190000.0 and rec["rank_ms_per_step_max"] == 63.9 and rec["build_mode"].startswith("build_id=")
    assert rec["n_gpus"] == 8 and rec["config"]["world_size"] == 8 and rec["config"]["backend"] == "nccl"
    assert "dp8" in rec["config"]["parallelism"] and rec["scaling"] == "weak"
    assert rec["parity"]["max_pose_depth"] == 3.9e-5 and rec["parity"]["max_last"] == 8.1e-5 and rec["parity"]["ok"] is True
    assert rec["roofline"]["syrk_form"].endswith("640x480:f16x2")


def test_bench_compact_line_carries_the_round6_parity_fields(tmp_path, capsys, monkeypatch):
    """1-rank line: per sweep entry [lambda / pose / damped depth, last coefficient, its float32 yardstick], `own_mask_max` where an
    entry needed the GPU's mask bits, the l2_base = 1 scene of the headline parity and the `backward` block."""
    import json
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_run12_bench_detail.json")))
    for name, rec in full["sweep"].items():
        if "parity" in rec:
            rec["parity"].update(max_pose_depth=1.8e-6, max_last=2.579e-3, max_last_ref32=1.728e-3,
                                 own_mask_max=[2.2e-4, 0.31] if name == "B256_2frame" else None)
    full["parity"]["large_step_scene"] = {"step_pose_depth": 4.1e-5, "max_last_ref32": 3.0e-3, "ok": True,
                                          "per_level": {"640x480": {"step_last": 1.2e-3}, "40x30": {"step_last": 7e-5}}}
    full["backward"] = {"ms": 99.0, "forward_only_ms": 13.5, "x_forward": 7.33, "peak_extra_GB": 19.99}
    monkeypatch.setenv("BANET_BENCH_DETAIL_DIR", str(tmp_path))
    bench.emit(full)
    last = capsys.readouterr().out.strip().split("\n")[-1]
    assert len(last) < 4096, len(last)
    rec = json.loads(last)
    for name, e in rec["sweep"].items():
        if "parity" in e:                          # [lambda / pose / damped depth, last coefficient, its float32 yardstick]
            assert e["parity"] == [1.8e-6, 2.579e-3, 1.728e-3], name
            assert e["parity"][0] <= 1e-4 and e["parity"][1] <= max(1e-4, 2 * e["parity"][2])
    assert rec["sweep"]["B256_2frame"]["own_mask_max"] == [2.2e-4, 0.31] and "own_mask_max" not in rec["sweep"]["B8_2frame"]
    assert rec["parity"]["l2_base_1"] == [4.1e-5, 1.2e-3, 3.0e-3, True]
    assert rec["backward"]["x_forward"] == 7.33


def test_strip_gather_register_contract():
    """ba_gather128s_kernel reserves v224..v255 behind the compiler's back (amdgpu_num_vgpr + inline asm naming them): the
    compiler-emitted instructions of every instantiation must not touch them and the default variants must not spill
    (tools/check_strip_regs.py; build.sh runs the same check whenever it recompiles gather128s.hip)."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_strip_regs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 violations" in r.stdout


def _level(capi, B, H, W, K, pairs, reserved=0, policy=0):
    lv = capi.Level()
    lv.policy = policy
    lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = B, H * W, 128, K, H, W
    lv.variant, lv.dense, lv.scale, lv.pairs, lv.normalize_rays = capi.BUNDLE, 1, 1.0, pairs, 1
    lv.flags = reserved
    return lv


def test_kernel_selection_table_of_the_baseline_configs(capi):
    """Which assembly (gather) and depth-block (SYRK) kernels BASELINE.json's configurations run, level by level -- the host-side
    plans need no GPU (256 CUs assumed when no device is visible).  banet_gather_selection: 1 direct tiles, 2 LDS patches,
    3 strip segments, 4 4x4-pixel items; banet_syrk_selection: 2 bf16 x 3 pieces, 3 wide-basis jobs (bf16), 4 fp16 x 2 pieces."""
    L = capi.lib()
    sel = lambda *a, **k: (L.banet_gather_selection(ctypes.byref(_level(capi, *a, **k))),       # noqa: E731
                           L.banet_syrk_selection(ctypes.byref(_level(capi, *a, **k))))
    pyr = [(30, 40), (60, 80), (120, 160), (240, 320), (480, 640)]
    # the metric's configuration: 2-frame, batch 32
    assert [sel(32, h, w, 128, 1) for h, w in pyr] == [(4, 2), (4, 2), (3, 2), (3, 4), (3, 4)]
    # batch 1: every level on the 4x4-item gather, exact SYRK (latency-bound launches)
    assert [sel(1, h, w, 128, 1) for h, w in pyr] == [(4, 2)] * 5
    # batch 8
    assert [sel(8, h, w, 128, 1) for h, w in pyr] == [(4, 2), (4, 2), (4, 2), (3, 2), (3, 4)]
    # cfg-3: 5-frame windows, batch 32 (frame-parallel strip gather from 160x120 up)
    assert [sel(32, h, w, 128, 4) for h, w in pyr] == [(4, 2), (2, 2), (3, 2), (3, 4), (3, 4)]
    # cfg-5 share: 8 windows x 8 frames, 1280x960, K = 256 (wide-basis SYRK jobs; fp16 form on the two finest levels)
    pyr5 = [(60, 80), (120, 160), (240, 320), (480, 640), (960, 1280)]
    assert [sel(8, h, w, 256, 7) for h, w in pyr5] == [(1, 3), (2, 3), (3, 3), (3, 4), (3, 4)]
    # cfg-1: 160x120, K = 32, one window
    assert sel(1, 120, 160, 32, 1) == (4, 0)
    # the documented development bits
    h, w = pyr[0]
    assert sel(32, h, w, 128, 1, reserved=1 << 30)[0] == 1 and sel(32, 480, 640, 128, 1, reserved=-2147483648)[1] == 2
    assert sel(32, h, w, 128, 1, reserved=(1 << 18) | 1024)[0] == 3
    assert sel(6, 240, 320, 128, 1)[0] == 1 and sel(6, 240, 320, 128, 1, reserved=1 << 25)[0] == 4      # (28800 items: beyond its limit -> tiles)


def test_batch_invariant_policy_selects_by_level_only(capi):
    """banet_level_t.policy = BANET_POLICY_BATCH_INVARIANT: the gather kernel and the SYRK form of a level are the ones a batch of
    BANET_CANONICAL_BATCH = 32 windows runs, whatever the launch's own batch -- so a window's arithmetic does not depend on its shard
    size (the bit-identity itself is a -m gpu test).  The default policy does depend on it (cfg-5: 64 windows on one GPU vs 8 per GPU)."""
    L = capi.lib()
    sel = lambda *a, **k: (L.banet_gather_selection(ctypes.byref(_level(capi, *a, **k))),       # noqa: E731
                           L.banet_syrk_selection(ctypes.byref(_level(capi, *a, **k))))
    pyr = [(30, 40), (60, 80), (120, 160), (240, 320), (480, 640)]
    for pairs in (1, 4):
        want = [sel(32, h, w, 128, pairs) for h, w in pyr]
        for B in (1, 2, 8, 13, 32, 64, 256):
            assert [sel(B, h, w, 128, pairs, policy=capi.POLICY_BATCH_INVARIANT) for h, w in pyr] == want, (pairs, B)
    # the dependence the policy removes: cfg-5's 320x240 level, K = 256, 7 target frames
    assert sel(64, 240, 320, 256, 7)[1] == 4 and sel(8, 240, 320, 256, 7)[1] == 3
    assert sel(64, 240, 320, 256, 7, policy=1) == sel(8, 240, 320, 256, 7, policy=1)
    # the field that used to be `pad_` still rejects garbage: the plans (workspace sizes, selections) refuse an unknown policy
    lv = _level(capi, 8, 30, 40, 128, 1, policy=7)
    assert L.banet_lm_level_workspace_bytes(ctypes.byref(lv)) == 0 and L.banet_ba_assemble_workspace_bytes(ctypes.byref(lv)) == 0
    assert L.banet_gather_selection(ctypes.byref(lv)) == -1 and L.banet_syrk_selection(ctypes.byref(lv)) == -1      # BANET_ERR_INVALID_ARG
    lv.policy = 1
    assert L.banet_lm_level_workspace_bytes(ctypes.byref(lv)) > 0 and L.banet_gather_selection(ctypes.byref(lv)) > 0
    # the old field name is an alias of the new one (tools/ still use it)
    lv.reserved_ = 1 << 18
    assert lv.flags == 1 << 18


def test_adjoint_accepts_the_sparse_reference_layout(capi):
    """banet_dense_adjoint_workspace_bytes (host-side plan, no launch): the backward of the assembly is compiled for the dense layout
    AND -- round 5 -- for the reference's own sparse layout (conv1 [B,N,C] at N sampled points, rays + per-point intrinsics, the
    [f|gx|gy] target map: bundlenet.py:332-399); the cell arrays of the per-texel gather are sized by H * W, not by N."""
    L = capi.lib()
    one = (ctypes.c_float * 4)()
    ptr = ctypes.cast(one, ctypes.c_void_p).value      # any non-null pointer: the plan only checks presence

    def ws(dense, tgt_has_grad, N, H, W, C=128, K=128, with_rays=True, variant=None):
        lv = capi.Level()
        lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = 2, N, C, K, H, W
        lv.variant = capi.BUNDLE if variant is None else variant
        lv.dense, lv.tgt_has_grad, lv.scale, lv.pairs, lv.normalize_rays = dense, tgt_has_grad, 1.0, 1, 1
        lv.src = lv.tgt = lv.depth = lv.basis = lv.intr = ptr
        if with_rays:
            lv.rays = lv.fx = lv.fy = lv.ox = lv.oy = ptr
        return L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv))

    dense = ws(1, 0, 48 * 64, 48, 64)
    sparse_small_map = ws(0, 1, 4096, 48, 64)
    sparse_big_map = ws(0, 1, 4096, 384, 512)
    assert dense > 0 and sparse_small_map > 0 and sparse_big_map > sparse_small_map      # H * W enters the size
    assert ws(0, 1, 4096, 384, 512, with_rays=False) == 0          # sparse points need rays and per-point intrinsics
    assert ws(0, 0, 4096, 384, 512) == 0 and ws(1, 1, 48 * 64, 48, 64) == 0      # the two mixed layouts are not compiled
    assert ws(0, 1, 4096, 384, 512, C=200, K=200) == 0             # sparse: not C > 128 together with K > 128 ...
    assert ws(0, 1, 4096, 384, 512, C=200, K=64) > 0 and ws(0, 1, 4096, 384, 512, C=64, K=200) > 0
    assert ws(0, 1, 4096, 384, 512, K=0, variant=capi.BUNDLE_CAMERA) > 0       # pose-only iteration


def test_round6_backward_entry_points_on_the_host(capi):
    """banet_dense_adjoint_workspace_bytes_ex / banet_small_step_adjoint_*: the host-side plans and argument checks (no launch).
    BANET_ADJOINT_FOLD_TARGET drops the 3C adjoint rows from the workspace, exists for the dense layout only; the small step is
    compiled for bundle (K >= 1) and bundle_camera (K = 0) with a system the SPD kernel can hold (or P < 32)."""
    L = capi.lib()
    FOLD = 4
    lv = capi.Level()
    lv.B, lv.N, lv.C, lv.K, lv.H, lv.W = 32, 480 * 640, 128, 128, 480, 640
    lv.variant, lv.dense, lv.scale, lv.pairs, lv.normalize_rays = capi.BUNDLE, 1, 1.0, 1, 1
    plain, fold = L.banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(lv), 0), L.banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(lv), FOLD)
    assert plain == L.banet_dense_adjoint_workspace_bytes(ctypes.byref(lv)) and 0 < fold < plain
    rows = 4 * 32 * 480 * 640 * 3 * 128                       # the [B, N, 3C] adjoint rows: 15.1 GB that no longer exist
    assert plain - fold > 0.95 * rows - 2 * 4 * 32 * 480 * 640 * 12
    assert L.banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(lv), 64 << 4) == 0          # unknown flag bits
    lv.dense, lv.tgt_has_grad = 0, 1                          # the reference's sparse layout: the row-gather path only
    keep = [ctypes.c_float(0)] * 5
    for name in ("rays", "fx", "fy", "ox", "oy"):
        setattr(lv, name, ctypes.addressof(keep[0]))
    lv.N = 4096
    assert L.banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(lv), 0) > 0 and L.banet_dense_adjoint_workspace_bytes_ex(ctypes.byref(lv), FOLD) == 0
    ws = L.banet_small_step_adjoint_workspace_bytes
    assert ws(capi.BUNDLE, 32, 307200, 128, 128, 1) > 0 and ws(capi.BUNDLE, 8, 307200, 128, 128, 4) > 0       # P = 134, 152
    assert ws(capi.BUNDLE_CAMERA, 4, 4096, 128, 0, 1) > 0                                     # P = 6: in-kernel Cholesky
    assert ws(capi.BUNDLE, 8, 1228800, 128, 256, 7) == 0                                      # P = 298: beyond the SPD kernel's LDS
    assert ws(capi.BUNDLE, 8, 1000, 300, 16, 1) == 0 and ws(capi.BUNDLE, 8, 1000, 64, 0, 1) == 0 and ws(capi.LEGACY_LM, 1, 100, 16, 0, 1) == 0
    assert L.banet_small_step_adjoint_f32(capi.BUNDLE, 2, 100, 16, 8, 1, 1000.0, None, *([None] * 14), None, None, 0, None) == -1


def test_committed_traffic_file_belongs_to_the_committed_sources():
    """profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic) names the build it was measured on; the digest build.sh
    bakes into the library -- every csrc/*.hip, *.hpp, include/banet_hip.h and the compile flags -- must be that build for the tree
    as committed: a source edit after the evidence run (even in a comment) would otherwise make bench.py report traffic = null."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "banet_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".hpp"))):
        h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.join(root, "include", "banet_hip.h"), "rb").read())
    flags = re.search(r'^FLAGS="([^"]*)"', open(os.path.join(csrc, "build.sh")).read(), re.M).group(1)
    h.update((flags + " \n").encode())           # build.sh: echo "$FLAGS ${EXTRA_HIPCC_FLAGS:-}" with no extra flags
    digest = h.hexdigest()[:16]
    traffic = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    assert traffic["build_id"] == digest, (traffic["build_id"], digest)
