#!/usr/bin/env python3
"""Extra seeds for the randomized parity tests (tests/test_gpu_parity.py::test_random_shape_sweep_assembly_matches_oracle and
the wide-basis / patch-kernel cases with random shapes): python tools/fuzz_parity.py [first_seed] [count]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_parity as T  # noqa: E402

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_shape_sweep_assembly_matches_oracle(seed)
    except Exception:
        bad += 1
        print("seed", seed, "FAILED")
        traceback.print_exc(limit=2)
print("random shape sweep: %d seeds, %d failures" % (count, bad))
import numpy as np  # noqa: E402
rng = np.random.RandomState(first)
cases_wide = [(128, int(rng.choice([128, 256])), int(rng.randint(1, 8))) for _ in range(max(2, count // 6))]
cases_wide = [(c, k, p if k == 256 else max(p, 5)) for c, k, p in cases_wide]          # K = 128 takes the job kernels with > 4 frames
for c in cases_wide:
    try:
        T.test_large_basis_windows_match_oracle(*c)
    except Exception:
        bad += 1
        print("wide SYRK case", c, "FAILED")
        traceback.print_exc(limit=2)
cases_patch = [(int(rng.randint(16, 72)), int(rng.randint(16, 100)), int(rng.choice([0, 4, 32, 128])), bool(rng.randint(2)),
                int(rng.randint(1, 4))) for _ in range(max(2, count // 6))]
for c in cases_patch:
    try:
        T.test_patch_gather_kernel_matches_oracle(*c)
    except Exception:
        bad += 1
        print("patch gather case", c, "FAILED")
        traceback.print_exc(limit=2)
print("wide SYRK cases %s, patch gather cases %s: %d failures so far" % (cases_wide, cases_patch, bad))
# round 3: the strip gather forced at random shapes (ragged strips / segments, 0..3 target frames, K up to 256, large motion)
import test_gpu_round3 as T3  # noqa: E402
cases_strip = [(int(rng.randint(8, 90)), int(rng.randint(21, 120)), int(rng.choice([0, 4, 16, 32, 64, 128, 256])), bool(rng.randint(2)),
                int(rng.randint(1, 4))) for _ in range(max(4, count // 3))]
for c in cases_strip:
    try:
        T3.test_strip_gather_kernel_matches_oracle(*c)
    except Exception:
        bad += 1
        print("strip gather case", c, "FAILED")
        traceback.print_exc(limit=2)
print("strip gather cases %s: %d failures so far" % (cases_strip, bad))
cases_eq = [(int(rng.randint(1, 4)), int(rng.randint(1, 700)), int(rng.choice([1, 3, 8, 64, 128, 200])), int(rng.choice([6, 7, 12, 38, 70, 134, 143, 144, 145, 160, 199, 200, 255, 262, 271, 272])))
            for _ in range(max(4, count // 4))]
for c in cases_eq:
    for fn in (T.test_equation_construction_matches_oracle, T.test_equation_construction_grad_matches_oracle):
        try:
            fn(*c)
        except Exception:
            bad += 1
            print("EquationConstruction case", c, fn.__name__, "FAILED")
            traceback.print_exc(limit=2)
cases_ss = [(int(rng.randint(1, 3)), int(rng.randint(8, 400)), int(rng.choice([1, 5, 64, 128, 256])), int(rng.randint(4, 30)), int(rng.randint(4, 40)))
            for _ in range(max(4, count // 6))]
for c in cases_ss:
    try:
        T.test_sample_stats_op_matches_the_torch_statements(*c)
    except Exception:
        bad += 1
        print("sample_stats case", c, "FAILED")
        traceback.print_exc(limit=2)
print("EquationConstruction cases %s, sample_stats cases %s: %d failures so far" % (cases_eq, cases_ss, bad))
# round 4: the 4x4-pixel-item gather forced at random shapes; the mask output of every gather kernel; the fp16 two-piece SYRK
# (both engines: K = 64 / 128 up to 4 frames, and the syrk_wide.hip jobs) with random column spans; the literal op up to P = 304
import test_gpu_round4 as T4  # noqa: E402
cases_quad = [(int(rng.randint(4, 90)), int(rng.randint(4, 120)), int(rng.choice([0, 4, 16, 32, 64, 128, 256])), bool(rng.randint(2)),
               int(rng.randint(1, 5))) for _ in range(max(4, count // 3))]
for c in cases_quad:
    try:
        T4.test_quad_gather_kernel_matches_oracle(*c)
    except Exception:
        bad += 1
        print("quad gather case", c, "FAILED")
        traceback.print_exc(limit=2)
cases_mask = [(int(rng.randint(24, 70)), int(rng.randint(24, 100)), int(rng.choice([0, 32, 128])), int(rng.randint(1, 5)), True)
              for _ in range(max(2, count // 8))]
for c in cases_mask:
    try:
        T4.test_mask_output_of_every_gather_kernel(*c)
    except Exception:
        bad += 1
        print("mask output case", c, "FAILED")
        traceback.print_exc(limit=2)
cases_f16 = [(int(k), int(rng.randint(24, 100)), int(rng.randint(32, 130)), int(rng.randint(1, 8 if k != 64 else 5)), float(rng.choice([0.0, 6.0, 12.0, 30.0, 60.0])))
             for k in rng.choice([64, 128, 256], size=max(4, count // 4))]
for c in cases_f16:
    try:
        T4.test_syrk_f16_two_piece_with_basis_columns_spanning_many_octaves(*c)
    except Exception:
        bad += 1
        print("fp16 SYRK case", c, "FAILED")
        traceback.print_exc(limit=2)
cases_eq2 = [(int(rng.randint(1, 3)), int(rng.randint(1, 300)), int(rng.choice([1, 8, 128])), int(rng.choice([273, 280, 298, 300, 304]))) for _ in range(max(2, count // 10))]
for c in cases_eq2:
    for fn in (T.test_equation_construction_matches_oracle, T.test_equation_construction_grad_matches_oracle):
        try:
            fn(*c)
        except Exception:
            bad += 1
            print("EquationConstruction case", c, fn.__name__, "FAILED")
            traceback.print_exc(limit=2)
print("quad gather cases %s, mask cases %s, fp16 SYRK cases %s, P > 272 op cases %s: %d failures in all" % (cases_quad, cases_mask, cases_f16, cases_eq2, bad))
sys.exit(1 if bad else 0)
