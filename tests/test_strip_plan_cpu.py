"""The rolling-window plan of the strip gather (banet_amd/csrc/strip_plan.hpp, shared by the HIP kernel and this host
build): the planned instruction stream is replayed against a model of the LDS ring and of the in-order VMEM counter.
For every pixel row served from the window: every texel row it reads is resident in its ring slot, was not overwritten by
this step's own loads, and has LANDED under the counted s_waitcnt the plan prescribes (loads retire in issue order, so
`vmcnt(n)` after S issued operations means operations 1 .. S - n are complete); likewise the row's source features."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("strip") / "libstrip_plan_host.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                           os.path.join(ROOT, "tests", "native", "strip_plan_host.cpp")])
    L = ctypes.CDLL(str(out))
    L.banet_test_strip_plan.restype = ctypes.c_int
    return L


def consts(lib):
    c = (ctypes.c_int32 * 8)()
    lib.banet_test_strip_consts(c)
    return dict(W=c[0], H=c[1], TEX=c[2], ROWS=c[3], ROWOPS=c[4], SRCOPS=c[5], MAXWAIT=c[6], AHEAD=c[7])


def plan(lib, stat, img_w):
    stat = np.ascontiguousarray(stat, np.int32)
    n = stat.shape[0]
    steps = np.zeros((n, 4), np.int32)
    xl = lib.banet_test_strip_plan(stat.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n, img_w,
                                   steps.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return xl, steps


def _static_window(row, xl, k, img_w):
    """the plan's static fit test (a row may still be demoted to direct by the rolling state; its source prefetch stays)"""
    ymin, ymax, xmin, xmax = (int(v) for v in row)
    return (xmin - 1 >= xl and xmax + 2 - xl <= k["TEX"] - 1 and (ymax + 2) - (ymin - 1) + 1 <= k["ROWS"] and img_w >= k["TEX"])


def replay(stat, xl, steps, k, img_w):
    """-> (#window steps, #direct steps, rows loaded).  Asserts the safety properties."""
    ring = {}            # slot -> (texel row, seq of its last operation)
    seq = 0
    src_seq = {}         # step -> seq of its source loads
    nwin = ndir = loaded = 0
    for r in range(min(k["AHEAD"], len(stat))):
        assert ((int(steps[r, 0]) >> 17) & 1) == int(stat[r][0] <= stat[r][1] and _static_window(stat[r], xl, k, img_w))
        if (int(steps[r, 0]) >> 17) & 1:
            seq += k["SRCOPS"]
            src_seq[r] = seq
    assert 0 <= xl <= max(img_w - k["TEX"], 0)
    for r in range(len(stat)):
        ctl, yfirst, ytop = int(steps[r, 0]), int(steps[r, 1]), int(steps[r, 2])
        mode, nrows, wait, src_next, mtop = ctl & 15, (ctl >> 4) & 15, (ctl >> 8) & 255, (ctl >> 16) & 1, (ctl >> 20) & 15
        ymin, ymax, xmin, xmax = (int(v) for v in stat[r])
        has = ymin <= ymax
        if src_next:
            seq += k["SRCOPS"]
            src_seq[r + k["AHEAD"]] = seq
        if mode != 1:
            assert nrows == 0                      # only window steps issue window rows
            assert (mode == 0) == (not has)
            ndir += mode == 2
            continue
        nwin += 1
        assert has
        yt, yb = ymin - 1, ymax + 2
        assert ytop == yt and mtop == yt % k["ROWS"]
        assert xmin - 1 >= xl and xmax + 2 <= xl + k["TEX"] - 1
        assert xl + k["TEX"] <= img_w
        assert yb - yt + 1 <= k["ROWS"]
        for y in range(yfirst, yfirst + nrows):
            slot = y % k["ROWS"]
            old = ring.get(slot)
            assert old is None or not (yt <= old[0] <= yb) or old[0] == y, "a row this step reads is overwritten"
            seq += k["ROWOPS"]
            ring[slot] = (y, seq)
            loaded += 1
        assert r in src_seq, "window step without prefetched source features"
        landed = seq - wait
        assert 0 <= wait <= k["MAXWAIT"]
        assert src_seq[r] <= landed
        for y in range(yt, yb + 1):
            got = ring.get(y % k["ROWS"])
            assert got is not None and got[0] == y, ("row not resident", r, y, got)
            assert got[1] <= landed, ("row not landed", r, y, got, landed)
    return nwin, ndir, loaded


def smooth_stats(n, y_start, x_start, scale_y=1.0, scale_x=1.0, shear=0.0, jitter=0.0, rng=None):
    st = np.zeros((n, 4), np.int32)
    for r in range(n):
        ys = [y_start + scale_y * r + shear * i + (rng.uniform(-jitter, jitter) if rng is not None else 0.0) for i in range(16)]
        xs = [x_start + scale_x * i + 0.02 * r for i in range(16)]
        st[r] = (int(np.floor(min(ys))), int(np.floor(max(ys))), int(np.floor(min(xs))), int(np.floor(max(xs))))
    return st


def test_unit_scale_translation_is_served_entirely_from_the_window(lib):
    k = consts(lib)
    assert (k["W"], k["H"]) == (16, 32)
    st = smooth_stats(32, 100.3, 200.7)
    xl, steps = plan(lib, st, 640)
    nwin, ndir, loaded = replay(st, xl, steps, k, 640)
    assert (nwin, ndir) == (32, 0)
    assert loaded == (st[-1, 1] + 2) - (st[0, 0] - 1) + 1           # every texel row exactly once
    # the ring is used to run ahead: in steady state the wait leaves at least one whole row in flight
    waits = [(int(c) >> 8) & 255 for c in steps[4:24, 0]]
    assert min(waits) >= k["ROWOPS"]


@pytest.mark.parametrize("sy,sx,shear", [(0.97, 0.98, 0.0), (1.04, 1.06, 0.01), (1.1, 1.1, -0.02), (0.9, 1.12, 0.03)])
def test_scaled_and_sheared_footprints(lib, sy, sx, shear):
    k = consts(lib)
    st = smooth_stats(32, 37.9, 11.2, sy, sx, shear)
    xl, steps = plan(lib, st, 320)
    nwin, ndir, loaded = replay(st, xl, steps, k, 320)
    assert nwin + ndir == 32 and nwin >= 30


def test_random_footprints_are_always_safe(lib):
    k = consts(lib)
    rng = np.random.RandomState(7)
    seen_direct = seen_skip = 0
    for trial in range(400):
        n = int(rng.choice([32, 32, 32, 17, 5, 1]))
        img_w = int(rng.choice([640, 320, 160, 80, 40, 21, 20]))
        st = smooth_stats(n, rng.uniform(2, 60), rng.uniform(1, max(img_w - 24, 2)), rng.uniform(0.85, 1.15),
                          rng.uniform(0.9, 1.12), rng.uniform(-0.04, 0.04), rng.uniform(0, 1.5), rng)
        for r in range(n):
            u = rng.uniform()
            if u < 0.08:
                st[r] = (1, 0, 1, 0)                               # no fast pixel in this row
            elif u < 0.14:
                st[r, 0:2] += int(rng.randint(-9, 10))             # a jump (depth discontinuity), up or down
            elif u < 0.18:
                st[r, 1] += int(rng.randint(3, 6))                 # a tall footprint that cannot fit
            elif u < 0.21:
                st[r, 3] += int(rng.randint(4, 9))                 # a wide footprint that cannot fit
        xl, steps = plan(lib, st, img_w)
        nwin, ndir, _ = replay(st, xl, steps, k, img_w)
        seen_direct += ndir
        seen_skip += sum(1 for r in range(n) if st[r, 0] > st[r, 1])
        if img_w < k["TEX"]:
            assert nwin == 0
    assert seen_direct > 50 and seen_skip > 50


def test_a_downward_jump_restarts_the_window_and_an_upward_jump_falls_back(lib):
    k = consts(lib)
    st = smooth_stats(32, 50.2, 100.1)
    st[16:, 0:2] += 12                        # rows 16.. read 12 texel rows further down: a fresh range, still window mode
    xl, steps = plan(lib, st, 640)
    nwin, ndir, _ = replay(st, xl, steps, k, 640)
    assert (nwin, ndir) == (32, 0)
    st = smooth_stats(32, 50.2, 100.1)
    st[16:20, 0:2] -= 10                      # rows 16..19 read rows the ring has already dropped: served directly
    xl, steps = plan(lib, st, 640)
    nwin, ndir, _ = replay(st, xl, steps, k, 640)
    assert ndir >= 1 and nwin + ndir == 32
