"""Dense multi-level window solver: every pixel of every pyramid level is a BA point
(SURVEY.md 8(d) "dense mode"; the reference samples N=1024-4096 points instead).

Per level the layer's inputs are the CNN's outputs in their native layouts -- source and
target feature maps NHWC `[B,H_l,W_l,C]`, depth `[B,H_l,W_l]`, basis `[B,H_l,W_l,K]` -- and
the whole coarse->fine LM schedule is enqueued on the current stream without a host sync.
Iteration bodies: `bundle` = bundlenet.py:193-278 (pose + depth basis), `bundle_camera` =
bundlenet.py:122-191, `legacy_lm` / `legacy_fixed` = legacy/ba.py:226-345 / :148-214.
"""
import os

import torch

from . import ops


class DenseLevel:
    def __init__(self, scale, src, tgt, depth, basis=None):
        """tgt [B,H,W,C] (2-frame) or [B,pairs,H,W,C] (multi-frame window: `pairs` target frames that
        share the key frame's src / depth / basis -- SURVEY.md 8(d))"""
        self.scale = float(scale)
        self.src, self.tgt, self.depth, self.basis = src, tgt, depth, basis
        self.pairs = tgt.shape[1] if tgt.dim() == 5 else 1
        self.B, self.H, self.W, self.C = tgt.shape[0], tgt.shape[-3], tgt.shape[-2], tgt.shape[-1]


class DenseBA:
    def __init__(self, intr, levels, lambda_weights, variant="bundle", l2_base=1000.0, batch_invariant=False):
        """intr [B,4] full-resolution (fx,fy,ox,oy); levels: list of DenseLevel coarse->fine;
        lambda_weights: list (one per level) of 5 (filters, biases) pairs.
        batch_invariant: banet_level_t.policy = BANET_POLICY_BATCH_INVARIANT on every level -- kernels, SYRK form and summation
        split chosen from the level alone, so a window's bits do not depend on the batch / shard it is solved in (callers that
        shard one batch unevenly over GPUs and compare results bit for bit; slower below ~8 windows per launch).  Default: the
        launch decides (fastest; results agree to rounding across batchings)."""
        self.variant = variant
        self.levels, self.lambda_weights = list(levels), list(lambda_weights)
        self.l2_base = float(l2_base) if variant == "bundle" else 1.0
        self.intr = intr.contiguous().float()
        dev = self.intr.device
        legacy = variant.startswith("legacy")
        self.problems, self.mlps = [], []
        for lv, lw in zip(levels, lambda_weights):
            B, H, W, C = lv.B, lv.H, lv.W, lv.C
            basis = lv.basis.reshape(B, H * W, -1) if (variant == "bundle") else None
            self.problems.append(ops.LevelProblem(variant, lv.src, lv.tgt, lv.depth.reshape(B, H * W), H, W, C,
                                                  basis=basis, intr=self.intr, scale=lv.scale, dense=True,
                                                  tgt_has_grad=False, normalize_rays=not legacy, pairs=lv.pairs))
            self.mlps.append(None if variant == "legacy_fixed" else ops.MlpWeights(lw, dev))
        if batch_invariant:
            for p in self.problems:
                p.c.policy = ops.capi.POLICY_BATCH_INVARIANT
        self.K = self.problems[0].K
        self.pairs = self.problems[0].pairs
        nb = max(ops.lm_level_workspace_bytes(p) for p in self.problems)
        if nb == 0:
            raise ops.capi.BanetError("DenseBA: unsupported level shape")
        self.ws = ops.capi.workspace(nb, dev)
        self.B = self.problems[0].B
        # The coarsest level (<= 1200 pixels) of a large batch as two half batches on two HIP streams (host-side orchestration only;
        # opt-in: BANET_SPLIT_COARSE=1).  Measured at 32 windows (profiles/r05_run13_*): 40x30 1.37 -> 1.08 ms per 10 iterations, but
        # 80x60 +7 % and 160x120 +13 % when split too -- net zero over a solve, +0.5 % with the coarsest level alone: not the default.
        # Under the default throughput policy the two B / 2 launches choose kernels and summation splits from THEIR batch, so enabling
        # it changes results in the low bits (equal to rounding, like any other batching: DESIGN.md section 6); BATCH_INVARIANT does not.
        self.split_coarse = os.environ.get("BANET_SPLIT_COARSE", "0") == "1"
        self._parts, self._side = {}, None
        self._variant_args = dict(variant=variant, legacy=legacy)

    def _level_parts(self, li, st):
        """two half-batch views of level li (problems over slices of the level tensors, workspaces) + views of the state"""
        lv = self.levels[li]
        # the halves alias the level's tensors: rebuilt whenever one of them (or the level's problem) has been replaced since
        key = (id(self.problems[li]),) + tuple(t.data_ptr() for t in (lv.src, lv.tgt, lv.depth) + ((lv.basis,) if lv.basis is not None else ()))
        if li not in self._parts or self._parts[li][0] != key:
            B = self.B
            cuts = [(0, B // 2), (B // 2, B)]
            parts = []
            for lo, hi in cuts:
                H, W, C = lv.H, lv.W, lv.C
                basis = lv.basis[lo:hi].reshape(hi - lo, H * W, -1) if self.variant == "bundle" else None
                prob = ops.LevelProblem(self.variant, lv.src[lo:hi], lv.tgt[lo:hi], lv.depth[lo:hi].reshape(hi - lo, H * W), H, W, C,
                                        basis=basis, intr=self.intr[lo:hi], scale=lv.scale, dense=True, tgt_has_grad=False,
                                        normalize_rays=not self._variant_args["legacy"], pairs=lv.pairs)
                prob.c.flags, prob.c.policy = self.problems[li].c.flags, self.problems[li].c.policy
                parts.append((lo, hi, prob, ops.capi.workspace(ops.lm_level_workspace_bytes(prob), self.intr.device)))
            self._parts[li] = (key, parts)
        out = []
        for lo, hi, prob, ws in self._parts[li][1]:
            prob.c.flags, prob.c.policy = self.problems[li].c.flags, self.problems[li].c.policy    # (follow later changes of the level's)
            sub = ops.capi.State()
            sub.R, sub.T = st.R[lo:hi].data_ptr(), st.T[lo:hi].data_ptr()
            sub.Wc = st.Wc[lo:hi].data_ptr() if st.Wc is not None else None
            sub.iters, sub.ratio = st.iters[lo:hi].data_ptr(), st.ratio[lo:hi].data_ptr()
            sub.lambda_out, sub.delta = st.lambda_out[lo:hi].data_ptr(), st.delta[lo:hi].data_ptr()
            holder = type("SubState", (), {})()
            holder.c = sub
            out.append((prob, ws, holder))
        return out

    def new_state(self, R=None, T=None, Wc=None):
        dev = self.intr.device
        B, K = self.B, self.K
        R = torch.eye(3, device=dev).repeat(B * self.pairs, 1, 1) if R is None else R
        T = torch.zeros(B * self.pairs, 3, 1, device=dev) if T is None else T
        if K > 0 and Wc is None:
            Wc = torch.zeros(B, K, 1, device=dev)
        R = R.reshape(B, -1, 3, 3) if self.pairs > 1 else R
        T = T.reshape(B, -1, 3, 1) if self.pairs > 1 else T
        return ops.LmState(R, T, Wc if K > 0 else None, P=6 * self.pairs + K, pairs=self.pairs)

    def solve(self, iters_per_level, state=None, early_termination=False, params=None, snapshots=None, level_events=None):
        """Enqueue the full schedule; returns the state (R,T,Wc updated in place) and the list
        of per-level iteration-count tensors.  params: ops.lm_params(...) (legacy/ba.py:5-9) or None for the
        reference's defaults.  snapshots: a list that receives, per level, a dict of clones of (R, T, W, delta, lam)
        after that level (device tensors, no host sync).  level_events: a list that receives one (start, end) pair of
        torch.cuda.Event per level, recorded on the current stream (per-level times without a host sync)."""
        st = state if state is not None else self.new_state()
        counts = []
        for prob, mlp, its in zip(self.problems, self.mlps, iters_per_level):
            if level_events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            li = len(counts)
            if self.split_coarse and self.B >= 16 and prob.N <= 1200 and li < len(self.levels):
                # two half batches, the second on a side stream; joined before the next level (the state tensors are shared)
                dev = self.intr.device
                cur = torch.cuda.current_stream(dev)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                (p0, w0, s0), (p1, w1, s1) = self._level_parts(li, st)
                self._side.wait_stream(cur)
                ops.lm_level(p0, mlp, self.l2_base, its, early_termination, s0, ws=w0, params=params)
                with torch.cuda.stream(self._side):
                    ops.lm_level(p1, mlp, self.l2_base, its, early_termination, s1, ws=w1, params=params)
                cur.wait_stream(self._side)
            else:
                ops.lm_level(prob, mlp, self.l2_base, its, early_termination, st, ws=self.ws, params=params)
            if level_events is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                level_events.append((e0, e1))
            counts.append(st.iters.clone())
            if snapshots is not None:
                snapshots.append(dict(R=st.R.clone(), T=st.T.clone(), W=None if st.Wc is None else st.Wc.clone(),
                                      delta=st.delta.clone(), lam=st.lambda_out.clone()))
        return st, counts

    def solve_differentiable(self, iters_per_level, R=None, T=None, Wc=None):
        """The same fixed-count schedule attached to the autograd graph: gradients flow to the levels' src / tgt / depth /
        basis tensors, to the initial (R, T, Wc) and to the lambda weights through the fused backward kernels
        (banet_amd/dense_train.py, csrc/adjoint.hip; the reference differentiates bundlenet.py:376-397 with tf.gradients +
        EquationConstructionGrad).  `bundle` (K <= 256) and the pose-only `bundle_camera` variant; two-frame and multi-frame
        windows."""
        from . import dense_train
        return dense_train.solve_differentiable(self, self.levels, self.lambda_weights, iters_per_level, R, T, Wc)

    def step_from(self, level_index, R, T, Wc=None):
        """ONE iteration of level `level_index` from the given state -> (state after it; .delta / .lambda_out hold the
        solved update and lambda).  Used by the parity checks to compare single updates from identical states."""
        st = self.new_state(R, T, Wc)
        ops.lm_level(self.problems[level_index], self.mlps[level_index], self.l2_base, 1, False, st, ws=self.ws)
        return st

    def algorithmic_bytes_per_iteration(self, level_index):
        """4*N_l*(C*F + K + 1) per window-iteration (F = 1 + pairs frames)."""
        p = self.problems[level_index]
        return 4 * p.N * (p.C * (1 + p.pairs) + p.K + 1)
