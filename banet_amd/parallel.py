"""One process per GPU; independent windows are sharded across ranks.

Every window is an independent LM problem, so the data path has NO collective: each rank
solves its contiguous shard of the batch.  The only exchange is one all-gather per solve of
the small per-window result record [R(9) | T(3) | W(K) | iters(L)] -- the "cross-window pose
reduction" a sequence-level consumer needs (cf. legacy/seq_example.py:170-173).  On MI355X
that is RCCL over xGMI (`backend="nccl"`); the CPU tests run the same code over gloo.

Shard size and bits: by default the library picks kernels, the SYRK's arithmetic form and the number of partial rows from the whole
launch (batch included), so a window solved in a shard of 8 and in a shard of 32 agrees to rounding (pose 1e-5, depth coefficients
1e-4 after a multi-level solve; measured 2.4e-5) but not bit for bit.  Equal shards select equal kernels (two ranks == one process,
tested); callers that need bit-identical results across UNEQUAL shards build their DenseBA with batch_invariant=True
(banet_level_t.policy = BANET_POLICY_BATCH_INVARIANT; tests/test_gpu_round5.py).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """contiguous shard [lo, hi) of `total` windows for `rank` (sizes differ by at most 1)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(R, T, Wc, iters_per_level):
    """-> [B_local, 12 pairs + K + L] float32 record (R [B,(pairs,)3,3], T [B,(pairs,)3,1])"""
    B = R.shape[0]
    parts = [R.reshape(B, -1), T.reshape(B, -1)]
    if Wc is not None:
        parts.append(Wc.reshape(B, -1))
    parts.append(torch.stack([c.to(torch.float32) for c in iters_per_level], dim=1))
    return torch.cat(parts, dim=1).contiguous()


def unpack_results(rec, K, L, pairs=1):
    B = rec.shape[0]
    shape = (B,) if pairs == 1 else (B, pairs)
    R = rec[:, 0:9 * pairs].reshape(*shape, 3, 3)
    T = rec[:, 9 * pairs:12 * pairs].reshape(*shape, 3, 1)
    o = 12 * pairs
    Wc = rec[:, o:o + K].reshape(B, K, 1) if K > 0 else None
    iters = rec[:, o + K:o + K + L].to(torch.int32)
    return R, T, Wc, iters


def gather_results(local_rec, total, group=None):
    """all-gather the per-window records of all ranks into one [total, F] tensor, ordered by
    window index (shards are contiguous).  One fused buffer, one collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_rec
    world = dist.get_world_size(group)
    F = local_rec.shape[1]
    sizes = [shard_range(total, r, world) for r in range(world)]
    maxb = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(maxb, F, dtype=local_rec.dtype, device=local_rec.device)
    pad[:local_rec.shape[0]] = local_rec
    out = torch.empty(world * maxb, F, dtype=local_rec.dtype, device=local_rec.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * maxb:r * maxb + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)
