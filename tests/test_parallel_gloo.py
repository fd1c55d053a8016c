"""N>1 path on CPU: world_size-2 gloo processes shard the windows and all-gather the result
records (the same code runs over RCCL on the GPUs)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, total, K, L, q):
    sys.path.insert(0, ROOT)
    from banet_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(total, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.float32)
    B = hi - lo
    R = torch.eye(3).repeat(B, 1, 1) * (idx + 1).reshape(B, 1, 1)
    T = idx.reshape(B, 1, 1).repeat(1, 3, 1) + 0.5
    W = idx.reshape(B, 1, 1).repeat(1, K, 1) * 2
    iters = [torch.full((B,), l + 1, dtype=torch.int32) + idx.to(torch.int32) for l in range(L)]
    rec = parallel.pack_results(R, T, W, iters)
    full = parallel.gather_results(rec, total)
    Rg, Tg, Wg, Ig = parallel.unpack_results(full, K, L)
    ok = full.shape == (total, 12 + K + L)
    exp = torch.arange(total, dtype=torch.float32)
    ok &= bool(torch.equal(Rg[:, 0, 0], exp + 1)) and bool(torch.equal(Tg[:, 2, 0], exp + 0.5))
    ok &= bool(torch.equal(Wg[:, K - 1, 0], exp * 2)) and bool(torch.equal(Ig[:, L - 1], (exp + L).to(torch.int32)))
    q.put((rank, ok, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    sys.path.insert(0, ROOT)
    from banet_amd import parallel
    for total in (1, 7, 8, 32, 255, 256):
        for world in (1, 2, 3, 8):
            r = [parallel.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_gather_results_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    total, K, L = 7, 4, 3                      # uneven shards: 4 + 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, K, L, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res)
    assert sorted((r[2], r[3]) for r in res) == [(0, 4), (4, 7)]


def test_bench_self_launches_its_ranks_without_a_torchrun_environment():
    """`python bench.py --gpus 2` outside torch.distributed.run (the driver's command shape) becomes the launcher of 2 ranks.
    There is no GPU here, so both ranks must stop at bench.py's own "needs a GPU" assertion -- which proves that two ranks
    were started with RANK / WORLD_SIZE set (the product path has no CPU fallback to fall through to)."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]
    assert "local_rank: 1" in r.stderr or "rank      : 1" in r.stderr or "rank: 1" in r.stderr, r.stderr[-2000:]
