"""The N > 1 path on CPU: two gloo ranks shard a frame by block-cyclic rows, each renders its rows with the
CPU oracle (standing in for its GPU), one gather to rank 0, and the assembled frame must equal the
single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, no_gather=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from micro_raytracer_amd import _abi, load_render, scenes
    from micro_raytracer_amd.dist import gather_frame, padded_rows, probe_gather, shard_row_index
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    if no_gather:          # a backend without gather: the one-time probe must switch every rank to all_gather
        def _refuse(*a, **k):
            raise NotImplementedError("gather is not implemented by this backend")
        dist.gather = _refuse
    use_all_gather = probe_gather()                 # once, agreed across ranks (ShardedSampler does this at construction)
    assert use_all_gather == bool(no_gather)
    render = load_render(scenes.cornell_box2(res=(40, 44), ssaa=1, sample=2))
    h = _abi.build_desc(render)
    o = oracle.Oracle(h, seed=4)
    nh, nw = o.nh, o.nw
    rows = shard_row_index(nh, rank, world)
    # render this rank's 8-row blocks (sample indices 0..1 for every block: a fresh count per block)
    full = np.zeros((nh, nw, 3), np.float32)
    for b0 in range(0, nh, 8):
        if (b0 // 8) % world == rank:
            o.reset()
            o.execute(2, threads=2, rows=(b0, min(nh, b0 + 8)))
            a, _ = o.accum()
            full[b0:b0 + 8] = a[b0:b0 + 8]
    local = torch.zeros((padded_rows(nh, world), nw, 3), dtype=torch.float32)
    local[: len(rows)] = torch.from_numpy(full[rows])
    # receive buffers allocated once and reused (ShardedSampler does this): a first exchange of other contents, then the real one
    parts = [torch.empty_like(local) for _ in range(world)] if (rank == 0 or use_all_gather) else None
    gather_frame(torch.full_like(local, 7.0), nh, nw, dst=0, use_all_gather=use_all_gather, parts=parts)
    frame = gather_frame(local, nh, nw, dst=0, use_all_gather=use_all_gather, parts=parts)
    if rank == 0:
        assert all(p.data_ptr() == q.data_ptr() for p, q in zip(parts, parts))      # still the caller's buffers
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_sharding_reassembles_the_frame(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    from micro_raytracer_amd import _abi, load_render, scenes

    out = str(tmp_path / "frame.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = np.load(out)
    h = _abi.build_desc(load_render(scenes.cornell_box2(res=(40, 44), ssaa=1, sample=2)))
    o = oracle_mod.Oracle(h, seed=4)
    o.execute(2)
    assert np.array_equal(got, o.accum()[0])


@pytest.mark.parametrize("world,no_gather", [(8, False), (3, True)])
def test_more_ranks_uneven_shards_and_the_all_gather_fallback(tmp_path, oracle_mod, world, no_gather):
    """8 ranks on a 44-row frame: six 8-row blocks, so two ranks own nothing but padding; and a backend that refuses
    gather (3 ranks): rank 0 must still assemble the single-process frame bit for bit."""
    import torch.multiprocessing as mp
    from micro_raytracer_amd import _abi, load_render, scenes

    out = str(tmp_path / "frame.npy")
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, no_gather)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = np.load(out)
    h = _abi.build_desc(load_render(scenes.cornell_box2(res=(40, 44), ssaa=1, sample=2)))
    o = oracle_mod.Oracle(h, seed=4)
    o.execute(2)
    assert np.array_equal(got, o.accum()[0])


def test_shard_row_index_partitions_every_row():
    from micro_raytracer_amd.dist import padded_rows, shard_row_index
    for nh, world, sr in ((1080, 8, 8), (2160, 8, 8), (44, 3, 8), (7, 4, 8), (100, 2, 16)):
        seen = np.zeros(nh, int)
        for r in range(world):
            rows = shard_row_index(nh, r, world, sr)
            assert len(rows) <= padded_rows(nh, world, sr)
            seen[rows] += 1
        assert (seen == 1).all()


def test_bench_gpus_n_starts_its_own_ranks_and_relays_failure():
    """`python bench.py --gpus 2` without WORLD_SIZE spawns two fresh ranks through torch.distributed.run before
    touching a GPU; without a device each rank stops with the no-CPU-path message and the exit code is non-zero."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU box by tests/test_gpu_dist.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") >= 2, out.stderr[-2000:]
