"""N>1 path on CPU: two gloo ranks each simulate their shard (device code under the lane
emulator), then all-gather the final states; rank 0 checks the result against a single-process
run of all instances.  Mirrors what bench.py does with RCCL on GPUs."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import rkfd_pkg
    from emu import EmuBatch
    R = rkfd_pkg.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = R.scenarios.config2(batch=total)
    lo, hi = R.sharding.shard_range(rank, world, total)
    eb = EmuBatch(sc["world"], hi - lo, max_rigid=0)
    eb.set_state(sc["dis"][lo:hi], sc["vel"][lo:hi])
    eb.update_init(); eb.update(1)
    d, v, _ = eb.get_state()
    out = R.sharding.gather_final_states(dist, torch.from_numpy(np.concatenate([d, v], axis=1)), total)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges(R):
    for total in (1, 5, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = R.sharding.shard_range(r, world, total)
                cover += list(range(lo, hi))
            assert cover == list(range(total))


def test_two_rank_gather_matches_single_process(R):
    import torch.multiprocessing as mp
    from emu import EmuBatch
    total, world = 5, 2          # uneven shards: 3 + 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    sc = R.scenarios.config2(batch=total)
    eb = EmuBatch(sc["world"], total, max_rigid=0)
    eb.set_state(sc["dis"], sc["vel"]); eb.update_init(); eb.update(1)
    d, v, _ = eb.get_state()
    assert np.array_equal(got, np.concatenate([d, v], axis=1))


def _worker8(rank, world, port, total, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import rkfd_pkg
    R = rkfd_pkg.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = R.sharding.shard_range(rank, world, total)
    # what bench.py does per rank: build ONLY this rank's shard of the seeded scenario (index-addressable stream)
    sc = R.scenarios.config5(batch=hi - lo, first=lo)
    final = np.concatenate([sc["dis"], sc["vel"] + rank], axis=1)          # config 5's state width: 2 x 54
    out = R.sharding.gather_final_states(dist, torch.from_numpy(final), total)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_uneven_shards_of_config5(R):
    """the 8-GPU layout of BASELINE config 5 rehearsed on CPU: 8 gloo ranks, 19 instances (shards of 3,3,3,2,2,2,2,2), every
    rank generating only its own shard of the scenario; the gathered [total, 108] block equals the scenario built whole"""
    import torch.multiprocessing as mp
    total, world = 19, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    sc = R.scenarios.config5(batch=total)
    assert got.shape == (total, 108)
    assert np.array_equal(got[:, :54], sc["dis"])
    ranks = np.concatenate([[r] * (R.sharding.shard_range(r, world, total)[1] - R.sharding.shard_range(r, world, total)[0]) for r in range(world)])
    assert np.array_equal(got[:, 54:], sc["vel"] + ranks[:, None])
