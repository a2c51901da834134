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
