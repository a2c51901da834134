#!/usr/bin/env python3
"""bench.py - sim-steps/sec of the batched rkFDUpdate hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload config4|config3|config2] [--batch B]

A "step" is one rkFDUpdate (4 RKG stage evaluations + the committing evaluation) of every
instance of the batch; state stays resident in HBM between steps.  One process per GPU; for
N>1 the driver launches this file under torch.distributed.run and every rank simulates its
own shard of instances (no data-path collective; one RCCL all-gather of the final states).
A step goes out as --split (default 3) launches of the step kernel over contiguous parts of the
batch on internal HIP streams: the instances are independent, so the thinly occupied tail of one
part's step overlaps the next step of another part (+25 % at 4096 instances per GPU).
`roofline.kernel_ms` is the average duration of ONE launch (HIP events on the stream it runs on),
`roofline.achieved` the algorithmic bytes of one launch over that duration; the launches of a step
overlap, `achieved_all_launches_of_a_step` relates the whole step's bytes to the step's duration.
The step kernel is compiled for the workload's world before the timing starts (rkfdBatchSpecialize, hipRTC, about 2 s;
--no-specialize keeps the library's generic kernel; same results either way).  --fuse N sends N steps per call.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {  # algorithmic HBM bytes per instance-step (SURVEY.md 8d / DESIGN.md)
    "config2": lambda m: 8 * 6 * m.ndof,
    "config3": lambda m: 8 * 6 * m.ndof + 8 * 80 + 24 * 24,
    "config4": lambda m: 8 * 6 * m.ndof + 8 * 80 + 24 * 24,
    "config4v": lambda m: 8 * 6 * m.ndof + 8 * 80 + 24 * 24,
    "config5": lambda m: 8 * 6 * m.ndof + 24 * 80 + 24 * 24,
    "config3_26": lambda m: 8 * 6 * m.ndof + 8 * 80 + 20 * 24,
    "config4_26": lambda m: 8 * 6 * m.ndof + 8 * 80 + 20 * 24,
}
HBM_PEAK_GBS = 8000.0


def measured_traffic(workload, batch):
    """HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in
    separate runs by tools/collect_traffic.sh and corrected as MI355X_MICROARCH.md prescribes);
    None when no measurement for this workload/batch is on file."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        with open(path) as fp:
            rec = json.load(fp).get(workload)
        if rec and rec.get("batch") == batch:
            return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    return None


def counted_flops(workload):
    """algorithmic flops per instance-step from the flop-counting build of the oracle
    (tools/count_flops.py -> profiles/r01_flops.json); None when not on file"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_flops.json")) as fp:
            return json.load(fp)[workload]["flops_per_instance_step"]
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(R, name, seconds=10.0):
    """the oracle (CPU restatement, 'port') timed on one host core on a bounded sample"""
    import numpy as np
    from oracle.pyoracle import Oracle
    sc = R.scenarios.CONFIGS[name](batch=4)
    nsteps = 0
    t0 = time.perf_counter()
    inst = 0
    while time.perf_counter() - t0 < seconds:
        o = Oracle(sc["world"].model)
        o.set_state(sc["dis"][inst % 4], sc["vel"][inst % 4])
        o.update_init()
        o.update_n(200)
        nsteps += 200
        inst += 1
    dt = time.perf_counter() - t0
    return dict(value=nsteps / dt, unit="sim-steps/sec", cores=1, kind="port",
                sample=f"{inst} instances x 200 steps of {name}, sequential on 1 core (oracle/rkfd_oracle.c, gcc -O3 -funroll-loops)")


def cpu_baseline_all_cores(R, name, seconds=8.0):
    """the same oracle on every host core this process may use: one OS thread per core over disjoint
    instances (the reference itself is single-threaded; this is the generous baseline).
    ctypes releases the GIL inside the C calls and the oracle keeps no global state."""
    import threading
    from oracle.pyoracle import Oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    sc = R.scenarios.CONFIGS[name](batch=4)
    done = [0] * cores
    t0 = time.perf_counter()

    def work(k):
        inst = k
        while time.perf_counter() - t0 < seconds:
            o = Oracle(sc["world"].model)
            o.set_state(sc["dis"][inst % 4], sc["vel"][inst % 4])
            o.update_init()
            o.update_n(200)
            done[k] += 200
            inst += cores

    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return dict(value=sum(done) / dt, unit="sim-steps/sec", cores=cores, kind="port",
                sample=f"{sum(done) // 200} instances x 200 steps of {name}, one thread per core on {cores} cores")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config4")
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse", type=int, default=1, help="steps per rkfdBatchUpdate call (must divide --steps).  1: a launch round per step.  "
                    "Worlds under the Vert plugin run the steps of a call fused in one kernel per part: the QP makes step times "
                    "vary widely between instances, and without a per-step barrier the variation averages out (config4v: 2.4 M -> 3.6 M steps/s)")
    ap.add_argument("--no-specialize", action="store_true", help="keep the generic step kernel instead of compiling it for the "
                    "workload's world (rkfdBatchSpecialize, hipRTC: same results, the world's dimensions as literals)")
    ap.add_argument("--split", type=int, default=3, help="launch each step as this many kernels over parts of the batch on internal HIP streams (1..8): the instances are independent, so the tail of one part's step overlaps the next step of another")
    args = ap.parse_args()

    import numpy as np
    if not args.no_specialize and os.path.exists("/opt/rocm/lib/libamd_comgr.so.3") and not os.environ.get("RKFD_BENCH_NO_COMGR_PRELOAD"):
        # rkfdBatchSpecialize compiles through hipRTC -> comgr.  PyTorch bundles an older comgr under the same soname, and
        # whichever is loaded first serves the whole process: with the bundled one the specialised kernel spills (437
        # VGPR spills, 3.9 M steps/s instead of 13.9 M).  Load the image's ROCm 7.2 compiler library before torch.
        import ctypes
        ctypes.CDLL("/opt/rocm/lib/libamd_comgr.so.3", mode=ctypes.RTLD_GLOBAL)
    import torch
    import rkfd_pkg
    R = rkfd_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    Bn = args.batch
    total = Bn * world                       # weak scaling: per-GPU work is fixed
    sc = R.scenarios.CONFIGS[args.workload](batch=total)
    lo, hi = R.sharding.shard_range(rank, world, total)
    sl = slice(lo, hi)
    b = R.Batch(sc["world"], Bn, device=local, max_rigid=sc["max_rigid"])
    specialized = False
    if not args.no_specialize and b.lds_bytes <= 64 * 1024:
        try:
            b.specialize()
            specialized = True
        except R.RkfdError as e:              # an optimisation, not a correctness path: say so and keep the generic kernel
            print("bench.py: %s" % e, file=sys.stderr)
    b.set_state(sc["dis"][sl], sc["vel"][sl])
    b.set_split(args.split)
    stream = torch.cuda.current_stream().cuda_stream
    b.update_init(stream)
    for _ in range(args.warmup):
        b.update(1, stream)
    assert b.status(stream) == 0
    b.time_launches(True)                     # HIP events around every kernel launch, on the stream it runs on

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    assert args.steps % args.fuse == 0, "--fuse must divide --steps"
    for _ in range(args.steps // args.fuse):
        b.update(args.fuse, stream)
    b.join(stream)                            # the current stream waits for the parts (no host sync)
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    st = b.status(stream)
    elapsed = t1 - t0
    step_ms = ev0.elapsed_time(ev1) / args.steps
    nlaunch, launch_ms = b.launch_timing()
    assert nlaunch > 0 and ( args.steps * args.split ) % nlaunch == 0, (nlaunch, args.steps, args.split)
    kernel_ms = launch_ms / nlaunch           # average duration of one launch of the step kernel
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the path's only collective: one all-gather of the final states (RCCL over xGMI), straight
        # from the device-resident state
        d_dis, d_vel, _ = b.dev_tensors()
        out = R.sharding.gather_final_states(dist, torch.cat([d_dis, d_vel], dim=1), total)
        torch.cuda.synchronize()
        assert out.shape[0] == total

    if rank == 0:
        m = sc["world"].model.contents
        alg = ALG_BYTES.get(args.workload, ALG_BYTES["config4"])(m)
        per_launch = Bn // args.split          # instances one launch processes
        achieved = alg * per_launch / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": "sim-steps/sec (node), 30-DoF humanoid + ground contact, batch=4096",
            "value": Bn * world * args.steps / elapsed, "unit": "sim-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": sc["name"], "instances_per_gpu": Bn, "ndof": m.ndof, "nlink": m.nlink,
                       "parallelism": f"instances sharded over {world} GPU(s), no data-path collective; {args.split} launches per step on internal streams",
                       "step_kernel": "compiled for this world (rkfdBatchSpecialize)" if specialized else "generic"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, per_launch),
                         "kernel": "rkfd_step_kernel_spec" if specialized else "rkfd_step_kernel", "kernel_ms": kernel_ms, "alg_bytes_per_instance_step": alg,
                         "instances_per_launch": per_launch, "launches_per_step": nlaunch / args.steps, "steps_per_launch": args.steps * args.split // nlaunch, "step_ms_hip_events": step_ms,
                         "lds_bytes_per_instance": b.lds_bytes, "resident_instances_per_cu": b.residency(),
                         "achieved_all_launches_of_a_step": alg * Bn / (step_ms * 1e-3) / 1e9},
            "device_status": st,
        }
        fl = counted_flops(args.workload)
        if fl is not None:
            # second axis SURVEY 8d asks for: fp64 vector throughput (dense peak 78.6 TFLOP/s, MI355X_MICROARCH.md)
            tf = fl * Bn / (step_ms * 1e-3) / 1e12      # whole step: the launches of one step overlap
            res["roofline_valu"] = {"bound": "valu_fp64", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                    "alg_flops_per_instance_step": fl}
        if world == 1 and not args.no_cpu_baseline:      # the CPU baseline is a single-GPU-run figure (rank 0, N = 1)
            res["cpu_baseline"] = cpu_baseline(R, args.workload)
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(R, args.workload)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
