#!/usr/bin/env python3
"""bench.py - sim-steps/sec of the batched rkFDUpdate hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload config4|config3|config2|config5|config4v|config5v|config1_volume|config4_volume|...] [--batch B]

A "step" is one rkFDUpdate (4 RKG stage evaluations + the committing evaluation) of every instance of the
batch; state stays resident in HBM between steps.

The workload is MPC-style rollouts (BASELINE.json north_star): every instance starts from its own standing
state (flat soles seated in the floor, all 8 sole vertices in rigid contact), is simulated for --horizon
steps (default 25), is put back to its start state by a device-side copy (rkfdBatchRestore, inside the timed
region) and simulated again.  Why a horizon: under the reference's algorithm (PGS with 10 sweeps and no warm
start + the K = 1000 deadbeat penetration compensation) a stiff body resting on more than three coplanar
vertices does not stay there - a 4-step chatter builds up and after 30-40 steps the robot rocks on 3-4
vertices (DESIGN.md "Scenario note", measured with the CPU oracle).  The horizon keeps the timed steps on
the contact problem the configuration names (Nc ~ 8 => a 24x24 MLCP); the JSON line reports what was really
solved: config.mean_rigid_contacts is read back from the device for exactly the timed steps, and the
algorithmic bytes are computed from it.  --horizon 0 runs one continuous trajectory instead (the rocking
regime, ~3.5 contacts; reported in DESIGN.md beside the headline).

--steps K is the block that is timed; the block is repeated until the timed region lasts --min-seconds
(default 6 s) so that the figure does not depend on a 4 ms measurement and a monitor sampling every 5 s sees the GPU busy; ms_per_step and value are averages
over all timed steps (config.timed_steps), the rollout boundaries continue across blocks.

One process per GPU; for N>1 the driver launches this file under torch.distributed.run and every rank
simulates its own shard of instances (no data-path collective; one RCCL all-gather of the final states).
Started by hand with --gpus N > 1 and no WORLD_SIZE, it spawns torch.distributed.run itself (as a child
process, before anything touches the GPU) and relays rank 0's line.
A step goes out as --split (default 3) launches of the step kernel over contiguous parts of the batch on
internal HIP streams: the instances are independent, so the thinly occupied tail of one part's step overlaps
the next step of another (+25 % at 4096 instances per GPU).
`roofline.kernel_ms` is the average duration of ONE launch (HIP events on the stream it runs on),
`roofline.achieved` the algorithmic bytes of one launch over that duration; the launches of a step overlap,
`achieved_all_launches_of_a_step` relates the whole step's bytes to the step's duration.
A launch carries --steps-per-launch (default 5) steps of its part of the batch (rkfdBatchSetStepsPerLaunch): the state stays in
LDS between them; roofline.steps_per_launch says what the timed launches really carried.
The step kernel is the one compiled for the workload's world (rkfdBatchSpecialize: ahead of time by `make spec`, else hipRTC
in about 2 s; --no-specialize keeps the library's generic kernel; same results either way), with one or two instances per
wavefront, whichever is faster for that world: --ipw 0 (default) measures both before the timed region
(rkfdBatchTuneInstancesPerWave; the two give the same bits) and reports the two times.
Counter-derived fields (roofline.traffic, valu_issue) come from the newest profiles/rNN_* files and only when those carry the
hash of this tree's device sources (else null + the reason).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def alg_bytes(m, mean_contacts, R):
    """algorithmic HBM bytes per instance-step: read dis, vel, u and write dis, vel, acc (8*6n);
    per contact vertex in contact r/w {type, _ref} and write f (80); per friction-pivot DoF r/w {type, prev_trq} (24).
    The contact count is the MEASURED mean over the timed steps."""
    jt = m.arr("jtype", m.nlink); mt = m.arr("mtype", m.nlink)
    n_fric = int(((jt == R.binding.JOINT_REVOL) | (jt == R.binding.JOINT_PRISM))[mt == 2].sum())
    return 8 * 6 * m.ndof + 80.0 * mean_contacts + 24 * n_fric, n_fric


def newest_profile(suffix):
    """(tag, record) of the newest profiles/rNN_<suffix> (tags sort by round), or (None, None)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    for f in reversed(fs):
        try:
            with open(f) as fp:
                return os.path.basename(f)[:3], json.load(fp)
        except (OSError, ValueError):
            continue
    return None, None


def stamped(rec, what):
    """(ok, reason): a counter-derived figure is reported only from a profile taken on THIS tree's device sources (the
    profile carries the sha256 of the sources it was collected on: rkfd_pkg.device_source_hash)"""
    import rkfd_pkg
    have = rec.get("device_source_sha256") if isinstance(rec, dict) else None
    cur = rkfd_pkg.device_source_hash()
    if have is None:
        return False, f"{what}: the profile on file carries no source stamp"
    if have != cur:
        return False, f"{what}: the profile on file was taken on other device sources (stamp {have}, this tree {cur}): re-collect with tools/collect_all.sh"
    return True, None


def measured_traffic(workload, per_launch):
    """(HBM bytes per launch, note) from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate runs by
    tools/collect_traffic.sh and corrected as MI355X_MICROARCH.md prescribes); (None, why) when no measurement of this
    workload / launch size taken on this tree's device sources is on file."""
    tag, allrec = newest_profile("hbm_traffic.json")
    if tag is None:
        return None, "no profiles/rNN_hbm_traffic.json on file"
    rec = allrec.get(workload)
    if not rec or rec.get("batch") != per_launch:
        return None, f"profiles/{tag}_hbm_traffic.json holds no measurement of {workload} at {per_launch} instances per launch"
    ok, why = stamped(allrec, f"profiles/{tag}_hbm_traffic.json")
    if not ok:
        return None, why
    return rec["hbm_bytes_per_launch"], f"profiles/{tag}_hbm_traffic.json (source stamp {allrec['device_source_sha256']})"


def counted_flops(workload, horizon):
    """algorithmic flops per instance-step from the flop-counting build of the oracle, counted over the same
    rollout window (tools/count_flops.py -> profiles/rNN_flops.json); None when not on file.  (A property of the
    reference's formulation as the oracle restates it, not of the device code: no source stamp needed.)"""
    tag, rec = newest_profile("flops.json")
    try:
        rec = rec[workload]
        return rec["flops_per_instance_step"] if rec.get("horizon") == horizon else None
    except (KeyError, TypeError):
        return None


def valu_issue(workload):
    """(VALU instructions per instance-step, note) from the rocprofv3 SQ pass on file (profiles/rNN_<workload>_rocprof_summary.json),
    only when it was taken on this tree's device sources"""
    tag, rec = newest_profile(f"{workload}_rocprof_summary.json")
    if tag is None:
        return None, "no rocprofv3 SQ summary of this workload on file"
    ok, why = stamped(rec, f"profiles/{tag}_{workload}_rocprof_summary.json")
    if not ok:
        return None, why
    try:
        return rec["SQ_INSTS_VALU"] / rec["SQ_WAVES"], f"profiles/{tag}_{workload}_rocprof_summary.json (source stamp {rec['device_source_sha256']})"
    except (KeyError, ZeroDivisionError):
        return None, "the summary on file lacks SQ_INSTS_VALU / SQ_WAVES"


def cpu_baseline(R, name, horizon, seconds=10.0):
    """the oracle (CPU restatement, 'port') timed on ONE host core on a bounded sample of the SAME workload: rollouts of
    `horizon` steps from the same standing states.  Threads and clock live in C (oracle/rkfd_oracle_mt.c)."""
    from oracle import pyoracle
    sc = R.scenarios.CONFIGS[name](batch=4)
    nsteps, dt = pyoracle.rollouts_mt(sc["world"].model, sc["dis"], sc["vel"], horizon, seconds, nthreads=1)
    what = f"rollouts of {horizon} steps from the standing states" if horizon > 0 else "trajectories of up to 1000 steps"
    return dict(value=nsteps / dt, unit="sim-steps/sec", cores=1, kind="port",
                sample=f"{nsteps} steps of {name} ({what}, 4 distinct instances), sequential on 1 core "
                       f"(oracle/rkfd_oracle.c, gcc -O3 -funroll-loops)")


def cpu_share():
    """(threads to use, description): the CPUs this process may really use - the affinity mask, capped by the cgroup's CPU quota
    (a GPU box shows all 256 hardware threads of its host in the affinity mask but grants the job a quota of about 16 CPUs:
    256 threads then run 9 x one core, throttled; 16 threads 14 x)"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fp:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fp.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                             # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fpd:
                q, per = float(fq.read()), float(fpd.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return n, f"{aff} hardware threads in the affinity mask, cgroup CPU quota " + ("none" if quota is None else f"{quota:.1f} CPUs")


def cpu_baseline_all_cores(R, name, horizon, seconds=8.0):
    """the same oracle on every host core this process may use: one OS thread per core over disjoint instances (the reference
    itself is single-threaded; this is the generous baseline).  The threads are pthreads INSIDE the oracle library
    (round 2 used Python threads around ctypes calls and measured the interpreter lock: 14.5 x one core on 256 hardware
    threads); the oracle keeps no global state."""
    from oracle import pyoracle
    cores, share = cpu_share()
    sc = R.scenarios.CONFIGS[name](batch=max(4, min(cores, 64)))
    nsteps, dt = pyoracle.rollouts_mt(sc["world"].model, sc["dis"], sc["vel"], horizon, seconds, nthreads=cores)
    return dict(value=nsteps / dt, unit="sim-steps/sec", cores=cores, kind="port", cpu_share=share,
                sample=f"{nsteps} steps of {name}, same rollouts, one pthread per CPU this job may use ({cores}: {share}); threads and clock inside the oracle library")


def spawn_ranks(args):
    """python bench.py --gpus N (N > 1) by hand: start torch.distributed.run as a CHILD process - this process has not
    touched the GPU - and relay its output; exit with its code"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, RKFD_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config4")
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=25, help="steps per rollout before the instances are put back to their start states "
                    "(device-side copy, timed); 0: one continuous trajectory")
    ap.add_argument("--min-seconds", type=float, default=6.0, help="repeat the --steps block until the timed region lasts this long (6 s: longer "
                    "than the 5 s sampling period of a utilisation monitor watching the run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-continuous", action="store_true", help="skip the second, shorter measurement of ONE continuous trajectory per instance "
                    "(no rollout boundaries; the regime --horizon 0 times), reported as `continuous_trajectory` in the same JSON line")
    ap.add_argument("--fuse", type=int, default=0, help="steps per rkfdBatchUpdate call (0: up to the end of the rollout / block).  Under split "
                    "launches a call of n steps goes out as n rounds of one-step launches, except for worlds under the Vert plugin, whose steps "
                    "run fused in one kernel per part: the QP makes step times vary widely between instances, and without a per-step barrier "
                    "the variation averages out")
    ap.add_argument("--no-specialize", action="store_true", help="keep the generic step kernel instead of compiling it for the "
                    "workload's world (rkfdBatchSpecialize: same results, the world's dimensions as literals)")
    ap.add_argument("--ipw", type=int, default=0, choices=(0, 1, 2), help="instances per wavefront of the world-specific step kernel: 1 = one instance has the "
                    "64 lanes; 2 = two instances share a wavefront, 32 lanes each (worlds of at most 32 links / joint coordinates; same results "
                    "to the last bit); 0 (default) = measure both before the timed region and keep the faster (rkfdBatchTuneInstancesPerWave)")
    ap.add_argument("--steps-per-launch", type=int, default=5, help="under split launches a call of n steps goes out as rounds of launches of at most "
                    "this many steps (rkfdBatchSetStepsPerLaunch; results do not depend on it)")
    ap.add_argument("--split", type=int, default=3, help="launch each step as this many kernels over parts of the batch on internal HIP streams (1..8)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import rkfd_pkg
    R = rkfd_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: measuring {world} GPU(s)", file=sys.stderr)
    dist = None
    if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    Bn = args.batch
    total = Bn * world                       # weak scaling: per-GPU work is fixed
    lo, hi = R.sharding.shard_range(rank, world, total)
    assert hi - lo == Bn
    sc = R.scenarios.CONFIGS[args.workload](batch=Bn, first=lo)      # this rank's shard only (the seeded stream is index-addressable)
    b = R.Batch(sc["world"], Bn, device=local, max_rigid=sc["max_rigid"])
    specialized = False; aot = False
    if not args.no_specialize and b.lds_bytes <= 64 * 1024:
        try:
            if args.ipw == 2:
                try:
                    b.set_instances_per_wave(2)
                except R.RkfdError as e:      # the world does not fit 32 lanes per instance: one per wavefront it is
                    print("bench.py: %s" % e, file=sys.stderr)
            b.specialize()
            specialized = True
            aot = bool(R.lib().rkfdSpecializeLastFromStore())
        except R.RkfdError as e:              # an optimisation, not a correctness path: say so and keep the generic kernel
            print("bench.py: %s" % e, file=sys.stderr)
    b.set_state(sc["dis"], sc["vel"])
    b.set_split(args.split)
    b.set_steps_per_launch(args.steps_per_launch)
    stream = torch.cuda.current_stream().cuda_stream
    b.update_init(stream)
    assert b.status(stream) == 0
    H = args.horizon
    ipw_tuning = None
    if specialized and args.ipw == 0:
        # untimed: which mapping is faster for THIS world (both give the same bits); the state comes back as it was
        try:
            chosen, ms = b.tune_instances_per_wave(min(H, 50) if H > 0 else 50)
            ipw_tuning = {"ms_1": round(ms[0], 4), "ms_2": round(ms[1], 4) if ms[1] >= 0 else None, "chosen": chosen}
        except R.RkfdError as e:              # an optimisation, not a correctness path: the kernel with one instance per wavefront stays
            print("bench.py: %s" % e, file=sys.stderr)
            ipw_tuning = {"error": str(e)}
        aot = aot and bool(R.lib().rkfdSpecializeLastFromStore())
        assert b.status(stream) == 0
    if H > 0:
        b.snapshot()                          # the rollouts' start: standing state + rkFDUpdateInit's committing evaluation

    pos = [0]                                 # steps since the last restore

    def run_steps(n):
        """n steps of the workload: rollouts of H steps, the state put back in between (device-side, stream-ordered)"""
        while n > 0:
            if H > 0 and pos[0] == H:
                b.restore(stream); pos[0] = 0
            k = n if H <= 0 else min(n, H - pos[0])
            if args.fuse > 0:
                k = min(k, args.fuse)
            b.update(k, stream)
            pos[0] += k; n -= k

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    assert b.status(stream) == 0
    # how many blocks make --min-seconds: one untimed block, the slowest rank decides
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    b.join(stream)
    barrier()
    t_block = time.perf_counter() - t0
    reps = max(1, int(np.ceil(args.min_seconds / max(t_block, 1e-6))))
    if dist is not None:
        t = torch.tensor([reps], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps = int(t.item())
    timed_steps = reps * args.steps
    # the timed steps must see the same mix of rollout phases whatever --steps is: start them at a rollout boundary
    if H > 0:
        run_steps(H - pos[0])
    b.contact_stats(reset=True)
    b.time_launches(True)                     # HIP events around every kernel launch, on the stream it runs on

    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(reps):
        run_steps(args.steps)
    b.join(stream)                            # the current stream waits for the parts (no host sync)
    ev1.record()
    barrier()
    t1 = time.perf_counter()
    st = b.status(stream)
    elapsed = t1 - t0
    step_ms = ev0.elapsed_time(ev1) / timed_steps
    nlaunch, launch_ms = b.launch_timing()
    assert nlaunch > 0
    kernel_ms = launch_ms / nlaunch           # average duration of one launch of the step kernel
    mean_rg, mean_el, counted = b.contact_stats()
    if sc["world"].model.contents.ncand > 0:       # (worlds without contact candidates keep no contact statistics)
        assert counted == Bn * timed_steps, (counted, Bn, timed_steps)
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

    # the same batch on ONE continuous trajectory per instance (what --horizon 0 times): the stiff body leaves the 8-vertex stance
    # after 30-40 steps and rocks on 3-4 vertices (DESIGN.md "Scenario note" - derived with this repository's oracle, not checked
    # against a build of the reference: [UNVERIFIED-DEP]), a lighter contact problem.  Reported beside the headline so that the
    # two regimes can be compared in one line.
    continuous = None
    if world == 1 and H > 0 and not args.no_continuous:
        ncand = sc["world"].model.contents.ncand
        b.set_state(sc["dis"], sc["vel"])
        if ncand > 0:
            b.set_contact(np.zeros((Bn, ncand), dtype=np.int32), np.zeros((Bn, ncand), dtype=np.int32), np.zeros((Bn, ncand, 3)))
        b.set_pivot(np.zeros((Bn, sc["world"].model.contents.nlink), dtype=np.int32), np.zeros((Bn, sc["world"].model.contents.nlink)))
        b.update_init(stream)
        cw, cs = 100, 0
        for _ in range(cw // 25):
            b.update(25, stream)
        b.join(stream); torch.cuda.synchronize()
        b.contact_stats(reset=True)
        c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
        tc0 = time.perf_counter(); c0.record()
        while time.perf_counter() - tc0 < min(1.5, max(args.min_seconds, 0.2)) or cs == 0:
            b.update(25, stream); cs += 25
            if cs % 200 == 0:
                b.join(stream); torch.cuda.synchronize()
        b.join(stream); c1.record(); torch.cuda.synchronize()
        cms = c0.elapsed_time(c1) / cs
        crg, cel, _cn = b.contact_stats()
        continuous = {"value": Bn / (cms * 1e-3), "unit": "sim-steps/sec", "ms_per_step": cms, "timed_steps": cs, "steps_before": cw,
                      "mean_rigid_contacts": crg, "mean_elastic_contacts": cel, "device_status": b.status(stream),
                      "note": "one trajectory per instance from the standing states, no rollout boundaries (--horizon 0)"}

    if rank == 0:
        m = sc["world"].model.contents
        has_contacts = m.ncand > 0
        if args.workload in ("config4", "config5", "config4v", "config4_26"):
            assert mean_rg > 0, "the timed region solved no rigid contact: the workload is not the one it names"
        alg, n_fric = alg_bytes(m, mean_rg + mean_el, R)
        steps_per_launch = 1 if (args.split > 1 and not (args.workload == "config4v")) else None
        per_launch = Bn // args.split          # instances one launch processes
        launches_per_step = nlaunch / timed_steps
        spl = args.split / launches_per_step   # steps one launch carries
        achieved = alg * per_launch * spl / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic(args.workload, per_launch)
        res = {
            "metric": "sim-steps/sec (node), 30-DoF humanoid + ground contact, batch=4096",
            "value": Bn * world * timed_steps / elapsed, "unit": "sim-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / timed_steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": sc["name"] + (f", MPC-style rollouts of {H} steps from the standing states" if H > 0 else ", one continuous trajectory"),
                       "instances_per_gpu": Bn, "instances_per_wavefront": b.instances_per_wave(), "instances_per_wavefront_tuning": ipw_tuning, "ndof": m.ndof, "nlink": m.nlink,
                       "rollout_horizon": H, "timed_steps": timed_steps, "blocks": reps,
                       "mean_rigid_contacts": mean_rg if has_contacts else 0.0, "mean_elastic_contacts": mean_el if has_contacts else 0.0,
                       "parallelism": f"instances sharded over {world} GPU(s), no data-path collective; {args.split} parts of the batch on internal streams, up to {args.steps_per_launch} steps per launch",
                       "step_kernel": ("compiled for this world ahead of time (make spec), loaded from roki-fd_amd/spec" if aot else "compiled for this world at run time (hipRTC)") if specialized else "generic"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "rkfd_step_kernel_spec" if specialized else "rkfd_step_kernel", "kernel_ms": kernel_ms,
                         "alg_bytes_per_instance_step": alg, "friction_pivot_dofs": n_fric,
                         "instances_per_launch": per_launch, "launches_per_step": launches_per_step, "steps_per_launch": spl, "step_ms_hip_events": step_ms,
                         "lds_bytes_per_instance": b.lds_bytes, "resident_instances_per_cu": b.residency(),
                         "achieved_all_launches_of_a_step": alg * Bn / (step_ms * 1e-3) / 1e9},
            "device_status": st,
        }
        if continuous is not None:
            res["continuous_trajectory"] = continuous
        del steps_per_launch
        fl = counted_flops(args.workload, H)
        if fl is not None and args.workload != "config4v":
            # second axis SURVEY 8d asks for: fp64 vector throughput (dense peak 78.6 TFLOP/s, MI355X_MICROARCH.md).  The flops are
            # those of the REFERENCE's formulation (counted on the oracle over the same rollout window: link-local ABA, column-probed
            # MLCP), not the device's instruction count - the device's innovations form needs fewer.  Not given for the Vert plugin:
            # the oracle's generic pseudo-inverse KKT solves cost ~10x what the device's structured solve does.
            tf = fl * Bn / (step_ms * 1e-3) / 1e12      # whole step: the launches of one step overlap
            res["roofline_valu"] = {"bound": "valu_fp64", "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
                                    "alg_flops_per_instance_step": fl, "flops_counted_on": "the reference's formulation (oracle), same rollout window"}
        vi, vi_note = valu_issue(args.workload)
        if vi is None:
            res["valu_issue"] = None; res["valu_issue_source"] = vi_note
        else:
            # what bounds the kernel: VALU instruction issue.  Instructions per instance-step from the rocprofv3
            # SQ_INSTS_VALU pass on file; one wave-wide VALU instruction occupies its SIMD for 4 cycles (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.02
            # quad-cycles measured); 256 CUs x 4 SIMDs.  The clock under this kernel is ~1.8-2.0 GHz, the nominal 2.4 GHz is used.
            # (SQ_INSTS_VALU / SQ_WAVES is per WAVEFRONT: with two instances per wavefront it serves both)
            ipw_now = max(1, b.instances_per_wave())
            # ... and per LAUNCH: a launch carries spl steps
            vi = vi / spl
            ips = vi * (Bn / ipw_now) / (step_ms * 1e-3)
            res["valu_issue"] = {"valu_insts_per_instance_step": vi / ipw_now, "valu_insts_per_wavefront_step": vi, "simd_busy_fraction_at_2.4GHz": ips * 4.0 / (256 * 4 * 2.4e9), "source": vi_note}
        if world == 1 and not args.no_cpu_baseline:      # the CPU baseline is a single-GPU-run figure (rank 0, N = 1)
            res["cpu_baseline"] = cpu_baseline(R, args.workload, H)
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(R, args.workload, H)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
