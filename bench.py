#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched rollout hot path on N MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): flamingo_light_v1 x 4096 envs per GPU,
flat terrain, fp32, precision "medium" (4 x 5 ms substeps per 50 Hz control step), GUI-default domain randomisation
(init/mass noise 0.05, action delay 0.05, frictions .8/.02/.01, friction loss .1, sensor noise "low"), per-env PD gains
x U(0.9, 1.1), synthetic actions a[n,j,t] = clip(0.25 sin(2 pi 0.5 Hz 0.02 t + phi[n,j])) with phi from
Philox(seed 1234, key (n, j)), auto-reset on.  One "step" = one control step of every env = one kernel launch.
Inputs (actions, commands) are resident in HBM before the timed region.  Weak scaling: 4096 envs per GPU; shards are
independent (no data-path collective); one all-reduce of reporter statistics closes the timed region.

Extra objects on the JSON line: "roofline" (algorithmic bytes of the step kernel / its mean launch duration from HIP
events on the launch stream, against the 8 TB/s HBM peak) and "cpu_baseline" (the fp64 CPU oracle on the CPU twins of the fleet's
own envs: one thread, then every host core; bounded sample; rank 0 at N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# The fleet is stepped as S range launches on S engine-owned HIP streams (see --streams).  The HIP runtime maps streams onto 4 hardware
# queues by default, and streams that share a queue run their kernels one after the other: 4 range streams + torch's own stream then
# step at HALF the speed of 2 (measured 7.7 M vs 13.5 M env-steps/s).  Eight queues let the four ranges overlap (13.9 M).  Must be set
# before the first HIP call of the process; the ranks of --gpus N inherit it.  (cosim_amd/__init__.py does the same for any caller.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ROBOT = "flamingo_light_v1"
HBM_PEAK_GBS = 8000.0
# name -> (robot, terrain, height_map, envs per GPU).  The default ("light_flat") is BASELINE.json configs[1], the
# configuration the metric is quoted on; the others are the remaining BASELINE configs, selectable for DESIGN.md numbers.
TIMING_STRIDE = 5
WORKLOADS = {
    "light_flat": ("flamingo_light_v1", "flat", False, 4096),
    "light_rocky": ("flamingo_light_v1", "rocky_hard", False, 4096),
    "w4_rocky": ("w4_p_v2", "rocky_hard", True, 4096),            # configs[2]
    "p_v3_flat": ("flamingo_p_v3", "flat", False, 4096),          # configs[3] per-GPU shard (flat; its terrain is not named)
    "humanoid_flat": ("humanoid_p_v0", "flat", False, 1024),
    "humanoid_stairs": ("humanoid_p_v0", "stairs_up_hard", False, 1024),   # configs[4] per-GPU shard: position-command mode, targets U([-3,3]^2)
}


def b_alg(nq, nv, nu, state_dim):
    """B_alg = 4 [2 (nq + nv) + 3 nu + 2 state_dim + 2 nv] bytes per env-step."""
    return 4 * (2 * (nq + nv) + 3 * nu + 2 * state_dim + 2 * nv)


def synthetic_actions(n_envs, env_id0, steps, nu, device):
    """a[t, n, j] for t in [0, steps); phases from Philox(seed 1234, key (n, j)) via the engine's host RNG twin."""
    import torch
    from cosim_amd import rng as crng
    gids = np.arange(env_id0, env_id0 + n_envs, dtype=np.uint64)[:, None]
    phi = 2 * np.pi * crng.uniform(1234, gids, 0, 5, np.arange(nu)[None, :]).astype(np.float32)
    t = torch.arange(steps, dtype=torch.float32, device=device)[:, None, None]
    a = 0.25 * torch.sin(2 * np.pi * 0.5 * 0.02 * t + torch.tensor(phi, device=device)[None])
    return a.clamp_(-1.0, 1.0).contiguous()


def workload_config(workload, n):
    """The configuration dict of a bench workload (shared by the GPU leg and the CPU baseline)."""
    from cosim_amd.config import make_config
    robot, terrain, hmap, _ = WORKLOADS[workload]
    poscmd = workload == "humanoid_stairs"
    cfg = make_config(robot, terrain=terrain, num_envs=n, seed=1234, height_map=hmap, position_command=poscmd)
    if poscmd:
        cfg["observation"]["command_dim"] = 2                        # envs/wrappers.py:357
    return cfg


def action_table_host(gids, steps, nu):
    """synthetic_actions() on the host (fp64): the same Philox phases, for the CPU twins of fleet envs `gids`."""
    from cosim_amd import rng as crng
    g = np.asarray(gids, dtype=np.uint64)[:, None]
    phi = 2 * np.pi * crng.uniform(1234, g, 0, 5, np.arange(nu)[None, :]).astype(np.float32)
    t = np.arange(steps, dtype=np.float32)[:, None, None]
    return np.clip(0.25 * np.sin((2 * np.pi * 0.5 * 0.02 * t + phi[None]).astype(np.float32)), -1.0, 1.0).astype(np.float64)


def cpu_baseline(workload="light_flat", budget_s=10.0, chunk=50):
    """The fp64 CPU oracle (a restatement of the reference's mj_step path: kind "port") on the GPU box's host cores, on the SAME
    inputs as the GPU leg: the CPU twins of the fleet's first envs (oracle/fleet.py) -- same randomised masses and PD gains, same
    init-noise and action-delay draws (host twin of the device's Philox streams), same action table, same termination rule and
    auto-reset.  Two legs, each bounded in time: one thread, then one env per thread on every host core (ctypes releases the GIL
    inside the C rollout).  Each env advances `chunk` control steps per call, starting from step 0 like the GPU leg's warm-up."""
    from concurrent.futures import ThreadPoolExecutor
    from cosim_amd.compile import compile_model
    from oracle.fleet import FleetEnvTwin
    robot = WORKLOADS[workload][0]
    cfg = workload_config(workload, WORKLOADS[workload][3])
    cm = compile_model(cfg)
    nu = cm.blob.nu
    ncores = os.cpu_count() or 1
    horizon = 1100                                                # the GPU leg's default run: 100 warm-up + 1000 timed steps

    class Runner:
        """One CPU twin at a time: env `gid` through the GPU leg's action table (steps 0 .. horizon), then the next env.  Only the
        rollout calls are timed (building a twin recompiles the mass-dependent model constants: set-up, not stepping)."""

        def __init__(self, ids):
            self.ids, self.busy, self.steps, self.last = ids, 0.0, 0, -1
            self._next()

        def _next(self):
            self.last = next(self.ids)
            self.tw = FleetEnvTwin(cfg, cm, 1234, self.last, gain_noise=0.1)
            self.acts = action_table_host([self.last], horizon, nu)[:, 0, :]
            self.tw.reset()
            self.done = 0

        def run(self, budget):
            while self.busy < budget:
                t0 = time.perf_counter()
                n = self.tw.rollout(self.acts[self.done:self.done + chunk])
                self.busy += time.perf_counter() - t0
                self.done += n
                self.steps += n
                if self.done >= horizon or n == 0:
                    self._next()
            return self

    # leg 1: one thread, envs 0, 1, 2, ... one after the other
    r1 = Runner(iter(range(10 ** 9))).run(0.4 * budget_s)
    n1, t1, e1 = r1.steps, r1.busy, r1.last
    # leg 2: every host core, one worker PROCESS per core (forked: they share the compiled model; threads would serialise on the GIL
    # in the per-chunk Python of the delay filter: 256 threads measured 146 k env-steps/s in 15.7 s of wall for 6 s of stepping),
    # one env per worker at a time, env ids handed out in blocks.  Rate = control steps / mean stepping time per worker.
    import multiprocessing as mp

    def worker(k):
        r = Runner(iter(range(1000 * (k + 1), 1000 * (k + 2)))).run(0.6 * budget_s)
        return r.steps, r.busy
    global _CPU_WORKER
    _CPU_WORKER = worker
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(ncores) as pool:
        res = pool.map(_cpu_worker_entry, range(ncores), chunksize=1)
    wall = time.perf_counter() - t0
    nall = sum(r[0] for r in res)
    tall = sum(r[1] for r in res) / len(res)
    ref_leg = None
    try:
        import mujoco  # the reference's own engine, if this host happens to have it (nothing is installed for it)
        from cosim_amd.model import get_field
        from tools.crosscheck_mujoco import build_xml
        mj_model = mujoco.MjModel.from_xml_string(build_xml(cfg))
        mj_data = mujoco.MjData(mj_model)
        mj_data.qpos[:] = np.array(get_field(cm.blob, "init_qpos")[:cm.blob.nq])
        mujoco.mj_forward(mj_model, mj_data)
        n_ref, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 5.0:                      # zero torque, frame_skip substeps per control step
            mujoco.mj_step(mj_model, mj_data, nstep=int(cm.blob.frame_skip))
            n_ref += 1
        ref_leg = {"value": n_ref / (time.perf_counter() - t0), "unit": "env-steps/s", "cores": 1, "kind": "reference",
                   "sample": f"mujoco {mujoco.__version__} mj_step x {int(cm.blob.frame_skip)} on the same MJCF (hull vertices inline), 1 env, zero torque, 5 s"}
        ref = "reference MuJoCo timed beside it (cpu_baseline.reference)"
    except Exception as e:  # noqa: BLE001
        ref = f"reference MuJoCo CPU path unavailable on this host ({type(e).__name__})"
    return {"value": n1 / t1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"workload {workload}: CPU twins of the GPU fleet's envs 0..{e1} ({robot}; same randomised masses / PD gains, same Philox "
                      f"init-noise and action-delay draws, same sinusoid action table over the GPU leg's steps 0..{horizon}, same termination + auto-reset), fp64 "
                      f"oracle, 1 thread of {ncores} host cores, {n1} control steps in {t1:.1f} s of stepping; the wrapper layer (observation, sensor "
                      f"noise) is not in this leg; {ref}",
            "all_cores": {"value": nall / tall, "unit": "env-steps/s", "cores": ncores,
                          "sample": f"one env at a time on each of {ncores} worker processes (one per host core), {nall} control steps in "
                                    f"{tall:.1f} s of stepping per worker ({wall:.1f} s of wall with building the twins)"},
            **({"reference": ref_leg} if ref_leg else {})}


_CPU_WORKER = None


def _cpu_worker_entry(k):
    return _CPU_WORKER(k)


def pmc_traffic(workload, split_substeps=0):
    """HBM bytes per launch of the step kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    passes over this same command: tools/pmc_run.sh -> profiles/rNN_pmc_summary_<workload>.txt).  Counters cannot be read from
    inside the timed run, so the figure is the one measured for the committed kernel; null for workloads without a PMC pass.
    The summary has one section per kernel, the one with the most GPU time first: that section is "the dominant kernel".
    split_substeps > 0 (the two-kernel pipeline): what one timed unit covers is a whole control step on one range stream, i.e.
    split_substeps launches of the narrowphase kernel and of the solver kernel: their sections are added up accordingly."""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    files = glob.glob(os.path.join(root, f"r*_pmc_summary_{workload}.txt"))
    if workload == "light_flat":
        files += glob.glob(os.path.join(root, "r*_pmc_summary_v*.txt"))           # round-1 naming
    files.sort(key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return {"traffic": None}
    sections, cur = [], None
    for line in open(files[-1]):
        if line.startswith("per launch of"):
            cur = {"kernel": line.split()[3].rstrip(",")}
            sections.append(cur)
            continue
        t = line.split()
        if cur is not None and len(t) >= 3 and t[0] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            cur[t[0]] = float(t[2]) * (1024.0 if t[0] != "SQ_INSTS_VALU" else 1.0)    # rocprofv3 reports the sizes in KiB
    if not sections:                                                  # round-1 / round-2 files: one unnamed section
        cur = {"kernel": "cosim::env_kernel"}
        for line in open(files[-1]):
            t = line.split()
            if len(t) >= 3 and t[0] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
                cur[t[0]] = float(t[2]) * (1024.0 if t[0] != "SQ_INSTS_VALU" else 1.0)
        sections = [cur]
    use = [x for x in sections if "FETCH_SIZE" in x and "WRITE_SIZE" in x]
    if split_substeps > 0:
        use = [x for x in use if "narrow" in x["kernel"] or "step_kernel" in x["kernel"]]
        mult = float(split_substeps)
    else:
        use, mult = use[:1], 1.0
    if not use:
        return {"traffic": None}
    fetch, write = mult * sum(x["FETCH_SIZE"] for x in use), mult * sum(x["WRITE_SIZE"] for x in use)
    valu = mult * sum(x.get("SQ_INSTS_VALU", 0.0) for x in use)
    extra = {"valu_insts_per_launch": valu} if valu else {}
    # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section): doubled, as the guide
    # prescribes, before it is compared with a byte count (an upper bound here: the guide calibrates the factor on 16-B-per-lane
    # streaming reads, this kernel reads a dword per lane); WRITE_SIZE is exact
    return {"traffic": 2.0 * fetch + write,
            "traffic_note": f"bytes per timed unit ({' + '.join(x['kernel'] for x in use)}{f' x {split_substeps} substeps' if split_substeps else ''}), "
                            f"{os.path.basename(files[-1])}: 2 x FETCH_SIZE {fetch:.3g} B (gfx950 correction of "
                            f"the guide; face value would be the lower bound) + WRITE_SIZE {write:.3g} B (state, observation stack "
                            "and info write-back, plus register-spill scratch in the kernels that spill)", **extra}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch_command(argv, n_ranks, port):
    """The torch.distributed.run command that re-runs this script as `n_ranks` ranks of one node (one process per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks as child processes and pass their
    output and exit code on.  Runs before anything in this process touches the GPU (no HIP call, no torch.cuda query): the
    parent only waits."""
    import subprocess
    cmd = relaunch_command(argv, args.gpus, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def launch_selftest(args):
    """--selftest-launch: the rendezvous / barrier / max-over-ranks / rank-0-prints skeleton of the bench with no GPU work
    (CPU test of the launcher: tests/test_distributed.py)."""
    import torch
    import torch.distributed as dist
    from cosim_amd.distributed import MetricsAccumulator, init_from_env
    rank, world = init_from_env("gloo")
    acc = MetricsAccumulator(["x"])
    acc.update(torch.full((4, 1), float(rank + 1), dtype=torch.float64))
    out = acc.reduce()
    t = torch.tensor([float(rank)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "max_rank": float(t.item()), "count": out["x"]["count"],
                          "mean": out["x"]["mean"]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--workload", default="light_flat", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=4,
                    help="env.step() issues each control step of the per-GPU fleet as S launches over contiguous env ranges on S "
                         "engine-owned HIP streams (BatchedEnv(ranges=S, deferred_join=True): a range's next control step fills the tail "
                         "of the others' launches); 1 = one launch on the caller's stream")
    ap.add_argument("--caller-streams", action="store_true",
                    help="A/B: the round-2 arrangement, the CALLER builds the S streams and calls env.step_range per range")
    ap.add_argument("--no-rollout", action="store_true", help="skip the extra cosim_rollout measurement after the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--report-every", type=int, default=16, help="reporter statistics are sampled every this many timed steps")
    ap.add_argument("--selftest-launch", action="store_true", help="launcher / collective skeleton only, no GPU work (CPU test)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))     # the children are the ranks; this process never touches the GPU
    if args.selftest_launch:
        return launch_selftest(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_line = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        # before anything in this process touches the GPU: the all-cores leg forks worker processes
        cpu_line = cpu_baseline(args.workload)

    import torch
    import torch.distributed as dist
    from cosim_amd.batched_env import BatchedEnv
    from cosim_amd.config import make_config
    from cosim_amd.distributed import init_from_env

    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): the two must agree")
    rank, world = init_from_env(args.backend)
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    robot, terrain, hmap, n_default = WORKLOADS[args.workload]
    n = args.envs_per_gpu or n_default
    env_id0 = rank * n

    S = max(1, min(args.streams, n))
    # contiguous ranges of n // S envs, the first n % S of them one env longer
    starts = [i * (n // S) + min(i, n % S) for i in range(S + 1)]
    ranges = [(starts[i], starts[i + 1] - starts[i]) for i in range(S)]
    ns = n / S                                                       # mean envs per launch (roofline.achieved is per launch)
    poscmd = args.workload == "humanoid_stairs"
    cfg = workload_config(args.workload, n)
    # ONE fleet (one engine handle, one set of [N, ...] buffers), stepped with the plain env.step(action) of the drop-in API.
    # S = 1: one launch per control step.  S > 1: the engine issues the step as S launches over contiguous env ranges, each range
    # on an engine-owned HIP stream, and (deferred join) does not make the caller's stream wait for them, so a range's next control
    # step fills the tail of the others' launches (a launch ends with its slowest env; envs never interact).  The actions are a
    # table resident in HBM: nothing between two steps depends on the whole fleet's last state.
    eng_ranges = 1 if args.caller_streams else S
    env = BatchedEnv(cfg, num_envs=n, device=local, seed=1234, auto_reset=True, env_id0=env_id0, gain_noise=0.1,
                     ranges=eng_ranges, deferred_join=eng_ranges > 1)
    envs = [env]
    main = torch.cuda.current_stream(env.device)
    if args.caller_streams:
        streams = [main] if S == 1 else [torch.cuda.Stream(device=env.device) for _ in range(S)]
    else:
        ranges = list(env.range_list)
        streams = [main] if S == 1 else list(env.range_streams)
    nu = env.action_dim
    total_steps = args.warmup + args.steps
    actions = synthetic_actions(n, env_id0, total_steps, nu, env.device)
    # reporter statistics (core/reporter.py's per-step scalars over the fleet): one launch of the fused reducer per sampled step,
    # sufficient statistics stay on the device until the one all-reduce that closes the timed region
    from cosim_amd.reporter import FleetReporter
    reporter = FleetReporter(env)
    if poscmd:   # per-env targets U([-3, 3]^2), keyed by global env id
        from cosim_amd import rng as crng
        gids = np.arange(env_id0, env_id0 + n, dtype=np.uint64)[:, None]
        env.receive_user_command((6.0 * crng.uniform(1234, gids, 0, 6, np.arange(2)[None, :]) - 3.0).astype(np.float32))
    else:
        env.receive_user_command(np.array([0.5, 0.0, 0.0, 0.0], dtype=np.float32)[:max(env.command_dim, 1)])
    env.reset()
    torch.cuda.synchronize()

    def fleet_step(t, report):
        a = actions[t]
        if args.caller_streams:
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    env.step_range(ranges[i][0], ranges[i][1], a)
                    if report:
                        reporter.write_info_range(ranges[i][0], ranges[i][1])
            return
        env.step(a)
        if report:   # reporter statistics of this step: each range's info rows are reduced on that range's own stream
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    reporter.write_info_range(ranges[i][0], ranges[i][1])
                if S > 1:
                    env.range_mark(i)

    if args.caller_streams:
        for st_ in streams:
            st_.wait_stream(main)
    for t in range(args.warmup):
        fleet_step(t, t == 0)             # t == 0: warm-up of the reducer (first use loads code objects: ~100 ms)
    torch.cuda.synchronize()
    reporter.acc.reduce()                 # warm-up of the torch kernels behind reduce(), then start from zero
    reporter.acc.buf.zero_()
    torch.cuda.synchronize()
    # kernel duration for the roofline: HIP event pairs around every TIMING_STRIDE-th launch of the timed region, on the launch's own
    # stream (an event pair around EVERY launch costs the pipeline 4 % of a 20-step run: tools/gpu_shortrun.py; 5 is coprime with the
    # four ranges, so every range is sampled in turn)
    for e_ in envs:
        e_.engine.set_param("timing_stride", np.array([float(TIMING_STRIDE)], dtype=np.float32))
    env.engine.set_timing(True)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for t in range(args.warmup, total_steps):
        fleet_step(t, (t - args.warmup) % args.report_every == 0)   # sampled: first timed step, then every k-th
    t_issue = time.perf_counter()
    env.join()                                          # the deferred join: the caller's stream now waits for every range
    torch.cuda.synchronize()
    t_gpu = time.perf_counter()
    sync()
    dt = time.perf_counter() - t0
    # the one collective, after the K steps: RCCL all-reduce of the reporter's (count, sum, sum^2) accumulated during them (the per-step
    # accumulation kernels run inside the timed region, on the range streams)
    fleet = reporter.acc.reduce()
    torch.cuda.synchronize()
    if os.environ.get("COSIM_BENCH_TRACE"):             # diagnostic: where the timed region's wall time went
        print(f"[bench trace] issue loop {1e3 * (t_issue - t0):.3f} ms, + join/sync {1e3 * (t_gpu - t_issue):.3f} ms, + barrier/sync "
              f"{1e3 * (t0 + dt - t_gpu):.3f} ms", file=sys.stderr, flush=True)
    kt = [e.engine.kernel_time() for e in envs]
    timed_launches = sum(k[1] for k in kt)                       # the sampled ones (an event pair each)
    kernel_ms = sum(k[0] * k[1] for k in kt) / max(1, timed_launches)
    launches = S * args.steps                                     # range launches of the timed region on this rank
    for e in envs:
        e.engine.set_timing(False)
    finite = all(bool(torch.isfinite(e.state).all().item()) for e in envs)
    sts = [e.solver_stats() for e in envs]
    st = {k: sum(x[k] for x in sts) for k in sts[0]}
    nsub = max(1, (st["step_count"] - n) * 4)    # the reset launch also bumps the step counter once per env

    tmax = torch.tensor([dt], dtype=torch.float64, device=env.device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # A second, separately labelled number (never `value`): the same fleet continuing for another args.steps control steps through
    # cosim_rollout -- the whole table in one launch per range, for callers whose actions do not depend on the last state.
    rollout = None
    if env.engine.query("rollout") == 1 and not args.caller_streams and not args.no_rollout:
        acts_r = synthetic_actions(n, env_id0, total_steps + args.steps, nu, env.device)[total_steps:].contiguous()
        env.rollout(acts_r[:min(2, args.steps)])            # code-object load of the rollout kernels
        sync()
        tr0 = time.perf_counter()
        env.rollout(acts_r)
        sync()
        tr = torch.tensor([time.perf_counter() - tr0], dtype=torch.float64, device=env.device)
        if world > 1:
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        str_ = env.solver_stats()
        rollout = {"env_steps_per_s": world * n * args.steps / float(tr.item()), "steps_per_launch": args.steps,
                   "api": "env.rollout(action_table) -> cosim_rollout", "dropped_contacts": str_["dropped_contacts"],
                   "finite": bool(torch.isfinite(env.state).all().item()),
                   "note": "same fleet, the next `steps` rows of the same action table, outputs for every step written; not the headline"}

    if rank == 0:
        value = world * n * args.steps / dt
        balg = b_alg(env.nq, env.nv, nu, env.state_dim)
        achieved = balg * ns / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0   # one launch processes ns envs
        split_waves = env.engine.query("split")
        pmc = pmc_traffic(args.workload, int(env.cm.blob.frame_skip) if split_waves else 0)
        valu = pmc.pop("valu_insts_per_launch", None)
        if valu:
            # VALU issue rate against the chip's issue slots: wave-level VALU instructions per launch (PMC pass of the committed kernel)
            # x this run's launches / this run's timed seconds, over 1024 SIMDs x 2.4 GHz / 2 (a wave64 fp32 VALU instruction holds its
            # SIMD's issue port for two cycles at the least): the bound that actually binds a latency / issue-bound solver
            pmc["valu_issue_frac"] = valu * launches / dt / (1024 * 2.4e9 / 2.0)
            pmc["valu_issue_note"] = (f"SQ_INSTS_VALU {valu:.4g} per launch (committed PMC pass) x {launches} launches / {dt:.4f} s timed, over "
                                      "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU issue")
        line = {
            "metric": f"env-steps/sec (whole node), {robot} xN envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{robot} x {n} envs per GPU, {terrain} terrain, precision medium (4 x 5 ms substeps), "
                                   "GUI-default domain randomisation + sensor noise low, sinusoid actions, auto-reset",
                       "envs_per_gpu": n, "global_envs": world * n, "substeps_per_s": value * 4, "parallelism": f"shard{world}",
                       "streams_per_gpu": S, "envs_per_launch": ns,
                       "finite": finite, "fleet_action_diff_RMSE": fleet["action_diff_RMSE"]["mean"],
                       "fleet_samples": fleet["action_diff_RMSE"]["count"], "fleet_abs_torque_0": fleet["abs_torque_0"]["mean"],
                       "solver_per_substep": {"rows": st["rows"] / nsub, "newton_iters": st["newton_iters"] / nsub,
                                              "ls_evals": st["ls_evals"] / nsub, "factorisations": st["factorisations"] / nsub},
                       "nan_resets": st["nan_resets"], "dropped_contacts": st["dropped_contacts"], "dropped_limit_rows": st["dropped_limit_rows"],
                       "max_contacts_per_env": max(x["max_contacts"] for x in sts), "contact_slots": env.engine.query("contact_slots"),
                       # control steps redone by the large-capacity kernel (contacts beyond the fleet kernel's slots); 0 slots: the
                       # workload's kernel has no such sibling and counts what it leaves out in dropped_contacts
                       "fixup_steps": st["fixup_steps"], "fixup_contact_slots": env.engine.query("fixup_contact_slots"),
                       "step_api": "caller streams + env.step_range" if args.caller_streams else "env.step",
                       "rollout": rollout},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, **pmc,
                         # per launch as the contract defines it (S launches overlap on the chip); the whole fleet per control step:
                         "achieved_fleet": balg * n / (dt / args.steps) / 1e9,
                         "frac_fleet": balg * n / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "kernel": (f"cosim::env_kernel<{env.nv},{env.cm.blob.nbody},...>" if not split_waves else
                                    f"cosim::env_narrow_kernel + env_step_kernel<{env.nv},{env.cm.blob.nbody},...>: one pair of launches per substep, "
                                    f"{split_waves} narrowphase waves per env; kernel_ms is the whole control step's launches on one range stream"),
                         "kernel_ms": kernel_ms, "launches": launches, "timed_launches": timed_launches,
                         "launches_note": f"kernel_ms is the mean over every {TIMING_STRIDE}th launch of the timed region (a HIP event pair each)",
                         "algorithmic_bytes_per_env_step": balg,
                         "note": "latency/VALU-bound small-state solver; HBM sees only the compulsory state traffic (SURVEY §8d)"},
        }
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for e in envs:
        e.close()


if __name__ == "__main__":
    main()
