#!/usr/bin/env python3
"""Headline benchmark: env-steps/s (forward + adjoint) of batched Laikago rollouts on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  * one process per GPU (for N > 1 launched by torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE from env)
  * a "step" = one forward rollout + one adjoint rollout over the rank's batch: bs envs x T sim steps
  * workload (BASELINE.json metric / SURVEY.md section 8(d) config C4): Laikago, mixed mi-trot / mi-spin
    mocap targets, T = 100 sim steps (4 frames, frame2step = 0,33,66,99), dt = 5e-4, fp32.
    --scaling strong (default; the BASELINE metric "batch=4096 at 1/2/4/8 MI355X" = SURVEY C4): 4096 envs GLOBAL, rank r rolls
    out the contiguous slice shard_envs(4096, N, r) (512 per GPU at N = 8).
    --scaling weak: 4096 envs PER GPU, global batch N x 4096.
    Either way every rank builds ITS slice of one global batch (per-env seeded: the slices concatenate to the
    same global batch for any N), envs are independent, and there is no collective on the data path
  * inputs are resident in HBM before the timed region; after the W warm-up steps the bench keeps stepping, untimed, for 0.25 s
    (clock ramp of an idle box; the line's "warmup_extra_steps" says how many steps that was), then times R = --repeats (10)
    BLOCKS of exactly K steps, each bracketed by barrier + synchronize, MAX over ranks per block.  value / ms_per_step are the
    MEDIAN block ("median of >= 10 repeats"); "blocks" carries min / max / all of them
  * rank 0 prints ONE JSON line (value = whole-job env-steps/s); for N > 1 the line also carries "other_scaling": the same
    measurement in the other mode (so one scaling run holds the strong AND the weak row)
  * "boundary": the same workload timed one level up, through the reference's autograd boundary -- ForwardWarp.apply(...) then
    .backward() on tensors that require grad, workspace / output / gradient allocation included (N = 1 only)

Extra objects on the line:
  roofline     dominant kernel (the adjoint rollout): algorithmic HBM bytes per launch / its average launch
               duration measured live with HIP events on the launch stream (pd_model_set_timing) on back-to-back
               launches in an extra un-timed pass; peak = 8 TB/s nominal (frac) and 6.3 TB/s achievable (frac_of_achievable);
               traffic = HBM bytes per launch from the PMC counters of the LAST COMMITTED PROFILE (traffic_source
               names it -- counters cannot be read inside this run); secondary = what actually bounds the kernel
               (waves per SIMD, VALU busy, LDS per workgroup) from the same profile + this run's launch geometry
  cpu_baseline the C oracle (oracle/ref_c, fp32, OpenMP over envs: "CPU restatement of the reference
               algorithm, not Warp") timed on this host, rank 0, N = 1 only, on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_BYTES = 8.0e12        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_BYTES = 6.3e12  # same guide: what a streaming kernel reaches in practice
VECTOR_PEAK_FLOPS = 157.3e12   # same guide: fp32 vector peak (packed v_pk_fma_f32: 256 CUs x 4 SIMDs x 16 lanes x 2 x 2 x 2.4 GHz)
VALU_CYCLES_PER_INST = 4.2    # self-measured issue interval of a non-packed fp32 wave64 instruction per SIMD (profiles/r06_micro_valu.txt); the guide lists 2
GLOBAL_BS = 4096               # BASELINE.json: "batch=4096"


def algorithmic_bytes(nb, nqd):
    """per env-step, fp32.  forward = state spill + controls read;
    adjoint = state read + controls read + control grads written."""
    c = 2 * nqd + 6 * nb
    fwd = 4 * (13 * nb + c)
    bwd = 4 * (13 * nb + 2 * c)
    return fwd, bwd


def shard_envs(global_bs, world, rank):
    """Contiguous env slice of a global batch for `rank`."""
    per = global_bs // world
    rem = global_bs % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def rank_inputs(tpl, robot, nsteps, world, rank, scaling, bs, seqs, seed=1000):
    """The rank's contiguous env slice of ONE global batch: weak = `bs` envs per rank out of
    world * bs, strong = shard_envs(bs, world, rank) out of `bs`.  Returns (inputs, (lo, hi), global batch size).
    tests/test_shard_gloo.py drives this same function."""
    from diffphys_amd import synth

    if scaling == "weak":
        gbs, (lo, hi) = world * bs, (rank * bs, (rank + 1) * bs)
    elif scaling == "strong":
        gbs, (lo, hi) = bs, shard_envs(bs, world, rank)
    else:
        raise ValueError("scaling must be 'weak' or 'strong'")
    return synth.make_env_inputs(tpl, robot, range(lo, hi), nsteps, seed=seed, seqs=seqs), (lo, hi), gbs


def cpu_baseline(tpl, robot, nsteps, seqs, budget_s=12.0):
    from diffphys_amd import synth
    from oracle import ref_c

    ref_c.build()
    rc = ref_c.RefC(tpl, np.float32)

    def run(bs):
        inp = synth.make_inputs(tpl, robot, bs=bs, nsteps=nsteps, seed=123, seqs=seqs)
        t0 = time.perf_counter()
        st = rc.rollout_forward(inp, nsteps, inp["frame2step"], inp["dt"])
        rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
        return time.perf_counter() - t0

    threads = rc.num_threads()
    probe_bs = 8 * threads
    run(probe_bs)  # warm (page faults, OpenMP pool)
    tp = run(probe_bs)
    bs = int(min(4096, max(probe_bs, probe_bs * budget_s / max(tp, 1e-6))))
    bs = max(threads, (bs // threads) * threads)
    reps, t = 0, 0.0
    while t < budget_s and reps < 64:  # many-core hosts finish 4096 envs in ~1 s: repeat up to the time budget
        t += run(bs)
        reps += 1
    return {
        "value": reps * bs * nsteps / t,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": "C oracle fp32 (CPU restatement of the reference algorithm, not Warp), OpenMP over envs: "
        "%d x (%d envs x %d steps fwd+adjoint) in %.1f s" % (reps, bs, nsteps, t),
    }


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): N fresh child processes of this same file, one per GPU,
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would -- children, never an
    exec, and started before this process makes any GPU call (it never makes one).  Rank 0's stdout (the one JSON line) is
    forwarded; the return code is non-zero if any rank failed.  SURVEY.md section 8(e): batch split only."""
    import socket
    import subprocess

    with socket.socket() as so:  # a free rendezvous port on the loopback
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PPR_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Poll ALL children: a rank that dies early (no GPU of its own, RCCL failure) leaves the others blocked in init_process_group until
    # the c10d timeout -- once any rank has exited non-zero the rest are terminated instead of waited for.  Rank 0's stdout is drained by
    # a thread so that a full pipe can never block it.
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            t_grace = time.time() + 15.0   # the others usually fail the same way on their own (and say why); then stop waiting
            while time.time() < t_grace and any(rcs[r] is None and p.poll() is None for r, p in enumerate(procs)):
                time.sleep(0.05)
            for r, p in enumerate(procs):
                if rcs[r] is None and p.poll() is not None:
                    rcs[r] = p.returncode
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: ranks failed (rank, rc): %s" % bad, file=sys.stderr)
        return 1
    return 0


def device_identity(dev):
    """what tells two GPUs apart: UUID and PCI address of this rank's device (whatever of it this torch build exposes)"""
    p = torch.cuda.get_device_properties(dev)
    ident = {"index": dev.index, "name": p.name}
    for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"):
        v = getattr(p, k, None)
        if v is not None:
            ident[k] = str(v)
    return ident


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=10, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="weak: --bs envs per GPU; strong: --bs envs in total, sharded across the GPUs")
    ap.add_argument("--bs", type=int, default=GLOBAL_BS, help="envs per GPU (weak) / global batch (strong)")
    ap.add_argument("--T", type=int, default=100, help="sim steps per rollout")
    ap.add_argument("--robot", default="laikago")
    ap.add_argument("--segw", type=int, default=0, help="lanes per articulation (0 = default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the autograd-boundary measurement")
    ap.add_argument("--both", action="store_true", help="also measure the other scaling mode at N = 1 (always done for N > 1)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # typed as `python bench.py --gpus N`: start the N ranks ourselves, BEFORE anything in this process touches the GPU
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world  # a launcher's WORLD_SIZE is authoritative
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not os.environ.get("PPR_BENCH_SHARE_GPU"):
            raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible); one process per GPU" % (rank, ndev))
        local_rank %= ndev  # test hook (tests/test_gpu_bench.py): the N-rank code path on a 1-GPU box, ranks sharing the device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    red_dev = dev
    backend, ranks_seen = None, 1
    if world > 1:
        backend = "nccl"
        try:  # RCCL (backend "nccl" on ROCm); only the barrier and the MAX of the elapsed time go through it
            dist.init_process_group(backend="nccl", device_id=dev)
        except Exception as e:  # keep the measurement alive if RCCL cannot come up on this node: same semantics over gloo,
            #                     and the line SAYS so ("collective_backend": "gloo")
            if rank == 0:
                print("bench.py: nccl/RCCL init failed (%s); using gloo for the barrier" % e, file=sys.stderr)
            dist.init_process_group(backend="gloo")
            red_dev = torch.device("cpu")
            backend = "gloo"
        one = torch.ones(1, dtype=torch.float64, device=red_dev)  # how many ranks the collective really spans
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
    # which DEVICES the ranks run on: a scaling line must prove N distinct GPUs
    devices_seen = [device_identity(dev)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices_seen[0])
        devices_seen = gathered
        keys = {tuple(sorted((k, v) for k, v in d.items() if k not in ("index", "name"))) or ("index", d["index"]) for d in devices_seen}
        if len(keys) != world and not os.environ.get("PPR_BENCH_SHARE_GPU"):
            raise SystemExit("bench.py: %d ranks but %d distinct GPUs (%s); one process per GPU" % (world, len(keys), devices_seen))

    from diffphys_amd import hip_backend, robots, synth

    tpl = robots.load_template(args.robot)
    seqs = ("mi-trot", "mi-spin") if args.robot == "laikago" else ("mi-pace",)
    T = args.T
    dm = hip_backend.DeviceModel(tpl)
    if args.segw:
        dm.set_segment_width(args.segw)

    def barrier():
        if world > 1:
            dist.barrier()

    def setup(scaling):
        """this rank's slice of the global batch, resident in HBM, and the step closure over it"""
        inp, (lo, hi), gbs = rank_inputs(tpl, args.robot, T, world, rank, scaling, args.bs, seqs)
        bs = hi - lo
        t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
        f2s = inp["frame2step"]
        fwd_args = [t[k] for k in ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass",
                                   "body_inertia", "body_inv_inertia")]
        bwd_args = [t[k] for k in ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass",
                                   "body_inertia", "body_inv_inertia")]
        adj_pos = torch.from_numpy(inp["adj_pos"]).to(dev)
        adj_vel = torch.from_numpy(inp["adj_vel"]).to(dev)
        # workspace, frame outputs and gradient buffers live outside the timed loop (the autograd boundary allocates them per
        # call through torch's caching allocator; the bench times the two launches, not the allocator)
        bufs = dm.alloc_rollout(bs, T, len(f2s), dev)

        def step():
            out = dm.rollout_forward(bs, T, inp["dt"], *fwd_args, frame2step=f2s, out=bufs)
            return dm.rollout_backward(bs, T, inp["dt"], *bwd_args, f2s, out[4], adj_pos, adj_vel, out=bufs)

        return step, bs, gbs

    extra_warm = [0]

    def timed(step):
        """W untimed steps (+ the clock ramp), then R blocks of exactly K steps, each between barrier + synchronize brackets, MAX over
        ranks per block; returns the per-block times (s) and the last gradients"""
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        # the first GPU process on a box that has been idle can run its first ~50 ms at a fraction of the clock (measured here:
        # 2.2 ms per step instead of 0.61 for the first bench process on some boxes, 0.61 for every later one): keep warming,
        # untimed, until a quarter of a second of work has gone through -- W steps of 0.6 ms do not wake the clocks up
        t_warm = time.perf_counter()
        extra_warm[0] = 0
        while time.perf_counter() - t_warm < 0.25:
            for _ in range(10):
                step()
            extra_warm[0] += 10
            torch.cuda.synchronize()
        blocks = []
        g = None
        for _ in range(max(1, args.repeats)):
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g = step()
            torch.cuda.synchronize()
            barrier()
            blocks.append(time.perf_counter() - t0)
        local_blocks[:] = blocks   # this rank's own clock, before the MAX over ranks (per_rank in the line)
        if world > 1:
            tt = torch.tensor(blocks, dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            blocks = [float(x) for x in tt.tolist()]
        return blocks, g

    def boundary_step_factory():
        """the same rank-0 workload through the reference's autograd boundary (diffphys_amd/dp_model.py ForwardWarp): tensors that
        require grad, workspace / outputs / gradients allocated per call by the op, loss.backward() through autograd"""
        from diffphys_amd import dp_model

        inp, (lo, hi), _ = rank_inputs(tpl, args.robot, T, world, rank, args.scaling, args.bs, seqs)
        nenv = hi - lo

        class Host:
            pass

        h = Host()
        h.env = robots.env_from_template(args.robot, nenv, device=dev)
        h.num_envs, h.steps_idx, h.frame2step, h.dt = nenv, range(T), inp["frame2step"], inp["dt"]
        t = {k: torch.from_numpy(inp[k]).to(dev).requires_grad_(True) for k in synth.INPUT_NAMES}
        adj_pos = torch.from_numpy(inp["adj_pos"]).to(dev)
        adj_vel = torch.from_numpy(inp["adj_vel"]).to(dev)
        args_ = [t[k] for k in synth.INPUT_NAMES]

        def step():
            for v in args_:
                v.grad = None
            pos, vel = dp_model.ForwardWarp.apply(*args_, h)
            torch.autograd.backward([pos, vel], [adj_pos.view_as(pos), adj_vel.view_as(vel)])
            return {"q_init": t["q_init"].grad}

        return step

    local_blocks = []
    step, bs, gbs = setup(args.scaling)
    blocks, g = timed(step)
    local_ms = float(np.median(local_blocks)) / args.steps * 1e3
    elapsed = float(np.median(blocks))  # the median block: ms_per_step x steps = one real K-step block
    warm_extra = extra_warm[0]
    bad = int(torch.isnan(g["q_init"]).sum().item())

    # per-kernel device time (HIP events on the launch stream around each launch, pd_model_set_timing), in an extra pass outside
    # the timed region: batches of 10 steps enqueued back to back, as the timed region runs them, and the durations of the LAST
    # forward and adjoint launch of each batch are read (the read synchronises).  Bracketing every launch with a host
    # synchronisation instead reads 0.27 + 0.33 ms where the back-to-back pair takes 0.56 (clocks sag in the idle gaps);
    # rocprofv3's per-dispatch durations of this same command (profiles/) are the quantity measured here.
    dm.set_timing(True)
    kf, kb = [], []
    for _ in range(max(4, min(args.steps // 2, 10))):
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        kf.append(dm.last_kernel_ms(0))
        kb.append(dm.last_kernel_ms(1))
    dm.set_timing(False)
    fwd_ms, bwd_ms = float(np.mean(kf)), float(np.mean(kb))

    # per rank: which kernel family ran the rank's shard and its OWN clock -- a scaling record can then tell a slow rank
    # from the latency floor of a small shard (512 envs per GPU at N = 8 take the quad-lane kernels)
    def family(kind):
        i = dm.last_launch_info(kind)
        waves, envs = i["threads_per_wg"] // 64, max(1, i["envs_per_wg"])
        if waves >= 2 * envs:
            return "quad-lane: one env per %d waves" % (waves // envs)
        epw = 64 // max(1, dm.segment_width())                 # envs per wave of the lane-per-body kernels
        roles = waves * epw // envs                            # waves per env group
        return "lane per body: %d envs per wave, %d wave%s per env group" % (epw, roles, "" if roles == 1 else "s")
    mine = {"rank": rank, "envs": bs, "ms_per_step": local_ms, "fwd_kernel_ms": fwd_ms, "bwd_kernel_ms": bwd_ms,
            "fwd_family": family(0), "bwd_family": family(1), "segment_lanes": dm.segment_width()}
    per_rank = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered

    # N > 1: the same measurement in the OTHER scaling mode, reported beside the main one (SURVEY section 8(d) writes C4 as
    # one 4096-env batch split over the GPUs = strong; the driver's contract quotes weak).  At N = 1 the two coincide.
    other = None
    if world > 1 or args.both:
        o_mode = "strong" if args.scaling == "weak" else "weak"
        o_step, o_bs, o_gbs = setup(o_mode)
        o_blocks, _ = timed(o_step)
        o_elapsed = float(np.median(o_blocks))
        other = {"scaling": o_mode, "value": o_gbs * T * args.steps / o_elapsed, "unit": "env-steps/s", "ms_per_step": o_elapsed / args.steps * 1e3,
                 "global_batch": o_gbs, "envs_per_gpu": o_bs,
                 "blocks": {"n": len(o_blocks), "min_value": o_gbs * T * args.steps / max(o_blocks), "max_value": o_gbs * T * args.steps / min(o_blocks)}}

    boundary = None
    if world == 1 and not args.no_boundary:
        b_blocks, gb = timed(boundary_step_factory())
        b_elapsed = float(np.median(b_blocks))
        boundary = {"value": gbs * T * args.steps / b_elapsed, "unit": "env-steps/s", "ms_per_step": b_elapsed / args.steps * 1e3,
                    "ratio_to_value": elapsed / b_elapsed,
                    "what": "ForwardWarp.apply + autograd backward on requires_grad tensors: workspace / output / gradient allocation, "
                            "save_for_backward and the in-kernel remove_nan included; median of %d blocks of %d steps" % (len(b_blocks), args.steps),
                    "nan_grads": int(torch.isnan(gb["q_init"]).sum().item())}

    if rank == 0:
        nb, nqd = int(tpl["nb"]), int(tpl["nqd"])
        bf, bb = algorithmic_bytes(nb, nqd)
        ach_bwd = bs * T * bb / (bwd_ms * 1e-3)
        ach_fwd = bs * T * bf / (fwd_ms * 1e-3)
        prof = {}
        # committed PMC profiles: the headline configuration, and the 512 envs a GPU holds at N = 8 (quad-lane kernels)
        pmc = os.path.join(ROOT, "profiles", "r06_l512_pmc_summary.json" if bs == 512 else "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                with open(pmc) as f:
                    prof = json.load(f)
            except Exception:
                prof = {}
        pb, pf = prof.get("k_rollout_bwd", {}), prof.get("k_rollout_fwd", {})
        same_cfg = args.robot == "laikago" and bs in (GLOBAL_BS, 512) and T == 100 and args.segw == 0  # a committed profile is of this configuration
        # ... and of THIS library: the profile carries the source half of pd_build_id() it was taken on (scripts/profile_gpu.sh); counters of
        # another build say nothing about the kernels timed here, so every counter-derived field goes null
        lib_hash = hip_backend.build_id().split("+")[-1]
        prof_hash = prof.get("source_hash")
        stale_profile = bool(prof) and prof_hash != lib_hash
        if stale_profile:
            same_cfg = False
        geo_b, geo_f = dm.last_launch_info(1), dm.last_launch_info(0)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count

        def secondary(geo, p):
            waves = geo["workgroups"] * geo["threads_per_wg"] / 64.0
            return {
                "waves_per_simd": waves / (cus * 4.0),                      # this run's launch geometry
                "lds_bytes_per_wg": geo["lds_bytes_per_wg"], "workgroups": geo["workgroups"],
                "threads_per_wg": geo["threads_per_wg"], "compute_units": cus,
                "valu_busy": p.get("valu_busy") if same_cfg else None,       # from the committed profile (not this run)
                "wave_wait_share": p.get("wait_share") if same_cfg else None,
            }

        def valu_roofline(p, ms):
            """The bound that actually binds: a SIMD issues one non-packed fp32 wave64 instruction per ~4.2
            cycles whatever the waves or their ILP (scripts/micro/valu_chain.hip, pk_issue.hip).  Instructions per launch from the
            committed profile (SQ_INSTS_VALU), duration from THIS run, clock taken as 2.4 GHz (GRBM_GUI_ACTIVE of the profile agrees)."""
            n = p.get("valu_insts_per_launch") if same_cfg else None
            if not n:
                return None
            peak = cus * 4.0 * 2.4e9 / VALU_CYCLES_PER_INST
            # the data-sheet number beside the self-measured ceiling: 157.3 TFLOP/s fp32 vector (MI355X_MICROARCH.md; v_pk_fma_f32
            # only: 2 FMAs per lane and issue).  Upper bound of what this kernel does against it: EVERY vector instruction counted as
            # one FMA per lane (2 flop x 64 lanes)
            return {"unit": "wave64 fp32 instructions/s", "achieved": n / (ms * 1e-3), "peak": peak, "frac": n / (ms * 1e-3) / peak,
                    "peak_source": "self-measured, raw output in profiles/r06_micro_valu.txt (scripts/micro/run_valu_micro.sh: valu_chain.hip, pk_issue.hip, "
                                   "issue_rate.hip built with the library's flags): one non-packed wave64 instruction per ~4.2 cycles and SIMD at 4 waves per "
                                   "SIMD, 4.5 at the two waves the rollout kernels have, 4.8-4.9 for a lone wave.  /opt/skills/guides/MI355X_MICROARCH.md lists "
                                   "v_fma_f32 wave64 at 2 cycles (SIMD-32): against THAT figure frac halves (frac_of_guide_rate)",
                    "frac_of_guide_rate": n / (ms * 1e-3) / (cus * 4.0 * 2.4e9 / 2.0),
                    "frac_of_vector_peak": n * 128.0 / (ms * 1e-3) / VECTOR_PEAK_FLOPS, "vector_peak_tflops": VECTOR_PEAK_FLOPS / 1e12,
                    "frac_of_vector_peak_note": "instructions x 64 lanes x 2 flop (every instruction priced as an FMA: an upper bound) / 157.3 TFLOP/s",
                    "insts_per_launch": n, "cycles_per_instruction": VALU_CYCLES_PER_INST, "clock_ghz_assumed": 2.4}

        line = {
            "metric": "env-steps/sec (fwd+adjoint), Laikago 12-DoF, batch=4096, 1/2/4/8 MI355X",
            "value": gbs * T * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "collective_backend": backend,   # "nccl" (= RCCL) | "gloo" (RCCL could not come up) | null at N = 1: barrier + MAX only
            "ranks_seen": ranks_seen,        # all-reduced count of ranks behind that backend
            "per_rank": per_rank,            # all-gathered: each rank's shard, kernel family, own ms_per_step and kernel durations
            "devices_seen": devices_seen,    # all-gathered identity (UUID / PCI address) of each rank's GPU: N distinct unless PPR_BENCH_SHARE_GPU (a test hook)
            "devices_distinct": len({json.dumps({k: v for k, v in d.items() if k not in ("index", "name")} or d, sort_keys=True) for d in devices_seen}),
            "launcher": "self" if os.environ.get("PPR_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else "none"),
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_extra_steps": warm_extra,
            "ms_per_step": elapsed / args.steps * 1e3,
            "blocks": {"n": len(blocks), "statistic": "median", "min_value": gbs * T * args.steps / max(blocks),
                       "max_value": gbs * T * args.steps / min(blocks), "ms_per_step_all": [b / args.steps * 1e3 for b in blocks]},
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s, mixed %s mocap targets, %s x %d sim steps (4 frames), dt=5e-4, "
                "fwd+adjoint, batch split only (no collective)" % (
                    args.robot, "/".join(seqs),
                    ("%d envs per GPU" % args.bs) if args.scaling == "weak" else ("%d envs in total (%d on rank 0)" % (gbs, bs)), T),
                "global_batch": gbs,
                "envs_per_gpu": bs,
                "sim_steps": T,
                "segment_lanes": dm.segment_width(),
                "nan_grads": bad,
            },
            "roofline": {
                # what binds is fp32 VALU issue (roofline.valu); achieved / peak / unit / frac below stay the HBM figures the contract
                # defines (algorithmic bytes / launch duration against 8 TB/s), repeated as hbm_frac
                "bound": "valu-issue",
                "hbm_frac": ach_bwd / HBM_PEAK_BYTES,
                "limiter": "fp32 VALU issue: a SIMD issues one non-packed wave64 fp32 instruction per ~4.2 cycles whatever the waves or "
                           "their ILP; the adjoint's two waves per SIMD keep it ~95 % busy (roofline.valu; the forward pass, three waves per SIMD -- "
                           "body, contact, cull -- ~56 %: the step's chain crosses two hand-overs).  frac of HBM is what the contract asks for",
                "valu": valu_roofline(pb, bwd_ms),
                "kernel": "k_rollout_bwd",
                "achieved": ach_bwd / 1e9,
                "peak": HBM_PEAK_BYTES / 1e9,
                "unit": "GB/s",
                "frac": ach_bwd / HBM_PEAK_BYTES,
                "frac_of_achievable": ach_bwd / HBM_ACHIEVABLE_BYTES,
                "peak_achievable": HBM_ACHIEVABLE_BYTES / 1e9,
                "traffic": pb.get("hbm_bytes_per_launch") if same_cfg else None,
                "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `bench.py --steps 10`), NOT measured in "
                                   "this run; taken on library source %s = the one loaded here" % (prof.get("tag", "pmc_summary.json"), prof_hash)) if (same_cfg and pb)
                                  else ("the committed profile is of another build (library source %s, loaded %s): counter-derived fields are null"
                                        % (prof_hash, lib_hash) if stale_profile else None),
                "avg_launch_ms": bwd_ms,
                # the event pass against the timed region: its forward + adjoint launch durations / the timed ms per step.  Above 1 = the
                # event pass runs slower than the timed blocks (the event records sit between the launches), i.e. `frac` UNDERSTATES the
                # kernel by about that factor; pair_frac_timed = the whole pair's algorithmic bytes / the TIMED step against 8 TB/s
                "event_pass_over_timed": (fwd_ms + bwd_ms) / (elapsed / args.steps * 1e3),
                "pair_frac_timed": bs * T * (bf + bb) / (elapsed / args.steps) / HBM_PEAK_BYTES,
                "algorithmic_bytes_per_env_step": bb,
                "secondary": secondary(geo_b, pb),
                "fwd_kernel": {"kernel": "k_rollout_fwd", "achieved": ach_fwd / 1e9, "frac": ach_fwd / HBM_PEAK_BYTES,
                               "frac_of_achievable": ach_fwd / HBM_ACHIEVABLE_BYTES, "avg_launch_ms": fwd_ms,
                               "algorithmic_bytes_per_env_step": bf, "secondary": secondary(geo_f, pf), "valu": valu_roofline(pf, fwd_ms)},
                "note": "VALU-issue bound (adjoint) / hand-over chain of three waves sharing a SIMD's issue slots (forward), not HBM-bound (roofline.valu, roofline.secondary, "
                        "DESIGN.md section 4); launch durations are per dispatch, measured on launches enqueued back to back like the timed region",
            },
        }
        if other is not None:
            line["other_scaling"] = other
        if boundary is not None:
            line["boundary"] = boundary
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(tpl, args.robot, T, seqs)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
