#!/usr/bin/env python3
"""Host-side cost of one rollout iteration at the boundary: time to ENQUEUE (no sync) against time to COMPLETE, for the raw backend
calls with preallocated buffers (what bench.py times), ForwardWarp through autograd, and ForwardWarpTrajLoss.
usage: gpu_host_overhead.py [robot:bs ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ppr-diffphys_amd")]
import numpy as np, torch
from diffphys_amd import dp_model, hip_backend, robots, synth

dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["laikago:512", "laikago:4096"]):
    name, bs = cfg.split(":"); bs = int(bs)
    tpl = robots.load_template(name); T = 100
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=0, seqs=("mi-trot", "mi-spin"))
    f2s = list(inp["frame2step"]); F = len(f2s); nb = int(tpl["nb"])

    class Host: pass
    h = Host()
    h.env = robots.env_from_template(name, bs, device=dev)
    h.num_envs, h.steps_idx, h.frame2step, h.dt = bs, range(T), f2s, inp["dt"]
    t = {k: torch.from_numpy(inp[k]).to(dev).requires_grad_(True) for k in synth.INPUT_NAMES}
    args = [t[k] for k in synth.INPUT_NAMES]
    dm = hip_backend.device_model(h.env)
    d = {k: t[k].detach() for k in synth.INPUT_NAMES}
    fa = [d[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
    ba = [d[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
    ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
    bufs = dm.alloc_rollout(bs, T, F, dev)
    with torch.no_grad():
        pos0, _ = dp_model.ForwardWarp.apply(*args, h)
    tgt = (pos0.reshape(F, bs, nb, 7).permute(1, 0, 2, 3) + 0.02 * torch.randn(bs, F, nb, 7, device=dev)).contiguous()
    outseq = torch.zeros(bs, F, dtype=torch.bool, device=dev)

    def raw():
        out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=f2s, out=bufs)
        dm.rollout_backward(bs, T, inp["dt"], *ba, f2s, out[4], ap, av, out=bufs)

    def autograd():
        for v in args: v.grad = None
        pos, vel = dp_model.ForwardWarp.apply(*args, h)
        torch.autograd.backward([pos, vel], [ap, av])

    def fused():
        for v in args: v.grad = None
        loss, _, _ = dp_model.ForwardWarpTrajLoss.apply(*args, tgt, outseq, h)
        loss.backward()

    # the raw pair captured in a HIP graph: one host call per iteration
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        raw()
        with torch.cuda.graph(graph, stream=side):
            raw()
    torch.cuda.current_stream().wait_stream(side)

    for tag, fn, mt in (("graph replay (raw pair)", graph.replay, True), ("raw backend calls", raw, True), ("ForwardWarp autograd", autograd, True), ("ForwardWarpTrajLoss", fused, True),
                        ("ForwardWarp, 1 thread", autograd, False), ("TrajLoss, 1 thread", fused, False)):
        torch.autograd.set_multithreading_enabled(mt)   # False: backward runs on the calling thread (no hand-over to the device thread)
        for _ in range(20): fn()
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("HOST %-8s bs=%-5d %-24s enqueue %.3f ms / iteration, complete %.3f ms / iteration" % (name, bs, tag, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
