#!/usr/bin/env python3
"""EXPERIMENT (not product code): one optimisation iteration of phys_model eager vs forward() + backward() captured in one HIP
graph (static frame_start / noise buffers, documented whole-network pattern), then update() eagerly.  Possible because
nothing between forward() and backward() synchronises the host any more.  Measured on MI355X / ROCm 7.2 / torch 2.10,
256 envs: eager 33.4 ms, captured 21.9 ms per iteration -- but the captured gradients are WRONG from the second replay on
(torch's own multi-block reductions, e.g. nn.Linear bias gradients: pure-torch reproducer scripts/micro/torch_graph_replay.py)
and capture_end can crash after eager iterations on the default stream, so the whole-iteration graph is not shipped.  The
library's own launches (FK, rollout, fused losses, adjoint) capture and replay bit-exactly:
tests/test_gpu_parity.py::test_empty_batch_and_graph_capture.   Usage: gpu_iter_graph.py [num_envs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import importlib.util
import numpy as np, torch
from diffphys_amd.dataloader import DataLoader
from diffphys_amd.phys_model import phys_model

nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = importlib.util.spec_from_file_location("pd_main", os.path.join(ROOT, "ppr-diffphys_amd", "main.py"))
pd_main = importlib.util.module_from_spec(spec); spec.loader.exec_module(pd_main)
opts = pd_main.get_opts(["--seqname", "mi-pace", "--urdf_template", "laikago", "--logroot", "/tmp/pprdp_prof/", "--logname", "g",
                         "--num_envs", str(nenv), "--frames_per_wdw", "4"])


def make():
    torch.manual_seed(0); np.random.seed(0)
    m = phys_model(opts, DataLoader(opts)).cuda(); m.train()
    m.reinit_envs(nenv, frames_per_wdw=4)
    return m


fs = (torch.arange(nenv, device="cuda") * 3) % 40
K = 20
res = {}
for mode in ("eager", "graph"):
    model = make()
    if mode == "graph":
        nq = model.n_dof + 7
        g_fs = torch.zeros(nenv, dtype=torch.long, device="cuda")
        g_noise = torch.zeros(nenv * nq, device="cuda")
        model._inv_norm_inertia(); model._frame_index()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                model.optimizer.zero_grad(set_to_none=True)
                model.forward(frame_start=g_fs, q_init_noise=g_noise)["total_loss"].backward()
            model.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                g_out = model.forward(frame_start=g_fs, q_init_noise=g_noise)
                g_out["total_loss"].backward()
        torch.cuda.current_stream().wait_stream(side)
        model.optimizer.zero_grad = lambda set_to_none=True: None  # the captured backward ASSIGNS its static gradient buffers
    losses = []

    def one(it):
        model.set_progress(it)
        np.random.seed(100 + it)  # the init noise is drawn on the host: same stream of numbers in both modes
        if mode == "graph":
            g_fs.copy_(fs)
            noise = model.make_q_init_noise()
            g_noise.copy_(noise) if noise is not None else g_noise.zero_()
            graph.replay()
            out = g_out
            model._pending_loss = out["total_loss"].detach()
        else:
            out = model.forward(frame_start=fs)
            model.backward(out["total_loss"])
        losses.append(out["total_loss"].detach().clone())
        model.update()

    for it in range(5):
        one(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(K):
        one(5 + it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    res[mode] = (dt, torch.stack(losses).cpu().numpy(), {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()})
    print("ITER %-5s num_envs=%d: %.2f ms per iteration (forward + backward + update)" % (mode, nenv, dt), flush=True)
le, lg = res["eager"][1], res["graph"][1]
print("loss trajectory eager vs graph: first %.6e / %.6e  last %.6e / %.6e  max rel diff %.2e" % (
    le[0], lg[0], le[-1], lg[-1], np.abs(le - lg).max() / np.abs(le).max()))
worst = max((np.abs(res["eager"][2][n] - res["graph"][2][n]).max() / (np.abs(res["eager"][2][n]).max() + 1e-12), n) for n in res["eager"][2])
print("parameters after %d iterations: max rel diff %.2e (%s)" % (K + 5, worst[0], worst[1]))
