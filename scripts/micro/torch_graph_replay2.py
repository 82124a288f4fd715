"""Which torch op replays wrongly inside a HIP graph on this stack (ROCm 7.2 / torch 2.10)?  Pure torch, no code of this repo.
Each candidate op is captured alone (static input buffers), replayed 4 times with fresh inputs, and compared with the eager result on
the same inputs.  Then the MLP forward + backward of torch_graph_replay.py in variants (bias through a ones column, autograd
multithreading off, reductions replaced by matmuls) to find a formulation that replays correctly.

    python scripts/micro/torch_graph_replay2.py
"""
import os
import sys

import torch
import torch.nn as nn

dev = "cuda"
torch.manual_seed(0)


def capture(fn, warm=3):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = fn()
    torch.cuda.current_stream().wait_stream(s)
    return g, out


def check_op(name, make_inputs, op):
    ins = make_inputs()
    g, out = capture(lambda: op(*ins))
    worst = []
    for it in range(4):
        fresh = make_inputs()
        for a, b in zip(ins, fresh):
            a.copy_(b)
        g.replay()
        torch.cuda.synchronize()
        ref = op(*[x.clone() for x in ins])
        torch.cuda.synchronize()
        o, r = (out, ref) if torch.is_tensor(out) else (out[0], ref[0])
        worst.append(float((o - r).abs().max() / (r.abs().max() + 1e-20)))
    print("op %-34s replay errors %s %s" % (name, " ".join("%.1e" % w for w in worst), "  <-- WRONG" if max(worst) > 1e-5 else ""), flush=True)


R = lambda *s: (lambda: [torch.randn(*s, device=dev)])
check_op("sum(0) [1024,128]", R(1024, 128), lambda x: x.sum(0))
check_op("sum(0) [25600,256]", R(25600, 256), lambda x: x.sum(0))
check_op("sum() [25600,256]", R(25600, 256), lambda x: x.sum())
check_op("mean() [1024,18]", R(1024, 18), lambda x: x.pow(2).mean())
check_op("norm [25600,256]", R(25600, 256), lambda x: x.norm())
check_op("x.t() @ y [256x25600x256]", lambda: [torch.randn(25600, 256, device=dev), torch.randn(25600, 256, device=dev)], lambda x, y: x.t() @ y)
check_op("addmm bias", lambda: [torch.randn(1024, 64, device=dev), torch.randn(128, 64, device=dev), torch.randn(128, device=dev)],
         lambda x, w, b: torch.nn.functional.linear(x, w, b))
check_op("foreach_norm", lambda: [torch.randn(300, 300, device=dev), torch.randn(70000, device=dev)], lambda a, b: torch.stack(torch._foreach_norm([a, b])))
check_op("median(1) [80,10]", R(80, 10), lambda x: x.median(1).values)
check_op("sort [64]", R(64), lambda x: x.sort().values)
check_op("cumsum [4096,4]", R(4096, 4), lambda x: x.cumsum(1))


class LinearOnes(nn.Module):
    """y = [x, 1] @ [W; b]: the bias gradient is a row of a GEMM, not a reduction kernel"""

    def __init__(self, i, o):
        super().__init__()
        l = nn.Linear(i, o)
        self.wb = nn.Parameter(torch.cat([l.weight.detach().t(), l.bias.detach()[None]], 0))

    def forward(self, x):
        return torch.cat([x, torch.ones_like(x[:, :1])], 1) @ self.wb


def mlp_case(tag, make_net, loss_fn, mt=True):
    torch.manual_seed(0)
    torch.autograd.set_multithreading_enabled(mt)
    net = make_net().to(dev)
    x = torch.randn(1024, 64, device=dev)

    def step():
        return loss_fn(net(x))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            net.zero_grad(set_to_none=True)
            step().backward()
        net.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            lg = step()
            lg.backward()
    torch.cuda.current_stream().wait_stream(s)
    static_grads = [p.grad for p in net.parameters()]
    errs = []
    for it in range(4):
        x.copy_(torch.randn(1024, 64, device=dev))
        g.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in static_grads]
        lgv = float(lg)
        for p in net.parameters():
            p.grad = None
        l = step()
        l.backward()
        torch.cuda.synchronize()
        e = max(float((a - p.grad).abs().max() / (p.grad.abs().max() + 1e-20)) for a, p in zip(got, net.parameters()))
        errs.append((e, abs(lgv - float(l)) / abs(float(l))))
        for p, t in zip(net.parameters(), static_grads):
            p.grad = t
    print("mlp %-46s grad errors %s | loss errors %s %s" % (tag, " ".join("%.1e" % e for e, _ in errs), " ".join("%.1e" % e for _, e in errs),
                                                             "  <-- WRONG" if max(e for e, _ in errs) > 1e-4 else ""), flush=True)
    torch.autograd.set_multithreading_enabled(True)


std = lambda: nn.Sequential(nn.Linear(64, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 18))
ones = lambda: nn.Sequential(LinearOnes(64, 128), nn.ReLU(), LinearOnes(128, 128), nn.ReLU(), LinearOnes(128, 18))
nobias = lambda: nn.Sequential(nn.Linear(64, 128, bias=False), nn.ReLU(), nn.Linear(128, 128, bias=False), nn.ReLU(), nn.Linear(128, 18, bias=False))
mlp_case("nn.Linear, mean loss", std, lambda y: y.pow(2).mean())
mlp_case("nn.Linear, mean loss, autograd single-threaded", std, lambda y: y.pow(2).mean(), mt=False)
mlp_case("nn.Linear bias=False, mean loss", nobias, lambda y: y.pow(2).mean())
mlp_case("ones-column Linear, mean loss", ones, lambda y: y.pow(2).mean())
mlp_case("ones-column Linear, single-threaded", ones, lambda y: y.pow(2).mean(), mt=False)
mlp_case("nn.Linear, loss by matmul (no reduction)", std, lambda y: (y.reshape(1, -1) @ y.reshape(-1, 1)).reshape(()) / y.numel())
print("torch", torch.__version__, "hip", torch.version.hip, "PYTORCH_NO_HIP_MEMORY_CACHING" in os.environ)
