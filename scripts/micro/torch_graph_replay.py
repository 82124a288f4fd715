"""Pure-torch reproducer (no code of this repo): forward + backward of a small MLP captured in one HIP graph replays correctly
once; from the second replay on the bias gradient of a Linear layer (a multi-block reduction) is wrong on ROCm 7.2 / torch 2.10.
This is why phys_model.capture_iteration() validates itself and why the whole-iteration graph is not the default."""
import torch, torch.nn as nn
torch.manual_seed(0)
dev = "cuda"
net = nn.Sequential(nn.Linear(64, 128), nn.ReLU(), nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 18)).to(dev)
x = torch.randn(1024, 64, device=dev)
def step():
    y = net(x)
    return (y.pow(2).mean() + (y * 0).sum())
# eager
net.zero_grad(set_to_none=True); l = step(); l.backward()
ge = [p.grad.clone() for p in net.parameters()]
le = float(l); del l
# graph
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        net.zero_grad(set_to_none=True); step().backward()
    net.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        lg = step(); lg.backward()
torch.cuda.current_stream().wait_stream(s)
for it in range(3):
    g.replay(); torch.cuda.synchronize()
    print("replay", it, le, float(lg), [float((a - p.grad).abs().max() / (a.abs().max() + 1e-20)) for a, p in zip(ge, net.parameters())])
