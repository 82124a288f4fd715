import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ppr-diffphys_amd"))
import numpy as np, torch
import bench
from diffphys_amd import hip_backend, robots, synth
t00 = time.perf_counter()
dev = torch.device("cuda:0")
tpl = robots.load_template("laikago"); T = 100
inp, (lo, hi), gbs = bench.rank_inputs(tpl, "laikago", T, 1, 0, "weak", 4096, ("mi-trot", "mi-spin"))
bs = hi - lo
dm = hip_backend.DeviceModel(tpl)
t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
f2s = inp["frame2step"]
fa = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ba = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
bufs = dm.alloc_rollout(bs, T, len(f2s), dev)
torch.cuda.synchronize()
print("setup %.1f s" % (time.perf_counter() - t00))
# the bench's pattern: 3 warm-up steps, synchronise, 20 steps enqueued back to back, synchronise
def step():
    out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=f2s, out=bufs)
    return dm.rollout_backward(bs, T, inp["dt"], *ba, f2s, out[4], ap, av, out=bufs)
for rep in range(4):
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: enqueue of 20 steps %.2f ms, until done %.2f ms (%.3f ms per step)" % (rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / 20))
ts = []
for i in range(60):
    t0 = time.perf_counter()
    out = dm.rollout_forward(bs, T, inp["dt"], *fa, frame2step=f2s, out=bufs)
    g = dm.rollout_backward(bs, T, inp["dt"], *ba, f2s, out[4], ap, av, out=bufs)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % x for x in ts))
