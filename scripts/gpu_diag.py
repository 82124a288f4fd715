#!/usr/bin/env python3
"""GPU diagnostic: HIP rollout fwd/bwd + FK vs the C oracle (fp32 and fp64). Prints error tables."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
import numpy as np, torch
from diffphys_amd import robots, synth, hip_backend, dp_model
from oracle.ref_c import RefC, build
build()

def err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    return d.max(), d.max() / (np.abs(b).max() + 1e-30)

def run(name, bs, nsteps, segw, seqs=("mi-pace",), perturb=True):
    tpl = robots.load_template(name)
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=nsteps, seed=3, seqs=seqs)
    if perturb:
        rng = np.random.RandomState(7)
        inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.5).astype(np.float32)
        inp["res_f"] = (rng.randn(*inp["res_f"].shape) * 0.5).astype(np.float32)
        inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * 0.1).astype(np.float32)
    f2s = inp["frame2step"]
    dm = hip_backend.DeviceModel(tpl)
    if segw: dm.set_segment_width(segw)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
    fos = list(f2s)
    args = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
    wp_pos, wp_vel, grf, jaf, ws = dm.rollout_forward(bs, nsteps, inp["dt"], *args, frame2step=fos)
    torch.cuda.synchronize()
    a2 = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
    g = dm.rollout_backward(bs, nsteps, inp["dt"], *a2, fos, ws, torch.from_numpy(inp["adj_pos"]).to(dev), torch.from_numpy(inp["adj_vel"]).to(dev))
    torch.cuda.synchronize()
    print("== %s bs=%d T=%d segw=%d" % (name, bs, nsteps, dm.segment_width()))
    for dt_, lab in ((np.float32, "c32"), (np.float64, "c64")):
        rc = RefC(tpl, dt_)
        st = rc.rollout_forward(inp, nsteps, f2s, inp["dt"])
        gg = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
        print("  vs %s: pos %.2e/%.2e vel %.2e/%.2e grf %.2e/%.2e jaf %.2e/%.2e" % ((lab,) + err(wp_pos.cpu(), st["wp_pos"]) + err(wp_vel.cpu(), st["wp_vel"]) + err(grf.cpu(), st["grf"]) + err(jaf.cpu(), st["jaf"])))
        for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia"):
            e = err(g[k].cpu().numpy().reshape(gg[k].shape), gg[k])
            print("     grad %-17s max|ref|=%.3e abs=%.2e rel=%.2e nan=%d" % (k, np.abs(gg[k]).max(), e[0], e[1], int(torch.isnan(g[k]).sum())))
    # FK
    rc = RefC(tpl, np.float64)
    n = 8
    rng = np.random.RandomState(1)
    jq = np.tile(inp["q_init"].reshape(bs, -1)[:1], (n, 1)).astype(np.float32) + (rng.randn(n, dm.nq) * 0.05).astype(np.float32)
    jqd = (rng.randn(n, dm.nqd) * 0.3).astype(np.float32)
    bq, bqd = dm.fk_forward(torch.from_numpy(jq).to(dev), torch.from_numpy(jqd).to(dev))
    rq, rqd = rc.fk_forward(jq, jqd)
    aq = rng.randn(n, dm.nb, 7).astype(np.float32); aqd = rng.randn(n, dm.nb, 6).astype(np.float32)
    gq, gqd = dm.fk_backward(torch.from_numpy(jq).to(dev), torch.from_numpy(jqd).to(dev), torch.from_numpy(aq).to(dev), torch.from_numpy(aqd).to(dev))
    rgq, rgqd = rc.fk_backward(jq, jqd, rq, aq, aqd)
    # pd_fk_backward returns the gradients with ForwardKinematics.backward's post-processing (dp_model.py:1109-1123 of the reference:
    # NaN -> 0, values above 1 -> 1) since ABI v4: the same on the oracle's raw gradients before comparing
    rgq, rgqd = np.minimum(np.nan_to_num(rgq, nan=0.0), 1.0), np.minimum(np.nan_to_num(rgqd, nan=0.0), 1.0)
    print("  FK: body_q %.2e/%.2e body_qd %.2e/%.2e g_q %.2e/%.2e g_qd %.2e/%.2e" % (err(bq.cpu(), rq) + err(bqd.cpu(), rqd) + err(gq.cpu(), rgq) + err(gqd.cpu(), rgqd)))

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run("laikago", 6, 40, 0)
    run("laikago", 37, 100, 0, perturb=False)
    run("laikago", 5, 40, 64)
    run("laikago", 5, 40, 32)
    run("human", 5, 40, 0)
    run("human", 3, 40, 64)
    run("quad", 7, 40, 0)
    # timing
    for name, bs, segw in (("laikago", 4096, 16), ("laikago", 4096, 32), ("laikago", 4096, 64), ("human", 1024, 32), ("quad", 8192, 32)):
        tpl = robots.load_template(name)
        nsteps = 100
        inp = synth.make_inputs(tpl, name, bs=bs, nsteps=nsteps, seed=0, seqs=("mi-trot", "mi-spin"))
        dm = hip_backend.DeviceModel(tpl); dm.set_segment_width(segw); dm.set_timing(True)
        dev = torch.device("cuda:0")
        t = {k: torch.from_numpy(inp[k]).to(dev) for k in synth.INPUT_NAMES}
        f2s = inp["frame2step"]; fos = list(f2s)
        args = [t[k] for k in ("q_init","qd_init","torques","res_f","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
        a2 = [t[k] for k in ("q_init","qd_init","torques","refs","target_ke","target_kd","body_inv_mass","body_inertia","body_inv_inertia")]
        ap = torch.from_numpy(inp["adj_pos"]).to(dev); av = torch.from_numpy(inp["adj_vel"]).to(dev)
        for it in range(3):
            out = dm.rollout_forward(bs, nsteps, inp["dt"], *args, frame2step=fos)
            g = dm.rollout_backward(bs, nsteps, inp["dt"], *a2, fos, out[4], ap, av)
            torch.cuda.synchronize()
            f, b = dm.last_kernel_ms(0), dm.last_kernel_ms(1)
        print("TIMING %s bs=%d segw=%d T=%d fwd %.3f ms bwd %.3f ms -> %.3e env-steps/s" % (name, bs, segw, nsteps, f, b, bs * nsteps / ((f + b) * 1e-3)))
