#!/usr/bin/env python3
"""Randomised parity sweep against the C oracle (float64; the fp32 oracle gauges the conditioning of a case that is off): robots x batch sizes x horizons x perturbations (kicks that force
the speculative sweep's redo path, drops into the ground that overflow hit lists, large joint velocities), plus run-to-run
bitwise repeats.  Prints one line per case and a summary; exits non-zero on a violation.  Usage: gpu_stress.py [ncases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from diffphys_amd import dp_model, hip_backend, robots, synth
from oracle import ref_c
from oracle.ref_c import RefC

INPUT_NAMES = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_mass", "body_inv_mass", "body_inertia",
               "body_inv_inertia")
FWD = ("q_init", "qd_init", "torques", "res_f", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
BWD = ("q_init", "qd_init", "torques", "refs", "target_ke", "target_kd", "body_inv_mass", "body_inertia", "body_inv_inertia")
dev = torch.device("cuda:0")

def gpu(dm, inp):
    bs = inp["q_init"].size // dm.nq
    T, f2s = inp["nsteps"], inp["frame2step"]
    t = {k: torch.from_numpy(np.ascontiguousarray(inp[k], dtype=np.float32)).to(dev) for k in INPUT_NAMES + ("adj_pos", "adj_vel")}
    fos = list(f2s)
    pos, vel, grf, jaf, ws = dm.rollout_forward(bs, T, inp["dt"], *[t[k] for k in FWD], frame2step=fos)
    g = dm.rollout_backward(bs, T, inp["dt"], *[t[k] for k in BWD], fos, ws, t["adj_pos"], t["adj_vel"])
    out = dict(wp_pos=pos.cpu().numpy(), wp_vel=vel.cpu().numpy(), grf=grf.cpu().numpy(), jaf=jaf.cpu().numpy())
    out.update({"g_" + k: v.cpu().numpy() for k, v in g.items()})
    return out

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b.reshape(a.shape)).max() / (np.abs(b).max() + 1e-30))

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ref_c.build()
bad = 0
tpls = {n: robots.load_template(n) for n in ("laikago", "human", "quad")}
for case in range(ncases):
    name = ("laikago", "laikago", "human", "quad")[case % 4]
    tpl = tpls[name]
    nb = int(tpl["nb"])
    bs = int(rng.choice([1, 3, 4, 5, 16, 17, 33, 64, 70]))
    T = int(rng.choice([1, 2, 7, 12, 20, 34]))
    spf = int(rng.choice([1, 3, 5, 11]))
    kind = ("plain", "kick", "drop", "spin")[rng.randint(4)]
    pen = float(rng.choice([0.0, 0.003, 0.01]))
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=T, seed=int(rng.randint(1 << 30)), steps_per_frame=spf, penetration=pen)
    inp["torques"] = (rng.randn(*inp["torques"].shape) * 0.5).astype(np.float32)
    inp["res_f"] = (rng.randn(*inp["res_f"].shape) * 0.5).astype(np.float32)
    inp["qd_init"] = (rng.randn(*inp["qd_init"].shape) * (2.0 if kind == "spin" else 0.1)).astype(np.float32)
    if kind == "kick":
        rf = inp["res_f"].reshape(T, bs, nb, 6)
        k = np.zeros((T, bs), np.float32); k[T // 4:, ::2] = float(rng.choice([800.0, 3000.0]))
        rf[..., 4] -= k[:, :, None] * inp["body_mass"].reshape(bs, nb)[None]
    if kind == "drop":
        q = inp["q_init"].reshape(bs, -1); q[:, 1] -= np.linspace(0.05, 0.3, bs).astype(np.float32)
    segw = int(rng.choice([0, 0, 32, 64])) if name == "laikago" else int(rng.choice([0, 64]))
    dm = hip_backend.DeviceModel(tpl)
    if segw:
        dm.set_segment_width(segw)
    o1 = gpu(dm, inp)
    o2 = gpu(dm, inp)
    same = all(np.array_equal(o1[k], o2[k]) for k in o1)
    rc = RefC(tpl, np.float64)  # the authority; round 3: the kernel's twist angle (atan2) is closer to float64 than the fp32 oracle's acos
    st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
    gr = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
    # violent regimes (kicks / drops: 500 N clamps, chaotic) get the loose bars of the dedicated tests
    hard = kind in ("kick", "drop") or T > 20
    e = dict(pos=rel(o1["wp_pos"], st["wp_pos"]), vel=rel(o1["wp_vel"], st["wp_vel"]), grf=rel(o1["grf"], st["grf"]),
             g_q=rel(o1["g_q_init"], gr["q_init"]), g_refs=rel(o1["g_refs"], gr["refs"]), g_m=rel(o1["g_body_inv_mass"], gr["body_inv_mass"]))
    lim = dict(pos=2e-4 if hard else 5e-5, vel=1e-2 if hard else 2e-3, grf=2e-2 if hard else 5e-3, g_q=5e-2 if hard else 2e-2,
               g_refs=5e-2 if hard else 2e-2, g_m=5e-2 if hard else 2e-2)
    finite = all(np.isfinite(v).all() for v in o1.values())
    if pen == 0.0:  # feet start exactly ON the ground: whether a point has c <= 0 is decided by the last bit, and d force / d height
        lim["g_q"] = float("inf")  # jumps by ke = 1e4 per point there -- the pose gradient is not comparable (everything else is)
    ok = same and finite and all(e[k] < lim[k] for k in e)
    note = ""
    if same and finite and not ok:
        # is the case itself ill-conditioned in fp32?  compare the oracle with itself in float64: if fp32 and fp64 oracles
        # disagree as much as the kernel does with the fp32 oracle, the mismatch is conditioning (contacts switching at
        # c = 0, chaotic impacts), not the kernel
        r32 = RefC(tpl, np.float32)
        s32 = r32.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
        g32 = r32.rollout_backward(s32, inp["adj_pos"], inp["adj_vel"])
        o = dict(pos=rel(s32["wp_pos"], st["wp_pos"]), vel=rel(s32["wp_vel"], st["wp_vel"]), grf=rel(s32["grf"], st["grf"]),
                 g_q=rel(g32["q_init"], gr["q_init"]), g_refs=rel(g32["refs"], gr["refs"]), g_m=rel(g32["body_inv_mass"], gr["body_inv_mass"]))
        # violent cases amplify rounding noise exponentially for a dozen steps (measured: kernel-vs-oracle and fp32-vs-fp64
        # oracle errors grow ~3x per step at the same rate after a 3000 m/s^2 kick, then decay): allow 30x the oracle's own spread
        slack = 30.0 if hard else 3.0
        if all(e[k] < lim[k] or e[k] < slack * o[k] for k in e):
            ok, note = True, "ill-conditioned (oracle fp32 vs fp64: " + " ".join("%s %.1e" % (k, o[k]) for k in e if e[k] >= lim[k]) + ")"
    if same and finite and not ok:
        # per env (tests/helpers.py, the machinery of test_config_size_gradients_every_tensor_per_env): every env whose gradient is off
        # must be explained by its measured conditioning or by a discrete branch the kernel took differently from float64
        from helpers import first_branch_difference, grad_env_errors, oracle_bundle, GRAD_LEAD
        from test_gpu_parity import gpu_rollout
        full = gpu_rollout(dm, inp, dev, keep_traj=True)
        ob = oracle_bundle(tpl, inp, bs)
        ee = grad_env_errors(full["grads"], ob["g64"], bs)
        w = np.max(np.stack([ee[k] for k in GRAD_LEAD]), 0)
        first = first_branch_difference(ob["rc64"], ob["st64"], full["traj"], inp, bs)
        # forward outputs env by env too (round 3, 12 000-case sweep: a branch difference -- a clamp, a contact, the Coulomb switch --
        # in a kicked or spinning robot moves THAT env's poses off the bars as well; the other envs of the batch stay inside them)
        F = len(inp["frame2step"])
        def fwd_env(a, r, wd):
            a = np.asarray(a, np.float64).reshape(F, bs, nb, wd); r = np.asarray(r, np.float64).reshape(F, bs, nb, wd)
            return np.abs(a - r).max((0, 2, 3)) / (np.abs(r).max() + 1e-30)
        st64 = ob["st64"]
        excess = np.maximum(np.maximum(fwd_env(full["wp_pos"], st64["wp_pos"], 7) / lim["pos"], fwd_env(full["wp_vel"], st64["wp_vel"], 6) / lim["vel"]),
                            np.maximum(fwd_env(full["grf"], st64["grf"], 6) / lim["grf"], w / 1e-3))  # > 1: off a bar
        unexpl = (excess > 1.0) & (first >= T) & (excess > 30 * ob["cond"] / 1e-3)
        if not unexpl.any():
            ok, note = True, "every env explained (%d of %d off a bar, %d with a branch difference)" % ((excess > 1.0).sum(), bs, (first < T).sum())
    # round 4: the airtight comparison on top -- the kernel's gradients against the float64 adjoint of the kernel's OWN trajectory with its
    # decisions forced (tests/helpers.own_trajectory_check): no chaos, every env.  Bar: 1e-2 per env, or no worse than twice the plain
    # fp32 evaluation of the same adjoint on the same trajectory (violent cases sit on the +-500 N clamps and the Coulomb switch, which
    # float64 re-decides); every touching candidate the restated decision finds must be in the kernel's hit log
    from helpers import own_trajectory_check
    chk = own_trajectory_check(dm, tpl, inp, dev, abs_floor=1e-8)   # (seeds are ~1e-3: a gradient below 1e-8 is zero for every purpose)
    own_w = chk["worst"]
    own_ok = chk["hitlog_missing"] == 0 and bool(np.all((own_w < 1e-2) | (own_w < 2.0 * chk["fp32_atan2"]) | (chk["coulomb"] < 1e-3) | (chk["force_clamp"] < 2e-2)))
    if not own_ok:
        ok = False
        note += " OWN-TRAJECTORY worst %.1e (fp32 %.1e) missing %d" % (own_w.max(), chk["fp32_atan2"].max(), chk["hitlog_missing"])
    note += " own %.0e" % own_w.max()
    bad += 0 if ok else 1
    if not ok:  # keep the failing case for offline inspection
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "stress_fail_%d.npz" % case), name=name, segw=segw, kind=kind, pen=pen,
                            **{"in_" + k: np.asarray(v) for k, v in inp.items()}, **{"gpu_" + k: v for k, v in o1.items()},
                            **{"ref_" + k: np.asarray(v) for k, v in st.items() if hasattr(v, "shape")},
                            **{"refg_" + k: np.asarray(v) for k, v in gr.items()})
    print("%-4s %-8s bs=%-3d T=%-3d spf=%-2d segw=%-2d %-5s  pos %.1e vel %.1e grf %.1e g_q %.1e g_refs %.1e g_m %.1e  repeat-%s %s" % (
        "ok" if ok else "FAIL", name, bs, T, spf, segw, kind, e["pos"], e["vel"], e["grf"], e["g_q"], e["g_refs"], e["g_m"],
        "same" if same else "DIFFERS", ("" if finite else "NONFINITE") + " pen=%g " % pen + note), flush=True)
print("stress: %d cases, %d failures" % (ncases, bad))
sys.exit(1 if bad else 0)
