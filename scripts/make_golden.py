#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the float64 torch oracle (autograd gradients).

SELF-GENERATED, NOT WARP: the reference has no tests or golden vectors and Warp cannot be run here
(SURVEY.md section 8(c)); these fixtures pin the build's own restatement so that the C oracle, the
HIP kernels and future refactors are all held to one set of numbers.  Inputs come from
diffphys_amd.synth (seeded), perturbed so that every gradient path is exercised.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
from diffphys_amd import robots, synth  # noqa: E402
from oracle import ref_torch as rt  # noqa: E402


def golden_inputs(tpl, name, bs, nsteps, seed):
    inp = synth.make_inputs(tpl, name, bs=bs, nsteps=nsteps, seed=seed, steps_per_frame=11, dtype=np.float64, penetration=0.003)
    rng = np.random.RandomState(seed + 100)
    inp["torques"] = rng.randn(*inp["torques"].shape) * 0.5
    inp["res_f"] = rng.randn(*inp["res_f"].shape) * 0.5
    inp["qd_init"] = rng.randn(*inp["qd_init"].shape) * 0.1
    # fixtures are float32-representable so fp32 implementations see identical inputs
    for k in synth.INPUT_NAMES + ("adj_pos", "adj_vel"):
        inp[k] = inp[k].astype(np.float32).astype(np.float64)
    return inp


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, bs, nsteps in (("laikago", 2, 34), ("human", 2, 34), ("quad", 2, 34)):
        tpl = robots.load_template(name)
        inp = golden_inputs(tpl, name, bs, nsteps, seed=11)
        T = rt.Template(tpl, torch.float64)
        tin = {k: torch.tensor(inp[k], dtype=torch.float64, requires_grad=True) for k in synth.INPUT_NAMES}
        pos, vel, grf, jaf = rt.rollout(T, *[tin[k] for k in synth.INPUT_NAMES], nsteps=nsteps,
                                        frame2step=inp["frame2step"], dt=inp["dt"])
        loss = (pos * torch.tensor(inp["adj_pos"])).sum() + (vel * torch.tensor(inp["adj_vel"])).sum()
        loss.backward()
        d = {"in_" + k: inp[k].astype(np.float32) for k in synth.INPUT_NAMES + ("adj_pos", "adj_vel")}
        d.update(frame2step=np.asarray(inp["frame2step"], np.int32), nsteps=np.int32(nsteps), dt=np.float64(inp["dt"]),
                 bs=np.int32(bs), wp_pos=pos.detach().numpy(), wp_vel=vel.detach().numpy(), grf=grf.detach().numpy(),
                 jaf=jaf.detach().numpy())
        for k in synth.INPUT_NAMES:
            g = tin[k].grad
            d["grad_" + k] = np.zeros_like(inp[k]) if g is None else g.numpy()
        # FK fixture: 3 articulations
        rng = np.random.RandomState(5)
        jq = np.tile(inp["q_init"].reshape(bs, -1)[:1], (3, 1)) + rng.randn(3, T.nq) * 0.05
        jqd = rng.randn(3, T.nqd) * 0.3
        jq, jqd = jq.astype(np.float32).astype(np.float64), jqd.astype(np.float32).astype(np.float64)
        tq, tqd = torch.tensor(jq, requires_grad=True), torch.tensor(jqd, requires_grad=True)
        bq, bqd = rt.eval_fk(T, tq, tqd)
        aq, aqd = rng.randn(3, T.nb, 7), rng.randn(3, T.nb, 6)
        ((bq * torch.tensor(aq)).sum() + (bqd * torch.tensor(aqd)).sum()).backward()
        d.update(fk_joint_q=jq.astype(np.float32), fk_joint_qd=jqd.astype(np.float32), fk_body_q=bq.detach().numpy(),
                 fk_body_qd=bqd.detach().numpy(), fk_adj_q=aq.astype(np.float32), fk_adj_qd=aqd.astype(np.float32),
                 fk_grad_q=tq.grad.numpy(), fk_grad_qd=tqd.grad.numpy())
        # adjoint seeds must be float32-exact too
        d["fk_grad_note"] = np.asarray("gradients of sum(body_q*fk_adj_q) + sum(body_qd*fk_adj_qd) with float32-rounded seeds")
        # recompute FK grads with the rounded seeds so fp32 implementations use identical seeds
        tq2, tqd2 = torch.tensor(jq, requires_grad=True), torch.tensor(jqd, requires_grad=True)
        bq2, bqd2 = rt.eval_fk(T, tq2, tqd2)
        ((bq2 * torch.tensor(d["fk_adj_q"].astype(np.float64))).sum() + (bqd2 * torch.tensor(d["fk_adj_qd"].astype(np.float64))).sum()).backward()
        d["fk_grad_q"], d["fk_grad_qd"] = tq2.grad.numpy(), tqd2.grad.numpy()
        np.savez_compressed(os.path.join(out, "rollout_%s.npz" % name), **d)
        print(name, "pos", pos.shape, "max|grad q_init|", np.abs(d["grad_q_init"]).max())


if __name__ == "__main__":
    main()
