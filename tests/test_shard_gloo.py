"""Multi-GPU path on CPU: world_size-2 gloo.  The rollout shards by env slice with no data-path collective
(SURVEY.md section 8(e)); here two ranks each roll out their slice with the C oracle and the concatenation
must equal the single-process result bit for bit, and the bench's timing reduction (MAX over ranks) works."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from diffphys_amd import robots, synth
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    gbs, T = 6, 12
    inp = synth.make_inputs(tpl, "laikago", bs=gbs, nsteps=T, seed=5, steps_per_frame=5, penetration=0.002)
    lo, hi = bench.shard_envs(gbs, world, rank)
    nb, nq, nqd = 13, 19, 18

    def sl(a, per, lead):
        return np.ascontiguousarray(a.reshape(lead + (gbs, per))[..., lo:hi, :].reshape(lead + ((hi - lo) * per,)))

    loc = dict(q_init=sl(inp["q_init"], nq, ()), qd_init=sl(inp["qd_init"], nqd, ()), torques=sl(inp["torques"], nqd, (T,)),
               refs=sl(inp["refs"], nqd, (T,)), res_f=np.ascontiguousarray(inp["res_f"].reshape(T, gbs, nb, 6)[:, lo:hi].reshape(T, -1, 6)),
               target_ke=sl(inp["target_ke"], nqd, ()), target_kd=sl(inp["target_kd"], nqd, ()),
               body_inv_mass=sl(inp["body_inv_mass"], nb, ()), body_mass=sl(inp["body_mass"], nb, ()),
               body_inertia=np.ascontiguousarray(inp["body_inertia"].reshape(gbs, nb, 3, 3)[lo:hi].reshape(-1, 3, 3)),
               body_inv_inertia=np.ascontiguousarray(inp["body_inv_inertia"].reshape(gbs, nb, 3, 3)[lo:hi].reshape(-1, 3, 3)))
    rc = RefC(tpl, np.float32)
    st = rc.rollout_forward(loc, T, inp["frame2step"], inp["dt"])
    pos = torch.from_numpy(st["wp_pos"].reshape(len(inp["frame2step"]), hi - lo, nb * 7).copy())
    gathered = [torch.zeros_like(pos) for _ in range(world)]
    dist.all_gather(gathered, pos)  # host-side concatenation of outputs; NOT on the measured path
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        full = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])["wp_pos"].reshape(len(inp["frame2step"]), gbs, nb * 7)
        np.save(out, np.array([float(np.abs(torch.cat(gathered, 1).numpy() - full).max()), float(t.item())]))
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_global_batch(tmp_path, oracle_libs):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, 29731, out), nprocs=2, join=True)
    err, tmax = np.load(out)
    assert err == 0.0          # envs are independent: sharding changes nothing, bit for bit
    assert abs(tmax - 0.2) < 1e-12


def test_shard_envs_partitions():
    import bench

    for gbs in (1, 7, 4096, 4099):
        for world in (1, 2, 4, 8):
            spans = [bench.shard_envs(gbs, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gbs
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    f, b = bench.algorithmic_bytes(13, 18)
    assert f + b == 2720  # SURVEY.md section 8(d) canonical figure for Laikago
