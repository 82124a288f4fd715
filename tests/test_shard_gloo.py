"""Multi-GPU path on CPU: world_size-2 gloo, driving the SAME input-building function bench.py uses (bench.rank_inputs).

The rollout shards by contiguous env slices with no data-path collective.  Here two ranks each
build their slice of the global batch in both scaling modes ("weak": bs envs per rank, "strong": bs envs in total), roll
it out (with the C oracle -- there is no GPU here; the same check on the HIP library is
tests/test_gpu_tight.py::test_two_shards_on_two_streams_equal_the_global_batch), gather, and rank 0 compares with the
single-process global batch bit for bit.  The bench's timing reduction (barrier + MAX over ranks) runs on the same group."""
import os
import sys

import numpy as np
import pytest
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
    from diffphys_amd import robots
    from oracle.ref_c import RefC

    tpl = robots.load_template("laikago")
    T, nb = 12, 13
    rc = RefC(tpl, np.float32)
    res = []
    for scaling, bs_arg in (("strong", 7), ("weak", 3)):
        inp, (lo, hi), gbs = bench.rank_inputs(tpl, "laikago", T, world, rank, scaling, bs_arg, ("mi-trot", "mi-spin"), seed=5)
        assert hi - lo == inp["q_init"].size // 19 and gbs == (7 if scaling == "strong" else 6)
        st = rc.rollout_forward(inp, T, inp["frame2step"], inp["dt"])
        g = rc.rollout_backward(st, inp["adj_pos"], inp["adj_vel"])
        F = len(inp["frame2step"])
        mine = [st["wp_pos"].reshape(F, hi - lo, nb * 7).transpose(1, 0, 2).copy(), g["q_init"].reshape(hi - lo, 19).copy()]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)  # host-side concatenation of outputs; NOT on the measured path
        if rank == 0:
            full_in, span, _ = bench.rank_inputs(tpl, "laikago", T, 1, 0, "strong", gbs, ("mi-trot", "mi-spin"), seed=5)
            assert span == (0, gbs)
            fs = rc.rollout_forward(full_in, T, full_in["frame2step"], full_in["dt"])
            fg = rc.rollout_backward(fs, full_in["adj_pos"], full_in["adj_vel"])
            pos = np.concatenate([p[0] for p in gathered], 0)
            gq = np.concatenate([p[1] for p in gathered], 0)
            res.append(float(np.abs(pos - fs["wp_pos"].reshape(F, gbs, nb * 7).transpose(1, 0, 2)).max()))
            res.append(float(np.abs(gq - fg["q_init"].reshape(gbs, 19)).max()))
    dist.barrier()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out, np.array(res + [float(t.item())]))
    dist.destroy_process_group()


def test_two_rank_shards_reproduce_global_batch(tmp_path, oracle_libs):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, 29731, out), nprocs=2, join=True)
    r = np.load(out)
    assert np.all(r[:4] == 0.0)      # envs are independent: sharding changes nothing, bit for bit (both modes)
    assert abs(r[4] - 0.2) < 1e-12   # MAX over ranks of the elapsed time


def test_shard_envs_partitions():
    import bench

    for gbs in (1, 7, 4096, 4099):
        for world in (1, 2, 4, 8):
            spans = [bench.shard_envs(gbs, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gbs
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert bench.shard_envs(4096, 8, 3) == (1536, 2048)   # 512 envs per GPU at 8 GPUs
    f, b = bench.algorithmic_bytes(13, 18)
    assert f + b == 2720  # SURVEY.md section 8(d) canonical figure for Laikago


def test_bench_gpus_n_self_launches_without_a_launcher():
    """`python bench.py --gpus 2` typed as the driver types N = 1 must start its own ranks.  On this GPU-less box both
    children must get as far as the product's "needs a GPU" refusal -- not the old "must be launched with torch.distributed.run"."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_bench.py covers the self-launch there")
    assert out.returncode != 0
    assert out.stderr.count("needs a GPU") == 2, out.stderr[-2000:]
    assert "ranks failed" in out.stderr and "torch.distributed.run" not in out.stderr
