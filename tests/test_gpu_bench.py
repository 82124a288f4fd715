"""The bench line itself (the driver parses it): one short run of bench.py on the GPU, checked for the contract's fields and for
internal consistency -- value against ms_per_step, the roofline fraction against the launch duration, the VALU-issue roofline
below its peak."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract_and_consistency():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--repeats", "3", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "boundary", "blocks"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["unit"] == "env-steps/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["global_batch"] == 4096 and d["config"]["sim_steps"] == 100 and d["config"]["nan_grads"] == 0
    # value = env-steps of one step / the median block's time per step
    assert abs(d["value"] - 4096 * 100 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["blocks"]["n"] == 3 and d["blocks"]["min_value"] <= d["value"] <= d["blocks"]["max_value"]
    r = d["roofline"]
    # "bound" names what binds; achieved / peak / unit / frac stay the contract's HBM figures, hbm_frac repeats frac
    assert r["bound"] == "valu-issue" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["hbm_frac"] == r["frac"]
    # achieved = algorithmic bytes per launch / the kernel's launch duration, measured in this run
    assert abs(r["achieved"] * 1e9 - 4096 * 100 * r["algorithmic_bytes_per_env_step"] / (r["avg_launch_ms"] * 1e-3)) < 1e-6 * r["achieved"] * 1e9
    assert 0.05 < r["frac"] < 1.0
    # the dominant kernel cannot take longer than the whole step; forward + adjoint launches are about the step
    f = r["fwd_kernel"]
    assert r["avg_launch_ms"] < d["ms_per_step"] and 0.8 * d["ms_per_step"] < r["avg_launch_ms"] + f["avg_launch_ms"] < 1.2 * d["ms_per_step"]
    # the counter-derived fields come from the committed profile ONLY when it was taken on the library that runs (a kernel
    # change without a re-profile must not carry stale counters into the line): profiles/pmc_summary.json names the library's source hash
    sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
    from diffphys_amd import hip_backend

    with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as fh:
        prof_hash = json.load(fh).get("source_hash")
    if prof_hash == hip_backend.build_id().split("+")[-1]:
        # the roofline that binds (fp32 VALU issue): below its peak, and the adjoint close to it
        assert r["valu"] is not None and 0.5 < r["valu"]["frac"] < 1.02, r["valu"]
        assert f["valu"] is not None and 0.2 < f["valu"]["frac"] < r["valu"]["frac"]
        # ... and against the data-sheet vector peak (157.3 TFLOP/s, packed FMAs only): instructions x 128 flop / time
        v = r["valu"]
        assert abs(v["frac_of_vector_peak"] - v["insts_per_launch"] * 128 / (r["avg_launch_ms"] * 1e-3) / 157.3e12) < 1e-9 and 0.2 < v["frac_of_vector_peak"] < 0.6
        assert "profiles/r06_micro_valu.txt" in v["peak_source"] and abs(v["frac_of_guide_rate"] - v["frac"] * 2.0 / v["cycles_per_instruction"]) < 1e-9
        assert r["traffic"] is not None and prof_hash in r["traffic_source"]
    else:
        assert r["valu"] is None and f["valu"] is None and r["traffic"] is None and "another build" in r["traffic_source"], r["traffic_source"]
        assert r["secondary"]["valu_busy"] is None and r["secondary"]["wave_wait_share"] is None
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["envs"] == 4096 and d["per_rank"][0]["fwd_family"].startswith("lane per body: 4 envs per wave, 3 waves")   # body, contact and cull wave
    assert d["per_rank"][0]["bwd_family"].startswith("lane per body: 4 envs per wave, 2 waves")
    assert abs(d["per_rank"][0]["ms_per_step"] - d["ms_per_step"]) < 1e-9   # one rank: its own clock IS the line's
    assert d["collective_backend"] is None and d["ranks_seen"] == 1 and d["launcher"] == "none"
    assert len(d["devices_seen"]) == 1 and d["devices_distinct"] == 1 and d["devices_seen"][0]["index"] == 0
    # the HIP-event pass against the timed region: reported, and the two agree within 15 %
    assert 0.9 < r["event_pass_over_timed"] < 1.15 and abs(r["pair_frac_timed"] - 4096 * 100 * 2720 / (d["ms_per_step"] * 1e-3) / 8e12) < 1e-9
    b = d["boundary"]
    assert 0.8 < b["ratio_to_value"] < 1.1 and b["nan_grads"] == 0


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` as the driver types it for N = 1 (no launcher): the parent starts two rank processes itself.  On this
    1-GPU box the ranks share the device (PPR_BENCH_SHARE_GPU test hook; RCCL refuses two ranks on one device, so the barrier falls back
    to gloo AND the line says so): the multi-rank code path -- sharding, barrier, MAX over ranks, both scaling modes -- runs for real."""
    # (the child ranks are pinned to ONE device -- the first this process may see -- so that the test means the same on a multi-GPU host: ADVICE r5)
    first = (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or "0").split(",")[0]
    one_gpu = dict(os.environ, HIP_VISIBLE_DEVICES=first)
    one_gpu.pop("CUDA_VISIBLE_DEVICES", None)
    env = dict(one_gpu, PPR_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "3"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["launcher"] == "self" and d["collective_backend"] in ("nccl", "gloo")
    # both ranks say which GPU they ran on; here (test hook) it is the same one, and the line shows that -- without the hook the run aborts
    assert len(d["devices_seen"]) == 2 and d["devices_distinct"] == 1 and d["devices_seen"][0] == d["devices_seen"][1]
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 4096 and d["config"]["envs_per_gpu"] == 2048
    assert abs(d["value"] - 4096 * 100 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    o = d["other_scaling"]
    assert o["scaling"] == "weak" and o["global_batch"] == 8192 and o["envs_per_gpu"] == 4096
    assert "cpu_baseline" not in d and "boundary" not in d   # N = 1 only
    # two ranks that would share a GPU WITHOUT the test hook: refused, and quickly (a dead rank no longer leaves the other in the rendezvous)
    env2 = {k: v for k, v in one_gpu.items() if k != "PPR_BENCH_SHARE_GPU"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "2"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300, env=env2)
    assert out.returncode != 0 and "no GPU of its own" in out.stderr
