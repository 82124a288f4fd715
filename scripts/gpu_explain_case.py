import sys, os
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import first_branch_difference, grad_env_errors, oracle_bundle, GRAD_LEAD
from test_gpu_parity import gpu_rollout
from diffphys_amd import hip_backend, robots
z=np.load(sys.argv[1], allow_pickle=True)
name=str(z["name"]); tpl=robots.load_template(name)
inp={k[3:]: z[k] for k in z.files if k.startswith("in_")}
for k in ("nsteps",): inp[k]=int(inp[k])
inp["dt"]=float(inp["dt"]); inp["frame2step"]=[int(x) for x in inp["frame2step"]]
bs=inp["q_init"].size//int(tpl["nq"])
dm=hip_backend.DeviceModel(tpl)
out=gpu_rollout(dm, inp, torch.device("cuda:0"), keep_traj=True)
ob=oracle_bundle(tpl, inp, bs)
e=grad_env_errors(out["grads"], ob["g64"], bs)
w=np.max(np.stack([e[k] for k in GRAD_LEAD]),0)
first=first_branch_difference(ob["rc64"], ob["st64"], out["traj"], inp, bs)
bad=np.argsort(-w)[:5]
for i in bad: print("env %d err %.2e cond %.2e first-branch-difference step %d (T=%d)"%(i,w[i],ob["cond"][i],first[i],inp["nsteps"]))
print("envs above 1e-3:", (w>1e-3).sum(), "unexplained:", ((w>np.maximum(30*ob["cond"],1e-3))&(first>=inp["nsteps"])).sum())
