#!/usr/bin/env python3
"""Compile the reference's URDF robots and AMP mocap files into small npz fixtures.

Run HERE (the container with /root/reference); outputs are committed:
  ppr-diffphys_amd/diffphys_amd/templates/{laikago,human,quad}.npz   articulation templates
  ppr-diffphys_amd/diffphys_amd/templates/mocap_laikago.npz          AMP frames of the 5 sequences

The npz files hold DATA only (flat arrays derived from the URDF / mesh / json
data files under /root/reference/data, Laikago meshes: PyBullet/Unitree
licence, see data/urdf_templates/laikago/license.txt there).  Nothing on the
GPU box reads /root/reference.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ppr-diffphys_amd"))
from diffphys_amd import robots  # noqa: E402

REF = os.environ.get("PPR_REFERENCE", "/root/reference")


def main():
    out = robots.TEMPLATE_DIR
    os.makedirs(out, exist_ok=True)
    for name in ("laikago", "human", "quad"):
        env, art, info = robots.make_env(name, os.path.join(REF, "data/urdf_templates"), 1, device="cpu")
        tpl = env.template()
        tpl["body_names"] = np.asarray(info["body_names"])
        tpl["kp"] = np.float32(info["kp"])
        tpl["kd"] = np.float32(info["kd"])
        tpl["mass_rule"] = np.asarray(info["mass_rule"])  # which reading of dp_model.py:185-191 produced body_mass (robots.MASS_RULES)
        np.savez_compressed(os.path.join(out, name + ".npz"), **tpl)
        print(
            "%-8s nb=%d nq=%d nqd=%d Nc=%d mass=%s"
            % (name, tpl["nb"], tpl["nq"], tpl["nqd"], len(tpl["contact_body"]), np.round(tpl["body_mass"], 3)[:6])
        )
    mocap = {}
    for seq in ("mi-pace", "mi-trot", "mi-spin", "mi-turn", "mi-sidesteps"):
        with open(os.path.join(REF, "data/motion_sequences/%s/amp-%s.txt" % (seq, seq))) as f:
            d = json.load(f)
        mocap[seq + "/frames"] = np.asarray(d["Frames"], dtype=np.float64)
        mocap[seq + "/frame_duration"] = np.float64(d["FrameDuration"])
        print(seq, mocap[seq + "/frames"].shape, d["FrameDuration"])
    np.savez_compressed(os.path.join(out, "mocap_laikago.npz"), **mocap)


if __name__ == "__main__":
    main()
