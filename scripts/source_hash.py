#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the sources of libpprdiffphys_hip.so, in the order of csrc/Makefile's SRCS -- the second half
of pd_build_id().  `python scripts/source_hash.py` prints it; diffphys_amd.hip_backend.source_hash() is the same function."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ppr-diffphys_amd", "csrc")
SRCS = ["pd_kernels.hip", "pd_host.hip", "pd_loss.hip", "pd_pose.hip", "pd_mlp.hip", "pd_math.h", "pd_device.h", "pd_args.h", "pd_se3.h", "pd_quad.h", "pd_trajloss.h", "../../include/ppr_diffphys.h", "Makefile"]


def source_hash(csrc=CSRC):
    h = hashlib.sha256()
    for f in SRCS:
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
