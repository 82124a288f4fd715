#!/bin/bash
# The reference's run.sh (five training runs of the Laikago imitation task, /root/reference/run.sh:9-14), on the HIP rollout.
# The reference starts an X server first (its renderer); this package has no renderer, so the five command lines are all there is.
#   bash run.sh [extra main.py flags ...]
set -e
cd "$(dirname "$0")"
rm -rf logdir/mi-*
for seq in mi-spin mi-trot mi-pace mi-sidesteps mi-turn; do
  HIP_VISIBLE_DEVICES=${HIP_VISIBLE_DEVICES:-0} python main.py --urdf_template laikago --seqname $seq --logname 0 "$@"
done
