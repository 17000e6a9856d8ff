#!/bin/bash
# A/B on one box: the un-dilated 7-tap convs on the 16x16x32 form (KX_DA_S16=1, default) against the 2 x 2-wave 32x32x16
# form (KX_DA_S16=2), two rounds, with the per-shape table.
cd $GRAFT_REPO_ROOT
exec bash tools/ab_env.sh ${1:-r04_s16k7} "KX_DA_S16=2" "KX_DA_S16=1"
