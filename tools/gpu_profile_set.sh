#!/bin/bash
# the profile set of the committed tree -> gpurun_out/<round dir> (copied to profiles/<round dir>)     usage: gpu_profile_set.sh [tag] [round dir]
TAG=${1:-r06a}; RD=${2:-r06}
bash tools/gpu_prof.sh $TAG $RD
bash tools/gpu_flavours.sh $TAG $RD > gpurun_out/$RD/${TAG}_flavours_log.txt 2>&1; tail -20 gpurun_out/$RD/flavours.txt | cut -c1-230
