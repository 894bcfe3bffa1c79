#!/bin/bash
# round 6: the profile set of the committed tree -> gpurun_out/r06 (copied to profiles/r06)
TAG=${1:-r06a}
bash tools/gpu_prof.sh $TAG r06
bash tools/gpu_flavours.sh $TAG r06 > gpurun_out/r06/${TAG}_flavours_log.txt 2>&1; tail -20 gpurun_out/r06/flavours.txt | cut -c1-230
