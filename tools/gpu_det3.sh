#!/bin/bash
mkdir -p gpurun_out/det3
bash tools/prof.sh d3 det3 sqA sqB > gpurun_out/det3/log.txt 2>&1
grep -E "^kernel|k_sweep|k_sphere_prep|k_history" gpurun_out/det3/d3_sqA_pmc.txt | cut -c1-300
grep -E "^kernel|k_sweep|k_sphere_prep|k_history" gpurun_out/det3/d3_sqB_pmc.txt | cut -c1-300
