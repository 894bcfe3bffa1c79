#!/bin/bash
# detection work of round 3: parity on small scenes (segmented key arena forced), full-size lists, then the kernel trace of the default bench
out=gpurun_out/r3h; mkdir -p $out
DEME_KEY_SEG_MIN=4096 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_mesh.py tests/test_fast_mode.py tests/test_fast_mode_features.py tests/test_decomp.py -x -q -m gpu > $out/seg_tests.log 2>&1; echo "rc $?" >> $out/seg_tests.log
tail -3 $out/seg_tests.log
timeout 1200 python -m pytest tests/test_full_size.py tests/test_config2_slabs.py -x -q -m gpu -k "contact_list or fast_mode or library or drifting or halo_loop or family_masks" > $out/full_tests.log 2>&1; echo "rc $?" >> $out/full_tests.log
tail -3 $out/full_tests.log
bash tools/prof.sh ${1:-det1} r3h trace
python bench.py --no-cpu-baseline --state-cache /tmp/deme_bed_${1:-det1}.npz > $out/${1:-det1}_default.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('$out/${1:-det1}_default.json').read().strip().split('\n')[-1]); k=d['kernels_ms']
print(f"step {d['ms_per_step']:.4f} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f}")
PY
