#!/bin/bash
# kernel trace + PMC passes of the default bench, the trace of the driver's exact command, detection timeline, the two bench
# lines with their CPU baseline.  Summaries -> gpurun_out/r05/ (copied to profiles/r05/).
TAG=${1:-r06a}; RD=${2:-r06}
out=gpurun_out/$RD; mkdir -p $out
bash tools/prof.sh $TAG $RD trace sqA sqB lds fetch write > $out/${TAG}_log.txt 2>&1
tail -30 $out/${TAG}_log.txt | cut -c1-220
BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline" bash tools/prof.sh ${TAG}_driver_shape $RD trace > $out/${TAG}_driver_shape_log.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tl && R=$GRAFT_REPO_ROOT && python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --state-cache /tmp/bed_tl.npz > /dev/null 2>&1 && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o p -- python $R/bench.py --steps 80 --warmup 10 --no-cpu-baseline --state-cache /tmp/bed_tl.npz > /tmp/tl.json 2>/tmp/tl.err; python $R/profiles/timeline.py $(find /tmp/prof_tl -name 'p_kernel_trace.csv' | head -1) $R/$out/${TAG}_detection_timeline.txt )
python bench.py --steps 20 --warmup 5 > $out/${TAG}_bench_driver_shape.json 2>/dev/null
python bench.py > $out/${TAG}_bench_default.json 2>/dev/null
python - $out/${TAG}_bench_driver_shape.json $out/${TAG}_bench_default.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().split('\n')[-1]); k=d['kernels_ms']; r=d['roofline']
    print(f"{f.split('/')[-1]:40s} step {d['ms_per_step']:.4f} ms value {d['value']:.3e} force {k['calc_forces']:.4f} integ {k['integrate']:.4f} det {k['detect_update']:.3f} frac {r['frac']:.3f} copy {r.get('attainable_copy_GBs')} cpu {d['cpu_baseline']['value']:.3e}")
PY
