#!/bin/bash
# short evidence run (when the round's GPU budget is nearly spent): whole GPU suite, smoke, the bench lines c2 (default command,
# live PMC + CPU baseline), c3, b1, one stream, and the one-stream rocprofv3 kernel stats
set -u
TAG=${1:-r3lite}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
O=gpurun_out/$TAG
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu tests exit: $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.log
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $O/bench_c3.json 2> $O/bench_c3.err
timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_b1.json 2> $O/bench_b1.err
timeout 300 python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/bench_s1.json 2> $O/bench_s1.err
for f in c2 c3 b1 s1; do python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$f.json') if l.startswith('{')][-1])
    print('$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'], d['config']['sampler_loop'], d['device']['clocks_during_timed_region'].get('sclk_mhz_avg'))
    for r in d['roofline']['classes']: print('   ', r['class'][:44], round(r['ms_per_pass'],2), 'ms', round(r['avg_launch_us'],1),'us', round(r['frac_of_mfma_peak'],4), r['launches_per_pass'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stream1 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --no-parity-check --streams 1 > $R/$O/stream1.log 2>&1
cd $R
f=$(find $O/stream1 -name "*kernel_stats.csv" | head -1)
cp "$f" $O/stream1_kernel_stats.csv
python tools/prof_summary.py $O/stream1_kernel_stats.csv 2 16
find $O/stream1 -name "*kernel_trace.csv" -delete
