#!/bin/bash
# Lean evidence run of a round (about twelve GPU-minutes): whole GPU suite, smoke, the default bench line (fp32mf `value` + `fp32_direct` + bf16x3 `split`, live PMC,
# CPU baseline), one clip / one stream lines, rocprofv3 kernel stats of the one-stream and the default command, SQ counter pass.
#   gpurun --timeout 1500 -- 'bash tools/gpu_evidence_lean.sh r05_final'      -> gpurun_out/<tag>/; copy what is quoted to profiles/
set -u
TAG=${1:-ev}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/gpu_tests.log 2>&1
echo "gpu tests exit: $?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $O/smoke.log
# (the stdout line of every run is the <= 4 KB driver line -> bench_<f>.line.json; the tables are in the side file bench_<f>.json)
timeout 600 python bench.py --detail $O/bench_c2.json > $O/bench_c2.line.json 2> $O/bench_c2.err
timeout 300 python bench.py --batch 1 --streams 1 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --detail $O/bench_b1.json > $O/bench_b1.line.json 2> $O/bench_b1.err
timeout 300 python bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-isolated --detail $O/bench_s1.json > $O/bench_s1.line.json 2> $O/bench_s1.err
if [ "${2:-}" = "all" ]; then
  timeout 400 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --detail $O/bench_c3.json > $O/bench_c3.line.json 2> $O/bench_c3.err
  timeout 400 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --detail $O/bench_c5.json > $O/bench_c5.line.json 2> $O/bench_c5.err
fi
for f in c2 b1 s1 c3 c5; do [ -f $O/bench_$f.json ] || continue; python - <<PY
import json
try:
    line=[l for l in open('$O/bench_$f.line.json') if l.startswith('{')][-1]
    assert len(line) <= 4097, len(line)
    d=json.load(open('$O/bench_$f.json'))
    sp=d.get('split') or {}
    pr=d['roofline'].get('path_roofline') or {}
    fd=d.get('fp32_direct') or {}
    print('$f', 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4), 'path', pr.get('frac') and round(pr['frac'],4), 'parity', d['parity_check'] and d['parity_check']['ok'],
          '| fp32_direct', fd and round(fd['value'],1), fd and fd['parity_check'] and fd['parity_check']['ok'],
          '| split', sp and round(sp['value'],1), sp and round(sp['ms_per_step'],2), sp and sp['parity_check'] and sp['parity_check']['ok'], d['device']['clocks_during_timed_region'].get('sclk_mhz_avg'))
    for r in d['roofline']['classes']+(sp.get('classes') or []): print('   ', r['class'][:44], round(r['ms_per_pass'],2), 'ms', round(r['avg_launch_us'],1),'us', round(r['frac_of_mfma_peak'],4), r['launches_per_pass'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
cd /tmp
prof() { # name args passes
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-pmc --no-parity-check $2 > $O/$1.log 2>&1
  f=$(find $O/$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${1}_kernel_stats.csv
  python $R/tools/prof_summary.py $O/${1}_kernel_stats.csv ${3:-3} 14
  find $O/$1 -name "*kernel_trace.csv" -delete
}
prof stream1_fp32mf "--streams 1 --vocoder-precision fp32mf"
prof stream1_fp32 "--streams 1 --vocoder-precision fp32"
prof default_all "" 7
cd $R
timeout 600 python tools/conv_mf_bench.py 8 > $O/mf_layer_bench.txt 2>&1; tail -4 $O/mf_layer_bench.txt
bash tools/gpu_sq_pmc.sh $TAG --vocoder-precision fp32mf 2>&1 | tail -40
