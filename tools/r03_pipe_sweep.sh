# unit kernel forms side by side on the headline workload: SANN_PIPE=0 (one unit per workgroup) and the pipelined kernel at
# 3 / 4 / 5 workgroups per CU; then the duplicate-heavy 1M-tweet corpus and the other algorithms.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pipe && O=gpurun_out/pipe
B="python bench.py --no-cpu-baseline --e2e-steps 0 --check-queries 8 --quality-queries 0 --steps 48 --warmup 8"
SANN_PIPE=0 $B > $O/fast.json 2> $O/fast.err || exit 1
for w in 3 4 5; do SANN_PIPE_WGS=$w $B > $O/pipe$w.json 2> $O/pipe$w.err || exit 1; done
SANN_PIPE=0 $B --tweets 1000000 > $O/fast_1m.json 2> $O/e1 || exit 1
$B --tweets 1000000 > $O/pipe_1m.json 2> $O/e2 || exit 1
SANN_PIPE=0 $B --alg logcosine > $O/fast_log.json 2> $O/e3 || exit 1
$B --alg logcosine > $O/pipe_log.json 2> $O/e4 || exit 1
SANN_PIPE=0 $B --alg dot > $O/fast_dot.json 2> $O/e5 || exit 1
$B --alg dot > $O/pipe_dot.json 2> $O/e6 || exit 1
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/pipe/*.json')):
    d=json.load(open(p)); r=d['roofline']
    print(p.split('/')[-1], 'ms/step %.4f unit %.1f us alone %.1f us merge %.1f desc %.1f frac %.3f alone %.3f parity %s fb %d' % (d['ms_per_step'], r['kernel_avg_ms']*1e3, r['kernel_avg_ms_alone']*1e3, r['merge_kernel_avg_ms']*1e3, r['desc_kernel_avg_ms']*1e3, r['frac'], r['frac_alone'], d['recall_at_400_parity'], d['fallback_units']))
PY
