# Round-3 evidence for the headline workload (BASELINE configs[2]; the timed loop rotates 8 distinct prepared batches):
# bench lines, rocprofv3 kernel traces of the same command (two batches in flight, one at a time), PMC traffic passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03final && O=gpurun_out/r03final && \
python bench.py > $O/bench.json 2> $O/bench.err && \
python bench.py --alg logcosine --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 > $O/bench_logcosine.json 2> $O/e1 && \
python bench.py --alg dot --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 > $O/bench_dot.json 2> $O/e2 && \
python bench.py --tweets 1000000 --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 > $O/bench_1M.json 2> $O/e3 && \
python bench.py --exercise-exchange --no-cpu-baseline --e2e-steps 0 > $O/bench_ex.json 2> $O/e4 && \
echo benches-done && \
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 --check-queries 0 --quality-queries 0 --steps 40 > $O/bench_kt.json 2> $O/kt.err && \
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 --check-queries 0 --quality-queries 0 --steps 40 --inflight 1 > $O/bench_kt1.json 2> $O/kt1.err && \
echo traces-done && \
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 --check-queries 0 --quality-queries 0 --inflight 1 --steps 16 --warmup 8 > $O/pf.json 2> $O/pf.err && \
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w -o w --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 --check-queries 0 --quality-queries 0 --inflight 1 --steps 16 --warmup 8 > $O/pw.json 2> $O/pw.err && \
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_f1 -o f1 --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 --check-queries 0 --quality-queries 0 --inflight 1 --rotate 1 --steps 16 --warmup 8 > $O/pf1.json 2> $O/pf1.err && \
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o sq --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --mb-requests 0 --quality-topical-tweets 0 --check-queries 0 --quality-queries 0 --inflight 1 --steps 8 --warmup 2 > $O/ps.json 2> $O/ps.err && \
echo pmc-done; \
python3 tools/pmc_summary.py $O/pmc_f $O/pmc_w $O/pmc_f1 $O/pmc_sq > $O/pmc_summary.txt 2>&1; grep -i "unit_fast\|merge_kernel\|desc_query\|^#" $O/pmc_summary.txt; \
python tools/show_bench.py $O/bench*.json; head -6 $O/kt/kt_kernel_stats.csv | cut -c1-200; head -6 $O/kt1/kt1_kernel_stats.csv | cut -c1-200
