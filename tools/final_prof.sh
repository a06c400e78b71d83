cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final && O=gpurun_out/final && \
python bench.py > $O/bench.json 2> $O/bench.err && \
python bench.py --alg logcosine --no-cpu-baseline --e2e-steps 0 > $O/bench_logcosine.json 2> $O/e1 && \
python bench.py --alg dot --no-cpu-baseline --e2e-steps 0 > $O/bench_dot.json 2> $O/e2 && \
python bench.py --tweets 1000000 --no-cpu-baseline --e2e-steps 0 > $O/bench_1M.json 2> $O/e3 && \
python bench.py --exercise-exchange --no-cpu-baseline --e2e-steps 0 > $O/bench_ex.json 2> $O/e4 && \
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --check-queries 0 --quality-queries 0 --steps 30 > $O/bench_kt.json 2> $O/kt.err && \
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt1 --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --check-queries 0 --quality-queries 0 --steps 30 --inflight 1 > $O/bench_kt1.json 2> $O/kt1.err && \
python tools/phase_prof.py > $O/phase.log 2>&1 && python tools/gather_probe.py > $O/probe.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_f -o f --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --check-queries 0 --quality-queries 0 --inflight 1 --steps 3 --warmup 1 > $O/pf.json 2> $O/pf.err && \
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_w -o w --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --check-queries 0 --quality-queries 0 --inflight 1 --steps 3 --warmup 1 > $O/pw.json 2> $O/pw.err && \
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o sq --output-format csv -- python3 bench.py --no-cpu-baseline --e2e-steps 0 --check-queries 0 --quality-queries 0 --inflight 1 --steps 3 --warmup 1 > $O/ps.json 2> $O/ps.err && \
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O/pmc_abl -o abl --output-format csv -- python3 tools/gather_probe.py > $O/probe_pmc.log 2>&1; \
python tools/show_bench.py $O/bench*.json
