# rocprofv3 evidence for BASELINE configs[3] (C4): the dense exhaustive search at 50M x 256, k = 200, and the HNSW walk
# at 1M x 256 (k = 10 / ef = 100 and k = 200 / ef = 800).  Kernel traces first, then separate --pmc passes (never combined
# with the trace domains).  Summaries are copied into profiles/r03_* by hand from gpurun_out/r03c4/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03c4 && O=gpurun_out/r03c4 && \
rocprofv3 --kernel-trace --stats -d $O/kt_dense -o kt --output-format csv -- python3 tools/dense_bench.py --steps 3 --cpu-vectors 0 > $O/dense_kt.json 2> $O/dense_kt.err && \
echo dense-trace-done && \
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 -d $O/pmc_dense_sq -o sq --output-format csv -- python3 tools/dense_bench.py --steps 1 --warmup 0 --cpu-vectors 0 > $O/dense_sq.json 2> $O/dense_sq.err && \
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU -d $O/pmc_dense_f -o f --output-format csv -- python3 tools/dense_bench.py --steps 1 --warmup 0 --cpu-vectors 0 > $O/dense_f.json 2> $O/dense_f.err && \
echo dense-pmc-done && \
rocprofv3 --kernel-trace --stats -d $O/kt_hnsw -o kt --output-format csv -- python3 tools/hnsw_bench.py --steps 2 --cpu-queries 0 > $O/hnsw_kt.jsonl 2> $O/hnsw_kt.err && \
echo hnsw-trace-done && \
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_hnsw_sq -o sq --output-format csv -- python3 tools/hnsw_bench.py --steps 1 --cpu-queries 0 --vectors 300000 > $O/hnsw_sq.jsonl 2> $O/hnsw_sq.err && \
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_hnsw_f -o f --output-format csv -- python3 tools/hnsw_bench.py --steps 1 --cpu-queries 0 --vectors 300000 > $O/hnsw_f.jsonl 2> $O/hnsw_f.err && \
echo hnsw-pmc-done && \
python3 tools/pmc_summary.py $O/pmc_dense_sq $O/pmc_dense_f $O/pmc_hnsw_sq $O/pmc_hnsw_f > $O/pmc_summary.txt 2>&1; \
find $O -name '*kernel_stats.csv' | xargs -I{} sh -c 'echo {}; head -12 {}'
