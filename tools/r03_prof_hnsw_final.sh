# HNSW at 1M x 256 with the round-3 walk kernel (hybrid candidate queue, integer keys) and the all-device builder:
# kernel trace + bench lines; then the 10M build + walk.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03h && O=gpurun_out/r03h && \
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 tools/hnsw_bench.py --steps 2 --cpu-queries 32 > $O/hnsw_1m_kt.jsonl 2> $O/kt.err && \
python3 tools/hnsw_bench.py --vectors 10000000 --cpu-queries 16 > $O/hnsw_10m.jsonl 2> $O/10m.err; \
find $O -name '*kernel_stats.csv' | xargs -I{} sh -c 'echo {}; head -8 {}'; tail -2 $O/10m.err; python3 tools/show_bench.py $O/hnsw_10m.jsonl 2>/dev/null | head -5
