# How much of the replayed batch was cache: the same bench line with the timed loop rotating 1 / 2 / 4 / 8 / 16 distinct
# prepared query batches (tools/r03_rotation_sweep.sh; summaries in profiles/r03_rotation_sweep.jsonl).
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/rot && O=gpurun_out/rot && \
for r in 1 2 4 8 16; do python bench.py --rotate $r --steps 48 --warmup 8 --no-cpu-baseline --e2e-steps 0 --check-queries 4 --quality-queries 0 > $O/rot$r.json 2> $O/rot$r.err || exit 1; done && \
python bench.py > $O/bench.json 2> $O/bench.err && python tools/show_bench.py $O/rot*.json $O/bench.json
