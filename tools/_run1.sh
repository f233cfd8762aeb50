cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_c3_gpu.py tests/test_golden_gpu.py tests/test_shard_gpu.py tests/test_stream_gpu.py tests/test_oracle_cpu.py -m gpu -x -q -s > $O/pytest_new.txt 2>&1; echo "pytest rc $?" >> $O/pytest_new.txt
timeout 300 python -m pytest tests/test_blocks_gpu.py -m gpu -x -q > $O/pytest_blocks.txt 2>&1; echo "pytest rc $?" >> $O/pytest_blocks.txt
timeout 60 tools/clk_lab > $O/clk_lab.txt 2>&1
timeout 300 python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --no-cpu > $O/bench_native_c3_eight_shards_one_gpu.json 2> $O/bench_native_c3.err
timeout 300 python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --workload fmchain --no-cpu > $O/bench_native_c3_eight_shards_one_gpu_fmchain.json 2>> $O/bench_native_c3.err
PCX_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --shard 8388608 --steps 50 --warmup 10 --no-cpu > $O/bench_eight_ranks_one_gpu_gloo.json 2> $O/bench_eight_ranks.err
tail -3 $O/pytest_new.txt; tail -2 $O/pytest_blocks.txt; cat $O/clk_lab.txt | tail -12; cat $O/bench_native_c3_eight_shards_one_gpu.json | cut -c1-400; tail -2 $O/bench_eight_ranks.err; cut -c1-300 $O/bench_eight_ranks_one_gpu_gloo.json
