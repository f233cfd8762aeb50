cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for w in fft4096 fmchain rotate direct255 decim8 interp4 fir255_i16; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
PCX_BENCH_BACKEND=gloo python bench.py --gpus 2 --shard 33554432 --steps 50 --warmup 10 --no-cpu > $O/bench_two_ranks_one_gpu_gloo.json 2> /dev/null
python bench.py --driver native --gpus 2 --native-devices 0,0 --shard 33554432 --no-cpu > $O/bench_native_two_shards_one_gpu.json 2> /dev/null
python bench.py --driver native --gpus 2 --native-devices 0,0 --shard 33554432 --workload fmchain --no-cpu > $O/bench_native_two_shards_one_gpu_fmchain.json 2> /dev/null
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --no-cpu > $O/bench_native_c3_eight_shards_one_gpu.json 2> /dev/null
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --workload fmchain --no-cpu > $O/bench_native_c3_eight_shards_one_gpu_fmchain.json 2> /dev/null
PCX_BENCH_BACKEND=gloo python bench.py --gpus 8 --shard 8388608 --steps 50 --warmup 10 --no-cpu > $O/bench_eight_ranks_one_gpu_gloo.json 2> /dev/null
cut -c1-300 $O/bench_default.json
