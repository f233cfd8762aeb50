cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; rm -rf $O; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for w in fft4096 fmchain rotate direct255 decim8 interp4 fir255_i16; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --workload fmchain --no-cpu > $O/bench_native_c3_eight_shards_one_gpu_fmchain.json 2> /dev/null
python bench.py --driver native --gpus 2 --native-devices 0,0 --shard 33554432 --workload fmchain --no-cpu > $O/bench_native_two_shards_one_gpu_fmchain.json 2> /dev/null
timeout 600 python tools/floor_table.py 1.0 > $O/floor_table.txt 2>/dev/null
tools/soffset_lab > $O/soffset_lab.txt 2>&1
cat $O/bench_default_time.txt; cat $O/soffset_lab.txt; cat $O/floor_table.txt
