# the native driver's lines on ONE box in one call: plain launch, configs[3] rehearsed (eight shards on one device) and two shards, each as shipped and
# double-buffered over two handles (pcx_shard_post_exchange / pcx_shard_compute):  gpurun -- bash tools/native_lines.sh [outdir]
O=${1:-gpurun_out/r04n}; mkdir -p $O
E8="--driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --no-cpu"
E2="--driver native --gpus 2 --native-devices 0,0 --shard 33554432 --no-cpu"
python bench.py --no-cpu --no-secondary 2>/dev/null | tail -1 > $O/bench_native_plain_same_box.json
python bench.py $E8 2>/dev/null | tail -1 > $O/bench_native_c3_eight_shards_one_gpu.json
python bench.py $E8 --native-pingpong 2>/dev/null | tail -1 > $O/bench_native_c3_eight_shards_one_gpu_double_buffered.json
python bench.py $E2 2>/dev/null | tail -1 > $O/bench_native_two_shards_one_gpu.json
python bench.py $E2 --native-pingpong 2>/dev/null | tail -1 > $O/bench_native_two_shards_one_gpu_double_buffered.json
python bench.py $E8 --workload fmchain 2>/dev/null | tail -1 > $O/bench_native_c3_eight_shards_one_gpu_fmchain.json
python bench.py $E8 --workload fmchain --native-pingpong 2>/dev/null | tail -1 > $O/bench_native_c3_eight_shards_one_gpu_fmchain_double_buffered.json
for f in $O/bench_native_*.json; do python -c "import json; d=json.load(open(\"$f\")); print(\"$f\", d[\"ms_per_step\"], d[\"value\"], d[\"roofline\"][\"frac\"])"; done
