# The judged evidence set, produced on the GPU box:  gpurun -- bash tools/refresh_profiles.sh [outdir]
# then, here:  python tools/collect_profiles.py <outdir> profiles/<round>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r05z}; rm -rf $O; mkdir -p $O
ulimit -c 0
make -s -C tools ubench ols_lab3 clk_lab floor_lab pcie_lab > /dev/null 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> /dev/null
for w in fft4096 fmchain rotate direct255 decim8 interp4 fir255_i16; do python bench.py --workload $w 2>/dev/null | tail -1 >> $O/bench_other_workloads.jsonl; done
PCX_BENCH_BACKEND=gloo python bench.py --gpus 2 --shard 33554432 --steps 50 --warmup 10 --no-cpu > $O/bench_two_ranks_one_gpu_gloo.json 2> /dev/null
python bench.py --driver native --gpus 2 --native-devices 0,0 --shard 33554432 --no-cpu > $O/bench_native_two_shards_one_gpu.json 2> /dev/null
python bench.py --driver native --gpus 2 --native-devices 0,0 --shard 33554432 --workload fmchain --no-cpu > $O/bench_native_two_shards_one_gpu_fmchain.json 2> /dev/null
# configs[3] rehearsed on this one GPU: eight shards of 64 Mi samples each behind the C ABI, and eight RANKS (gloo, reduced shard)
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --no-cpu > $O/bench_native_c3_eight_shards_one_gpu.json 2> /dev/null
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --workload fmchain --no-cpu > $O/bench_native_c3_eight_shards_one_gpu_fmchain.json 2> /dev/null
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --native-pingpong --no-cpu > $O/bench_native_c3_eight_shards_one_gpu_double_buffered.json 2> /dev/null
python bench.py --driver native --gpus 2 --native-devices 0,0 --shard 33554432 --native-pingpong --no-cpu > $O/bench_native_two_shards_one_gpu_double_buffered.json 2> /dev/null
PCX_BENCH_BACKEND=gloo python bench.py --gpus 8 --shard 8388608 --steps 50 --warmup 10 --no-cpu > $O/bench_eight_ranks_one_gpu_gloo.json 2> /dev/null
# the pass of a MIDDLE rank of an RCCL world on this one GPU (halo sent to the rank itself): plain, pipelined (the N > 1 default), unpipelined
for w in fir255 fmchain; do s=""; [ $w = fmchain ] && s="_fmchain"
  python bench.py --workload $w --no-cpu --no-secondary 2>/dev/null | tail -1 > $O/bench_rccl_rank_rehearsal_plain_same_box$s.json
  python bench.py --workload $w --no-cpu --no-secondary --rehearse-rccl-rank --no-autotune 2>/dev/null | tail -1 > $O/bench_rccl_rank_rehearsal$s.json
  python bench.py --workload $w --no-cpu --no-secondary --rehearse-rccl-rank --no-pingpong 2>/dev/null | tail -1 > $O/bench_rccl_rank_rehearsal_unpipelined$s.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 bench.py --steps 2000 --warmup 50 --no-cpu > $O/bench_kt.log 2>&1
bash tools/prof.sh fir255 $O/fir255 ols4096 > /dev/null 2>&1
bash tools/prof.sh fft4096 $O/fft4096 fft_r16 > /dev/null 2>&1
bash tools/prof.sh fmchain $O/fmchain fmchain > /dev/null 2>&1
bash tools/prof.sh decim8 $O/decim8 decim > /dev/null 2>&1
bash tools/prof.sh interp4 $O/interp4 interp > /dev/null 2>&1
bash tools/prof.sh fir255_i16 $O/fir255_i16 fir_cf64_ols > /dev/null 2>&1
bash tools/prof.sh direct255 $O/direct255 fir_cf32_direct > /dev/null 2>&1
bash tools/prof.sh rotate $O/rotate map_kernel > /dev/null 2>&1
timeout 60 tools/clk_lab > $O/clk_lab.txt 2>&1
timeout 300 tools/ols_lab3 3 > $O/ols_lab3_summary.txt 2>&1
python tools/transient_probe.py 20 24 0 2>/dev/null > $O/transient_probe.txt
python tools/transient_probe.py 20 24 5 2>/dev/null >> $O/transient_probe.txt
python tools/shard_probe.py 2>/dev/null | grep shards > $O/shard_probe.txt
python tools/two_streams_probe.py 2>/dev/null | grep launches >> $O/shard_probe.txt
bash tools/ab_gated_slots.sh >> $O/shard_probe.txt 2>/dev/null
( export PCX_PROBE_SHORT=1; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt_shard4 -- python3 tools/shard_probe.py 4 > /dev/null 2>&1; python tools/trace_pass.py $O/kt_shard4 40 > $O/shard4_trace.txt; rm -rf $O/kt_shard4 )
tools/ubench > $O/ubench_roofs.txt 2>&1
timeout 600 python tools/floor_table.py 1.0 > $O/floor_table.txt 2>/dev/null
timeout 300 python tools/chain_path.py > $O/chain_path.txt 2>/dev/null
( export PCX_PROBE_TOTAL=536870912; for G in 1 2 4 8; do timeout 120 python tools/shard_probe.py $G 2>/dev/null | grep shards; done ) > $O/shard_probe_c3.txt
python tools/sweep_fir.py > $O/sweep_fir_taps.txt 2>/dev/null
python tools/sweep_fft.py 16 64 256 1024 2048 4096 8192 16384 > $O/sweep_fft_sizes.txt 2>/dev/null
python tools/sweep_fft_q15_large.py > $O/sweep_fft_q15_large.txt 2>/dev/null
python tools/real_probe.py > $O/real_f32_fir.txt 2>/dev/null
# round 5: the host path (PCIe) and the native driver's host time
timeout 300 python tools/host_path.py 2>/dev/null > $O/host_path.txt
timeout 120 tools/pcie_lab > $O/pcie_lab.txt 2>&1
timeout 60 examples/c_pcie_probe > $O/c_pcie_probe.txt 2>&1
( for t in "" 1; do PCX_PROBE_THREADS=$t PCX_PROBE_TOTAL=536870912 timeout 120 python tools/shard_probe.py 8 2>/dev/null | grep shards; PCX_PROBE_THREADS=$t timeout 120 python tools/shard_probe.py 2 2>/dev/null | grep shards; done ) > $O/shard_probe_threads.txt
python bench.py --driver native --gpus 8 --native-devices 0,0,0,0,0,0,0,0 --native-submit-threads --no-cpu > $O/bench_native_c3_eight_shards_one_gpu_submit_threads.json 2> /dev/null
find $O -name "*.csv" -size +2M -delete
find $O -name "*agent_info*" -delete
du -sh $O
