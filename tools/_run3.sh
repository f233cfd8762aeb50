cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
( export PCX_PROBE_TOTAL=536870912
for G in 2 3 4 6; do for s in 256 512 768 1024; do echo "G=$G, each shard's launch on $s workgroups"; PCX_HIP_LIBRARY=$D PCX_GATED_SLOTS=$s PCX_DEALT_SLOTS=$s timeout 120 python tools/shard_probe.py $G 2>/dev/null | grep shards; done; done
export PCX_PROBE_TOTAL=67108864
for G in 2 4 8; do for s in 512 1024; do echo "64 Mi total: G=$G, each shard's launch on $s workgroups"; PCX_HIP_LIBRARY=$D PCX_GATED_SLOTS=$s PCX_DEALT_SLOTS=$s timeout 120 python tools/shard_probe.py $G 2>/dev/null | grep shards; done; done
) > $O/shard_probe_c3_slots.txt 2>&1
cat $O/shard_probe_c3_slots.txt
timeout 600 python bench.py > $O/bench_default_new.json 2> $O/bench_default_new.err; tail -3 $O/bench_default_new.err; cat $O/bench_default_new.json
timeout 300 python bench.py --workload direct255 --no-cpu > $O/bench_direct.json 2>> $O/bench_default_new.err; cat $O/bench_direct.json
