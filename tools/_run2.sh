cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
timeout 600 python -m pytest tests/test_c3_gpu.py -m gpu -x -q -s > $O/pytest_c3.txt 2>&1; echo "pytest rc $?" >> $O/pytest_c3.txt
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_shard_gpu.py tests/test_stream_gpu.py tests/test_oracle_cpu.py -m gpu -x -q > $O/pytest_new.txt 2>&1; echo "pytest rc $?" >> $O/pytest_new.txt
( export PCX_PROBE_TOTAL=536870912
echo "8 x 64 Mi on one device, product library (128 slots per shard)"; timeout 120 python tools/shard_probe.py 8 2>/dev/null | grep shards
echo "GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 timeout 120 python tools/shard_probe.py 8 2>/dev/null | grep shards
echo "GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout 120 python tools/shard_probe.py 8 2>/dev/null | grep shards
for s in 256 384 512 1024; do echo "each shard's launch on $s workgroups"; PCX_HIP_LIBRARY=$D PCX_GATED_SLOTS=$s PCX_DEALT_SLOTS=$s timeout 120 python tools/shard_probe.py 8 2>/dev/null | grep shards; done
echo "4 x 128 Mi"; timeout 120 python tools/shard_probe.py 4 2>/dev/null | grep shards
echo "2 x 256 Mi"; timeout 120 python tools/shard_probe.py 2 2>/dev/null | grep shards
echo "1 x 512 Mi"; timeout 120 python tools/shard_probe.py 1 2>/dev/null | grep shards
) > $O/shard_probe_c3.txt 2>&1
cat $O/shard_probe_c3.txt; tail -3 $O/pytest_c3.txt; tail -3 $O/pytest_new.txt
