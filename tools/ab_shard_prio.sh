D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
echo "product library"; python tools/shard_probe.py 2>/dev/null | grep shards
echo "diag, halo streams at high priority"; PCX_HIP_LIBRARY=$D python tools/shard_probe.py 2>/dev/null | grep shards
echo "diag, halo streams at default priority"; PCX_HIP_LIBRARY=$D PCX_SHARD_HALO_PRIO=0 python tools/shard_probe.py 2>/dev/null | grep shards
echo "diag, default priority, 1024 workgroups"; PCX_HIP_LIBRARY=$D PCX_SHARD_HALO_PRIO=0 PCX_GATED_SLOTS=1024 python tools/shard_probe.py 2>/dev/null | grep shards
