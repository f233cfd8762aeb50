# workgroups of a gated launch (1024 = every resident slot taken) against the pass time of G shards on one device
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
for s in 1024 1016 1008 992 960; do echo "gated launch on $s workgroups"; PCX_HIP_LIBRARY=$D PCX_GATED_SLOTS=$s python tools/shard_probe.py 2>/dev/null | grep shards; done
