# Does the kernel a gate waits for (the halo's copy, the signal) find room beside a launch that fills every resident slot?
# Two shards on ONE device stand in for two devices: each shard's launch gets HALF the slots, so that -- as on a device of its
# own -- all of its workgroups are resident and what it leaves free stays free.
D=$PWD/pothoscomms_amd/libpcx_hip_diag.so
for s in 512 508 504 496 480; do echo "each shard's launch on $s workgroups (2 x $s of 1024 slots)"; PCX_HIP_LIBRARY=$D PCX_GATED_SLOTS=$s PCX_DEALT_SLOTS=$s python tools/shard_probe.py 2 2>/dev/null | grep shards; done
