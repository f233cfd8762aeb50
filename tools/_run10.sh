cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_special_values_gpu.py tests/test_shard_gpu.py tests/test_c3_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q -n 4 -k "chain or fm or Chain" > $O/pytest_chain.txt 2>&1; echo "rc $?" >> $O/pytest_chain.txt; tail -3 $O/pytest_chain.txt
bash tools/ab_lib.sh $PWD/tools/_ab/libpcx_hip_r04base.so fmchain 2>&1 | tee $O/ab_fmchain_tail.txt
