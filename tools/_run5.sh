cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python tools/floor_table.py 1.0 > $O/floor_table.txt 2>&1
cat $O/floor_table.txt
timeout 300 python tools/chain_path.py > $O/chain_path.txt 2>&1; cat $O/chain_path.txt
