#!/bin/bash
# Round 6: the N > 1 path of bench.py end to end on the one GPU there is: two ranks on cuda:0 over gloo (RCCL refuses two
# ranks per device), self-spawned and under torchrun, trajectory gather on.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/r06_call15; rm -rf $R; mkdir -p $R
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 2 --same-device --dist-backend gloo --steps 40 --warmup 10 --envs 2048 > $R/two_ranks_spawned.json 2> $R/two_ranks_spawned.err
echo "spawned rc=$?"; tail -c 600 $R/two_ranks_spawned.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --same-device --dist-backend gloo --steps 40 --warmup 10 --envs 2048 > $R/two_ranks_torchrun.json 2> $R/two_ranks_torchrun.err
echo "torchrun rc=$?"; tail -c 300 $R/two_ranks_torchrun.json; tail -5 $R/two_ranks_torchrun.err
