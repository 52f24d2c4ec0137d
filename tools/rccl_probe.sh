export MASTER_ADDR=127.0.0.1 MASTER_PORT=29520 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout -k 10 150 python tools/host_time.py --rccl > gpurun_out/host_time_rccl.log 2>&1; tail -2 gpurun_out/host_time_rccl.log
cd /tmp && export TMPDIR=/tmp
SVAE_DP_REHEARSAL=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_rccl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_rccl.log 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_rccl.log | cut -c1-300
