#!/bin/bash
# bench.py on N ranks SHARING this box's one GPU over gloo (MSC_BENCH_ONE_GPU: the exchange logic and its host-side cost per step,
# not a scaling measurement -- the ranks' kernels queue on the same device) next to one rank on the same sequences.
#   tools/bench_ranks_one_gpu.sh <ranks> [bench.py arguments ...]
set -e
N=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 rank : %.1f M pairs/s, %.2f ms per step' % (d['value'] / 1e6, d['ms_per_step']))"
MSC_BENCH_BACKEND=gloo MSC_BENCH_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$N ranks on one GPU (gloo): %.1f M pairs/s, %.2f ms per step' % (d['value'] / 1e6, d['ms_per_step']))"
