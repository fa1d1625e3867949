#!/bin/bash
# Round 6: the N-rank code path of bench.py as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE GPU box:
# two and four ranks sharing the GPU over gloo (correctness of the sharded step, the shard set-up's all-reduce, the exchange selection, the
# sharded SSSP loop; times mean nothing), and one rank over RCCL.
OUT=gpurun_out/r06r; mkdir -p $OUT; export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python -c "
import sys,json
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2', 'n_gpus', d.get('n_gpus'), 'ms', round(d['ms_per_step'],4), 'verified', d['verified'], 'exchange', (d.get('exchange') or {}).get('method'), (d.get('exchange') or {}).get('methods_of_timed_steps'), 'setup', bool(d.get('shard_setup')), 'rehearsal', d.get('rehearsal'))
for e in d.get('extra', []):
    print('    extra:', (e.get('workload') or '')[:50], e.get('verified'), {k: (v.get('methods'), v.get('iterations')) for k, v in e.items() if isinstance(v, dict) and 'methods' in v})
"; }
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 5 --warmup 3 --scale 22 --backend gloo --share-gpus --no-cpu-baseline > $OUT/gloo_x$n.json 2> $OUT/gloo_x$n.err; echo "rc=$?"; show $OUT/gloo_x$n.json gloo_x$n
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 3 --scale 22 --backend gloo --share-gpus --ranked --no-cpu-baseline --no-extra > $OUT/gloo_x2_ranked.json 2> $OUT/gloo_x2_ranked.err; echo "rc=$?"; show $OUT/gloo_x2_ranked.json gloo_x2_ranked
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 3 --scale 22 --workload mxv_lor_land_masked --backend gloo --share-gpus --no-cpu-baseline --no-extra > $OUT/gloo_x2_bfs.json 2> $OUT/gloo_x2_bfs.err; echo "rc=$?"; show $OUT/gloo_x2_bfs.json gloo_x2_bfs
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-dist --steps 5 --warmup 3 --no-cpu-baseline --no-extra > $OUT/rccl_x1.json 2> $OUT/rccl_x1.err; echo "rc=$?"; show $OUT/rccl_x1.json rccl_x1
timeout 900 python bench.py --gpus 1 --steps 5 > $OUT/plain_x1.json 2> $OUT/plain_x1.err; echo "rc=$?"; show $OUT/plain_x1.json plain_x1
