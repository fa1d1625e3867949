#!/bin/bash
# round 3: rehearsal of the N-rank bench code path on a 1-GPU box (gloo, ranks share the GPU): self-spawn, verify, extras
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/rehearsal; mkdir -p $O
( time timeout 1500 python bench.py --gpus 2 --backend gloo --share-gpus --steps 5 --no-cpu-baseline ) > $O/n2_selfspawn.json 2> $O/n2_selfspawn.err
tail -c 1500 $O/n2_selfspawn.err
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --backend gloo --share-gpus --steps 5 --no-extra --scale 22 ) > $O/n4_torchrun_s22.json 2> $O/n4_torchrun_s22.err
tail -c 600 $O/n4_torchrun_s22.err
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --backend gloo --share-gpus --steps 5 --no-extra --scale 22 --workload mxv_lor_land_masked --overlap-chunks 4 ) > $O/n2_bfs_s22.json 2> $O/n2_bfs_s22.err
tail -c 600 $O/n2_bfs_s22.err
python - <<'PY'
import json
for f in ("n2_selfspawn", "n4_torchrun_s22", "n2_bfs_s22"):
    try:
        d = json.loads(open(f"gpurun_out/rehearsal/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "NO JSON", e); continue
    print(f, {k: d.get(k) for k in ("n_gpus", "value", "ms_per_step", "verified", "rehearsal")}, d.get("exchange"))
    for e in d.get("extra", []):
        print("   extra:", e["workload"][:40], e.get("value"), e.get("ms_per_step"), e.get("verified"), e.get("error"))
PY
