#!/bin/bash
# Round 5: the block-cyclic ranked step through the N-rank code path (sharded.OverlappedMxv(cyclic=...)) on ONE GPU box:
#  * --force-dist: one rank over RCCL (the collective is a copy to itself; the dealing copies and the chunking are real);
#  * --gpus 2 --share-gpus --backend gloo: two ranks on the one GPU, host-staged exchange -- correctness of the 2-rank dealing on device images.
set -u
cd /root/repo
out=gpurun_out/r05_cyclic; mkdir -p $out
export MASTER_ADDR=127.0.0.1
[ -n "${SKIP_PYTEST:-}" ] || timeout 900 python -m pytest tests/test_vertex_order.py tests/test_sharded_gloo.py -m gpu -x -q 2>&1 | tail -3
for wl in ${WLS:-mxv_min_plus_masked mxv_lor_land_masked}; do
  for rk in "" "--ranked"; do
    tag=$wl$( [ -n "$rk" ] && echo _ranked )
    timeout 600 python bench.py --workload $wl --scale 24 --force-dist $rk --steps 50 --warmup 10 --no-cpu-baseline > $out/fd_$tag.json 2> $out/fd_$tag.err || tail -5 $out/fd_$tag.err
    python - <<PY
import json
try:
    x=json.loads(open("$out/fd_$tag.json").read().strip().splitlines()[-1]); print("force-dist $tag", x["ms_per_step"], x["verified"], x.get("exchange"), x["stats"].get("ordered"))
except Exception as e: print("fd $tag failed", e)
PY
  done
done
for rk in "" "--ranked"; do
  tag=g2$( [ -n "$rk" ] && echo _ranked )
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-gpus --backend gloo --scale 22 --workload ${G2WL:-mxv_min_plus_masked} $rk --steps 10 --warmup 3 --no-cpu-baseline > $out/$tag.json 2> $out/$tag.err || tail -5 $out/$tag.err
  python - <<PY
import json
try:
    x=json.loads(open("$out/$tag.json").read().strip().splitlines()[-1]); print("gloo x2 $tag", x["ms_per_step"], x["verified"], x["stats"].get("ordered"))
except Exception as e: print("$tag failed", e)
PY
done
