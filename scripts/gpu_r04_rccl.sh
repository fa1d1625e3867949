#!/bin/bash
# Round 4: RCCL on the library's memory with ONE rank: the one-rank nccl world script, the bench's N-rank path under
# --force-dist (masked relaxation + BFS step with presence words + the sharded scale-22 SpGEMM extra), and the pytest wrapper.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_rccl; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tests/nccl_one_rank.py > "$OUT/nccl_one_rank.json" 2> "$OUT/nccl_one_rank.err"; echo "nccl_one_rank rc=$?"; tail -1 "$OUT/nccl_one_rank.json"; tail -3 "$OUT/nccl_one_rank.err"
timeout 1200 python bench.py --force-dist --backend nccl --no-cpu-baseline > "$OUT/bench_force_dist_nccl.json" 2> "$OUT/bench_force_dist_nccl.err"; echo "bench --force-dist rc=$?"
python - "$OUT/bench_force_dist_nccl.json" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("ms_per_step", "verified", "exchange", "rehearsal")})
    for e in d.get("extra", []):
        print("  extra:", e.get("workload", "")[:60], {k: e.get(k) for k in ("ms_per_step", "verified", "error")})
except Exception as e:
    print("FAILED", e)
PY
tail -3 "$OUT/bench_force_dist_nccl.err"
timeout 1200 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k rccl_one_rank 2>&1 | tail -3
