#!/bin/bash
# The scaling curve of the frame-sharded path in ONE command, for whoever has a multi-GPU MI355X node (no round of this
# build had one: every multi-GPU number in DESIGN.md is a wire-less single-GPU measurement or an estimate).
#
#   tools/scale.sh [out-dir]            N = 1, 2, 4, 8; backends nccl (torch.distributed) and native (tf_rank_pivotal)
#   GPUS="1 2 8" BACKENDS="nccl" STEPS=20 WARMUP=5 tools/scale.sh out
#
# Per (N, backend) one bench.py line (JSON) in <out-dir>/scale_<backend>_<N>.json: `ms_per_step` / `value` = the faster of
# the two verified forms of the rank's attention (`value_form` names it), `ms_per_step_bit_identical` = the one-pass form
# (equal to the bit-stable single-GPU run bit for bit), `ms_per_step_split` = the split form; then the two-GPU RCCL tests
# (skipped on a 1-GPU box) and a table of value / the two timings / speed-up over N = 1.
set -u
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/scale}
GPUS=${GPUS:-"1 2 4 8"}
BACKENDS=${BACKENDS:-"nccl native"}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
mkdir -p "$OUT"
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NDEV"
for b in $BACKENDS; do
  for n in $GPUS; do
    if [ "$n" -gt "$NDEV" ]; then echo "skip N=$n ($NDEV GPUs visible)"; continue; fi
    extra=""; [ "$n" -gt 1 ] && extra="--backend $b"
    [ "$n" -eq 1 ] && [ "$b" != "$(echo $BACKENDS | cut -d' ' -f1)" ] && continue     # N = 1 has no backend
    timeout 1800 python bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-yardstick $extra \
      > "$OUT/scale_${b}_${n}.json" 2> "$OUT/scale_${b}_${n}.err" || echo "N=$n backend=$b FAILED (see $OUT/scale_${b}_${n}.err)"
  done
done
if [ "$NDEV" -ge 2 ]; then
  timeout 1800 python -m pytest tests/test_sharded_gpu.py -q -k "rccl or two_gpus" -p no:cacheprovider > "$OUT/two_gpu_tests.txt" 2>&1
  tail -3 "$OUT/two_gpu_tests.txt"
fi
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = []
for path in sorted(glob.glob(os.path.join(sys.argv[1], "scale_*.json"))):
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception:
        continue
    b = os.path.basename(path).split("_")[1]
    rows.append((b, d["n_gpus"], d["value"], d.get("ms_per_step_bit_identical") or d["ms_per_step"], d.get("ms_per_step_split")))
base = next((r[3] for r in rows if r[1] == 1), None)
print("backend  N  frames/s  ms/step (bit-identical)  ms/step (split)  speed-up (bit-identical / split)")
for b, n, v, ms, mss in sorted(rows, key=lambda r: (r[0], r[1])):
    su = f"{base / ms:5.2f}x" if base else "  -  "
    sus = f"{base / mss:5.2f}x" if (base and mss) else "  -  "
    print(f"{b:7s} {n:2d} {v:9.1f} {ms:12.3f} {(mss if mss else float('nan')):22.3f}   {su} / {sus}")
PY
