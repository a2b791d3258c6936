#!/bin/bash
# select_kernel block size (and the winners' final order: counting up to 256 / 512 winners, else a bitonic sort) at a large top_k
export DPQ_DEV=1
mkdir -p gpurun_out
for cfg in "--m 8 --topk 1000" "--m 16 --topk 1000" "--m 8 --topk 512" "--m 8 --topk 300" "--m 8 --topk 2048"; do
for t in 256 512 1024 r512; do
if [ $t = r512 ]; then env="DPQ_LIB_PATH=$PWD/variants/lib_rank512.so DPQ_SELECT_THREADS=512"; else env="DPQ_SELECT_THREADS=$t"; fi
env $env python bench.py --no-cpu-baseline --reps 4 $cfg > gpurun_out/st.json 2>gpurun_out/st.err || { tail -5 gpurun_out/st.err; continue; }
python - <<PY
import json
d=json.loads(open("gpurun_out/st.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$cfg select threads $t:", round(d["value"]), "q/s", round(d["ms_per_step"],4), "ms/step scan", round(r["scan_ms_per_step"],4), "select+boot", round(r["select_ms_per_step"],4), "parity", d["parity_checked_queries"], flush=True)
PY
done
done 2>&1 | tee gpurun_out/select_threads.txt
