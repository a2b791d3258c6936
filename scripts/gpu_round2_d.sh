#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02d_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02d_pytest_gpu.log
tail -22 gpurun_out/r02d_pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r02d_bench.json'));print({k:d[k] for k in ('value','ms_per_step','scaling','parity_checked_queries','repetitions')});r=d['roofline'];print({k:r[k] for k in ('bound','achieved','peak','frac','traffic','hbm','scan_ms_per_step','select_ms_per_step','candidates_per_query')});print(d['cpu_baseline'])"
tail -3 gpurun_out/r02d_bench.err
