#!/bin/bash
# round 2, GPU call J: polynomial erf-GELU (csrc/common.h::gelu_erf2) A/B against the Abramowitz-Stegun build at the tower GEMM shapes,
# the kernels' own tests, 32x32x16 vs 16x16x32 matrix-pipe burn, headline bench
TAG=${1:-r02j}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && (/tmp/mfma_peak 2 150 > $OUT/mfma_peak.txt 2>&1; cat $OUT/mfma_peak.txt)
for r in 1 2; do
  python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$|l14 fc1|epi fc1 (bias|gelu)" > $OUT/gemm_poly_$r.txt 2>&1
  MARQO_AMD_LIB=$PWD/tools/probes/libmarqo_hip_gelu_as.so python tools/gemm_bench.py --only "b32 (qkv|out|fc1|fc2)$|l14 fc1|epi fc1 (bias|gelu)" > $OUT/gemm_as_$r.txt 2>&1
done
for f in gemm_poly_1 gemm_as_1 gemm_poly_2 gemm_as_2; do echo "== $f"; grep -v amdgpu.ids $OUT/$f.txt; done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_variants_gpu.py tests/test_small_m_gpu.py tests/test_towers_gpu.py tests/test_fp8_gpu.py tests/test_ref_parity_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|Error" $OUT/pytest_sel.log | tail -12
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_poly.json 2> $OUT/bench_poly.err; echo "bench rc=$?"
MARQO_AMD_LIB=$PWD/tools/probes/libmarqo_hip_gelu_as.so timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_as.json 2> $OUT/bench_as.err
python - <<PY
import json
for n in ("bench_poly", "bench_as"):
    try:
        d = json.load(open("$OUT/%s.json" % n)); r = d["roofline"]
        print(n, d["value"], d["ms_per_step"], "frac", r["frac"], {k: round(v["ms_per_step"], 3) for k, v in r["per_family"].items()})
        print("   e2e", json.dumps(d.get("e2e_vectorise")))
    except Exception as e:
        print(n, "failed", e)
PY
