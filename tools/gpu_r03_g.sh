#!/bin/bash
# round 3, call G: the GPU suite and the default bench line under the per-model residual-stream policy (auto)
tag=${1:-r03g}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -30 | tee $out/pytest_gpu.txt
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.err; python - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %.1f emb/s %.3f ms/step gemm %.1f TF frac %.3f; cpu %s; cos %.2e" % (d["value"], d["ms_per_step"], r["achieved"], r["frac"], d.get("cpu_baseline", {}).get("value"), d.get("cos_err_vs_cpu", float("nan"))))
print("e2e", json.dumps(d.get("e2e_vectorise"))[:900])
for a in d.get("also", []):
    print("also", a.get("workload", "")[:50], a.get("value"), a.get("gemm_frac"), a.get("cos_err_vs_cpu"))
PY
