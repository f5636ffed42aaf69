#!/bin/bash
# round 3, call E: bf16 residual stream as default candidate (16-byte bf16-input LayerNorm A/B, the GPU suite under it), the new bench workloads, the tightened asserts
tag=${1:-r03e}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for rb in 0 1; do for wide in 0 1; do
  [ $rb = 0 ] && [ $wide = 0 ] && continue
  MQ_RESIDUAL_BF16=$rb MQ_LN_BF16_WIDE=$wide timeout 200 python bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('residual_bf16=$rb ln_wide=$wide  %9.1f emb/s %8.3f ms/step  gemm %6.1f TF (frac %.3f)  fam %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/residual_ln_ab.txt
done; done
for wl in clip_text_b32 vit_l14_image; do for rb in 0 1; do
  MQ_RESIDUAL_BF16=$rb timeout 200 python bench.py --workload $wl --steps 15 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$wl residual_bf16=$rb  %9.1f emb/s %8.3f ms/step  gemm frac %.3f  fam %s' % (d['value'], d['ms_per_step'], r['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['per_family'].items()}))
" 2>&1 | tee -a $out/residual_ln_ab.txt
done; done
timeout 300 python bench.py --workload vit_l14_chunked_fp8 --steps 10 --warmup 3 --cpu-seconds 12 > $out/bench_chunked_fp8.json 2> $out/bench_chunked_fp8.err; tail -c 1500 $out/bench_chunked_fp8.json; tail -3 $out/bench_chunked_fp8.err
timeout 300 python bench.py --workload add_documents_stream --steps 80 --warmup 4 --cpu-seconds 12 > $out/bench_stream.json 2> $out/bench_stream.err; tail -c 1800 $out/bench_stream.json; tail -3 $out/bench_stream.err
MQ_RESIDUAL_BF16=1 timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gemm_variants_gpu.py 2>&1 | tail -25 | tee $out/pytest_gpu_residual_bf16.txt
