# flow detokenizers: parity + chunk timing; k_rows_gemm variants (8-wave K split, two row tiles per block)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q13}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py tests/test_gpu_hift.py tests/test_gpu_worker.py tests/test_gpu_codec.py tests/test_gpu_snac.py -q -x 2>&1 | tail -4) > $O/parity.log
cat $O/parity.log
run() {  # tag, env...
  tag=$1; shift
  for b in 1 8; do
  env "$@" timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b${b}_$tag.json 2> $O/cv_b${b}_$tag.err
  env "$@" timeout 600 python tools/bench_glm.py --batch $b --greedy --steps 150 > $O/glm_b${b}_$tag.json 2> $O/glm_b${b}_$tag.err
  done
}
run A VOX_ROWS_WV8=0 VOX_ROWS_MT2=0
run B VOX_ROWS_WV8=1 VOX_ROWS_MT2=0
run C VOX_ROWS_WV8=1 VOX_ROWS_MT2=1024
run D VOX_ROWS_WV8=1 VOX_ROWS_MT2=512
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k or "samples" in k)})
    except Exception as e: print(f,"ERR",e)
PY
