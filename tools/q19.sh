cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q19}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_qwen3.py tests/test_gpu_csm.py tests/test_gpu_ops.py tests/test_gpu_icl.py tests/test_gpu_hift.py tests/test_gpu_codec.py tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py -q -x 2>&1 | tail -4) > $O/parity.log
cat $O/parity.log
for b in 1 8; do
  timeout 600 python tools/bench_glm.py --batch $b --greedy --steps 150 > $O/glm_b$b.json 2> $O/glm_b$b.err
  timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b$b.json 2> $O/cv_b$b.err
done
VOX_CG_BK128=0 timeout 600 python tools/bench_cosyvoice2.py --batch 1 > $O/cv_b1_bk64.json 2> $O/cv_b1_bk64.err
timeout 600 python tools/bench_csm.py --batch 16 > $O/csm_b16.json 2> $O/csm_b16.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k or "samples" in k)})
    except Exception as e: print(f,"ERR",e)
PY
