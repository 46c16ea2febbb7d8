cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q20}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_qwen3.py tests/test_gpu_csm.py tests/test_gpu_ops.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for V in 0 1 0 1; do
  VOX_MFMA_ONESEG=$V timeout 600 python tools/bench_cosyvoice2.py --batch 8 > $O/cv_b8_$V.json 2> $O/cv_b8_$V.err
  VOX_MFMA_ONESEG=$V timeout 600 python tools/bench_csm.py --batch 16 > $O/csm_b16_$V.json 2> $O/csm_b16_$V.err
  python - <<PY
import json
for f in ["cv_b8_$V.json","csm_b16_$V.json"]:
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, {k:round(v,3) for k,v in d.items() if isinstance(v,float) and ("ms" in k)})
PY
done
timeout 600 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8.json 2> $O/glm_b8.err; tail -c 400 $O/glm_b8.json
