cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q27}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py tests/test_gpu_worker.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for V in 1024 256 512 1024 256; do
  VOX_ROWS_NT2=$V timeout 600 python tools/bench_cosyvoice2.py --batch 1 > $O/cv_b1_$V.json 2> $O/cv_b1_$V.err
  VOX_ROWS_NT2=$V timeout 600 python tools/bench_glm.py --batch 1 --greedy --steps 150 > $O/glm_b1_$V.json 2> $O/glm_b1_$V.err
  python - <<PY
import json
for f in ["cv_b1_$V.json","glm_b1_$V.json"]:
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("chunk_ms" in k or "window_ms" in k)})
PY
done
