cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q28}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py tests/test_gpu_worker.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for b in 1 2 8; do
  timeout 600 python tools/bench_glm.py --batch $b --greedy --steps 150 > $O/glm_b$b.json 2> $O/glm_b$b.err
done
for b in 1 8; do timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b$b.json 2> $O/cv_b$b.err; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k or "samples" in k)})
    except Exception as e: print(f,"ERR",e)
PY
