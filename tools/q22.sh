cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q22}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for V in 0 8192 4096; do
  VOX_ROWS_NT8=$V timeout 600 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8_$V.json 2> $O/glm_b8_$V.err
  VOX_ROWS_NT8=$V timeout 600 python tools/bench_cosyvoice2.py --batch 8 > $O/cv_b8_$V.json 2> $O/cv_b8_$V.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k or "samples" in k)})
    except Exception as e: print(f,"ERR",e)
PY
