cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q26}; mkdir -p $O
run() { tag=$1; shift
  for b in 2 4 8; do env "$@" timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b${b}_$tag.json 2> $O/cv_b${b}_$tag.err; done
  env "$@" timeout 600 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8_$tag.json 2> $O/glm_b8_$tag.err
  env "$@" timeout 600 python tools/bench_glm.py --batch 2 --greedy --steps 150 > $O/glm_b2_$tag.json 2> $O/glm_b2_$tag.err
}
run B VOX_ROWS_NT2=512 VOX_ROWS_NT4=2048
run E VOX_ROWS_NT2=256 VOX_ROWS_NT4=2048
run F VOX_ROWS_NT2=128 VOX_ROWS_NT4=2048
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("chunk_ms" in k or "window_ms" in k)})
    except Exception as e: print(f,"ERR",e)
PY
