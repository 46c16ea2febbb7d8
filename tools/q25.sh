cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q25}; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout 600 python tools/bench_cosyvoice2.py --batch 8 > $O/cv_b8_$tag.json 2> $O/cv_b8_$tag.err
  env "$@" timeout 600 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8_$tag.json 2> $O/glm_b8_$tag.err
  env "$@" timeout 600 python tools/bench_cosyvoice2.py --batch 4 > $O/cv_b4_$tag.json 2> $O/cv_b4_$tag.err
}
run A VOX_ROWS_NT2=1024 VOX_ROWS_NT4=2048
run B VOX_ROWS_NT2=512 VOX_ROWS_NT4=2048
run C VOX_ROWS_NT2=512 VOX_ROWS_NT4=1024
run D VOX_ROWS_NT2=256 VOX_ROWS_NT4=768
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], {k:round(v,2) for k,v in d.items() if isinstance(v,float) and ("ms" in k or "samples" in k)})
    except Exception as e: print(f,"ERR",e)
PY
