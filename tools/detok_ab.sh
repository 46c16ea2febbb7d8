# A/B of the few-row GEMM's K step on the detokenizers and the Qwen3 frame (run through gpurun); prints JSON lines.
cd $GRAFT_REPO_ROOT; O=gpurun_out/f12; mkdir -p $O
Q="--no-cpu-baseline --ttfa-requests 0 --serving-ttfa-requests 0 --no-other-configs"
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py tests/test_gpu_hift.py tests/test_gpu_codec.py tests/test_gpu_snac.py -q -x 2>&1 | tail -8) > $O/parity.log
for V in 0 1; do
  VOX_SKINNY_BK256=$V timeout 600 python tools/bench_cosyvoice2.py --batch 1 > $O/cv_b1_bk$V.json 2> $O/cv_b1_bk$V.err
  VOX_SKINNY_BK256=$V timeout 600 python bench.py --batch 1 --steps 60 --warmup 10 $Q > $O/q3_b1_bk$V.json 2> $O/q3_b1_bk$V.err
done
timeout 600 python tools/bench_cosyvoice2.py --batch 8 > $O/cv_b8.json 2> $O/cv_b8.err
for b in 1 8; do timeout 600 python tools/bench_glm.py --batch $b --greedy --steps 150 > $O/glm_b$b.json 2> $O/glm_b$b.err; done
timeout 600 python tools/bench_clone.py > $O/clone.json 2> $O/clone.err
cat $O/parity.log; for f in cv_b1_bk0 cv_b1_bk1 cv_b8 glm_b1 glm_b8 clone; do echo $f; cut -c1-700 $O/$f.json; done
for V in 0 1; do python -c "import json;d=json.load(open('$O/q3_b1_bk$V.json'));print('q3 bk$V',d['value'],d['ms_per_step'],d.get('roofline',{}).get('frac'))"; done
