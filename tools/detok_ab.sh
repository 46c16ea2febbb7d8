# A/B of the few-row GEMM (k_rows_gemm vs the LDS-staged kernel; row bound of the flows) on the detokenizers (run through gpurun)
cd $GRAFT_REPO_ROOT; O=gpurun_out/d3; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py tests/test_gpu_hift.py tests/test_gpu_codec.py tests/test_gpu_snac.py tests/test_gpu_csm.py tests/test_gpu_spkenc.py tests/test_gpu_codec_encoder.py tests/test_gpu_worker.py -q -x 2>&1 | tail -15) > $O/parity.log
cat $O/parity.log
for V in 1000000; do
  VOX_FLOW_ROWS=$V timeout 600 python tools/bench_cosyvoice2.py --batch 8 > $O/cv_b8_r$V.json 2> $O/cv_b8_r$V.err
  VOX_FLOW_ROWS=$V timeout 600 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8_r$V.json 2> $O/glm_b8_r$V.err
done
python - <<PY
import json
for f in ["cv_b8_r512","cv_b8_r1000000","glm_b8_r512","glm_b8_r1000000"]:
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, {k:v for k,v in d.items() if "ms" in k or "samples" in k})
    except Exception as e: print(f,"ERR",e)
PY
