cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q31}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_ops.py tests/test_gpu_qwen3.py tests/test_gpu_csm.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for V in 1 3 1 3; do
  VOX_MFMA_DEPTH=$V timeout 600 python tools/bench_glm.py --batch 8 --greedy --steps 150 > $O/glm_b8_$V.json 2> $O/glm_b8_$V.err
  python -c "
import json; d=json.loads(open('$O/glm_b8_$V.json').read().strip().splitlines()[-1]); print('glm b8 depth=$V', round(d['lm_graph_ms'],3), round(d['audio_samples_per_s']))"
  VOX_MFMA_DEPTH=$V timeout 600 python tools/lm_timing.py 16 > $O/lm16_$V.txt 2>&1; tail -1 $O/lm16_$V.txt
done
