# A/B: greedy pick of codebook i inside the persistent launch of step i + 1 (VOX_DEPTH_PICK=0: a sampler launch in between)
for i in 1 2 3; do
  for m in 1 0; do echo -n "VOX_DEPTH_PICK=$m "; VOX_DEPTH_PICK=$m LM_KV=200 python tools/lm_timing.py 2>/dev/null | tail -1; done
done
timeout 600 python -m pytest tests/test_gpu_qwen3.py tests/test_gpu_worker.py -x -q 2>&1 | tail -3
VOX_DEPTH_PICK=0 timeout 600 python -m pytest tests/test_gpu_qwen3.py -x -q 2>&1 | tail -1
