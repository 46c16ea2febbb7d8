# after the 16-byte greedy scan: GLM-4-Voice B=8 step, the sampler / LM tests, and the default one-request frame
python tools/bench_glm.py --steps 60 --greedy 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ('lm_graph_ms','ms_per_step','detokenizer_window_ms','lm_tokens_per_s')})"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_lm.py tests/test_gpu_qwen3.py tests/test_gpu_csm.py -x -q 2>&1 | tail -3
LM_KV=200 python tools/lm_timing.py 2>/dev/null | tail -3
