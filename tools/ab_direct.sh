# A/B of fragment-major weights in k_linear_mfma_stream (VOX_STREAM_FRAG=0: row-major) on GLM-4-Voice / CosyVoice2 B=8, then the tests that cover it
for m in 1 0; do
  echo "== VOX_STREAM_FRAG=$m glm B=8"; VOX_STREAM_FRAG=$m python tools/bench_glm.py --steps 60 --greedy 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ('lm_graph_ms','ms_per_step','detokenizer_window_ms','lm_tokens_per_s')})"
  echo "== VOX_STREAM_FRAG=$m cosyvoice2 B=8"; VOX_STREAM_FRAG=$m python tools/bench_cosyvoice2.py --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ('lm_graph_ms','ms_per_token_step','detokenizer_chunk_ms','audio_samples_per_s')})"
done
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_lm.py -x -q 2>&1 | tail -4
