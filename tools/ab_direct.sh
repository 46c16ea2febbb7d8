# A/B of the no-LDS-tile form of k_linear_mfma (VOX_MFMA_DIRECT=0: the staged kernel): CosyVoice2 / GLM-4-Voice LM steps under their graphs
for m in 1 0; do
  echo "== VOX_MFMA_DIRECT=$m cosyvoice2 B=8"; VOX_MFMA_DIRECT=$m python tools/bench_cosyvoice2.py --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ('lm_graph_ms','ms_per_token_step','detokenizer_chunk_ms','audio_samples_per_s')})"
  echo "== VOX_MFMA_DIRECT=$m glm B=8"; VOX_MFMA_DIRECT=$m python tools/bench_glm.py --steps 60 --greedy 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in d if k in ('lm_graph_ms','ms_per_step','detokenizer_window_ms','lm_tokens_per_s')})"
done
