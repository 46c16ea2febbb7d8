cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q32}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_worker.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for V in 0 1 0 1; do
  for b in 1 8; do VOX_ATTN_HEADSPLIT=$V timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b${b}_$V.json 2> $O/cv_b${b}_$V.err; python -c "
import json; d=json.loads(open('$O/cv_b${b}_$V.json').read().strip().splitlines()[-1]); print('cv b$b split=$V lm', round(d['lm_graph_ms'],3), round(d['audio_samples_per_s']))"; done
done
