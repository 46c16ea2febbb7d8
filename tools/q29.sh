cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q29}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_glm_decoder.py tests/test_gpu_worker.py tests/test_gpu_hift.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for b in 1 1 8; do timeout 600 python tools/bench_cosyvoice2.py --batch $b > $O/cv_b$b.json 2> $O/cv_b$b.err; python -c "
import json; d=json.loads(open('$O/cv_b$b.json').read().strip().splitlines()[-1]); print('cv b$b', round(d['detokenizer_chunk_ms'],2), round(d['audio_samples_per_s']))"; done
