cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q23}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_qwen3.py tests/test_gpu_ops.py tests/test_gpu_icl.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
Q="--no-cpu-baseline --ttfa-requests 0 --serving-ttfa-requests 0 --no-other-configs"
for V in 0 1 0 1; do
  VOX_FULLK_CT2=$V timeout 600 python tools/lm_timing.py 32 > $O/lm_$V.txt 2>&1; tail -4 $O/lm_$V.txt
done
for V in 0 1; do
  VOX_FULLK_CT2=$V timeout 600 python bench.py --batch 32 --steps 100 --warmup 20 $Q > $O/bench_b32_$V.json 2> $O/bench_b32_$V.err
  python -c "
import json; d=json.loads(open('$O/bench_b32_$V.json').read().strip().splitlines()[-1]); print('bench b32 CT2=$V', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
