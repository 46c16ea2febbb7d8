cd $GRAFT_REPO_ROOT; O=gpurun_out/${OUT:-q24}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_qwen3.py tests/test_gpu_ops.py tests/test_gpu_lm.py tests/test_gpu_csm.py -q -x 2>&1 | tail -3) > $O/parity.log
cat $O/parity.log
for V in 0 1 0 1; do
  VOX_FULLK_CT2=$V timeout 600 python tools/lm_timing.py 8 > $O/lm8_$V.txt 2>&1; tail -1 $O/lm8_$V.txt
  VOX_FULLK_CT2=$V timeout 600 python tools/lm_timing.py 16 > $O/lm16_$V.txt 2>&1; tail -1 $O/lm16_$V.txt
done
timeout 600 python tools/lm_timing.py 1 > $O/lm1.txt 2>&1; tail -1 $O/lm1.txt
timeout 600 python tools/lm_timing.py 32 > $O/lm32.txt 2>&1; tail -1 $O/lm32.txt
