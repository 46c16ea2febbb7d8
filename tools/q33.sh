# MFMA busy cycles of the detokenizer kernels (CosyVoice2 B=8, GLM B=8): a PMC pass of its own (--pmc with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/q33; mkdir -p $O
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_cv -o p -- python tools/bench_cosyvoice2.py --batch 8 --steps 50 --warmup 0 > $O/pmc_cv.log 2>&1
python tools/mfma_summary.py $(find $O/pmc_cv -name "*counter_collection.csv") $(find $O/pmc_cv -name "*kernel_trace.csv") > $O/mfma_cosyvoice2_b8.json 2>&1
rm -rf $O/pmc_cv
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_glm -o p -- python tools/bench_glm.py --batch 8 --greedy --steps 60 --warmup 5 > $O/pmc_glm.log 2>&1
python tools/mfma_summary.py $(find $O/pmc_glm -name "*counter_collection.csv") $(find $O/pmc_glm -name "*kernel_trace.csv") > $O/mfma_glm_b8.json 2>&1
rm -rf $O/pmc_glm
head -c 1500 $O/mfma_cosyvoice2_b8.json
