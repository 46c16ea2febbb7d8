cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q7; mkdir -p $O
for tp in 0 1; do for B in 1 4; do
  VOX_CONV_TAPS=$tp python $R/tools/codec_chunk_prof.py $B 2 graph >> $O/times.txt 2>&1
  VOX_CONV_TAPS=$tp rocprofv3 --kernel-trace --stats -d $O/prof_${B}_$tp -o p -- python $R/tools/codec_chunk_prof.py $B 2 > $O/prof_${B}_$tp.log 2>&1
done; done
