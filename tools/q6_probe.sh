cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q6; mkdir -p $O
export VOX_LIB=$R/tools/bin/libvoxhip_dev.so
for dv in 0 1 3 7 15 6; do
  VOX_CODEC_DEV=$dv rocprofv3 --kernel-trace --stats -d $O/prof_$dv -o p -- python $R/tools/codec_chunk_prof.py 32 2 > $O/prof_$dv.log 2>&1
done
