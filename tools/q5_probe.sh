cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q5; mkdir -p $O
python $R/tools/codec_planes_probe.py > $O/planes.txt 2>&1
for cfg in "32 2" "1 2"; do
  set -- $cfg
  python $R/tools/codec_chunk_prof.py $1 $2 graph >> $O/times.txt 2>&1
  rocprofv3 --kernel-trace --stats -d $O/prof_$1_$2 -o p -- python $R/tools/codec_chunk_prof.py $1 $2 > $O/prof_$1_$2.log 2>&1
done
cd $R; timeout 400 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -5 > $O/codec_tests.txt
