cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o c32 -- python tools/codec_chunk_prof.py 32 2 > $O/run.log 2>&1
tail -2 $O/run.log
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_chunk.py $DB 15 > $O/chunk_b32.txt 2>&1
cat $O/chunk_b32.txt | cut -c1-150
rm -rf $O/prof
