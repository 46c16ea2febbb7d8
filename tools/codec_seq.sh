#!/bin/bash
# ordered kernel sequence of one Qwen3 codec chunk at B=32 (two-term operands): name, grid, duration
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/cseq; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -o c -- python tools/codec_chunk_prof.py ${CSEQ_B:-32} 2 > $O/run.log 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/cseq/t/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last chunk: find the last k_rvq
idx=[i for i,r in enumerate(rows) if 'k_rvq' in r['Kernel_Name']]
a=idx[-1]
seq=rows[a:]
t0=int(seq[0]['Start_Timestamp'])
with open('gpurun_out/cseq/seq.txt','w') as o:
    for r in seq:
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        o.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f} us grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']} lds {r.get('LDS_Block_Size','')} {r['Kernel_Name'][:70]}\n")
print(open('gpurun_out/cseq/seq.txt').read())
PY
rm -rf $O/t
