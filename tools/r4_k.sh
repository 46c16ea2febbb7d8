cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -4
for v in 0 1; do
  VOX_ATTN_MERGE_ROW=$v timeout 600 python bench.py --no-other-configs --no-cpu-baseline --serving-ttfa-requests 0 --sub-batches "" --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('merge_row=$v', {b: {k: round(v['frame_ms'],3) for k,v in d['kv_sweep'][b].items()} for b in ('batch1','batch32')})"
done
