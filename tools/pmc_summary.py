"""Per-frame HBM traffic of the LM graph from a rocprofv3 --pmc pass (FETCH_SIZE or WRITE_SIZE).
usage: pmc_summary.py <counter_collection.csv> [COUNTER] [frame-marker kernel, default k_qwen3_feedback; k_csm_feedback / k_lm_feedback
for the CSM / single-stack engines]
Counts frames by the number of frame-marker dispatches; sums the counter over the LM kernels (everything that is
not a codec / torch kernel) and over the codec kernels separately.  Units: the counter is reported in KiB-like units of
1024 B by rocprofv3's derived metric; MI355X_MICROARCH.md ('HBM [CDNA4]'): on gfx950 FETCH_SIZE tallies 128-B requests
at 64 B for wide coalesced streaming reads -> corrected = 2 x raw.  WRITE_SIZE is uncalibrated (reported raw)."""
import csv, sys, json
from collections import defaultdict
path = sys.argv[1]
counter = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
marker = sys.argv[3] if len(sys.argv) > 3 else "k_qwen3_feedback"
rows = list(csv.DictReader(open(path)))
# engine-creation kernels (fragment-major weight copies, depth-loop projection tables) run before the first embedding
# kernel of the first prefill: they are not part of any frame
first = min((int(r["Dispatch_Id"]) for r in rows if any(t in r["Kernel_Name"] for t in ("k_qwen3_mix", "k_frame_init", "k_csm_embed", "k_lm_feedback"))), default=0)
by = defaultdict(float)
cnt = defaultdict(int)
for r in rows:
    if r["Counter_Name"] != counter or int(r["Dispatch_Id"]) < first:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    by[name] += float(r["Counter_Value"])
    cnt[name] += 1
frames = sum(v for k, v in cnt.items() if marker in k)
codec = lambda k: any(t in k for t in ("conv_gemm", "codec", "snake", "dwconv", "rvq", "state_update", "_f32", "final_conv", "pos_advance", "k_mimi", "k_flow",
                                       "k_hift", "k_glmflow"))
torchk = lambda k: k.startswith("at::") or "rocclr" in k or "elementwise" in k or "k_swizzle_frag" in k   # (+ engine-creation kernels)
lm = sum(v for k, v in by.items() if not codec(k) and not torchk(k))
cd = sum(v for k, v in by.items() if codec(k))
unit = 1024.0
out = {"counter": counter, "frames": frames, "lm_raw_bytes_per_frame": lm * unit / max(frames, 1),
       "codec_raw_bytes_total": cd * unit, "correction": 2.0 if counter == "FETCH_SIZE" else 1.0}
out["lm_corrected_bytes_per_frame"] = out["lm_raw_bytes_per_frame"] * out["correction"]
print(json.dumps(out))
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k[:50]:50s} n={cnt[k]:6d} raw {v*unit/1e6:10.1f} MB total, {v*unit/max(cnt[k],1)/1e6:8.3f} MB/dispatch")
