# same-box A/B: one-request frame time, library before / after (VOX_LIB=tools/bin/libvoxhip_old.so = the previous build)
for i in 1 2 3; do
  echo -n "old: "; VOX_LIB=tools/bin/libvoxhip_old.so LM_KV=200 python tools/lm_timing.py 2>/dev/null | tail -1
  echo -n "new: "; LM_KV=200 python tools/lm_timing.py 2>/dev/null | tail -1
done
