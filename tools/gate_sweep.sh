# A/B of the idle event record in front of the vocab projection over the bench configurations
for cfgname in beam3_b512_fp32 beam3_b512_fp32split beam3_b512_fp32_gru searcher_b512_fp32 greedy_b32_fp32 beam10_v50k_fp32 beam5_b512_bf16; do
  for g in 0 1; do
    ICAP_VOCAB_GATE=$g timeout -k 10 120 python bench.py --config $cfgname --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfgname gate=$g', d['value'])" || exit 1
  done
done
