run() {
  echo "$*"
  env "$@" timeout -k 10 150 python bench.py --config beam5_b512_bf16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['value'], d['roofline'].get('avg_launch_us'))
" || exit 1
}
run ICAP_MERGED=0
for tb in 0 256 512 768 1024 1280; do run ICAP_BF16_ATTQ_TILES_BEFORE=$tb; done
for tb in 0 352 768; do run ICAP_BF16_ATTQ_TILES_BEFORE=$tb ICAP_SIDE_ATT_WGS=256; done
