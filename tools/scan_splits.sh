run() { python bench.py --config $1 --steps 10 --warmup 2 --cpu-seconds 0 --secondary 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4))"; }
for sp in "" 14,14 15,11 15,10 15,9 16,8 16,7; do BLR_MI355X_GRAM_SPLITS=$sp run c3 "[$sp]"; done
for sp in "" 3,3 4,2 4,1; do BLR_MI355X_GRAM_SPLITS=$sp run c5 "[$sp]"; done
