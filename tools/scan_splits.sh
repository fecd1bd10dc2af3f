# bench c3 / c5 over split factors of the Gram launch ("off-diagonal,diagonal"; run on the GPU box from the repo root)
run() { python bench.py --config $1 --steps 10 --warmup 3 --cpu-seconds 0 --secondary 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4))"; }
for sp in ${C3_SPLITS:-14,14 15,11 14,14 15,11}; do BLR_MI355X_GRAM_SPLITS=$sp run c3 "[$sp]"; done
for sp in ${C5_SPLITS:-3,3 7,7 8,5 11,11 12,7 15,15 16,9 15,15}; do BLR_MI355X_GRAM_SPLITS=$sp run c5 "[$sp]"; done
