# bench c3 / c5 over split factors of the Gram launch (run on the GPU box from the repo root).
# BLR_MI355X_GRAM_SPLITS = "off-diagonal,diagonal" column ranges per tile, or "off-diagonal,diagonal,nlong": three kinds of
# work items, dispatched diagonal tiles first, then the nlong strictly lower tiles that have ONE RANGE LESS, then the others
# (multi-round launches, plan_gram_rounds in blr_abi.hip; BLR_MI355X_PLAN_DEBUG=1 prints the plan the library picks).
# Measured at c5's shape (D = 2048, N = 16384, ms per update, three runs each): 7,7 1.045 | 8,4,64 0.997 | 8,8,64 0.999 |
# 8,6,56 0.999 | 8,6,64 0.997 | 8,8,72 1.001 | 8,5,64 1.011 | 9,9,60 1.012 | 8,4,56 1.032 | 8,8,48 1.061 | 8,8,56 1.065 | 8,7,64 1.076
run() { python bench.py --config $1 --steps 10 --warmup 3 --cpu-seconds 0 --secondary 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4))"; }
for sp in ${C3_SPLITS:-14,14 15,11 14,14 15,11}; do BLR_MI355X_GRAM_SPLITS=$sp run c3 "[$sp]"; done
for sp in ${C5_SPLITS:-7,7 8,8,64 8,4,64 8,6,56 8,8,56 7,7 8,8,64}; do BLR_MI355X_GRAM_SPLITS=$sp run c5 "[$sp]"; done
