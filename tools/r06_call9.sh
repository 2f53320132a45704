# round 6, GPU call 9: (1) the new fp16 guards (saturating stores, optimizer skip) + every fp16 test; (2) same-box A/B of the attention forward's loop variants
# (gpurun_variants/lib_{A,B,C,D}.so: A = round-5 loop, B = lag / first-tile copies, C = B + spelled-out issue order, D = C without the split K-fragment request);
# (3) the notebook's recipe trained in fp16 at loss-scale targets 2 (default), 6 (round-6 first version, now saturating) and -2
mkdir -p gpurun_out/r06c9
O=gpurun_out/r06c9
timeout 1200 python -m pytest tests -m gpu -q -x -k "fp16 or clip_adam or saturat or train_entry or training_loop" 2>&1 | tail -8 > $O/pytest_fp16.log
tail -4 $O/pytest_fp16.log
for v in A B C D A; do
  PFN_LIB=gpurun_variants/lib_$v.so timeout 300 python tools/bench_attn.py 2>&1 | grep -E "attn_fwd |shape" | tee -a $O/attn_variants.txt
  PFN_LIB=gpurun_variants/lib_$v.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $v', d['value'], d['ms_per_step'])" | tee -a $O/attn_variants.txt
done
for tg in 2 6 -2; do
  timeout 900 python tools/train_pfn.py --stage notebook5 --light --precision fp16 --tune 15=$tg --epochs 80 --steps-per-epoch 100 --batch 64 --lr 3e-4 --eval-datasets 128 \
      --out $O/trained_fp16_target$tg.json 2>&1 | grep -v "^Using\|^(tensor" | tail -30 | tee $O/train_target$tg.txt
done
