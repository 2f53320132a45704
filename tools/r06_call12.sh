# round 6, GPU call 12: with the pre-LayerNorm sums in fp16 -- (a) 64-row LayerNorm-fused tiles at emsize 512 (PFN_TUNE_GEMM_LN_ROWS) and (b) the wide LayerNorm-fused
# GEMMs at emsize 1024 (PFN_TUNE_FUSE_LN_WIDE, configs[4]) again, same box, interleaved; (c) the notebook recipe trained with the new default
mkdir -p gpurun_out/r06c12
O=gpurun_out/r06c12
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], d['ms_per_step'], 'final_loss', d['config'].get('final_loss'))"; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras"
for rep in 1 2; do
  timeout 600 python bench.py $Q 2>/dev/null | line "config2 default rep$rep" | tee -a $O/ab.txt
  timeout 600 python bench.py $Q --tune 7=1 2>/dev/null | line "config2 64-row LN tiles rep$rep" | tee -a $O/ab.txt
  timeout 600 python bench.py --config 5 $Q 2>/dev/null | line "config5 default rep$rep" | tee -a $O/ab.txt
  timeout 600 python bench.py --config 5 $Q --tune 5=1 2>/dev/null | line "config5 wide fused LN (fp16 sums) rep$rep" | tee -a $O/ab.txt
  timeout 600 python bench.py --config 5 $Q --tune 5=1,16=0 2>/dev/null | line "config5 wide fused LN (f32 sums) rep$rep" | tee -a $O/ab.txt
done
timeout 900 python tools/train_pfn.py --stage notebook5 --light --precision fp16 --epochs 80 --steps-per-epoch 100 --batch 64 --lr 3e-4 --eval-datasets 128 \
    --out $O/trained_fp16_sums16.json 2>&1 | grep -v "^Using\|^(tensor" | tail -8 | tee $O/train_sums16.txt
