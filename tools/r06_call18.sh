# round 6, GPU call 18: the notebook's recipe TRAINED IN FP16 through train.train with the final arithmetic (saturating stores, loss-scale target 2, fp16 sums), scored
# against the exact GP, with the parity of both 16-bit training forwards and of the f32 inference kernels ON THE TRAINED WEIGHTS at the configs[1] model size;
# short bf16 / fp16 / f32 loss curves from identical seeds
mkdir -p gpurun_out/r06c18
O=gpurun_out/r06c18
timeout 2400 python tools/train_pfn.py --stage notebook5 curves --precision fp16 --epochs 80 --steps-per-epoch 100 --batch 64 --lr 3e-4 --out $O/trained_fp16_notebook5.json 2>&1 | grep -v "^Using\|^(tensor" | tail -40 | tee $O/train_log.txt
