# round 6, GPU call 8: whole GPU suite (assert mode, bounds recorded), then the notebook's recipe TRAINED IN FP16 through train.train (512 k datasets), scored against
# the exact GP, with the parity of both 16-bit training forwards on the trained weights; short bf16 / fp16 / f32 loss curves from identical seeds
mkdir -p gpurun_out/r06c8
O=gpurun_out/r06c8
PFN_RECORD_BOUNDS=$O/parity_measured.json timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
tail -6 $O/pytest.log
timeout 1500 python tools/train_pfn.py --stage notebook5 curves --precision fp16 --epochs 80 --steps-per-epoch 100 --batch 64 --lr 3e-4 --out $O/trained_fp16_notebook5.json 2>&1 | grep -v "^Using\|^(tensor" | tail -40 | tee $O/train_log.txt
