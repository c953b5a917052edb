#!/bin/bash
# Run ON THE GPU BOX: numeric modes compared on a fine-tuned ("trained-like") weight set, where rankings are far from chance.
#   1. fine-tune the 7B synthetic model (LoRA r 8 + visual_head, fp16) on a 64-pair synthetic set until it has memorised the pairs;
#   2. evaluate that checkpoint (all six passes, CPN, top-16) in fp16, bf16 (compensated parity mode), plain bf16, fp8 (default for a fine-tuned
#      checkpoint: MLP only, f8_mask 12) and fp8 on every GEMM (f8_mask 31);
#   3. print the recall tables and the per-pass worst relative deviation of every mode from the fp16 run (tools/compare_scores.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/f8trained; CK=/tmp/f8trained          # checkpoints stay on the box (80 MB each)
N=${1:-64}; EPOCHS=${2:-20}; WARM=${3:-2}
mkdir -p $OUT $CK; cd $R
COMMON="--synthetic $N --synthetic_7b --topk 16 --cpn --alpha 0.4 0.8 --c 0.3 0.6 0.9 0.7"
python3 -m blim_amd.main $COMMON --synthetic_same --lr 2e-4 --epochs $EPOCHS --warmup_epochs $WARM --batch_size 16 --output_dir $CK > $OUT/train.log 2>&1
grep -E "loss|Training time" $OUT/train.log | tail -8
for dt in f16 bf16 f8; do
  python3 -m blim_amd.main $COMMON --eval --resume $CK/epoch$((EPOCHS-1)).pth --dtype $dt --dump_scores $OUT/scores_$dt.npz --output_dir $OUT/eval_$dt > $OUT/eval_$dt.log 2>&1
done
python3 -m blim_amd.main $COMMON --eval --resume $CK/epoch$((EPOCHS-1)).pth --dtype bf16 --vtg_precise none --dump_scores $OUT/scores_bf16plain.npz --output_dir $OUT/eval_bf16plain > $OUT/eval_bf16plain.log 2>&1
python3 -m blim_amd.main $COMMON --eval --resume $CK/epoch$((EPOCHS-1)).pth --dtype f8 --f8_mask 31 --dump_scores $OUT/scores_f8all.npz --output_dir $OUT/eval_f8all > $OUT/eval_f8all.log 2>&1
python3 tools/compare_scores.py $OUT/scores_f16.npz $OUT/scores_bf16.npz $OUT/scores_bf16plain.npz $OUT/scores_f8.npz $OUT/scores_f8all.npz | tee $OUT/compare.txt
for dt in f16 bf16 bf16plain f8 f8all; do echo "== $dt"; grep -A6 "t2v_r1" $OUT/eval_$dt.log | head -8; done | tee -a $OUT/compare.txt
