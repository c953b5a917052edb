#!/bin/bash
# Run ON THE GPU BOX: numeric modes compared on a fine-tuned ("trained-like") weight set, where rankings are far from chance.
#   1. fine-tune the 7B synthetic model (LoRA r 8 + visual_head, fp16) on a 64-pair synthetic set until it has memorised the pairs;
#   2. evaluate that checkpoint (all six passes, CPN, top-16) with the adapters kept APART (fp16, bf16 compensated parity mode, fp8) and MERGED into the
#      engine's weights (fp16, bf16 parity mode, fp8 with f8_mask 12 = MLP only and 31 = every GEMM);
#   3. print the recall tables and the per-pass worst relative deviation of every mode from the fp16 run (tools/compare_scores.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/f8trained; CK=/tmp/f8trained          # checkpoints stay on the box (80 MB each)
N=${1:-64}; EPOCHS=${2:-20}; WARM=${3:-2}
mkdir -p $OUT $CK; cd $R
COMMON="--synthetic $N --synthetic_7b --topk 16 --cpn --alpha 0.4 0.8 --c 0.3 0.6 0.9 0.7"
python3 -m blim_amd.main $COMMON --synthetic_same --lr 2e-4 --epochs $EPOCHS --warmup_epochs $WARM --batch_size 16 --output_dir $CK > $OUT/train.log 2>&1
grep -E "loss|Training time" $OUT/train.log | tail -8
EV="python3 -m blim_amd.main $COMMON --eval --resume $CK/epoch$((EPOCHS-1)).pth"
# adapters kept APART (the default since round 4): fp16 = the yardstick; bf16 parity mode; fp8 (the adapted projections then run in fp16 whatever the mask says)
for dt in f16 bf16 f8; do
  $EV --dtype $dt --vtg_precise $([ $dt = bf16 ] && echo full || echo none) --dump_scores $OUT/scores_$dt.npz --output_dir $OUT/eval_$dt > $OUT/eval_$dt.log 2>&1
done
# adapters MERGED into the engine's 16-bit weights (round 3's only mode): fp16, bf16 parity mode, fp8 with the MLP only / every GEMM in e4m3
$EV --dtype f16 --vtg_precise none --lora_mode merge --dump_scores $OUT/scores_f16merge.npz --output_dir $OUT/eval_f16merge > $OUT/eval_f16merge.log 2>&1
$EV --dtype bf16 --vtg_precise full --lora_mode merge --dump_scores $OUT/scores_bf16merge.npz --output_dir $OUT/eval_bf16merge > $OUT/eval_bf16merge.log 2>&1
$EV --dtype f8 --lora_mode merge --f8_mask 12 --dump_scores $OUT/scores_f8merge12.npz --output_dir $OUT/eval_f8merge12 > $OUT/eval_f8merge12.log 2>&1
$EV --dtype f8 --lora_mode merge --f8_mask 31 --dump_scores $OUT/scores_f8merge31.npz --output_dir $OUT/eval_f8merge31 > $OUT/eval_f8merge31.log 2>&1
$EV --dtype f16 --dump_scores $OUT/scores_f16auto.npz --output_dir $OUT/eval_f16auto > $OUT/eval_f16auto.log 2>&1       # --vtg_precise auto (the driver's default)
grep "vtg_precise auto" $OUT/eval_f16auto.log | tee $OUT/compare.txt
python3 tools/compare_scores.py $OUT/scores_bf16.npz $OUT/scores_f16.npz $OUT/scores_f16auto.npz $OUT/scores_f16merge.npz $OUT/scores_bf16merge.npz $OUT/scores_f8.npz $OUT/scores_f8merge12.npz $OUT/scores_f8merge31.npz | tee -a $OUT/compare.txt
for dt in f16 bf16 f16merge bf16merge f8 f8merge12 f8merge31; do echo "== $dt"; grep -A6 "t2v_r1" $OUT/eval_$dt.log | head -8; grep "evaluation:" $OUT/eval_$dt.log | cut -c1-200; done | tee -a $OUT/compare.txt
