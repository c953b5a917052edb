"""Thin evaluation driver with the reference's flags (main.py:31-75, eval branch :146-174).

    python -m blim_amd.main --eval --dataset MSRVTT --model_path ./pretrained/VideoChat-Flash-Qwen2-7B_res448 \\
        --resume ./checkpoint/msrvtt.pth --topk 16 --batch_size_eval 16 --cpn --alpha 0.4 0.8 --c 0.3 0.6 0.9 0.7
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m blim_amd.main ...        (one process per GPU, RCCL)

Reads ./data/<DS>/..., ./scores/<ds>[_zeroshot].pth exactly as the reference does.  `--synthetic N` replaces checkpoint,
tokenizer, dataset and first-stage scores by seeded synthetic ones (no downloads): an end-to-end dry run of the same code path.
Without --eval it fine-tunes as the reference's main.py:155-195 does: per epoch train_one_epoch (blim_amd/training.py: LoRA adapters +
visual_head on the engine's trainer), save `epoch<N>.pth`, hand the adapters to the scoring engine (apart; --lora_mode merge folds them in),
val_one_epoch, keep `checkpoint_best.pth`, append to <output_dir>/log.txt.
"""
from __future__ import annotations

import argparse
import json
import os
import time
import types


def get_args_parser():
    p = argparse.ArgumentParser("BLiM evaluation on the MI355X engine", add_help=True)
    # the reference's flags with the reference's defaults (main.py:31-75), so that one of its command lines gives the same table
    p.add_argument("--batch_size_eval", default=64, type=int)
    p.add_argument("--num_workers", default=4, type=int)
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)
    p.add_argument("--model_path", default="./pretrained/VideoChat-Flash-Qwen2-7B_res448", type=str)
    p.add_argument("--dataset", default="DiDeMo", type=str, choices=["DiDeMo", "ActivityNet", "LSMDC", "MSRVTT"])
    p.add_argument("--output_dir", default="./checkpoint", type=str)
    p.add_argument("--resume", default="", type=str, help="fine-tuned LoRA / visual_head checkpoint (empty = zero-shot)")
    p.add_argument("--eval", action="store_true")
    p.add_argument("--topk", default=10, type=int)
    p.add_argument("--num_clips", default=4, type=int)
    p.add_argument("--cpn", action="store_true")
    p.add_argument("--alpha", default=[0.0, 0.0], type=float, nargs="+", help="CPN weights (t2v, v2t)")
    p.add_argument("--c", default=[0.0, 0.0, 0.0, 0.0], type=float, nargs="+", help="ensemble weights")
    p.add_argument("--lora_r", default=8, type=int)
    p.add_argument("--lora_alpha", default=32, type=int)
    # training flags (main.py:33-43, 62-65), used without --eval
    p.add_argument("--batch_size", default=64, type=int, help="batch size per GPU")
    p.add_argument("--epochs", default=5, type=int)
    p.add_argument("--accum_iter", default=1, type=int)
    p.add_argument("--weight_decay", default=0.05, type=float)
    p.add_argument("--lr", default=None, type=float, help="absolute learning rate")
    p.add_argument("--min_lr", default=0.0, type=float)
    p.add_argument("--warmup_epochs", default=40, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--lora_drop", default=0.05, type=float)
    # flags of the reference's launcher: accepted so that its command lines parse, not used
    for flag, kw in (("--device", dict(default="cuda")), ("--world_size", dict(default=1, type=int)), ("--local_rank", dict(default=-1, type=int)),
                     ("--dist_on_itp", dict(action="store_true")), ("--dist_url", dict(default="env://"))):
        p.add_argument(flag, help="accepted for compatibility with the reference's command lines; unused", **kw)
    p.add_argument("--lora_mode", default="apart", choices=["apart", "merge"],
                   help="how --resume's LoRA adapters enter the scoring path.  apart (default): kept as separate matrices, y = W x + (alpha / r) B (A x), as the "
                        "reference evaluates them (main.py:96-105) -- the rank-r term rides in the base GEMM's accumulation (64 extra K columns on q/k/v/o_proj, lm_head "
                        "and the projector MLPs: about +1 %% time).  merge: W + (alpha / r) B A folded into the engine's 16-bit weight at load time (no per-call cost; the sum "
                        "is rounded to the engine's format: fine in fp16 for a bf16 base checkpoint, 8 %% of the update in bf16, most of it in e4m3)")
    p.add_argument("--allow_partial_resume", action="store_true", help="load a resume file that lacks some of the expected adapters (warn instead of fail)")
    # engine-side options
    p.add_argument("--dtype", default=None, choices=["f16", "bf16", "f8"])
    p.add_argument("--max_tokens", default=32768, type=int, help="packed tokens per engine call")
    p.add_argument("--f8_mask", default=None, type=int,
                   help="fp8 mode: which GEMMs take e4m3 operands (bit 0 qkv, 1 o_proj, 2 gate|up, 3 down, 4 lm_head).  Default 31 (all) on a base checkpoint, "
                        "12 (the MLP only) when --resume names a fine-tuned checkpoint: LoRA adapts q/k/v/o_proj and lm_head, whose merged rank-8 update is below one "
                        "e4m3 step of the base weight -- those GEMMs stay in fp16, the MLP (87 %% of a layer's flops, not adapted) runs in fp8")
    p.add_argument("--vtg_precise", default="auto", choices=["auto", "none", "full"],
                   help="compensated (hi + lo) activations on the VTG calls.  auto (default): measured on the loaded checkpoint before the first pass -- up to 256 pairs of the "
                        "evaluation are scored plain and fully compensated (which sits at 2e-6 .. 1e-4 of the fp32 reference) and plain is kept if its largest deviation, 4.5 x its "
                        "RMS deviation and the largest deviation predicted for the whole evaluation's entries are inside 1e-3 (PairScorer.calibrate_vtg; the table is printed; measured "
                        "again whenever weights or adapters change).  none = plain 16-bit: the reference's own numerics, and what auto keeps on an fp16 engine unless the "
                        "checkpoint has massive activations (tests/golden/sink.npz, heavy7b.npz: plain fp16 is 3 - 5e-3 from the fp32 result there).  full = every activation "
                        "(fp16 engines: second pass on the e2m3 MFMA, 0.67x the plain rate; the mode in which a bf16 engine holds 1e-3 at 7B depth, at 0.5x)")
    p.add_argument("--tvg_precise", default="auto", choices=["auto", "attn", "full"],
                   help="how much of the TVG calls' MLP branch runs compensated (their embeddings, QKV, attention, o_proj and head always do on a 16-bit engine).  auto (default): "
                        "measured like --vtg_precise auto, on the TVG likelihood and prior of up to 256 pairs; attn = MLP plain (1.6x faster than full), "
                        "full = everything (what weights with massive residual channels need: tests/golden/heavy7b.npz)")
    p.add_argument("--second_pass", default=None, choices=["e2m3", "16bit", "auto"],
                   help="what the compensated calls' second walk over K runs in.  e2m3: the block-scaled MFMA on 6-bit operand tiles (fp16 engines: the default; bf16 engines: "
                        "opt-in, 0.69x the plain rate with about one fp16 rounding's accuracy).  16bit: a second walk in the engine's own format (bf16 engines: the default and the "
                        "parity mode, 0.5x the plain rate at 1 - 3e-6).  auto (bf16 engines): measured on the loaded checkpoint like --vtg_precise auto -- the e2m3 form is kept "
                        "when its scores stay inside the bar of the 16-bit form's on the evaluation's own calibration pairs")
    p.add_argument("--masked_query_zero", action="store_true",
                   help="PARITY-UNPINNED: masked query positions write a zero attention output, as the reference's flash-attention-2 class does (modeling_qwen2_flash.py:526-563) "
                        "where its eager / SDPA classes -- the semantics this engine's parity is pinned to -- compute them like any other row.  Changes the TVG-CPN prior only "
                        "(its first gathered row is a masked position).  For comparing against numbers produced by a reference run with flash-attn installed; "
                        "`python -m blim_amd.first_contact` reports which attention class a checkpoint's config selects")
    p.add_argument("--literal", action="store_true", help="run the reference's per-batch control flow instead of the fused PairScorer")
    p.add_argument("--compat_allreduce_offset", action="store_true")
    p.add_argument("--no_dedup", action="store_false", dest="dedup", help="score the pairs both directions share twice, as the reference does")
    p.add_argument("--shard", default=None, type=int, nargs=2, metavar=("W", "RANK"),
                   help="play rank RANK of a W-process job in this single process: the rank's own row blocks, no merge, no recall table "
                        "(timing of the 8-GPU configurations on one GPU)")
    p.add_argument("--synthetic", default=0, type=int, help="N > 0: dry run on N synthetic videos/texts (tiny model unless --synthetic_7b)")
    p.add_argument("--synthetic_7b", action="store_true")
    p.add_argument("--synthetic_same", action="store_true", help="synthetic training: train on the evaluation set itself (a run that memorises its pairs: "
                                                                 "rankings far from chance, for comparing numeric modes on R@k)")
    p.add_argument("--dump_scores", default=None, type=str, help="write the evaluation's score matrices to this .npz")
    return p


def dims_from_config(model_path: str, num_clips: int = 4):
    from .synth import ModelDims
    c = json.load(open(os.path.join(model_path, "config.json")))
    if c.get("mm_llm_compress", False):
        raise NotImplementedError("checkpoint enables PyramidDrop token compression (mm_llm_compress): outside the scoring path")
    # the one splice the scoring path builds (modeling_videochat_flash.py:209-243): video backbone features, 'spatial_*pad*' merge with no newline
    # token -> VTG rows take the clips' tokens flattened, TVG rows their per-clip mean.  Any other combination changes the rows; refuse it
    vet, mpt = c.get("vision_encode_type", "image"), c.get("mm_patch_merge_type", "flat")
    nlp, far = c.get("mm_newline_position", "nothing"), c.get("frame_aspect_ratio", "square")
    if vet != "video_image" or not (mpt.startswith("spatial") and "pad" in mpt) or nlp != "nothing" or "anyres" in far:
        raise NotImplementedError(f"checkpoint config outside the scoring path's splice: vision_encode_type={vet!r} (need 'video_image'), mm_patch_merge_type={mpt!r} "
                                  f"(need 'spatial_*pad*'), mm_newline_position={nlp!r} (need 'nothing'), frame_aspect_ratio={far!r} (no 'anyres')")
    return ModelDims(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                     num_layers=c["num_hidden_layers"], num_heads=c["num_attention_heads"], num_kv_heads=c["num_key_value_heads"],
                     rms_eps=c.get("rms_norm_eps", 1e-6), rope_theta=c.get("rope_theta", 1e6), mm_hidden_size=c.get("mm_hidden_size", 1024),
                     num_clips=num_clips)


def load_tokenizer(model_path: str):
    """main.py:94: the checkpoint's own tokenizer (Qwen2 BPE + ChatML specials)."""
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(model_path, trust_remote_code=True)


def main(args):
    import numpy as np
    import torch
    from . import distributed as D
    from . import synth
    from .modeling import BlimModel, DDPLike
    from .training_utils import val_one_epoch

    rank, world, local = D.init_distributed_mode()
    D.limit_host_threads(world)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if not args.eval and args.lr is None:
        raise SystemExit("--lr is required for training (main.py:41: absolute learning rate)")
    if not args.eval and args.dtype == "f8":
        raise SystemExit("training needs a 16-bit engine (--dtype f16 | bf16)")
    t0 = time.time()
    train_loader = None
    if args.synthetic > 0:
        dims = synth.ModelDims(num_clips=args.num_clips) if args.synthetic_7b else synth.ModelDims(
            vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64, num_clips=args.num_clips)
        model = BlimModel(dims, dtype=args.dtype)
        model.engine.init_synthetic_weights(0)
        prob = synth.make_problem(1, args.synthetic, dims, tok_per_clip=64 if args.synthetic_7b else 8, fast_video=args.synthetic > 256)
        T = torch.from_numpy
        loader = synth.ProblemLoader(prob, args.batch_size_eval)
        tokenizer = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
        # get_recall treats a matrix holding an exact 0 as "not computed" (training_utils.py:174-175); the synthetic first-stage
        # scores are sums of integers and hit 0.0 about once per 10^5 entries, real InternVideo2 similarities do not
        nz = lambda a: np.where(a == 0, np.float32(1e-6), a)
        args.iv2_scores = {"v2t": T(nz(prob.v2t_sims)), "t2v": T(nz(prob.t2v_sims))}
        if args.eval and args.resume and os.path.isfile(args.resume):     # adapters of a (synthetic) training run
            # (any other non-empty --resume only selects the fine-tuned score combination, training_utils.py:150-167)
            if args.lora_mode == "apart":
                from .checkpoint import apply_resume
                apply_resume(model.engine, dims, args.resume, lora_r=args.lora_r, lora_alpha=float(args.lora_alpha), strict_resume=not args.allow_partial_resume)
            else:                                                         # merged on the device by the trainer's own merge kernel
                from .training import Trainer
                tr = Trainer(model.engine, lora_r=args.lora_r, lora_alpha=float(args.lora_alpha), lora_dropout=0.0)
                tr.load_checkpoint_state(torch.load(args.resume, map_location="cpu", weights_only=False))
                tr.merge_into_engine()
                tr.close()
        if not args.eval:     # synthetic training set: other videos / captions of the same generator, this rank's share
            tprob = prob if args.synthetic_same else \
                synth.make_problem(2 + rank, max(args.batch_size, args.synthetic), dims, tok_per_clip=64 if args.synthetic_7b else 8, fast_video=args.synthetic > 256)
            train_loader = synth.ProblemLoader(tprob, args.batch_size)
    else:
        from .checkpoint import load_checkpoint, summarize_report
        from .dataloader import load_data
        tokenizer = load_tokenizer(args.model_path)
        dims = dims_from_config(args.model_path, args.num_clips)
        cfg_json = json.load(open(os.path.join(args.model_path, "config.json")))
        if cfg_json.get("tokenizer_padding_side", "right") != "right":
            # modeling_videochat_flash.py:472-485: with "left" the reference left-pads the spliced rows and the decoder then numbers positions from the
            # pad (retrieval_utils.py:93 passes no position_ids), so a row's score depends on its batch neighbours; only the default is built
            raise NotImplementedError(f"config.tokenizer_padding_side = {cfg_json['tokenizer_padding_side']!r}: only 'right' (the reference's default) is supported")
        model = BlimModel(dims, dtype=args.dtype, tokenizer_model_max_length=cfg_json.get("tokenizer_model_max_length"))   # modeling_videochat_flash.py:452
        # evaluation merges the resume file's adapters at load time; training keeps the base weights pristine (the trainer owns the adapters)
        report = load_checkpoint(model.engine, dims, args.model_path, (args.resume or None) if args.eval else None, lora_r=args.lora_r,
                                 lora_alpha=args.lora_alpha, strict_resume=not args.allow_partial_resume, lora_mode=args.lora_mode)
        if rank == 0:
            print("weights: " + summarize_report(report))
        loader = load_data(args, tokenizer=tokenizer, split="test")
        if not args.eval:
            train_loader = load_data(args, tokenizer=tokenizer, split="train")
    if getattr(args, "masked_query_zero", False):
        model.masked_query_zero = True
        print("[note] --masked_query_zero: flash-attention-2 semantics for masked query rows (parity-unpinned; the default, eager / SDPA semantics, is what the goldens pin)")
    if getattr(args, "second_pass", None) and model.engine.can_precise:
        model.second_pass = args.second_pass
    if model.engine.can_precise:
        model.tvg_precise = args.tvg_precise
    if args.vtg_precise is not None and model.engine.can_precise:
        model.vtg_precise = None if args.vtg_precise == "none" else args.vtg_precise        # "auto": resolved by evaluation() on the loaded weights
    if model.engine.dtype == "f8":
        finetuned_file = bool(args.resume) and os.path.isfile(args.resume)
        mask = args.f8_mask if args.f8_mask is not None else (12 if finetuned_file else 31)
        model.engine.set_option("f8_mask", mask)
        if rank == 0 and finetuned_file:
            print(f"fp8 mode on a fine-tuned checkpoint: f8_mask = {mask}, lora_mode = {args.lora_mode} "
                  + ("(adapters apart: the adapted projections q/k/v/o and lm_head run in fp16 with the adapters as separate 16-bit operands whatever the mask says; "
                     "the MLP, 87 % of a layer's flops and not adapted, runs in e4m3)" if args.lora_mode == "apart" else
                     "(MLP only: the adapted projections q/k/v/o and lm_head stay in fp16, so the merged adapters survive)" if mask == 12 else
                     "(adapters merged BEFORE the e4m3 quantisation: a rank-8 update is mostly below one e4m3 step of the base weight -- 2 - 4 points of R@1 lost, "
                     "profiles/r03_modes_trained_weights.md)"))
    if rank == 0:
        print(f"model + data ready in {time.time() - t0:.1f}s ({model.engine.dtype}, world size {world})")
    if not args.eval:
        results = train_loop(args, model, train_loader, loader, tokenizer, device, rank, world)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        model.engine.close()
        return results
    torch.cuda.synchronize()
    t1 = time.time()
    torch.cuda.reset_peak_memory_stats()
    results = val_one_epoch(DDPLike(model), loader, None, device, 0, None, tokenizer=tokenizer, args=args)
    torch.cuda.synchronize()
    if rank == 0:
        st = getattr(args, "_eval_stats", {})
        dt = time.time() - t1 - float(getattr(args, "_dump_seconds", 0.0))       # (--dump_scores compresses 8 N x N matrices to disk: not part of an evaluation)
        free_b, total_b = torch.cuda.mem_get_info()
        # executed GEMM FLOPs of this process's engine calls (retrieval_utils.executed_flops) against the dense MFMA peaks: the compensated calls' e2m3 second pass at
        # the fp6 peak, an fp8 engine's calls at the fp8 peak, the rest at the 16-bit one (MI355X_MICROARCH.md: 2.5 / 5 / 10 PFLOP/s)
        fl, f6 = float(st.get("executed_flops", 0.0)), float(st.get("executed_flops_lo6", 0.0))
        at_peak = ((fl - f6) / (5.0e15 if model.engine.dtype == "f8" else 2.5e15) + f6 / 1.0e16)
        print(f"evaluation: {st.get('pairs_requested', 0)} (query, candidate) pairs of this rank's row blocks, {st.get('pairs_scored', 0)} scored by the engine "
              f"(the rest shared between directions), in {dt:.2f}s = {st.get('pairs_requested', 0) / dt:.0f} pairs/s per process "
              f"(world {st.get('world', world)}, host planning and loading included); executed {fl / 1e12:.1f} TFLOP = {at_peak / dt:.3f} of the MFMA peak; "
              f"device memory in use {(total_b - free_b) / 2**30:.1f} GiB, torch peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        if st.get("host_marks"):
            print("host marks (stage, seconds since the evaluation began; device work is asynchronous, so a stage's time is its planning + launching): "
                  + ", ".join(f"{k} {v:.2f}" for k, v in st["host_marks"]))
        if st.get("calibration_whole_sample"):
            # `--shard W r` on its own: nobody to gather the calibration sample from, so this process measured all of it (the decision is the job's); a real rank scores 1 / W
            cal, Wj = float(st.get("calibration_seconds", 0.0)), int(st.get("world", 1))
            print(f"calibration: {cal:.2f}s for the job's whole sample (this process has no peers); at a rank's 1/{Wj} share of it the evaluation above takes "
                  f"{dt - cal * (1.0 - 1.0 / Wj):.2f}s = {st.get('pairs_requested', 0) / (dt - cal * (1.0 - 1.0 / Wj)):.0f} pairs/s per process")
    if args.shard is not None:
        model.engine.close()
        return None                                           # one rank's share: the matrices are partial, no recall table
    if rank == 0:
        import pandas as pd
        os.makedirs(args.output_dir, exist_ok=True)
        table = pd.DataFrame(results).T                                                    # main.py:170-173
        print(table.to_string())
        with open(os.path.join(args.output_dir, "log.txt"), "a") as f:
            f.write(table.to_string() + "\n")
    if D.is_dist_avail_and_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    model.engine.close()
    return results


def train_loop(args, model, train_loader, val_loader, tokenizer, device, rank: int, world: int):
    """main.py:118-195: LoRA set-up, optional resume of adapters / optimizer / scaler, then per epoch train -> save -> evaluate."""
    import datetime
    import pandas as pd
    import torch
    from .modeling import DDPLike
    from .training import Trainer, save_model, train_one_epoch
    from .training_utils import val_one_epoch
    vh = None
    if not args.synthetic:                                                                           # visual_head of the base checkpoint, when it has one
        from .checkpoint import open_base_checkpoint
        keys, get = open_base_checkpoint(args.model_path)
        vh = get("visual_head.weight") if "visual_head.weight" in keys else None
    trainer = Trainer(model.engine, lora_r=args.lora_r, lora_alpha=float(args.lora_alpha), lora_dropout=args.lora_drop, seed=args.seed,
                      weight_decay=args.weight_decay, visual_head=vh)                              # seed: same adapters on every rank (DDP broadcasts rank 0's)
    if args.resume:                                                                                  # util/misc.py:303-316
        ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
        trainer.load_checkpoint_state(ckpt)
        # The reference's load_model restores the weights only and restarts at --start_epoch (util/misc.py:303-316).  A file written by
        # THIS trainer also carries its optimizer moments (flat layout): only then is the run continued after the saved epoch.
        opt = ckpt.get("optimizer")
        if isinstance(opt, dict) and "exp_avg" in opt and "epoch" in ckpt:
            args.start_epoch = int(ckpt["epoch"]) + 1
        elif rank == 0:
            print(f"resume: weights only (no optimizer state in this repo's layout): starting at --start_epoch {args.start_epoch}")
    if args.start_epoch >= args.epochs and rank == 0:
        print(f"warning: nothing to train: start epoch {args.start_epoch} >= --epochs {args.epochs}")
    eff = args.batch_size * args.accum_iter * world                                                  # main.py:133-139
    if rank == 0:
        n_train = sum(int(__import__("numpy").prod(s)) for _, s in trainer.layout.values())
        print(f"Trainable params: {n_train:,}")
        print("base lr: %.2e" % (args.lr * 256 / eff)); print("actual lr: %.2e" % args.lr)
        print("accumulate grad iterations: %d" % args.accum_iter); print("effective batch size: %d" % eff)
        print(f"Start training for {args.epochs} epochs")
    start, best_r1, results = time.time(), 0.0, None
    for epoch in range(args.start_epoch, args.epochs):
        sampler = getattr(train_loader, "sampler", None)
        if world > 1 and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)                                                                 # main.py:160-161
        stats = train_one_epoch(trainer, train_loader, epoch, args, world_size=world, log=print if rank == 0 else (lambda *_: None))
        if rank == 0:
            save_model(args, epoch, trainer, name=f"epoch{epoch}")                                   # main.py:165
        if args.lora_mode == "merge":
            trainer.merge_into_engine()                                                              # W + alpha/r * B A folded into the scoring weights (rounded to the engine's format)
        else:
            trainer.adapters_into_engine()                                                           # adapters apart on the pristine base weights, like the reference's live peft model
        model.clear_cache()                                                                          # projector outputs cached under the previous weights
        results = val_one_epoch(DDPLike(model), val_loader, None, device, epoch, None, tokenizer=tokenizer, args=args)
        if rank == 0:
            cur = results["blim"]["t2v_r1"] + results["blim"]["v2t_r1"]                              # main.py:176-181
            if best_r1 < cur:
                best_r1 = cur
                save_model(args, epoch, trainer, name="checkpoint_best")
            table = pd.DataFrame(results).transpose().to_string()
            log_stats = {"epoch": epoch, **{f"train_{k}": v for k, v in stats.items()}, **{f"val_{k}": v for k, v in results.items()}}
            os.makedirs(args.output_dir, exist_ok=True)
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n" + table + "\n")
            print("\n" + table)
    if rank == 0:
        print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start)))))
    trainer.close()
    return results


if __name__ == "__main__":
    main(get_args_parser().parse_args())
