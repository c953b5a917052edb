"""Offline video-feature extraction with the reference's flags (extract.py:12-21) on the MI355X vision encoder.

    python -m blim_amd.extract --dataset MSRVTT --num_chunk 8 --chunk_idx 0 [--model_path ./pretrained/VideoChat-Flash-Qwen2-7B_res448]

Reads ./data/<DS>/videos/* (LSMDC: ./data/LSMDC/videos/*/*), samples 16 frames per video (np.linspace(0, vlen - 2, 16); DiDeMo videos
are cut at 30 s: extract.py:47-54), preprocesses them as UMTImageProcessor does, runs 4 clips x 4 frames through the vision tower and
ToMe on the GPU and writes ./data/<DS>/features/<vid>.pth = fp16 [4, 64, 1024] -- the files blim_amd.dataloader (and the reference's
dataloader, base_dataset.py:23-31) read.  Chunking over processes as in the reference (--num_chunk / --chunk_idx: one process per GPU).

Video decoding needs `decord` (as in the reference); where it is not installed, pre-decoded frames are read instead:
./data/<DS>/frames/<vid>.npy = uint8 [n, H, W, 3] with the frames ALREADY sampled (--frames_dir overrides the directory).
`--synthetic SEED` uses seeded random tower weights instead of the checkpoint (dry run without downloads)."""
from __future__ import annotations

import argparse
import glob
import os
import time

import numpy as np


def get_args_parser():
    p = argparse.ArgumentParser(description="Video feature extractor (UMT-L + ToMe) on the MI355X engine")
    p.add_argument("--dataset", default="DiDeMo", type=str, choices=["DiDeMo", "ActivityNet", "LSMDC", "MSRVTT"])
    p.add_argument("--model_path", type=str, default="./pretrained/VideoChat-Flash-Qwen2-7B_res448")
    p.add_argument("--num_frames", type=int, default=16)
    p.add_argument("--num_chunk", required=True, type=int)
    p.add_argument("--chunk_idx", required=True, type=int)
    p.add_argument("--batch_size", type=int, default=1, help="videos per engine call")
    p.add_argument("--save_iter", type=int, default=10, help="accepted for compatibility (features are written as they are produced)")
    p.add_argument("--clear", action="store_true", help="clear the feature folder")
    p.add_argument("--frames_dir", default=None, help="directory of pre-decoded <vid>.npy frame stacks (used when decord is unavailable)")
    p.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    p.add_argument("--num_workers", type=int, default=4, help="host threads decoding + preprocessing ahead of the GPU (the reference's DataLoader uses 4 workers)")
    p.add_argument("--synthetic", default=None, type=int, help="seed of synthetic tower weights (dry run)")
    return p


def video_id(path: str, dataset: str) -> str:
    """extract.py:66-69."""
    base = os.path.basename(path)
    return base[:-4] if dataset == "LSMDC" else base.split(".")[0]


def chunk_of(items, num_chunk: int, chunk_idx: int):
    """extract.py:83-90: equal chunks, the last one takes the remainder."""
    size = len(items) // num_chunk
    start = size * chunk_idx
    end = len(items) if chunk_idx == num_chunk - 1 else min(size * (chunk_idx + 1), len(items))
    return items[start:end]


def list_sources(args):
    """(video id, source path) pairs: video files when decord is importable, else pre-decoded frame stacks."""
    try:
        import decord  # noqa: F401
        have_decord = args.frames_dir is None
    except Exception:
        have_decord = False
    if have_decord:
        pat = f"./data/{args.dataset}/videos/*/*" if args.dataset == "LSMDC" else f"./data/{args.dataset}/videos/*"     # extract.py:76-79
        files = sorted(glob.glob(pat))
        return [(video_id(f, args.dataset), f) for f in files], True
    d = args.frames_dir or f"./data/{args.dataset}/frames"
    files = sorted(glob.glob(os.path.join(d, "*.npy")))
    return [(os.path.basename(f)[:-4], f) for f in files], False


def read_frames(path: str, is_video: bool, dataset: str, num_frames: int) -> np.ndarray:
    """uint8 [num_frames, H, W, 3]."""
    if not is_video:
        a = np.load(path)
        if a.shape[0] != num_frames:
            from .vision import sample_frame_indices
            a = a[sample_frame_indices(a.shape[0] + 1, num_frames).clip(0, a.shape[0] - 1)]
        return a
    from decord import VideoReader
    from .vision import sample_frame_indices
    vr = VideoReader(path, num_threads=1)
    vlen, fps = len(vr), vr.get_avg_fps()
    if vlen / float(fps) > 30 and dataset == "DiDeMo":                    # extract.py:50-52
        vlen = 30 * fps
    return vr.get_batch(sample_frame_indices(vlen, num_frames)).asnumpy()    # vlen stays a float for cut DiDeMo videos (30 * fps), as extract.py:52-54


def main(args):
    import torch
    from .vision import VisionDims, VisionEncoder, preprocess
    out_dir = f"./data/{args.dataset}/features"
    os.makedirs(out_dir, exist_ok=True)
    if args.clear:                                                        # extract.py:23-27
        for f in glob.glob(os.path.join(out_dir, "*.pth")):
            os.remove(f)
        print("clear the feature folder!!!")
    sources, is_video = list_sources(args)
    print(f"Number of videos: {len(sources)}")
    sources = chunk_of(sources, args.num_chunk, args.chunk_idx)
    print(f"num_chunk: {args.num_chunk}\nchunk_size: {len(sources):,}\nUsing Batch size: {args.batch_size}")
    dims = VisionDims()
    if args.num_frames % dims.num_frames:
        raise ValueError(f"--num_frames must be a multiple of {dims.num_frames} (clips of mm_local_num_frames frames)")
    enc = VisionEncoder(dims, dtype=args.dtype)
    if args.synthetic is not None:
        enc.init_synthetic_weights(args.synthetic)
    else:
        enc.load_checkpoint(args.model_path)
    t0, n_done = time.time(), 0
    clips = args.num_frames // dims.num_frames

    def load_one(item):                                                   # host side: decode + resize + normalise (PIL releases the GIL)
        vid, path = item
        try:
            return vid, preprocess(read_frames(path, is_video, args.dataset, args.num_frames), dims.image_size)
        except Exception as e:                                            # extract.py:73-75 skips unreadable videos
            print(f"Error loading video {path}: {e}")
            return vid, None

    import collections
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max(1, args.num_workers))
    ahead = 2 * max(1, args.num_workers) + args.batch_size               # bounded look-ahead: a preprocessed video is 19 MB
    pending, nxt = collections.deque(), 0

    def refill():
        nonlocal nxt
        while nxt < len(sources) and len(pending) < ahead:
            pending.append(pool.submit(load_one, sources[nxt])); nxt += 1

    refill()
    for lo in range(0, len(sources), args.batch_size):
        frames, vids = [], []
        for _ in range(min(args.batch_size, len(sources) - lo)):
            vid, fr = pending.popleft().result()
            refill()
            if fr is not None:
                frames.append(fr); vids.append(vid)
        if not vids:
            continue
        tome, _ = enc.encode(torch.cat(frames, dim=0))
        feats = tome.to(torch.float16).cpu().reshape(len(vids), clips, dims.tome_tokens, dims.hidden_size)
        for vid, f in zip(vids, feats):
            torch.save(f.clone(), os.path.join(out_dir, f"{vid}.pth"))  # extract.py:107-110
        n_done += len(vids)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{n_done} videos in {dt:.1f}s ({n_done / max(dt, 1e-9):.1f} videos/s, decoding and preprocessing included)")
    pool.shutdown()
    enc.close()
    return n_done


if __name__ == "__main__":
    main(get_args_parser().parse_args())
