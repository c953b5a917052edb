"""GPU: the eval driver end to end on a synthetic on-disk tree -- sharded safetensors checkpoint + config.json, a resume file
with LoRA adapters and visual_head, ./data/MSRVTT features + annotations, ./scores/msrvtt.pth -- i.e. the same files, names and
flags a user of the reference has (main.py:146-174), with the stand-in tokenizer of tests/dataset_fixture.py."""
import json
import os

import numpy as np
import pytest
import torch

from blim_amd import checkpoint as CK
from blim_amd import main as driver
from blim_amd import synth
from dataset_fixture import CAPTIONS, StubTokenizer, build_tree
from test_checkpoint import _adapters, _resume_state

pytestmark = pytest.mark.gpu


def _tree(root):
    dims = synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=1024)
    w = synth.synthetic_weights(dims, 3)
    ck = os.path.join(root, "pretrained", "tiny")
    CK.save_hf_checkpoint(w, ck, shards=2, dtype="bf16")
    json.dump({"vocab_size": dims.vocab_size, "hidden_size": dims.hidden_size, "intermediate_size": dims.intermediate_size,
               "num_hidden_layers": dims.num_layers, "num_attention_heads": dims.num_heads, "num_key_value_heads": dims.num_kv_heads,
               "rms_norm_eps": 1e-6, "rope_theta": 1e6, "mm_hidden_size": 1024, "mm_llm_compress": False, "vision_encode_type": "video_image",
               "mm_patch_merge_type": "spatial_nopad", "mm_newline_position": "nothing", "tokenizer_padding_side": "right", "tokenizer_model_max_length": 32768},
              open(os.path.join(ck, "config.json"), "w"))
    build_tree(root, "MSRVTT")
    n = len(CAPTIONS)
    rs = np.random.RandomState(0)
    sims = rs.randn(n, n).astype(np.float32) + 3 * np.eye(n, dtype=np.float32)
    os.makedirs(os.path.join(root, "scores"), exist_ok=True)
    for name in ("msrvtt.pth", "msrvtt_zeroshot.pth"):
        torch.save({"v2t": torch.from_numpy(sims), "t2v": torch.from_numpy(sims.T.copy())}, os.path.join(root, "scores", name))
    os.makedirs(os.path.join(root, "checkpoint"), exist_ok=True)
    ad = _adapters(dims, 5, CK.expected_adapters(dims))
    vh = (np.random.RandomState(7).randn(1024, dims.hidden_size) * 0.02).astype(np.float32)
    torch.save(_resume_state(ad, vh), os.path.join(root, "checkpoint", "msrvtt.pth"))
    return ck


def _run(argv):
    return driver.main(driver.get_args_parser().parse_args(argv))


def test_eval_driver_on_a_synthetic_tree(tmp_path, monkeypatch, capsys):
    ck = _tree(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(driver, "load_tokenizer", lambda path: StubTokenizer())
    base = ["--eval", "--dataset", "MSRVTT", "--model_path", ck, "--topk", "4", "--batch_size_eval", "3", "--num_workers", "0",
            "--output_dir", str(tmp_path / "out")]
    tuned = base + ["--resume", "./checkpoint/msrvtt.pth", "--cpn", "--alpha", "0.4", "0.8", "--c", "0.3", "0.6", "0.9", "0.7"]
    fused = _run(tuned)
    literal = _run(tuned + ["--literal"])
    zero = _run(base)                                                  # zero-shot: VTG passes only, ./scores/msrvtt_zeroshot.pth
    out = capsys.readouterr().out
    assert "model + data ready" in out
    assert fused == literal                                            # fused PairScorer and the reference's per-batch control flow agree on every R@k
    for res in (fused, zero):
        assert set(res) >= {"t2v", "v2t"} or len(res) > 0
        flat = [v for d in res.values() for v in (d.values() if isinstance(d, dict) else [d])]
        assert all(0.0 <= float(x) <= 100.0 for x in flat)
    log = open(tmp_path / "out" / "log.txt").read()
    assert log.count("R@1") >= 3 or log.count("R1") >= 3 or len(log) > 100


def test_two_ranks_give_the_single_process_recall_table(tmp_path):
    """One process per 'GPU' x 2 (both on cuda:0, gloo in place of RCCL so that two ranks can share the device): row-block sharding,
    text-sharded prior and the fused all-gather with the REAL engine reproduce the single-process recall table exactly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--eval", "--synthetic", "19", "--cpn", "--resume", "x", "--alpha", "0.4", "0.8", "--c", "0.3", "0.6", "0.9", "0.7", "--topk", "5"]
    env = dict(os.environ, PYTHONPATH=root)
    r1 = subprocess.run([sys.executable, "-m", "blim_amd.main"] + common + ["--output_dir", str(tmp_path / "w1"), "--dump_scores", str(tmp_path / "w1.npz")], cwd=root, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    env2 = dict(env, BLIM_DIST_BACKEND="gloo", BLIM_FORCE_DEVICE="0")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29541", "-m", "blim_amd.main"] + common + ["--output_dir", str(tmp_path / "w2"), "--dump_scores", str(tmp_path / "w2.npz")], cwd=root, env=env2,
                        capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    t1, t2 = open(tmp_path / "w1" / "log.txt").read(), open(tmp_path / "w2" / "log.txt").read()
    assert "blim" in t1 and t1 == t2
    # ... and every score matrix bit for bit: pair ownership, the text-sharded prior and the all-gathered TVG clip features (each rank projects its own
    # block of videos, PairScorer.share_tvg_feats) change who computes what, not what is computed
    import numpy as np
    a, b = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w2.npz")
    assert set(a.files) == set(b.files) and len(a.files) == 8
    for k in a.files:
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k


def _same_npz(a, b, rtol=0.0):
    assert set(a.files) == set(b.files) and len(a.files) == 8
    for k in a.files:
        if rtol:
            assert np.allclose(a[k], b[k], rtol=rtol, atol=0.0) and np.array_equal(a[k] == -100.0, b[k] == -100.0), k
            continue
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k


@pytest.mark.parametrize("n, extra", [(19, []), (19, ["--no_dedup"]), (5, [])], ids=["N19-short-and-empty-blocks", "N19-six-passes", "N5-fewer-rows-than-ranks"])
def test_eight_ranks_through_the_real_engine_give_the_single_process_matrices(tmp_path, n, extra):
    """The 8-rank job the driver's scaling run starts, with the REAL engine in every rank: eight processes on cuda:0 (gloo in place of RCCL so that they can share the
    device), `--cpn`, both settings of pair ownership.  N = 19 at W = 8 is the reference's `step = N // W + 1 = 3` (retrieval_utils.py:213-215): six blocks of three
    rows, one of ONE row and an EMPTY one; N = 5: three ranks own nothing at all.  Calibration (`--vtg_precise / --tvg_precise auto`) is sharded over the eight ranks and
    gathered, the TVG clip features are projected per block and all-gathered, the prior is text-sharded, all matrices merge in one all-gather -- and the recall table
    and every score matrix must equal the one-process run bit for bit (/root/reference/retrieval_utils.py:213-215, 233-235, 252-262).  `--no_dedup` keeps the reference's
    six row-sharded passes: there a v2t TVG pair shares its merged sequence (PairScorer._plan_tvg: the candidates of one text, segments of ONE sequence) with whichever
    other candidates of that text fall into the RANK's row block, so its attention sums associate differently from the one-process run's: equal to 2e-6 relative (observed:
    8 ulp on one entry), same computed / not-computed pattern, same recall table."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--eval", "--synthetic", str(n), "--cpn", "--resume", "x", "--alpha", "0.4", "0.8", "--c", "0.3", "0.6", "0.9", "0.7", "--topk", "4"] + extra
    env = dict(os.environ, PYTHONPATH=root)
    r1 = subprocess.run([sys.executable, "-m", "blim_amd.main"] + common + ["--output_dir", str(tmp_path / "w1"), "--dump_scores", str(tmp_path / "w1.npz")], cwd=root, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    env8 = dict(env, BLIM_DIST_BACKEND="gloo", BLIM_FORCE_DEVICE="0", OMP_NUM_THREADS="1")
    r8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                         "--master-port", str({(19, 0): 29571, (19, 1): 29573, (5, 0): 29575}[(n, len(extra))]), "-m", "blim_amd.main"] + common + ["--output_dir", str(tmp_path / "w8"), "--dump_scores", str(tmp_path / "w8.npz")],
                        cwd=root, env=env8, capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stderr[-3000:]
    assert "world size 8" in r8.stdout
    t1, t8 = open(tmp_path / "w1" / "log.txt").read(), open(tmp_path / "w8" / "log.txt").read()
    assert "blim" in t1 and t1 == t8
    _same_npz(np.load(tmp_path / "w1.npz"), np.load(tmp_path / "w8.npz"), rtol=2e-6 if extra else 0.0)


def test_bench_gpus_8_on_one_gpu_keeps_the_json_contract(tmp_path):
    """`python bench.py --gpus 8` as the driver's scaling run invokes it -- the parent starts torch.distributed.run with eight ranks -- here with all eight on cuda:0
    over gloo (the 7B replica is 15 GB: eight fit one MI355X): one JSON line from rank 0, `n_gpus: 8`, whole-job pairs/s = 8 ranks' pairs / the slowest rank's time,
    weak scaling, the strong-scaling leg's fixed job run by all eight ranks with its collectives."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, BLIM_DIST_BACKEND="gloo", BLIM_FORCE_DEVICE="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--queries", "8", "--strong-n", "96"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 8 * d["config"]["pairs_per_step_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]        # whole-job aggregate over the 8 ranks
    assert "cpu_baseline" not in d and "roofline" in d and d["roofline"]["frac"] > 0
    ss = d["strong_scaling"]
    assert ss["world"] == 8 and ss["finite"] is True and ss["pairs"] == 6 * 96 * 16 and "emulated_world" not in ss


def _rccl_env(root, port):
    """One rank, world size 1, backend nccl (= RCCL) on cuda:0, every collective branch forced (blim_amd/distributed.py:force_collective)."""
    env = dict(os.environ, PYTHONPATH=root, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               BLIM_FORCE_COLLECTIVE="1", NCCL_DEBUG="VERSION", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("BLIM_DIST_BACKEND", None); env.pop("BLIM_FORCE_DEVICE", None)
    return env


def test_rccl_runs_every_collective_of_the_evaluation_at_world_size_one(tmp_path, capsys):
    """The multi-GPU flow's collectives on the REAL backend: a process group with backend 'nccl' (RCCL) at world size 1 on cuda:0 and
    BLIM_FORCE_COLLECTIVE=1, so that the evaluation runs merge_row_blocks_many (the all-gather of the score blocks), the text-sharded prior's
    gather and PairScorer.share_tvg_feats on device tensors through RCCL -- the calls an 8-GPU job makes (util/misc.py:199-229,
    retrieval_utils.py:252-262) -- and every score matrix must equal the run without a process group bit for bit.  NCCL_DEBUG=VERSION makes
    the library announce itself: the log must carry its version line."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--eval", "--synthetic", "19", "--cpn", "--resume", "x", "--alpha", "0.4", "0.8", "--c", "0.3", "0.6", "0.9", "0.7", "--topk", "5"]
    r1 = subprocess.run([sys.executable, "-m", "blim_amd.main"] + common + ["--output_dir", str(tmp_path / "w1"), "--dump_scores", str(tmp_path / "w1.npz")], cwd=root,
                        env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "blim_amd.main"] + common + ["--output_dir", str(tmp_path / "rc"), "--dump_scores", str(tmp_path / "rc.npz")], cwd=root,
                        env=_rccl_env(root, 29561), capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    log = r2.stdout + r2.stderr
    m = re.search(r"(RCCL|NCCL) version[^\n]*", log)
    with capsys.disabled():
        print("\n[rccl world-size-1 evaluation] " + (m.group(0) if m else "no version line; log tail: " + log[-400:]))
    assert m, log[-2000:]
    a, b = np.load(tmp_path / "w1.npz"), np.load(tmp_path / "rc.npz")
    assert set(a.files) == set(b.files) and len(a.files) == 8
    for k in a.files:
        assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), k
    assert open(tmp_path / "w1" / "log.txt").read() == open(tmp_path / "rc" / "log.txt").read()


def test_rccl_runs_the_bench_collectives_at_world_size_one(tmp_path, capsys):
    """bench.py under the same environment: the timed region's all_gather of the score rows, the MAX all-reduce of the step time, the barriers and the
    strong-scaling leg's collectives all go through RCCL on cuda:0; the JSON line keeps its contract."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--queries", "8", "--strong-n", "96", "--no-cpu-baseline"], cwd=root,
                       env=_rccl_env(root, 29563), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    m = re.search(r"(RCCL|NCCL) version[^\n]*", r.stdout + r.stderr)
    with capsys.disabled():
        print("\n[rccl world-size-1 bench] " + (m.group(0) if m else "no version line"))
    assert m, (r.stdout + r.stderr)[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["strong_scaling"]["finite"] is True and d["strong_scaling"]["world"] == 1
    assert "RCCL" in d["config"]["parallelism"]


def test_bench_watchdog_keeps_the_headline_line_when_a_later_leg_does_not_finish(tmp_path):
    """bench.py --leg-timeout: the legs behind the timed steps (the strong-scaling leg's collectives have never met more than one GPU) must not be able to cost the headline.
    With a 1-second limit the watchdog fires inside the strong-scaling leg: still exactly one JSON line, the measured headline in it, the timeout recorded where the leg's
    object would be, exit code 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--queries", "8", "--strong-n", "1000", "--leg-timeout", "1"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["steps"] == 2 and d["roofline"]["frac"] > 0 and "watchdog" in d["strong_scaling"]["error"] and "cpu_baseline" not in d
    assert "watchdog" in r.stderr


def test_bench_prints_one_json_line_with_the_contract_fields(tmp_path):
    """bench.py's output contract: exactly one JSON line on stdout with the driver's keys, the roofline and (at N = 1) the CPU
    baseline objects.  Small workload (8 queries) so that the test takes seconds; the numbers themselves are not checked."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--queries", "8", "--strong-n", "96"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "strong_scaling"):
        assert k in d, k
    ss = d["strong_scaling"]                                             # the fixed-size evaluation + the per-rank shares of an 8-process job
    assert ss["scaling"] == "strong" and ss["world"] == 1 and ss["pairs"] == 6 * 96 * 16 and ss["finite"] is True and ss["seconds"] > 0
    assert ss["emulated_world"] == 8 and len(ss["emulated_rank_seconds"]) == 8 and ss["predicted_seconds"] == max(ss["emulated_rank_seconds"])
    assert abs(ss["predicted_speedup"] - ss["seconds"] / ss["predicted_seconds"]) < 0.02 and ss["pairs_scored_rank0"] <= ss["pairs"]
    # the fixed job has a roofline fraction of its own (executed GEMM FLOPs of all its engine calls / time / peak), and so has every emulated rank
    # the e2m3 second pass of the compensated (TVG) calls is priced at the fp6 peak, everything else at the 16-bit one
    at_peak = (ss["executed_tflop_job"] - ss["executed_tflop_job_e2m3_pass"]) / 2500.0 + ss["executed_tflop_job_e2m3_pass"] / 10000.0
    assert 0 < ss["executed_tflop_job_e2m3_pass"] < 0.2 * ss["executed_tflop_job"]
    assert 0 < ss["frac_mfma_peak"] < 1 and abs(ss["frac_mfma_peak"] - at_peak / ss["seconds"]) < 3e-3 and len(ss["emulated_rank_frac_mfma_peak"]) == 8
    assert abs(ss["executed_tflop_job"] / ss["seconds"] - ss["executed_tflops_per_gpu"]) < 0.02 * ss["executed_tflops_per_gpu"] + 0.2
    # the headline step's executed FLOPs leave out the last layer's o_proj / MLP on the rows nobody reads (prune_last)
    tok, pairs_ = d["config"]["tokens_per_step_per_gpu"], d["config"]["pairs_per_step_per_gpu"]
    assert d["executed_gflop_per_pair"] * pairs_ * 1e9 < 28 * 466092032 * tok + 1089994752 * 32 * pairs_
    assert d["metric"].startswith("candidate-pairs/sec") and d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f16"
    assert "workload" in d["config"] and "model" not in d["config"]
    cm = d["compensated_mode"]              # the same step with fully compensated VTG calls, reported beside the headline (never as `value`)
    assert cm["vtg_compensated"] == "full" and cm["finite"] is True and 0.4 * d["value"] < cm["value"] < 0.9 * d["value"] and cm["second_pass"].startswith("e2m3")
    assert 0 < cm["frac_mfma_peak_whole_step"] < 1
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "pairs/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["tiny_config_evaluation"]["agree_1e-3"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_bench_gpus_n_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver invokes it (no launcher around it): the parent spawns the two ranks, rank 0
    prints the single JSON line with n_gpus = 2 and the all-gather inside the timed region.  With one visible device both ranks share
    cuda:0 (gloo stands in for RCCL, which needs one device per rank); with >= 2 devices it is the real RCCL path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env.update(BLIM_DIST_BACKEND="gloo", BLIM_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", "8", "--strong-n", "96"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["pairs_per_step_per_gpu"] == 8 * 8 and "cpu_baseline" not in d      # 8 queries x top-min(16, 8 texts)
    ss = d["strong_scaling"]                                             # the same fixed-size evaluation split over the two ranks, merge inside
    assert ss["scaling"] == "strong" and ss["world"] == 2 and ss["pairs"] == 6 * 96 * 16 and ss["finite"] is True and "predicted_speedup" not in ss


def test_training_driver_on_a_synthetic_tree(tmp_path, monkeypatch, capsys):
    """main.py without --eval (main.py:155-195): two epochs of LoRA fine-tuning on the on-disk tree through the real train dataloader,
    epoch checkpoints in the reference's format, and `--eval --resume <epoch file>` (adapters merged by blim_amd/checkpoint.py at load)
    reproducing the in-training validation (adapters merged by the trainer's kernel)."""
    ck = _tree(str(tmp_path))
    fname, annos = __import__("dataset_fixture").annotations("MSRVTT")
    train = [a for i, a in enumerate(annos) if i != 2] * 2                      # the video without a feature file is dropped by the train split anyway
    json.dump(train, open(os.path.join(str(tmp_path), "data", "MSRVTT", "msrvtt_ret_train.json"), "w"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(driver, "load_tokenizer", lambda path: StubTokenizer())
    out_dir = str(tmp_path / "ft")
    common = ["--dataset", "MSRVTT", "--model_path", ck, "--topk", "4", "--batch_size_eval", "3", "--num_workers", "0", "--output_dir", out_dir,
              "--cpn", "--alpha", "0.4", "0.8", "--c", "0.3", "0.6", "0.9", "0.7"]
    last = _run(common + ["--lr", "2e-3", "--epochs", "2", "--warmup_epochs", "1", "--batch_size", "4", "--lora_drop", "0.05", "--seed", "1"])
    out = capsys.readouterr().out
    assert "Trainable params" in out and "Training time" in out
    logs = [json.loads(l) for l in open(os.path.join(out_dir, "log.txt")) if l.startswith("{")]
    assert [l["epoch"] for l in logs] == [0, 1]
    assert logs[1]["train_loss"] < logs[0]["train_loss"], logs                   # it learns (12 samples, lr 2e-3)
    for name in ("epoch0.pth", "epoch1.pth", "checkpoint_best.pth"):
        assert os.path.exists(os.path.join(out_dir, name)), name
    ckpt = torch.load(os.path.join(out_dir, "epoch1.pth"), map_location="cpu", weights_only=False)
    keys = list(ckpt["model"])
    assert "base_model.model.visual_head.weight" in keys
    assert "base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in keys
    assert "base_model.model.model.mm_projector.tvg_mlp.base_model.model.2.lora_B.default.weight" in keys
    assert ckpt["epoch"] == 1 and "scaler" in ckpt and ckpt["optimizer"]["step"] == 6
    again = _run(["--eval", "--resume", os.path.join(out_dir, "epoch1.pth")] + common)                              # adapters kept apart (the default)
    assert again == last, (again, last)
    merged = _run(["--eval", "--resume", os.path.join(out_dir, "epoch1.pth"), "--lora_mode", "merge"] + common)      # ... and merged on the host at load
    assert merged == last, (merged, last)


def test_two_ranks_train_with_averaged_gradients(tmp_path):
    """Fine-tuning with one process per 'GPU' x 2 (both on cuda:0, gloo in place of RCCL): each rank steps on its own synthetic batches,
    the flat gradient buffer is averaged with one all-reduce, so both ranks hold the same adapters and print the same validation."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, BLIM_DIST_BACKEND="gloo", BLIM_FORCE_DEVICE="0")
    out = str(tmp_path / "ft2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
                        "-m", "blim_amd.main", "--synthetic", "8", "--lr", "1e-3", "--epochs", "2", "--warmup_epochs", "1", "--batch_size", "4", "--topk", "4",
                        "--cpn", "--alpha", "0.4", "0.8", "--c", "0.3", "0.6", "0.9", "0.7", "--output_dir", out], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "effective batch size: 8" in r.stdout and "Training time" in r.stdout
    logs = [json.loads(l) for l in open(os.path.join(out, "log.txt")) if l.startswith("{")]
    assert [l["epoch"] for l in logs] == [0, 1] and all(np.isfinite(l["train_loss"]) for l in logs)
    assert os.path.exists(os.path.join(out, "epoch1.pth"))


def test_training_resumes_from_an_epoch_checkpoint(tmp_path, monkeypatch):
    """util/misc.py:303-316 (load_model): adapters, AdamW moments, step count, loss scale and the epoch counter come back from
    `epoch0.pth`; the resumed second epoch ends where the uninterrupted run ends (up to the order of floating-point atomics)."""
    ck = _tree(str(tmp_path))
    fname, annos = __import__("dataset_fixture").annotations("MSRVTT")
    json.dump([a for i, a in enumerate(annos) if i != 2] * 2, open(os.path.join(str(tmp_path), "data", "MSRVTT", "msrvtt_ret_train.json"), "w"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(driver, "load_tokenizer", lambda path: StubTokenizer())
    common = ["--dataset", "MSRVTT", "--model_path", ck, "--topk", "4", "--batch_size_eval", "3", "--num_workers", "0", "--lr", "1e-3", "--warmup_epochs", "1",
              "--batch_size", "4", "--lora_drop", "0.05", "--seed", "1"]
    _run(common + ["--epochs", "2", "--output_dir", str(tmp_path / "a")])
    _run(common + ["--epochs", "1", "--output_dir", str(tmp_path / "b")])
    _run(common + ["--epochs", "2", "--output_dir", str(tmp_path / "b"), "--resume", str(tmp_path / "b" / "epoch0.pth")])
    A = torch.load(str(tmp_path / "a" / "epoch1.pth"), map_location="cpu", weights_only=False)
    B = torch.load(str(tmp_path / "b" / "epoch1.pth"), map_location="cpu", weights_only=False)
    assert A["epoch"] == B["epoch"] == 1 and A["optimizer"]["step"] == B["optimizer"]["step"] == 6
    for k in A["model"]:
        a, b = A["model"][k].float(), B["model"][k].float()
        d = (a - b).abs()
        assert d.median() <= 1e-6 and d.max() <= 3e-4, (k, d.median().item(), d.max().item())   # Adam normalises: an element with ~zero gradient moves by O(lr) on atomics' rounding order


def test_first_contact_go_and_no_go(tmp_path, monkeypatch, capsys):
    """blim_amd/first_contact.py on the synthetic on-disk tree (stand-in tokenizer): config / tokenizer constants / row shapes / key naming / resume-file
    checks, the numeric-mode table measured on the loaded checkpoint and the fused-vs-literal comparison end in GO; a resume file whose keys carry an
    unknown wrapper prefix is a NO-GO before any weight is loaded."""
    from blim_amd import first_contact as FC
    ck = _tree(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    argv = ["--model_path", ck, "--resume", "./checkpoint/msrvtt.pth", "--dataset", "MSRVTT", "--topk", "4", "--batch_size_eval", "3"]
    assert FC.main(argv, tokenizer=StubTokenizer()) == 0
    out = capsys.readouterr().out
    assert "GO:" in out and "NO-GO" not in out and "vtg_precise auto" in out and "fused PairScorer == literal" in out and "adapters kept apart" in out
    st = torch.load("./checkpoint/msrvtt.pth", map_location="cpu", weights_only=False)
    st["model"] = {"module." + k: v for k, v in st["model"].items()}
    torch.save(st, "./checkpoint/drift.pth")
    assert FC.main(["--model_path", ck, "--resume", "./checkpoint/drift.pth", "--dataset", "MSRVTT", "--topk", "4"], tokenizer=StubTokenizer()) == 1
    out = capsys.readouterr().out
    assert "NO-GO" in out and "map onto no engine tensor" in out
