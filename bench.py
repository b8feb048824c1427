"""Headline benchmark: utterances/s for connected-digit decode of synthetic 16 kHz four-digit
clips (BASELINE.json config 3: MFCC + 58-state free-loop Viterbi, between-word penalty -100),
audio resident in HBM, one process per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (audio -> MFCC -> emission scores -> Viterbi -> label
records [-> all-gather of the label records when N > 1]) over this rank's 10 000 clips.
Rank 0 prints one JSON line.
"""
import argparse
import functools
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cs-304-speech-recognition-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from loe_speech_recognition_amd import synthetic  # noqa: E402

CONFIG_ID = 3
N_DIGITS = 4
UTTS_PER_GPU = 10000
BETWEEN = -100
# other BASELINE.json configurations, selectable with --config (the default and the driver's run is 3)
CONFIGS = {
    2: dict(utts=1024, digits=1, what="1 024 one-digit clips: MFCC + loop decode + isolated 11-model classification"),
    3: dict(utts=10000, digits=4, what="10 000 four-digit clips: MFCC + S=58 loop decode, between=-100"),
    4: dict(utts=12000, digits=[1, 2, 3, 4, 5, 7], what="one continuous-training iteration (forced alignment + statistics + M-step) over 12 000 clips of 1-7 digits"),
    5: dict(utts=50000, digits=7, what="50 000 seven-digit clips: MFCC + S=58 loop decode, between=-100"),
}
D = 39
FP64_MATRIX_PEAK_TFLOPS = 78.6       # MI355X dense FP64 matrix (= vector) peak, AMD datasheet
CPU_SAMPLE = 48
POOL = 16


def _gen(args):
    return synthetic.batch(*args)


def generate(rank: int, count: int, workers: int, config_id: int = CONFIG_ID, digits=N_DIGITS):
    chunk = 240
    jobs = [(config_id, min(chunk, count - s), digits, rank * count + s) for s in range(0, count, chunk)]
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(_gen, jobs)
    return [s for p in parts for s in p[0]], [d for p in parts for d in p[1]]


# ---- CPU baseline: the oracle with the reference's loop structure, fanned out per label -------------
def _cpu_predict(model_arrays, item):
    """One pool task = one digit string and its clips, as scripts/project5_test_ndigits_with_sil.py:18-42."""
    from oracle import hmm_oracle as O
    from oracle import mfcc_oracle as M

    label, signals = item
    words = [O.WordModelRef(lab, *model_arrays[lab]) for lab in synthetic.LOOP_ORDER]
    ltp, gauss, bounds = O.stack_models(words)
    out = []
    t_mfcc = t_dec = 0.0
    for s in signals:
        t0 = time.perf_counter()
        x = M.mfcc_features(s).T
        t1 = time.perf_counter()
        _, path = O.viterbi_loop_ref(x, ltp, gauss, bounds, BETWEEN)
        t2 = time.perf_counter()
        out.append("".join(bounds.get_labels(path)))
        t_mfcc += t1 - t0
        t_dec += t2 - t1
    return label, out, t_mfcc, t_dec


def cpu_baseline(signals, truth, workers: int):
    arrays = synthetic.load_word_arrays()
    labeled = synthetic.group_by_label(signals, truth)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(functools.partial(_cpu_predict, arrays), list(labeled.items()))
    wall = time.perf_counter() - t0
    decoded = {}
    for label, out, _, _ in res:
        decoded[label] = out
    cpu_s = sum(r[2] + r[3] for r in res)
    return {"value": len(signals) / wall, "unit": "utterances/s", "cores": workers, "kind": "port",
            "sample": f"{len(signals)} of the workload's four-digit clips, oracle with the reference's per-(frame,state) "
                      f"SciPy logpdf loop, one pool task per digit string on {workers} processes; "
                      f"{cpu_s / len(signals):.3f} CPU-s per utterance ({sum(r[2] for r in res) / len(signals) * 1e3:.1f} ms MFCC)",
            "per_core": len(signals) / cpu_s}, decoded, labeled


def _setup(args):
    """Process-group and engine set-up shared by all configurations."""
    import torch
    import torch.distributed as dist

    from loe_speech_recognition_amd import _native

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LSR_DIST_BACKEND", "nccl")
    dev = local % torch.cuda.device_count()          # more ranks than GPUs only in gloo rehearsals
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    return torch, dist, _native.Engine(dev), rank, world, backend


def other_config(args):
    """Configs 2, 4 and 5: same timing contract, reported with their own metric names."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cfg = CONFIGS[args.config]
    utts = args.utts if args.utts != UTTS_PER_GPU else cfg["utts"]
    workers = max(1, POOL // max(1, min(world, 8)))
    sigs, truth = generate(rank, utts, workers, args.config, cfg["digits"])
    torch, dist, eng, rank, world, backend = _setup(args)
    from loe_speech_recognition_amd import (MFCC, HiddenMarkovModelInference, HiddenMarkovModelTrainContinuous,
                                            ModelCollection, _native)
    _native._engine, _native._engine_pid = eng, os.getpid()
    models = synthetic.load_word_models()
    ab = eng.audio_batch(sigs)
    extra = {}
    if args.config in (2, 5):
        inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
        inf._log_transition_probability_between_words = BETWEEN
        mc = ModelCollection()
        mc._models = [models[lab] for lab in synthetic.WORDS]
        state = {}

        def step():
            inf._bind()
            state["out"] = eng.recognize(ab, want_paths=False, label_cap=16)
            if args.config == 2:
                state["iso"] = mc.predict_batch(eng.mfcc(ab))
    else:
        tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
        for lab in synthetic.WORDS + ["S"]:
            tc.add_model(models[lab])
        fb = eng.mfcc(ab)
        labeled = {"__prepared__": None}
        tc.__dict__["_data_cache"] = {"key": id(labeled), "fb": fb, "owner": truth}

        def step():
            tc._iteration(labeled)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.config in (2, 5):
        names = inf._model_boundaries._labels
        lab, cnt = state["out"][2].cpu().numpy(), state["out"][3].cpu().numpy()
        strings = ["".join(names[w] for w in lab[i, :min(int(cnt[i]), 16)]) for i in range(ab.n)]
        extra["seq_accuracy_vs_truth"] = float(np.mean([a == b for a, b in zip(strings, truth)]))
        if args.config == 2:
            extra["isolated_accuracy_vs_truth"] = float(np.mean([a == b for a, b in zip(state["iso"], truth)]))
    if rank == 0:
        print(json.dumps({"metric": f"utterances/sec, BASELINE config {args.config}", "value": world * ab.n * args.steps / elapsed,
                          "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": cfg["what"], "utterances_per_gpu": ab.n, "frames_per_gpu": ab.total_frames},
                          **extra}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--utts", type=int, default=UTTS_PER_GPU, help="utterances per GPU (default: the config-3 size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    args = ap.parse_args()
    if args.config != 3:
        return other_config(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    workers = max(1, POOL // max(1, min(world, 8))) if world > 1 else POOL

    # host-side work that forks happens before this process touches the GPU
    sigs, truth = generate(rank, args.utts, workers)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, cpu_decoded, cpu_labeled = cpu_baseline(sigs[:CPU_SAMPLE], truth[:CPU_SAMPLE], POOL)

    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native

    torch, dist, eng, rank, world, backend = _setup(args)
    _native._engine, _native._engine_pid = eng, os.getpid()
    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = BETWEEN
    inf._bind()

    t0 = time.perf_counter()
    ab = eng.audio_batch(sigs)
    torch.cuda.synchronize()
    upload_s = time.perf_counter() - t0
    label_cap = 16
    out = (eng.empty((ab.n,), torch.float32), None, eng.empty((ab.n, label_cap), torch.uint8),
           eng.empty((ab.n,), torch.int32))
    from loe_speech_recognition_amd.distributed import gather_label_records
    state = {}

    def step():
        eng.recognize(ab, want_paths=False, label_cap=label_cap, out=out)
        if world > 1:
            # the exchange step of the path: every rank ends up with the label records (word ids + count)
            # of the whole job, one all-gather of 20-byte records per step
            if backend == "nccl":
                state["records"] = gather_label_records(out[2], out[3], world * ab.n)
            else:                                    # gloo rehearsal: records travel through host memory
                state["records"] = gather_label_records(out[2].cpu(), out[3].cpu(), world * ab.n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile_reset()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # results of the last step: strings, accuracy vs the generating strings and vs the CPU path
    names = inf._model_boundaries._labels
    lab = out[2].cpu().numpy()
    cnt = out[3].cpu().numpy()
    strings = ["".join(names[w] for w in lab[i, :min(int(cnt[i]), label_cap)]) for i in range(ab.n)]
    n_right = sum(a == b for a, b in zip(strings, truth))
    if world > 1:
        t = torch.tensor([n_right], dtype=torch.float64, device=eng.device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_right = float(t.item())
        g_lab, g_cnt = state["records"]
        assert g_lab.shape[0] == world * ab.n and g_cnt.shape[0] == world * ab.n
        lo = rank * ab.n                             # this rank's slice of the gathered records is its own output
        assert torch.equal(g_lab[lo:lo + ab.n].cpu()[:, :1], out[2].cpu()[:, :1]) and torch.equal(g_cnt[lo:lo + ab.n].cpu(), out[3].cpu())
    acc_truth = float(n_right) / (world * ab.n)

    if rank == 0:
        kernels = {}
        for k in ("frame_index", "mel_power", "mfcc_finish", "emission", "viterbi"):
            ms, n = eng.profile_get(k)
            kernels[k] = {"ms_per_launch": ms / max(n, 1), "launches": n}
        total_frames = ab.total_frames
        S = 58
        flops = float(S) * total_frames * (2 * D * D + 3 * D)        # algorithmic flops of one emission launch
        em_s = kernels["emission"]["ms_per_launch"] * 1e-3
        achieved = flops / em_s / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "emission_traffic.json")
        if os.path.exists(tpath) and args.utts == UTTS_PER_GPU:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        line = {
            "metric": "utterances/sec (whole node), 4-digit connected-digit decode (MFCC + 58-state loop Viterbi)",
            "value": world * ab.n * args.steps / elapsed,
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"config 3: {ab.n} synthetic four-digit 16 kHz clips per GPU, S=58 loop decode, between=-100",
                       "utterances_per_gpu": ab.n, "frames_per_gpu": int(total_frames),
                       "audio_mb_per_gpu": round(float(ab.n_samples.sum()) * 4 / 1e6, 1),
                       "parallelism": f"utterance-sharded x{world}" + (", all-gather of label records" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": "lsr_emission_kernel", "achieved": achieved,
                         "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
                         "traffic": traffic, "flops_per_launch": flops, "ms_per_launch": em_s * 1e3,
                         # the kernel evaluates the same quadratic form through a triangular factor in 4 x 4 blocks:
                         # 220 MFMAs of 512 flop per (64 frames, state) are issued, fewer than the dense definition counts
                         "executed_tflops": 220 * 512.0 * S * ((total_frames + 63) // 64) / em_s / 1e12},
            "kernels_ms_per_launch": {k: round(v["ms_per_launch"], 4) for k, v in kernels.items()},
            "seq_accuracy_vs_truth": acc_truth,
            "h2d_upload_s": round(upload_s, 4),
            "value_including_h2d": world * ab.n / (elapsed / args.steps + upload_s),
        }
        if cpu is not None:
            # strings of the CPU sample, matched by position within each label group
            seen = {}
            agree = 0
            for i in range(CPU_SAMPLE):
                k = seen.get(truth[i], 0)
                seen[truth[i]] = k + 1
                agree += strings[i] == cpu_decoded[truth[i]][k]
            line["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample")}
            line["cpu_baseline"]["per_core_utterances_per_s"] = cpu["per_core"]
            line["strings_equal_to_cpu_path"] = f"{agree}/{CPU_SAMPLE}"
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
