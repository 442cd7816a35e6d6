#!/usr/bin/env python3
"""bench.py — Valley visual-token hot path on MI355X.

A "step" is ONE pass of the hot path over one batch of synthetic clips resident in HBM:
    frames [B,T,3,224,224] bf16 -> CLIP ViT-L/14 encode (23 of 24 layers, hidden_states[-2])
    -> temporal mean pool + per-frame CLS -> (N>1: RCCL all-gather of pooled tokens, 1024-wide)
    -> mm_projector -> token-embedding gather + visual splice -> Llama prefill over [visual || text]
    -> lm_head on all positions.
Default workload = BASELINE.json configs[2] "Valley-13b-v1: 16 frames x batch 8, ViT-L/14 + Vicuna-13B
prefill, bf16, 1xMI355X" (S = 320 + T = 336 per SURVEY.md §8d) — the configuration BASELINE.json's metric
("prefill tokens/sec (13B)") is quoted on; it fits one GPU.  `--config c2` runs the 7B case (configs[1]).

Multi-GPU (`--gpus N`, launched by torch.distributed.run, one rank per GPU): frames shard by whole
clips (every rank encodes its own B clips: weak scaling), ONE all-gather reassembles the pooled
visual tokens of all N*B clips on every rank, every rank projects them, then prefills its own B
sequences on its Llama replica (`--prefill replicated` prefills all N*B on every rank instead, the
literal reading of configs[3]).

One JSON line on rank 0.  `value` = frames pushed through the WHOLE path per second over all
ranks; `stages` gives the ViT-encode frames/s and prefill tokens/s measured with HIP events inside
the same timed steps; `roofline` is for the dominant kernel (the MFMA GEMM instantiation with the
largest share of time), achieved = algorithmic flop per launch / mean launch duration, both measured
live with HIP events around every GEMM launch of the timed steps; `cpu_baseline` times the CPU oracle
(oracle/valley_oracle.py, kind "port") on a bounded sample of the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (B clips per GPU, T frames, hidden, heads, intermediate, layers, eps, label)
    "c2": dict(B=4, T=8, H=4096, heads=32, I=11008, L=32, eps=1e-5, label="Valley2-7b: 8 frames x batch 4, ViT-L/14 + Llama-2-7B prefill"),
    "c3": dict(B=8, T=16, H=5120, heads=40, I=13824, L=40, eps=1e-6, label="Valley-13b-v1: 16 frames x batch 8, ViT-L/14 + Vicuna-13B prefill"),
    # configs[3]: 64 clips x 32 frames over 8 GPUs = 8 clips per GPU; run with --gpus 8 --prefill replicated for the
    # literal configuration (all 64 sequences prefilled on every rank), default --prefill sharded for weak scaling
    "c4": dict(B=8, T=32, H=5120, heads=40, I=13824, L=40, eps=1e-6, label="Valley-13b-v1: 32 frames x 8 clips per GPU, ViT-L/14 + Vicuna-13B prefill"),
    # configs[4]: 13B decode behind an 8-frame visual prefix (run with --decode 256)
    "c5": dict(B=1, T=8, H=5120, heads=40, I=13824, L=40, eps=1e-6, label="Valley-13b-v1: 8-frame visual prefix, Vicuna-13B"),
    "tiny": dict(B=2, T=4, H=256, heads=2, I=512, L=2, eps=1e-5, label="tiny plumbing config"),
}
VOCAB_TEXT = 32000
PEAK_BF16_TFLOPS = 2500.0           # MI355X dense bf16 MFMA, MI355X_MICROARCH.md
VIT_GFLOP_PER_FRAME = 155.29        # patch 0.308 + 23 x 6.738


def prefill_flop(S, H, I, L, V):
    """S*[L*(8H^2+6HI)+2HV] + L*2*S*(S+1)*H (causal-half attention)."""
    return S * (L * (8 * H * H + 6 * H * I) + 2 * H * V) + L * 2 * S * (S + 1) * H


def _host_info():
    """CPU model, physical cores and logical CPUs of this host (Linux /proc)."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
            elif not ln.strip() and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    return model, (len(phys) or logical), logical


def _pick_threads(physical):
    """torch's CPU GEMM does not scale to every core of a big host (round 1: 830 GFLOP/s at 16 threads, 111 at 128 on
    the 256-core GPU box): time one ViT-fc1-shaped GEMM at a few thread counts, keep the fastest."""
    a, w = torch.randn(2056, 1024), torch.randn(4096, 1024)
    best, rates = 1, {}
    for th in sorted({t for t in (8, 16, 32, 64, physical) if t <= physical} or {physical}):
        torch.set_num_threads(th)
        torch.nn.functional.linear(a, w)
        t0 = time.perf_counter()
        for _ in range(4):
            torch.nn.functional.linear(a, w)
        rates[th] = round(4 * 2 * 2056 * 1024 * 4096 / (time.perf_counter() - t0) / 1e9, 1)
        if rates[th] > rates.get(best, 0):
            best = th
    return best, rates


def cpu_baseline():
    """the oracle (kind "port", oracle/valley_oracle.py) on configs[0] EXACTLY — 1 clip x 8 frames
    (224^2) -> ViT-L/14 (23 contributing layers) -> mean pool + mm_projector -> splice -> Llama-2-7B-shape prefill
    (32 layers, S = 328) -> lm_head on all positions; random weights, fp32 (bf16 weights if the host cannot hold
    27 GB); 1 warm-up pass + the median of 3 timed passes (bounded to ~30 s of CPU work); thread count =
    the fastest of a short GEMM calibration, reported with the CPU model, the physical core count and the torch
    version.  The 32 decoder layers alias ONE random layer's weight buffers (0.81 GB fp32 per layer — larger than any
    host L3, so every layer still streams its weights from DRAM; identical arithmetic; allocating and faulting in 27 GB
    took 2.5 min of setup on the authoring host and bought nothing)."""
    from oracle import valley_oracle as O
    model, physical, logical = _host_info()
    threads, rates = _pick_threads(physical)
    torch.set_num_threads(threads)
    T, H, I, L, heads, V = 8, 4096, 11008, 32, 32, VOCAB_TEXT + 6
    S = 320 + T
    g = torch.Generator().manual_seed(0)
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std  # noqa: E731
    vw = {"embeddings.class_embedding": rn(1024), "embeddings.patch_embedding.weight": rn(1024, 3, 14, 14),
          "embeddings.position_embedding.weight": rn(257, 1024), "pre_layrnorm.weight": torch.ones(1024),
          "pre_layrnorm.bias": torch.zeros(1024)}
    for i in range(23):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            vw[p + f"self_attn.{n}.weight"], vw[p + f"self_attn.{n}.bias"] = rn(1024, 1024), rn(1024)
        for n in ("layer_norm1", "layer_norm2"):
            vw[p + n + ".weight"], vw[p + n + ".bias"] = torch.ones(1024), torch.zeros(1024)
        vw[p + "mlp.fc1.weight"], vw[p + "mlp.fc1.bias"] = rn(4096, 1024), rn(4096)
        vw[p + "mlp.fc2.weight"], vw[p + "mlp.fc2.bias"] = rn(1024, 4096), rn(1024)
    base = {"q": rn(H, H), "k": rn(H, H), "v": rn(H, H), "o": rn(H, H), "gate": rn(I, H), "up": rn(I, H), "down": rn(H, I)}
    lw = {"model.norm.weight": torch.ones(H), "model.embed_tokens.weight": rn(V, H), "lm_head.weight": rn(V, H),
          "model.mm_projector.weight": rn(H, 1024), "model.mm_projector.bias": torch.zeros(H)}
    for i in range(L):
        p = f"model.layers.{i}."
        for n in "qkvo":
            lw[p + f"self_attn.{n}_proj.weight"] = base[n]
        for n in ("gate", "up", "down"):
            lw[p + f"mlp.{n}_proj.weight"] = base[n]
        lw[p + "input_layernorm.weight"], lw[p + "post_attention_layernorm.weight"] = torch.ones(H), torch.ones(H)
    from valley_amd import weights as W
    tok = O.TokenIds(**W.SPECIAL_IDS(VOCAB_TEXT))
    ids = torch.from_numpy(W.synthetic_prompt(7, T, VOCAB_TEXT)).view(1, S)
    px = torch.randn((1, T, 3, 224, 224), generator=g)
    vcfg = O.VisionCfg(layers=24)
    lcfg = O.LlamaCfg(hidden=H, heads=heads, intermediate=I, layers=L, vocab=V, eps=1e-5)
    def one_pass():
        with torch.no_grad():
            t0 = time.perf_counter()
            feats = O.vit_select(px[0], vw, vcfg, -2)
            proj = O.mm_project(feats, lw)
            tv = time.perf_counter() - t0
            t0 = time.perf_counter()
            emb = torch.nn.functional.embedding(ids, lw["model.embed_tokens.weight"])
            emb = O.splice_visual_tokens(ids, emb, [proj], tok, "mean", lw)
            hidden, _ = O.llama_forward(emb, lw, lcfg)
            lg = torch.nn.functional.linear(hidden, lw["lm_head.weight"])
            return tv, time.perf_counter() - t0, lg

    # 1 warm-up + >= 3 timed passes, median (bounded: a pass is 5-8 s on the GPU box's EPYC, so the
    # whole leg stays under ~30 s; a slower host gets fewer timed passes, never fewer than one)
    budget_s = float(os.environ.get("VALLEY_CPU_BASELINE_BUDGET_S", "30"))
    t_start = time.perf_counter()
    tv, tl, logits = one_pass()                                       # warm-up (page faults, thread pool, allocator)
    passes = []
    while len(passes) < 3 and (not passes or time.perf_counter() - t_start + (tv + tl) < budget_s):
        tv, tl, logits = one_pass()
        passes.append((tv + tl, tv, tl))
    passes.sort()
    _, t_vit, t_l = passes[len(passes) // 2]                          # the median pass
    assert torch.isfinite(logits).all()
    return {"value": round(T / (t_vit + t_l), 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"configs[0] exactly: 1 clip x {T} frames ViT-L/14 23 layers + projector ({t_vit:.2f}s), then splice + "
                      f"7B-shape prefill L={L} S={S} + lm_head on all positions ({t_l:.2f}s); oracle fp32, 1 warm-up + "
                      f"median of {len(passes)} timed passes, the {L} layers alias one layer's 0.81 GB of weights",
            "timed_passes": len(passes), "pass_seconds": [round(p[0], 2) for p in passes],
            "vit_frames_per_s": round(T / t_vit, 3), "prefill_tokens_per_s": round(S / t_l, 2),
            "cpu_model": model, "physical_cores": physical, "logical_cpus": logical, "threads": threads,
            "thread_calibration_GFLOPs": rates, "torch": torch.__version__}


def live_traffic(args, kernel_name):
    """Hardware counters of `kernel_name`, measured now: three `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE do not fit the
    TCC slot budget together — MI355X_MICROARCH.md §rocprofv3 PMC slots —, the SQ / GRBM pair has its own pass, and nothing but
    --kernel-trace rides along) over a short child run of this same bench.
      HBM-side bytes per launch: both counters x1024, FETCH_SIZE x2 on gfx950 (128-byte requests tallied at 64 B), WRITE_SIZE as is
      (the guide's HBM section);
      matrix-core utilisation: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), both from the SAME pass;
      sustained clock: GRBM_GUI_ACTIVE / 8 / the launch's duration in that pass's kernel trace.
    -> (bytes per launch | None, note, {"mfma_busy", "sustained_clock_GHz", ...} | {})."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", {}
    out = tempfile.mkdtemp(prefix="vly_pmc_", dir="/tmp")
    key = kernel_name.replace(" ", "")
    per = {}
    extra = {}
    for tag, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                          ("MFMA", ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"])):
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", tag, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--config", args.config, "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
               "--no-kernel-events", "--traffic", "none", "--also", "none", "--pack-weights", str(args.pack_weights)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, timeout=900)
        except Exception as e:  # noqa: BLE001
            return None, f"rocprofv3 --pmc {tag} failed: {e!r}", extra
        files = glob.glob(os.path.join(out, "**", f"{tag}_counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            if tag == "MFMA":                                      # the traffic passes stand on their own
                extra = {"pmc_note": f"rocprofv3 --pmc {' '.join(counters)}: rc={r.returncode}, no counter file"}
                break
            return None, f"rocprofv3 --pmc {tag}: rc={r.returncode}, no counter file ({r.stderr.decode(errors='replace')[-200:]})", extra
        rows = [row for row in csv.DictReader(open(files[0])) if key in row["Kernel_Name"].replace(" ", "")]
        for c in counters:
            vals = [float(row["Counter_Value"]) for row in rows if row["Counter_Name"] == c]
            if not vals:
                if tag == "MFMA":
                    extra = {"pmc_note": f"{kernel_name} not in the {c} pass"}
                    break
                return None, f"{kernel_name} not in the {c} pass", extra
            per[c] = (sum(vals) / len(vals), len(vals))
        if tag == "MFMA" and "SQ_VALU_MFMA_BUSY_CYCLES" in per and "GRBM_GUI_ACTIVE" in per:
            cycles = per["GRBM_GUI_ACTIVE"][0] / 8.0               # the counter sums the 8 XCDs' active cycles
            extra = {"mfma_busy": round(per["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * cycles), 4),
                     "pmc_launches": per["GRBM_GUI_ACTIVE"][1],
                     "pmc_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), one rocprofv3 --pmc pass of a 2-step child run; "
                                 "clock = GRBM_GUI_ACTIVE / 8 / launch duration in that pass (profiled runs clock ~3 % under un-profiled ones)"}
            tfiles = glob.glob(os.path.join(out, "**", "MFMA_kernel_trace.csv"), recursive=True)
            if tfiles:
                durs = [int(row["End_Timestamp"]) - int(row["Start_Timestamp"]) for row in csv.DictReader(open(tfiles[0]))
                        if key in row["Kernel_Name"].replace(" ", "")]
                if durs:
                    extra["sustained_clock_GHz"] = round(cycles / (sum(durs) / len(durs)), 3)
                    extra["pmc_avg_launch_us"] = round(sum(durs) / len(durs) / 1e3, 2)
    shutil.rmtree(out, ignore_errors=True)
    b = 2 * 1024 * per["FETCH_SIZE"][0] + 1024 * per["WRITE_SIZE"][0]
    return int(b), (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two passes of `bench.py --config {args.config} --steps 2` on this "
                    f"box ({per['FETCH_SIZE'][1]} launches), x1024, FETCH_SIZE x2 (gfx950); fetch {2 * 1024 * per['FETCH_SIZE'][0] / 1e6:.1f} MB "
                    f"+ write {1024 * per['WRITE_SIZE'][0] / 1e6:.1f} MB per launch"), extra


ALSO = {
    # key in the JSON line: (child arguments, BASELINE.json configuration it is)
    "c2": (["--config", "c2", "--steps", "10", "--warmup", "3"], "configs[1]"),
    "c4_n1": (["--config", "c4", "--steps", "5", "--warmup", "2"], "configs[3], per-GPU shape at N = 1"),
    "c5_decode": (["--config", "c5", "--decode", "256", "--warmup", "8"], "configs[4]"),
    # the same step with 8 live requests (serving.ContinuousBatcher; the reference's worker admits 5 at a time, model_worker.py:467-474)
    "c5_decode_b8": (["--config", "c5", "--decode", "128", "--decode-batch", "8", "--warmup", "8"], "configs[4] x 8 concurrent requests"),
    # the headline workload on the fp32 validation engines (valley_amd/precise.py): what the 1e-3 logit bound costs
    "c3_fp32": (["--config", "c3", "--steps", "2", "--warmup", "1"], "configs[2] on the fp32 validation engines (logits within 1e-3)",
                {"VALLEY_PRECISION": "fp32"}),
    # the same engines with every GEMM as three 16-bit partial products on the production MFMA kernels (ops_f32, VALLEY_F32_GEMM=x3):
    # what the 1e-3 bound costs once the contractions leave the f32-input MFMA
    "c3_x3": (["--config", "c3", "--steps", "3", "--warmup", "1"], "configs[2] on the split-operand engines (fp32 tensors, GEMMs = hi.hi + hi.lo + lo.hi on the bf16 MFMA; logits within 1e-3)",
              {"VALLEY_PRECISION": "fp32", "VALLEY_F32_GEMM": "x3"}),
    # the headline workload on libvalley_hip_f16.so: IEEE fp16 storage is the reference's own inference dtype
    # (valley/inference/run_valley.py:39 `torch_dtype=torch.float16`); the child is told through VALLEY_PRECISION
    "c3_fp16": (["--config", "c3", "--steps", "10", "--warmup", "3"], "configs[2] at the reference's inference dtype (fp16 storage)",
                {"VALLEY_PRECISION": "fp16"}),
}


def run_also(names, pack_weights):
    """The other BASELINE.json configurations, each measured by a child run of this script on the same box right after the
    timed region (the parent's engines are freed first), trimmed to the keys that carry the numbers: value / unit,
    ms_per_step, stages and that run's own roofline."""
    import subprocess
    out = {}
    for name in names:
        extra, what = ALSO[name][:2]
        env = dict(os.environ, **(ALSO[name][2] if len(ALSO[name]) > 2 else {}))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *extra, "--no-cpu-baseline", "--traffic", "none", "--also", "none",
               "--pack-weights", str(pack_weights)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            line = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": f"rc={r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"}
                continue
            d = json.loads(line[-1])
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
            continue
        keep = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "stages", "wall_ms_per_token") if k in d}
        keep["baseline_config"] = what
        keep["workload"] = d.get("config", {}).get("workload")
        rf = d.get("roofline")
        if rf:
            keep["roofline"] = {k: rf[k] for k in ("bound", "kernel", "shape", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "hbm_GBps", "mfma_busy",
                                                  "sustained_clock_GHz", "launches", "share_of_step_time", "bytes_per_token", "note") if k in rf}
        keep["child_wall_s"] = round(time.perf_counter() - t0, 1)
        out[name] = keep
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=list(CONFIGS),
                    help="default: c3 (BASELINE configs[2], the configuration the metric is quoted on) at --gpus 1; c4 = configs[3] at --gpus > 1")
    ap.add_argument("--prefill", default=None, choices=["sharded", "replicated"],
                    help="N > 1: 'replicated' (default) = configs[3] as written — every rank prefills all N x B sequences; 'sharded' = each rank "
                         "prefills its own clips (weak scaling of the whole step; measured as well and attached under also.sharded_prefill)")
    ap.add_argument("--simulate-ranks", type=int, default=1, metavar="R",
                    help="N = 1 only, with --prefill replicated: run THIS rank's share of an R-rank job — local clips encoded once, "
                         "their pooled tokens tiled R times in place of the all-gather, then the replicated prefill of all R x B "
                         "sequences (configs[3] at R = 8: B = 64, M = 22528 rows per GEMM).  `value` counts the local frames only.")
    ap.add_argument("--decode", type=int, default=0, metavar="N",
                    help="instead of the prefill step: prefill once, then time N greedy hipGraph decode steps (configs[4])")
    ap.add_argument("--decode-batch", type=int, default=1, metavar="R",
                    help="with --decode: R concurrent requests (<= 8) on ONE captured step through serving.ContinuousBatcher — the reference's "
                         "worker admits 5 concurrent requests (serve/model_worker.py:467-474); the weight stream of a step is shared")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VALLEY_BENCH_STREAMS", "1")),
                    help="run the batch as this many independent sub-batches on separate HIP streams (tail filling)")
    ap.add_argument("--pack-weights", type=int, default=int(os.environ.get("VALLEY_PACK_WEIGHTS", "1")), choices=[0, 1],
                    help="1 (default): the Llama prefill GEMMs read a second, block-ordered copy of the weights (ops.PackedWeight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default="live", choices=["live", "file", "none"],
                    help="roofline.traffic of the dominant kernel: measured now by two rocprofv3 --pmc child passes (live, N=1 "
                         "only), read from the newest profiles/**/r*_traffic_<config>.json (file), or null (none)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (rocprof runs)")
    ap.add_argument("--also", default="auto",
                    help="extra workloads measured after the timed region, each in a child run of this script, and attached to the "
                         "line under `also` (N = 1 only): comma list of c2 (configs[1]), c4 (configs[3]'s per-GPU shape at N = 1), "
                         "decode (configs[4]: 13B, 256 tokens, hipGraph step), c3_fp16 (the headline workload on the fp16 library); "
                         "'auto' = all four for the default c3 run, 'none' = off")
    args = ap.parse_args()
    os.environ["VALLEY_PACK_WEIGHTS"] = str(args.pack_weights)      # read by the engines when they load their weights

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        # started without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1) — the same command line the
        # driver uses, so `python bench.py --gpus N` alone is a complete multi-GPU run
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    default_multi = world > 1 and args.prefill is None and args.config is None and not args.decode     # configs[3] as written + the sharded form
    if args.config is None:
        args.config = "c3" if world == 1 else "c4"
    if args.prefill is None:
        args.prefill = "sharded" if world == 1 else "replicated"
    # VALLEY_BENCH_SAME_DEVICE=1 + VALLEY_BENCH_BACKEND=gloo: plumbing test of the N>1 path on a 1-GPU box
    if os.environ.get("VALLEY_BENCH_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VALLEY_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from valley_amd import ops, parallel, runtime
    from valley_amd import valley_model as vm
    # compute dtype of the line: the library's 16-bit storage / MFMA operand type (bf16 default; VALLEY_PRECISION=fp16 runs
    # the same step on libvalley_hip_f16.so — the reference's own inference dtype)
    DT = runtime.PRECISION                                    # "bf16" | "fp16" | "fp32" (the validation engines)
    if DT == "fp32":
        from valley_amd import ops_f32
        if ops_f32.GEMM_MODE == "x3":
            DT = "fp32 tensors; GEMMs as 3 bf16 partial products (hi.hi + hi.lo + lo.hi), fp32 accumulation"
    from valley_amd import weights as W

    cfg = CONFIGS[args.config]
    B, T, H, I, L = cfg["B"], cfg["T"], cfg["H"], cfg["I"], cfg["L"]
    V = VOCAB_TEXT + 6
    S = 320 + T
    config = vm.ValleyConfig(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=L,
                             num_attention_heads=cfg["heads"], num_key_value_heads=cfg["heads"], rms_norm_eps=cfg["eps"],
                             max_position_embeddings=2048)
    config.use_mm_proj, config.mm_hidden_size, config.mm_vision_select_layer = True, 1024, -2
    model = vm.ValleyLlamaForCausalLM(config, device=dev)
    mm = model.get_model()
    mm.llama.init_random(seed=0)
    tower = vm.build_vision_tower(None, device=dev)
    tower.init_random(seed=0, layers=23)                     # layer 24 never contributes to hidden_states[-2]
    for k, v in W.SPECIAL_IDS(VOCAB_TEXT).items():
        setattr(tower.config, k, v)
    mm.initialize_vision_modules(tower, -2)

    # synthetic inputs, resident in HBM before the timed region
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    frames = torch.randn((B, T, 3, 224, 224), generator=g, device=dev).to(runtime.HALF)
    ids = torch.from_numpy(W.synthetic_prompt(7, T, VOCAB_TEXT)).view(1, S)
    sim = args.simulate_ranks if (world == 1 and args.prefill == "replicated") else 1
    Bp = B * world * sim if args.prefill == "replicated" else B
    input_ids = ids.repeat(Bp, 1)
    cache = mm.llama.new_cache(Bp, S)
    Ts_all = [T] * (B * world * sim)

    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    stage_events = []
    gather_events = []

    if args.decode and args.decode_batch > 1:
        # continuous batching: R live requests, each with its own T-frame visual prefix and KV rows, advance together
        # through ONE captured decode step; the step streams every weight once whatever R is, so the aggregate rate scales with R
        # until the R attention passes and the GEMVs' R accumulator rows show
        from valley_amd.serving import ContinuousBatcher
        R, n_new = args.decode_batch, args.decode
        cb = ContinuousBatcher(model, slots=R, ctx_max=S + n_new + args.warmup + 8)
        for r in range(R):
            cb.add(input_ids[:1], images=frames[:1])
        for _ in range(args.warmup):
            cb.step()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n_new):
            toks = cb.step()                                  # (one D2H read of the R tokens per step: the serving loop's own sync)
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        assert len(toks) == R
        ms_step = e0.elapsed_time(e1) / n_new
        wbytes = 2.0 * (L * (4 * H * H + 3 * H * I) + H * mm.llama.Vpad)
        ctx_mid = S + args.warmup + n_new / 2
        kvbytes = 2.0 * 2 * L * H * ctx_mid * R
        ach = (wbytes + kvbytes) / (ms_step * 1e-3) / 1e9
        print(json.dumps({
            "metric": "decode tokens/sec, all live requests (KV-cache, greedy, one hipGraph step per token)", "value": round(R * 1e3 / ms_step, 2),
            "unit": "tokens/s", "n_gpus": 1, "steps": n_new, "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DT, "data": "synthetic",
            "config": {"workload": f"{cfg['label']} -> continuous batching, {R} live requests on one captured step, prefix S={S}, {n_new} tokens each",
                       "name": args.config, "requests": R},
            "tokens_per_s_per_request": round(1e3 / ms_step, 2), "wall_ms_per_step": round(wall / n_new * 1e3, 4),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                         "traffic": None, "bytes_per_token": int((wbytes + kvbytes) / R), "bytes_per_step": int(wbytes + kvbytes),
                         "note": "algorithmic bytes per STEP = all matmul weights once (16-bit) + K and V of the mean context of every request"}}),
            flush=True)
        return

    if args.decode:
        # configs[4]: T-frame visual prefix, B=1, N generated tokens through the captured decode step
        from valley_amd.decode import DecodeSession
        n_new = args.decode
        dcache = mm.llama.new_cache(1, S + n_new + 8)
        out = model(input_ids=input_ids[:1], images=frames[:1], past_key_values=dcache, use_cache=True)
        sess = DecodeSession(mm.llama, dcache, use_graph=True)
        sess.begin(out.logits[:, -1].argmax(-1))
        for _ in range(args.warmup):
            sess.step()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n_new):
            sess.step()
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms_tok = e0.elapsed_time(e1) / n_new
        wbytes = 2.0 * (L * (4 * H * H + 3 * H * I) + H * mm.llama.Vpad)
        ctx_mid = S + args.warmup + n_new / 2
        kvbytes = 2.0 * 2 * L * H * ctx_mid
        ach = (wbytes + kvbytes) / (ms_tok * 1e-3) / 1e9
        print(json.dumps({
            "metric": "decode tokens/sec (KV-cache, greedy, hipGraph step)", "value": round(1e3 / ms_tok, 2), "unit": "tokens/s",
            "n_gpus": 1, "steps": n_new, "warmup": args.warmup, "ms_per_step": round(ms_tok, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DT, "data": "synthetic",
            "config": {"workload": f"{cfg['label']} -> autoregressive decode, B=1, prefix S={S}, {n_new} tokens", "name": args.config},
            "wall_ms_per_token": round(wall / n_new * 1e3, 4),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                         "traffic": None, "bytes_per_token": int(wbytes + kvbytes),
                         "note": "algorithmic bytes = all matmul weights once (bf16) + K and V of the mean context"}}), flush=True)
        return

    NS = max(1, min(args.streams, B, Bp))
    side = [torch.cuda.Stream(device=dev) for _ in range(NS)] if NS > 1 else []
    cuts_b = [B * i // NS for i in range(NS + 1)]                      # clips per stream (ViT)
    cuts_p = [Bp * i // NS for i in range(NS + 1)]                     # sequences per stream (prefill)
    sub_caches = [mm.llama.new_cache(cuts_p[i + 1] - cuts_p[i], S) for i in range(NS)] if NS > 1 else []

    def fork_join(fn):
        """Run fn(i) for every sub-batch on its own stream; the main stream waits for all of them."""
        main = torch.cuda.current_stream()
        outs = []
        for i, st in enumerate(side):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(fn(i))
        for st in side:
            main.wait_stream(st)
        return outs

    def step(record):
        e0, e1, e2 = (ev(), ev(), ev()) if record else (None, None, None)
        if record:
            e0.record()
        if NS > 1:                                                       # independent sub-batches side by side
            pooled = torch.cat(fork_join(lambda i: mm.encode_clips(frames[cuts_b[i]:cuts_b[i + 1]])[0]), 0)
        else:
            pooled, _ = mm.encode_clips(frames)                          # ViT encode + temporal pool (local clips)
        if world > 1:
            if record:
                g0, g1 = ev(), ev()
                g0.record()
            pooled = parallel.all_gather_rows(pooled, [pooled.shape[0]] * world)
            if record:
                g1.record()
                gather_events.append((g0, g1, pooled.shape[0] // world * pooled.shape[1] * pooled.element_size()))
        if sim > 1:                                                      # stand-in for the all-gather of R ranks' pooled tokens
            pooled = pooled.repeat(sim, 1)
        if record:
            e1.record()
        visual = mm.project_pooled(pooled)                               # all N*B clips' tokens, on every rank
        if args.prefill == "sharded" and world > 1:
            n = B * (256 + T)
            visual = visual[rank * n:(rank + 1) * n]
        if NS > 1:
            nv = 256 + T

            def pre(i):
                c = sub_caches[i]
                c.seq_len = 0
                return model(input_ids=input_ids[cuts_p[i]:cuts_p[i + 1]], past_key_values=c, use_cache=True,
                             visual_tokens=visual[cuts_p[i] * nv:cuts_p[i + 1] * nv],
                             frames_per_clip=Ts_all[cuts_p[i]:cuts_p[i + 1]])
            outs = fork_join(pre)
            out = outs[-1]
        else:
            cache.seq_len = 0
            out = model(input_ids=input_ids, past_key_values=cache, use_cache=True, visual_tokens=visual,
                        frames_per_clip=Ts_all[:Bp])
        if record:
            e2.record()
            stage_events.append((e0, e1, e2))
        return out

    # untimed: the online GEMM tuner (ops.gemm) tries its candidate kernels in place, one per call and shape;
    # run steps until every shape on the path is decided (all ranks take the same number of passes)
    tune_passes = 0
    while True:
        step(False)
        torch.cuda.synchronize()
        tune_passes += 1
        pend = ops.tuning_pending()
        if world > 1:
            tp = torch.tensor([pend], device=dev, dtype=torch.int64)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            pend = int(tp.item())
        if pend == 0 or tune_passes >= 600:
            break
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    rec = None if args.no_kernel_events else []
    # per-launch HIP events serialise the queue (~3.4 us per instrumented GEMM: 1.46 ms of a 28.6 ms c2 step when
    # every launch carries a pair), so only every EVENT_STRIDE-th timed step is instrumented
    EVENT_STRIDE = int(os.environ.get("VALLEY_BENCH_EVENT_STRIDE", "5"))
    rec_steps = max(1, len(range(0, args.steps, EVENT_STRIDE)))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        ops.set_recorder(rec if (rec is not None and it % EVENT_STRIDE == 0) else None)
        out = step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.set_recorder(None)
    assert torch.isfinite(out.logits[:, -1]).all()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    sharded_also = None
    if default_multi:
        # the weak-scaling form of the same step (each rank prefills only its own B clips), timed the same way in the same processes
        ids_s = ids.repeat(B, 1)
        cache_s = mm.llama.new_cache(B, S)
        nvs = B * (256 + T)

        def step_sharded():
            pooled, _ = mm.encode_clips(frames)
            pooled = parallel.all_gather_rows(pooled, [pooled.shape[0]] * world)
            visual = mm.project_pooled(pooled)[rank * nvs:(rank + 1) * nvs]
            cache_s.seq_len = 0
            return model(input_ids=ids_s, past_key_values=cache_s, use_cache=True, visual_tokens=visual, frames_per_clip=Ts_all[:B])
        passes = 0
        while True:                                                       # the online tuner's untimed passes for the new shapes
            step_sharded()
            torch.cuda.synchronize()
            passes += 1
            tp = torch.tensor([ops.tuning_pending()], device=dev, dtype=torch.int64)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            if int(tp.item()) == 0 or passes >= 600:
                break
        for _ in range(args.warmup):
            step_sharded()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            out_s = step_sharded()
        torch.cuda.synchronize()
        dist.barrier()
        tsh = torch.tensor([time.perf_counter() - ts0], device=dev, dtype=torch.float64)
        dist.all_reduce(tsh, op=dist.ReduceOp.MAX)
        assert torch.isfinite(out_s.logits[:, -1]).all()
        sharded_also = {"value": round(B * T * world / (float(tsh.item()) / args.steps), 2), "unit": "frames/s",
                        "ms_per_step": round(float(tsh.item()) / args.steps * 1e3, 3), "prefill_batch_per_gpu": B, "tune_passes": passes,
                        "what": "the same step with each rank prefilling only its own clips: weak scaling of the WHOLE step (compare with the N = 1 "
                                "line's value x N); the headline above is configs[3] as written, whose replicated prefill grows with N"}
        del cache_s

    vit_enc_ms_max = 0.0
    if world > 1:
        # encode stage of THIS rank without its all-gather (e0 -> e1 minus g0 -> g1), MAX over ranks
        mine = (sum(a.elapsed_time(b) for a, b, _ in stage_events) - sum(a.elapsed_time(b) for a, b, _ in gather_events)) / max(1, len(stage_events))
        t = torch.tensor([mine], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vit_enc_ms_max = float(t.item())
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        frames_total = B * T * world
        vit_ms = sum(a.elapsed_time(b) for a, b, _ in stage_events) / len(stage_events)
        pre_ms = sum(b.elapsed_time(c) for _, b, c in stage_events) / len(stage_events)
        vit_fps = B * T / (vit_ms * 1e-3)
        pre_tps = Bp * S / (pre_ms * 1e-3)
        vit_tf = vit_fps * VIT_GFLOP_PER_FRAME / 1e3
        pre_tf = prefill_flop(S, H, I, L, V) * Bp / (pre_ms * 1e-3) / 1e12
        result = {
            "metric": "frames/sec ViT-L/14 encode + prefill tokens/sec" + (" (13B)" if H == 5120 else " (7B)" if H == 4096 else ""),
            "value": round(frames_total / (elapsed / args.steps), 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DT, "data": "synthetic",
            "config": {"workload": cfg["label"] + f", S={S}, end-to-end hot path (encode+pool+project+splice+prefill+lm_head)",
                       "name": args.config, "clips_per_gpu": B, "frames_per_clip": T, "prefill_batch_per_gpu": Bp,
                       "seq_len": S, "tune_passes": tune_passes, "streams": NS, "weights": "row-major + packed64" if args.pack_weights else "row-major", "parallelism": f"frame-dp{world}" + ("+replicated-prefill" if args.prefill == "replicated" else "") + (f" (one rank of a simulated {sim}-rank job)" if sim > 1 else "")},
            "stages": {"vit_frames_per_s_per_gpu": round(vit_fps, 1), "vit_ms": round(vit_ms, 3),
                       "vit_TFLOPs": round(vit_tf, 1), "vit_frac_of_bf16_peak": round(vit_tf / PEAK_BF16_TFLOPS, 4),
                       "prefill_tokens_per_s_per_gpu": round(pre_tps, 1), "prefill_ms": round(pre_ms, 3),
                       "prefill_TFLOPs": round(pre_tf, 1), "prefill_frac_of_bf16_peak": round(pre_tf / PEAK_BF16_TFLOPS, 4)},
        }
        if rec:
            agg = {}
            shapes = {}
            for name, flop, a, b, shp in rec:
                dt = a.elapsed_time(b) * 1e-3
                d = agg.setdefault(name, [0.0, 0.0, 0])
                d[0] += dt
                d[1] += flop
                d[2] += 1
                d = shapes.setdefault("%dx%dx%d/e%d" % shp, [0.0, 0.0, 0, name])
                d[0] += dt
                d[1] += flop
                d[2] += 1
            # the dominant kernel = the GEMM call site (shape) with the largest share of the step; an instantiation
            # that serves several shapes (e.g. the plain 192x128 tile: ViT out-proj / fc2 and the Llama o-proj / down-proj
            # split-K pairs) is not one workload, its per-name average in rocprofv3 mixes them (kernel_all_shapes below)
            dshape, (tsum, fsum, n, name) = max(shapes.items(), key=lambda kv: kv[1][0])
            ach = fsum / tsum / 1e12
            traffic, traffic_src, pmc = None, None, {}
            if args.traffic == "live" and world == 1:
                traffic, traffic_src, pmc = live_traffic(args, name)
            if traffic is None and args.traffic != "none":           # newest committed PMC run of this config
                import glob
                for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "**", f"r*_traffic_{args.config}.json"), recursive=True),
                                    key=os.path.basename, reverse=True):
                    tk = json.load(open(tpath)).get("kernels", {})
                    hit = [v for k, v in tk.items() if k.replace(" ", "") == name.replace(" ", "")]
                    if hit:
                        traffic = hit[0]["hbm_bytes_per_launch"]
                        traffic_src = (f"{os.path.relpath(tpath, ROOT)} (another box; rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, x1024, "
                                       f"separate passes; tools/pmc_traffic.sh)" + (f"; live attempt: {traffic_src}" if traffic_src else ""))
                        break
            M_, N_, K_ = (int(x) for x in dshape.split("/")[0].split("x"))
            e_ = int(dshape.split("/e")[1])
            algo_bytes = 2 * (M_ * K_ + N_ * K_) + 2 * M_ * (N_ // 2 if e_ == 2 else N_)
            result["roofline"] = {"bound": "mfma", "kernel": name, "shape": dshape,
                                  "kernel_all_shapes": {"launches": agg[name][2], "avg_launch_us": round(agg[name][0] / agg[name][2] * 1e6, 2),
                                                        "note": "what rocprofv3 --stats lists under this kernel name"},
                                  "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                                  "traffic_source": traffic_src, "algorithmic_bytes": algo_bytes,
                                  "traffic_over_algorithmic": round(traffic / algo_bytes, 2) if traffic else None,
                                  # north_star's two rocprof quantities, from the live PMC passes: HBM-side GB/s of this kernel
                                  # (counter bytes / the live events' launch time) and its matrix-core utilisation + sustained clock
                                  "hbm_GBps": round(traffic / (tsum / n) / 1e9, 1) if traffic else None,
                                  "mfma_busy": pmc.get("mfma_busy"), "sustained_clock_GHz": pmc.get("sustained_clock_GHz"),
                                  "pmc": {k: v for k, v in pmc.items() if k not in ("mfma_busy", "sustained_clock_GHz")} or None,
                                  "launches": n, "avg_launch_us": round(tsum / n * 1e6, 2),
                                  "avg_flop_per_launch": round(fsum / n / 1e9, 3),
                                  "share_of_step_time": round(tsum / rec_steps / (ms_step * 1e-3), 3),
                                  "instrumented_steps": rec_steps,
                                  "all_gemm_kernels": {k: {"TFLOPs": round(v[1] / v[0] / 1e12, 1), "launches": v[2],
                                                           "ms_per_step": round(v[0] / rec_steps * 1e3, 3)}
                                                       for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])},
                                  "gemm_shapes": {k: {"TFLOPs": round(v[1] / v[0] / 1e12, 1), "avg_us": round(v[0] / v[2] * 1e6, 1),
                                                      "ms_per_step": round(v[0] / rec_steps * 1e3, 3), "kernel": v[3]}
                                                  for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][0])}}
        if world > 1:
            # the N > 1 run proves itself: what torch.distributed reports, and the one collective of the path timed with HIP
            # events inside the same timed steps (rank 0's view)
            result["dist"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                              "rccl": bool(dist.get_backend() == "nccl"), "prefill": args.prefill,
                              "devices": torch.cuda.device_count(),
                              "allgather_us_per_step": round(sum(a.elapsed_time(b) for a, b, _ in gather_events) / max(1, len(gather_events)) * 1e3, 1),
                              "allgather_bytes_per_rank": gather_events[0][2] if gather_events else None,
                              "allgather_calls": len(gather_events)}
            # the north star's ">= 6x frame-encode throughput at 8 GPUs vs 1" read off this one line, whatever the prefill mode:
            # aggregate encode rate = all ranks' frames / the SLOWEST rank's encode stage (its all-gather excluded), and what the
            # collective costs the step
            ag_ms = result["dist"]["allgather_us_per_step"] * 1e-3
            result["dist"].update({
                "vit_ms_max_over_ranks": round(vit_enc_ms_max, 3),
                "vit_frames_per_s_all_gpus": round(frames_total / (vit_enc_ms_max * 1e-3), 1),
                "vit_frames_per_s_per_gpu_min": round(B * T / (vit_enc_ms_max * 1e-3), 1),
                "allgather_share_of_step": round(ag_ms / ms_step, 5),
                "compare_with": "stages.vit_frames_per_s_per_gpu of the N = 1 line (same per-GPU workload: weak scaling)"})
        also = [] if (args.also == "none" or world > 1) else \
            (list(ALSO) if args.config == "c3" else []) if args.also == "auto" else [a for a in args.also.split(",") if a]
        also = [{"c4": "c4_n1", "decode": "c5_decode"}.get(a, a) for a in also]
        bad = [a for a in also if a not in ALSO]
        if bad:
            raise SystemExit(f"--also: unknown workload(s) {bad}; choose from c2, c4, decode, c3_fp16")
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"error": repr(e)}
        if sharded_also is not None:
            result["also"] = {"sharded_prefill": sharded_also}
        if also:
            # free this run's engines (26 + 25 GB of 13B weights, caches, workspaces) before the children build theirs
            del out, model, mm, tower, cache, frames
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            result["also"] = run_also(also, args.pack_weights)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
