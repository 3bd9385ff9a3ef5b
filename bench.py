#!/usr/bin/env python3
"""bench.py -- decode tokens/s + Q4_0 GEMV GB/s vs the HBM roofline, Llama-3-8B shape, on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run (one rank per GPU).  A "step" is one batch-1 greedy decode token: one pass of the
hot path (32 layers x 7 block-quantized GEMVs + classifier, RMSNorm/RoPE/softmax/KV-cache attention)
through the HIP backend.  W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
device synchronize; rank 0 prints ONE JSON line.  Weights are synthetic (no network): random GGUF-layout
Q4_0 blocks of the Llama-3-8B shapes, resident in HBM before timing starts.

Multi-GPU (`--gpus N`): Llama-3-8B Q4_0 (4.2 GB) fits one GPU, so per BASELINE/SURVEY 8(e) the ranks are
independent replicas decoding their own sequence (no data-path collective): weak scaling, value =
total tokens / max-over-ranks time.

Extra objects on the JSON line:
  roofline      the Q4_0 GEMV kernel: algorithmic bytes / HIP-event kernel time, measured live in an
                instrumented pass right after the timed region (event pairs on the kernel's own stream;
                kept out of the timed region so they do not perturb tokens/s).  peak = 8.0 TB/s HBM3E.
  cpu_baseline  the reference's SIMD CPU path restated in C (oracle/, AVX2 lane order, crabml's row-split
                thread pool) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the achievable copy rate
# HBM bytes per GEMV launch from rocprofv3 --pmc (FETCH_SIZE x2 correction for gfx950), see profiles/; None until measured
KERNEL_SOURCES = ("fused_ffn.hpp", "fused_common.hpp", "gemv_core.hpp", "devutil.hpp")  # what k_gateup_q is made of


def kernel_code_hash():
    """sha256 over the sources of the dominant kernel: profiles/pmc_traffic.json is stamped with it when the PMC pass is
    taken (tools/pmc_traffic.py), so a traffic figure measured on other code is never quoted."""
    import hashlib

    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "crabml_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass -- or None when that pass was
    taken on different kernel code (hash mismatch)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("kernel_code_hash") != kernel_code_hash():
            return {"stale": True, "source": pmc.get("source"), "measured_on": pmc.get("kernel_code_hash")}
        return pmc
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="llama3-8b", help="shape key in crabml_amd.synth.SHAPES")
    ap.add_argument("--wtype", default="Q4_0")
    ap.add_argument("--gguf", default=None,
                    help="decode a llama GGUF FILE instead of synthetic weights (C++ loader, crabml_amd/csrc/host/gguf.hpp); "
                         "the JSON line then describes that file -- not the BASELINE workload")
    ap.add_argument("--output-type", default=None,
                    help="GGML type of output.weight when it differs from --wtype (llama.cpp's Q4_0 files: Q6_K)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: truncate the layer count (INVALID as a result)")
    ap.add_argument("--no-affinity", action="store_true", help="do not move the process to the GPU's NUMA node")
    ap.add_argument("--no-trait", action="store_true", help="skip the reference-API leg (traced runs)")
    ap.add_argument("--path", default="auto", choices=["auto", "trait", "fused"],
                    help="trait = one launch per Tensor op (Llama2Runner unchanged); fused = fused decode step")
    ap.add_argument("--no-norm-epilogue", action="store_true", help="A/B: keep RMSNorm+quantize as its own launch")
    ap.add_argument("--flags", type=int, default=0, help="extra CRABML_HIP_LLAMA_* flags for A/B runs")
    ap.add_argument("--no-prefetch", action="store_true", help="A/B: disable Infinity-Cache weight prefetch")
    ap.add_argument("--tp", action="store_true",
                    help="WORLD_SIZE > 1: the ranks form ONE tensor-parallel group (RCCL all-reduce after wo / ffn_down) that "
                         "decodes a single token stream, instead of independent replicas; meant for --model llama3-70b")
    ap.add_argument("--tp-rccl", action="store_true",
                    help="with --tp: all-reduce through RCCL (ncclAllReduce per segment, the baseline) instead of the one-shot "
                         "P2P collective fused into the wo / ffn_down kernels")
    ap.add_argument("--tp-same-gpu", action="store_true",
                    help="with --tp: every rank uses GPU 0 and torch.distributed runs over gloo (the P2P group's hipIpc mapping works "
                         "between processes on one device; RCCL does not) -- the self-test of exactly this code path on a 1-GPU box")
    ap.add_argument("--tp-fail-p2p", action="store_true", help=argparse.SUPPRESS)  # test hooks: pretend the P2P / RCCL group cannot
    ap.add_argument("--tp-fail-rccl", action="store_true", help=argparse.SUPPRESS)  # be formed on rank 1 (the vote must carry it)
    ap.add_argument("--tp-dry", type=int, default=0,
                    help="measure ONE rank of a tensor-parallel group of this size with its all-reduces skipped "
                         "(per-rank kernel time; not a tokens/s result)")
    ap.add_argument("--tp-split-vocab", action="store_true",
                    help="with --tp-dry / --tp: the classifier split by vocabulary (each rank streams 1 / tp of output.weight)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the batched-prefill measurement")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--repeats", type=int, default=5, help="the K-step timed region is repeated this many times (same positions)")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--no-context", action="store_true", help="skip the long-context decode points")
    ap.add_argument("--no-c3", action="store_true",
                    help="skip the 128-step run over positions 0..127 (rocprofv3 --kernel-trace segfaults inside the tracer on 128 "
                         "back-to-back graph launches; tools/gpu_profile.sh passes this)")
    ap.add_argument("--no-gemv-points", action="store_true", help="skip the single-kernel GEMV points (SURVEY.md 8d)")
    ap.add_argument("--selftest-dist", action="store_true", help="CPU/gloo self test of the rank aggregation")
    return ap.parse_args()


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


class Dist:
    """barrier + max-over-ranks, via torch.distributed when WORLD_SIZE > 1 (RCCL on GPU, gloo on CPU)."""

    def __init__(self, world, local, cpu_only=False):
        self.world = world
        self.torch = None
        self.dev = None
        if world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            use_nccl = (not cpu_only) and torch.cuda.is_available()
            if use_nccl:
                try:  # RCCL over xGMI: only used for the barrier and the max/sum of two scalars
                    torch.cuda.set_device(local)
                    self.dev = torch.device("cuda", local)
                    dist.init_process_group("nccl", device_id=self.dev)
                except Exception as e:  # pragma: no cover - keep the bench alive on an odd fabric setup
                    print(f"[bench] nccl init failed ({e!r}); falling back to gloo", file=sys.stderr)
                    use_nccl = False
            if not use_nccl:
                self.dev = torch.device("cpu")
                dist.init_process_group("gloo")

    def barrier(self):
        if self.world > 1:
            if self.dev.type == "cuda":
                self.dist.barrier(device_ids=[self.dev.index])
                self.torch.cuda.synchronize()
            else:
                self.dist.barrier()

    def max_sum(self, elapsed, units):
        """(max over ranks of elapsed, sum over ranks of units)"""
        if self.world == 1:
            return elapsed, units
        t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.dev)
        u = self.torch.tensor([float(units)], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(u, op=self.dist.ReduceOp.SUM)
        return float(t.item()), float(u.item())

    def gather(self, obj):
        """every rank's object, in rank order"""
        if self.world == 1:
            return [obj]
        box = [None] * self.world
        self.dist.all_gather_object(box, obj)
        return box

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def selftest_dist(args):
    rank, world, local = dist_env()
    d = Dist(world, local, cpu_only=True)
    d.barrier()
    t, u = d.max_sum(1.0 + rank, 10)
    d.barrier()
    if rank == 0:
        print(json.dumps({"selftest": "dist", "n_gpus": world, "max_elapsed": t, "units": u}))
    d.close()


def cpu_baseline(model, steps_budget_s):
    """crabml's CPU path (restated, oracle/) on this box's host cores: full-model greedy decode, AVX2 dot
    order, crabml's row-split pool.  Bounded sample: a few tokens per thread count."""
    from oracle import oracle as o
    from tests.helpers import to_oracle

    ncpu = os.cpu_count() or 1
    results = []
    # crabml's CLI default (-T 2, main.rs:49-50) and 32 threads.  (Round 3 also timed one thread per host CPU: 1.1-1.4 tok/s on
    # 256 CPUs -- the row-split pool oversubscribes the memory channels -- for 8 s of the run; dropped.)
    legs = sorted({2, min(ncpu, 32)})
    for threads in legs:
        odev = o.OracleDevice(thread_num=threads, use_avx2=True)
        conf, w = to_oracle(model, odev)
        r = o.OracleLlamaRunner(conf, w, odev, 64, True)
        r.forward([1], 0)  # touch all pages once (untimed)
        tok, pos, n = o.argmax_last(r.logits), 1, 0
        t0 = time.perf_counter()
        budget = steps_budget_s / len(legs)
        while True:
            r.forward([tok], pos)
            tok = o.argmax_last(r.logits)
            pos += 1
            n += 1
            el = time.perf_counter() - t0
            if el > budget or n >= 32:
                break
        results.append((n / el, threads, n, el))
        del r, w, odev
    best = max(results)
    return {
        "value": round(best[0], 3), "unit": "tokens/s", "cores": best[1], "kind": "port",
        "host_cpus": ncpu, "avx2": bool(o.lib().co_have_avx2()),
        "sample": "full-model greedy decode, same synthetic weights; " + "; ".join(
            f"T={t}: {n} tokens in {el:.2f}s = {v:.2f} tok/s" for v, t, n, el in results),
        "note": "reference is Rust nightly (no rustc here): its AVX2 CPU path restated in C (oracle/crabml_oracle.c)",
    }


def gemv_points(ca, synth, dev, fname="Q4_0"):
    """SURVEY.md 8(d), the second half of BASELINE's metric ("Q4_0 GEMV GB/s vs HBM roofline"): crabml_hip_matmul_vec alone on the
    four Llama-3-8B shapes.  Every launch reads a DIFFERENT weight buffer out of >= 512 MB of distinct copies (cycled, so the
    256 MB Infinity Cache never holds the next one), 20 warm-up + 200 timed launches, kernel time from the dispatch's own
    start / stop events: median, p10, p90, algorithmic GB/s = (m k / 32 * 18 + 4 k + 4 m) / median, fraction of 8 TB/s."""
    import numpy as np

    typ = synth.TYPE_BY_NAME[fname]
    gt = {synth.Q4_0: ca.GGMLType.Q4_0, synth.Q8_0: ca.GGMLType.Q8_0, synth.Q4_1: ca.GGMLType.Q4_1}[typ]
    rng = np.random.default_rng(3)
    pts = {}
    for (m, k) in [(4096, 4096), (14336, 4096), (4096, 14336), (128256, 4096)]:
        wbytes = m * k // synth.BLOCK_ELEMS[typ] * synth.BLOCK_BYTES[typ]
        algo = wbytes + 4 * k + 4 * m
        ncopies = max(2, -(-536_870_912 // wbytes))
        raw = synth.random_blocks(rng, m * k, typ)
        ws = [ca.HipTensor.from_cpu(np.roll(raw, 4096 * c), [m, k], gt, dev) for c in range(ncopies)]
        x = ca.HipTensor.new(rng.standard_normal(k).astype(np.float32), [k], dev)
        for i in range(20):
            ws[i % ncopies].matmul_vec(x)
        dev.sync()
        dev.prof_enable(True)
        for i in range(200):
            ws[i % ncopies].matmul_vec(x)
        ms = dev.prof_read_launches()
        dev.prof_enable(False)
        us = np.sort(ms.astype(np.float64) * 1e3)
        med, p10, p90 = float(np.median(us)), float(us[len(us) // 10]), float(us[len(us) * 9 // 10])
        pts[f"{m}x{k}"] = {"algo_MB": round(algo / 1e6, 2), "distinct_weight_MB": round(ncopies * wbytes / 1e6), "launches": int(len(us)),
                           "median_us": round(med, 2), "p10_us": round(p10, 2), "p90_us": round(p90, 2),
                           "GBps": round(algo / med / 1e3, 1), "frac_of_peak": round(algo / med / 1e3 / HBM_PEAK_GBS, 4)}
        del ws, x
    return {"format": fname, "kernel": "crabml_hip_matmul_vec (k_gemv, one launch per call)", "peak_GBps": HBM_PEAK_GBS, "points": pts,
            "method": "200 timed launches per shape after 20 warm-ups, each on a different weight buffer (cycling >= 512 MB), "
                      "hipExtLaunchKernelGGL start/stop events"}


def parity_check(ca, synth, model, conf, weights, dev, ordinal, n_pos=4):
    """The benchmarked path against the oracle, on the benchmarked model AND on its zero-mean twin: teacher-forced on fixed tokens,
    (a) the fast fused step (hipGraph, 5-kernel layers, hop-free norm -- what `value` times) and the same step with RMSNorm's
    division kept in the producing launch (EXACT_NORM) -> max relative logit error and greedy-token agreement, next to the distance
    between the reference's OWN two builds (scalar and AVX2 order of the block dots, both restated in oracle/) on the same tokens in
    the same run: the yardstick for what any re-association costs on that model; (b) a STRICT-order device
    (CRABML_HIP_FLAG_STRICT_ORDER) -> bit-identical logits, with its tokens/s.  The oracle is test infrastructure: checker only,
    outside every timed region."""
    import numpy as np

    from oracle import oracle as o
    from tests.helpers import to_oracle

    EXACT_NORM = 8388608
    toks = [1, 365, 400, 282, 9906, 7, 9, 11][:n_pos]
    ncpu = os.cpu_count() or 1

    def oracle_logits(m, avx2):
        odev = o.OracleDevice(thread_num=max(2, min(ncpu, 64)), use_avx2=avx2)
        oconf, ow = to_oracle(m, odev)
        orr = o.OracleLlamaRunner(oconf, ow, odev, 64, True)
        return [orr.forward([t], i).copy() for i, t in enumerate(toks)]

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

    def one_model(m, hconf, hw, tag):
        t0 = time.perf_counter()
        ref = oracle_logits(m, False)
        t_oracle = time.perf_counter() - t0
        avx = oracle_logits(m, True)
        spread = [rel(a, r) for a, r in zip(avx, ref)]
        row = {"model": tag, "oracle_s": round(t_oracle, 2),
               "reference_avx2_vs_scalar_per_pos": [float("%.3g" % e) for e in spread], "reference_avx2_vs_scalar_max": float("%.3g" % max(spread)),
               "reference_avx2_vs_scalar_tokens_equal": [bool(o.argmax_last(a) == o.argmax_last(r)) for a, r in zip(avx, ref)]}
        for name, flags in (("fast", 0), ("fast_exact_norm", EXACT_NORM)):
            f = ca.HipLlamaRunner(hconf, hw, dev, 64, True, extra_flags=flags)
            errs, equal = [], []
            for i, t in enumerate(toks):
                lg = f.forward(t, i)
                errs.append(rel(lg, ref[i]))
                equal.append(bool(o.argmax_last(lg) == o.argmax_last(ref[i])))
            del f
            row[name + "_rel_logit_err_per_pos"] = [float("%.3g" % e) for e in errs]
            row[name + "_max_rel_logit_err"] = float("%.3g" % max(errs))
            row[name + "_tokens_equal"] = equal
            row[name + "_over_reference_spread"] = round(max(errs) / max(spread), 3) if max(spread) > 0 else None
        # `value` is timed through the reference's API: the UNCHANGED runner, its Tensor calls recorded and served by the fused step.
        # Its logits on these tokens must be the fused entry point's, bit for bit (same launches, same buffers' worth of arithmetic),
        # and every token must have been served by the fused step -- asserted in tests/, recorded here on the benchmarked weights
        try:
            f = ca.HipLlamaRunner(hconf, hw, dev, 64, True)
            r = ca.Llama2Runner(hconf, hw, dev, 64, True)
            st0 = dev.lazy_stats()["fused_tokens"]
            same = []
            for i, t in enumerate(toks):
                a_ = np.asarray(f.forward(t, i)).copy()
                b_ = np.asarray(r.forward([t], i)).copy()
                same.append(bool(np.array_equal(a_.view(np.uint32), b_.view(np.uint32))))
            row["reference_api_equals_fused_entry_point_bitwise"] = same
            row["reference_api_tokens_served_by_fused_step"] = int(dev.lazy_stats()["fused_tokens"] - st0)
            del f, r
        except Exception as e:  # pragma: no cover
            row["reference_api_check_error"] = repr(e)
        return row, ref

    bench_row, ref = one_model(model, conf, weights, "the benchmark model (synthetic Q4_0 blocks, every block scale d > 0)")
    out = {"tokens_compared": len(toks), "fast_max_rel_logit_err": bench_row["fast_max_rel_logit_err"],
           "fast_rel_logit_err_per_pos": bench_row["fast_rel_logit_err_per_pos"], "fast_tokens_equal": bench_row["fast_tokens_equal"],
           "oracle": "scalar-order restatement of the reference CPU path (AVX2 order: the same restatement with the AVX2 build's association)",
           "oracle_s": bench_row["oracle_s"], "models": [bench_row],
           "yardstick": "reference_avx2_vs_scalar: the distance between the reference's own two builds on the same model and tokens, computed in "
                        "this run; *_over_reference_spread = the HIP fast path's error divided by it (re-association + a truncating rhs quantizer: "
                        "one ulp moves a block's largest element between the levels 126 and 127)"}
    if model.wtype == synth.Q4_0:
        # the zero-mean twin: block scales of either sign (the hard case; the benchmark's d > 0 blocks carry a common-mode component).
        # The flip is an involution; it is undone only if it completed (a half-flipped model must not reach the timed legs: then
        # the run stops here)
        flipped = False
        try:
            synth.flip_scale_signs(model, 5)
            flipped = True
            zconf, zw = synth.to_hip(model, dev)
            zrow, _ = one_model(model, zconf, zw, "the same blocks with d of either sign (zero-mean weights)")
            out["models"].append(zrow)
            del zw
        except Exception as e:  # pragma: no cover
            out["zero_mean_error"] = repr(e)
            if not flipped:
                raise RuntimeError("parity_check: the sign flip of the benchmark model failed part-way; its weights are not the benchmark's any more") from e
        finally:
            if flipped:
                synth.flip_scale_signs(model, 5)  # (the benchmark model is back)
    try:
        sdev = ca.HipTensorDevice(ordinal, False, 0, True)
        sconf, sw = synth.to_hip(model, sdev)
        strict = ca.HipLlamaRunner(sconf, sw, sdev, 160, True)
        ident = []
        for i, t in enumerate(toks):
            lg = strict.forward(t, i)
            ident.append(bool(np.array_equal(lg.view(np.uint32), ref[i].view(np.uint32))))
        strict.reset()
        strict.decode_greedy(1, 8)  # (graph capture + warm-up)
        sdev.sync()
        best = None
        for _ in range(3):
            strict.reset()
            strict.decode_greedy(1, 8)
            sdev.sync()
            t0 = time.perf_counter()
            strict.decode_greedy(1, 64)
            sdev.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["strict_bit_identical"] = ident
        out["strict_tokens_per_s"] = round(64 / best, 2)
        out["strict_note"] = ("CRABML_HIP_FLAG_STRICT_ORDER: every sum in the reference's scalar order (block terms added in block order, "
                              "RMSNorm scan and chunk order, sequential softmax sums, f16 attention chains); 64 greedy steps after 8 warm-up steps, "
                              "best of 3; the same device through the reference's unchanged runner: strict_reference_api_tokens_per_s")
        # the bit-identical tier through the reference's own API (Llama2Runner<HipTensor>, recorded-op queue -> fused strict step)
        tr = ca.Llama2Runner(sconf, sw, sdev, 160, True)
        tok = int(tr.timed_decode(1, 8)[0][-1])
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            tok = int(tr.timed_decode(tok, 32)[0][-1])
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["strict_reference_api_tokens_per_s"] = round(32 / best, 2)
        del strict, tr, sw
    except Exception as e:  # pragma: no cover
        out["strict_error"] = repr(e)
    return out


def tp_dry_run(args, ca, synth, local):
    """One rank of a tensor-parallel group, all-reduces skipped: what the kernels of a rank cost per token.
    NOT a tokens/s result (the collective is the other half, and is not measurable on a 1-GPU box)."""
    from crabml_amd import tp as tp_mod

    shape = synth.SHAPES[args.model]
    wtype = synth.TYPE_BY_NAME[args.wtype]
    n = args.tp_dry
    tp_mod.check_tp(shape, n, wtype, True)
    model = synth.build_model(shape, wtype, seed=8, n_layers=args.layers, tp=n, tp_split_vocab=args.tp_split_vocab)
    dev = ca.HipTensorDevice(device_ordinal=local)
    conf, weights = synth.to_hip(model, dev)
    seq_len = args.warmup + args.steps + 16
    r = ca.HipLlamaRunner(conf, weights, dev, seq_len, True, True, not args.no_prefetch, tp_size=n, tp_rank=0,
                          extra_flags=128 | (1048576 if args.tp_split_vocab else 0) | args.flags)
    r.decode_greedy(1, args.warmup)
    dev.sync()
    t0 = time.perf_counter()
    r.decode_greedy(1, args.steps)
    dev.sync()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    local_bytes = sum(t.data.nbytes for name, t in model.tensors.items()
                      if not name.endswith("_norm.weight") and name != "token_embd.weight")
    # per-stage kernel times of the same rank (eager replay, dispatch-timestamp events)
    stage_names = {1: "qkv", 2: "wo", 3: "gate|up", 4: "ffn_down", 5: "classifier", 6: "norm", 7: "attention", 8: "softmax", 9: "pv"}
    per_stage = None
    try:
        del r
        e = ca.HipLlamaRunner(conf, weights, dev, 64, True, False, not args.no_prefetch, tp_size=n, tp_rank=0,
                              extra_flags=128 | (1048576 if args.tp_split_vocab else 0) | args.flags)
        e.decode_greedy(1, args.warmup)
        dev.sync()
        dev.prof_enable(True)
        e.decode_greedy(1, 8)
        recs = dev.prof_read()
        dev.prof_enable(False)
        per_stage = {stage_names.get(x["stage"], str(x["stage"])): {"avg_us": round(x["kernel_ms"] * 1e3 / x["launches"], 2),
                                                                     "launches_per_token": x["launches"] / 8,
                                                                     "algo_MB": round(x["algo_bytes"] / x["launches"] / 1e6, 2)}
                     for x in sorted(recs, key=lambda x: x["stage"])}
    except Exception as ex:
        per_stage = {"error": repr(ex)}
    print(json.dumps({
        "measurement": "tensor-parallel dry run: one rank's kernels per token, all-reduces skipped",
        "per_stage": per_stage,
        "model": shape.name, "wtype": args.wtype, "tp_size": n, "ms_per_token_rank_kernels": round(ms, 4),
        "rank_weight_bytes_per_token": local_bytes,
        "rank_effective_GBps": round(local_bytes / ms / 1e6, 1),
        "all_reduces_per_token": 2 * conf.n_layers, "all_reduce_bytes": shape.dim * 4,
        "launches_per_layer": 5, "classifier": "split by vocabulary (1 / tp of output.weight per rank)" if args.tp_split_vocab else "replicated",
        "note": "not tokens/s: the rank runs the fused-collective layer (5 launches: q|k|v, attention, wo + exchange + norm, gate|up, "
                "down + exchange + norm) with the exchange itself skipped; add 2 x n_layers one-shot exchanges of dim x 8 bytes "
                "per peer over xGMI (not measurable on one GPU)",
    }))


def tp_group_run(args, ca, synth, dist, rank, world, local):
    """--tp under torchrun: one tensor-parallel group over all ranks (BASELINE config 5).  Every rank
    builds the LOCAL shard shapes with the same seed (identical bytes on every rank: a consistent model whose shards
    happen to be equal, so all ranks sample the same tokens) and decodes ONE token stream with two all-reduces per layer.

    Collective: the one-shot P2P all-reduce over hipIpc-mapped inboxes, fused into the wo / ffn_down kernels (the production
    form) -- and if ANY rank cannot form that group or cannot complete a first step through it (peer mapping refused, a fault
    raised by a bounded poll), every rank falls back to RCCL (ncclAllReduce per segment) and the line says which one ran and
    why (`config.collective`, `config.collective_fallback`).  The node's hipDeviceCanAccessPeer matrix and every rank's own
    exchange time ride along, so that one run on a real node is enough to read what happened."""
    from crabml_amd import tp as tp_mod

    shape = synth.SHAPES[args.model]
    wtype = synth.TYPE_BY_NAME[args.wtype]
    tp_mod.check_tp(shape, world, wtype, True)
    ordinal = 0 if args.tp_same_gpu else local
    dev = ca.HipTensorDevice(device_ordinal=ordinal)
    seq_len = args.warmup + args.steps + 16

    def all_ok(ok, why=""):
        """(every rank succeeded, the first failure text)"""
        votes = dist.gather((bool(ok), str(why)))
        bad = [f"rank {r}: {w}" for r, (o, w) in enumerate(votes) if not o]
        return not bad, "; ".join(bad)

    def build(kind):
        """model shards, communicator and runner for one collective kind; raises on this rank's own failure"""
        split_vocab = args.tp_split_vocab and kind == "p2p"
        model = synth.build_model(shape, wtype, seed=8, n_layers=args.layers, tp=world, tp_split_vocab=split_vocab)
        conf, weights = synth.to_hip(model, dev)
        if kind == "rccl":
            if args.tp_fail_rccl and rank == world - 1:
                raise RuntimeError("RCCL group refused (test hook --tp-fail-rccl)")
            if args.tp_fail_rccl:
                raise RuntimeError("skipped: a peer cannot join (test hook --tp-fail-rccl)")
            comm = tp_mod.init_tp_comm(dev, rank, world, tp_mod.torch_broadcast(rank))
        else:
            if args.tp_fail_p2p and rank == world - 1:
                raise RuntimeError("P2P group refused (test hook --tp-fail-p2p)")
            if args.tp_fail_p2p:
                raise RuntimeError("skipped: a peer cannot join (test hook --tp-fail-p2p)")
            comm = tp_mod.init_tp_p2p(dev, rank, world, shape.dim, tp_mod.torch_all_gather(world))
        xflags = args.flags | (1048576 if split_vocab else 0)
        r = ca.HipLlamaRunner(conf, weights, dev, seq_len, True, True, not args.no_prefetch, tp_size=world, tp_rank=rank, comm=comm,
                              extra_flags=xflags)
        return model, conf, weights, comm, r, xflags, split_vocab

    kinds = ["rccl"] if args.tp_rccl else ["p2p", "rccl"]
    fallback = None
    state = None
    for kind in kinds:
        err = ""
        try:
            state = build(kind)
        except Exception as e:  # noqa: BLE001 -- whatever stops this rank must reach the vote
            state, err = None, f"{type(e).__name__}: {e}"
        ok, why = all_ok(state is not None, err)
        if ok:  # one step through the collective before anything is timed: a fault raised by a bounded poll surfaces here
            err = ""
            try:
                state[4].decode_greedy(1, 1)
                dev.sync()
                state[4].reset()
            except Exception as e:  # noqa: BLE001
                err = f"first step: {type(e).__name__}: {e}"
            ok, why = all_ok(not err, err)
        if ok:
            break
        state = None
        fallback = f"{kind} unavailable ({why})"
        if rank == 0:
            print(f"[bench --tp] {fallback}; trying the next collective", file=sys.stderr, flush=True)
    if state is None:
        if rank == 0:
            print(json.dumps({"metric": "decode tokens/sec, tensor-parallel", "value": None, "n_gpus": world, "error": fallback}), flush=True)
        dist.close()
        return 2
    model, conf, weights, comm, r, xflags, split_vocab = state
    kind_ran = kind

    peers = None
    if dist.torch is not None and dist.torch.cuda.is_available():
        n_dev = dist.torch.cuda.device_count()
        try:
            peers = [[1 if i == j or dist.torch.cuda.can_device_access_peer(i, j) else 0 for j in range(n_dev)] for i in range(n_dev)]
        except Exception:  # noqa: BLE001
            peers = None

    ids_w = r.decode_greedy(1, args.warmup) if args.warmup > 0 else [1]
    tok = int(ids_w[-1])
    dev.sync()
    dist.barrier()
    t0 = time.perf_counter()
    ids = r.decode_greedy(tok, args.steps)
    dev.sync()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max, _ = dist.max_sum(elapsed, args.steps)
    ids_all = dist.gather([int(t) for t in ids])
    # the same rank with its exchanges skipped (same weights, same kernels, TP_DRY_RUN): step time minus this = what the
    # collectives (2 per layer + the sampler's pair exchange) cost per token on this node -- one run reports both
    dry = ca.HipLlamaRunner(conf, weights, dev, seq_len, True, True, not args.no_prefetch, tp_size=world, tp_rank=rank,
                            extra_flags=xflags | 128)
    dry.decode_greedy(1, args.warmup)
    dev.sync()
    dist.barrier()
    t0 = time.perf_counter()
    dry.decode_greedy(1, args.steps)
    dev.sync()
    dry_own = time.perf_counter() - t0
    dry_max, _ = dist.max_sum(dry_own, args.steps)
    per_rank = dist.gather({"step_ms": round(elapsed / args.steps * 1e3, 4), "rank_kernels_ms": round(dry_own / args.steps * 1e3, 4),
                            "exchange_ms": round((elapsed - dry_own) / args.steps * 1e3, 4), "device": ordinal})
    del dry
    if rank == 0:
        local_bytes = sum(t.data.nbytes for name, t in model.tensors.items()
                          if not name.endswith("_norm.weight") and name != "token_embd.weight")
        tps = args.steps / elapsed_max  # ONE token stream for the whole group
        print(json.dumps({
            "metric": f"decode tokens/sec (batch-1 greedy), {shape.name} shape {args.wtype}, tensor-parallel over {world} GPUs",
            "value": round(tps, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "rank_kernels_ms_per_step": round(dry_max / args.steps * 1e3, 4),
            "exchange_ms_per_step": round((elapsed_max - dry_max) / args.steps * 1e3, 4),
            "per_rank": per_rank,
            "ranks_sampled_the_same_tokens": all(x == ids_all[0] for x in ids_all),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.wtype, "data": "synthetic",
            "config": {"workload": f"{shape.name}-shape all-{args.wtype} synthetic weights, one batch-1 greedy token stream, f16 KV cache, "
                                   f"positions {args.warmup}..{args.warmup + args.steps - 1}",
                       "parallelism": f"tp{world}", "all_reduces_per_token": 2 * conf.n_layers, "all_reduce_bytes": shape.dim * 4,
                       "collective": "RCCL ncclAllReduce per segment" if kind_ran == "rccl" else
                                     "one-shot P2P all-reduce over hipIpc-mapped inboxes, fused into the wo / ffn_down epilogue",
                       "collective_kind": kind_ran, "collective_fallback": fallback,
                       "same_gpu_self_test": bool(args.tp_same_gpu),
                       "peer_access_matrix": peers,
                       "classifier": "split by vocabulary, per-shard arg-max + 8-byte pair exchange" if split_vocab else "replicated",
                       "rank_weight_bytes_per_token": local_bytes},
            "roofline": {"bound": "hbm", "achieved": round(tps * local_bytes / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(tps * local_bytes / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "note": "per-rank effective weight bandwidth over the whole step (kernels + all-reduces), not one kernel"},
        }), flush=True)
    del r
    del comm
    dist.close()
    return 0


def main():
    args = parse()
    if args.selftest_dist:
        return selftest_dist(args)
    rank, world, local = dist_env()
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = Dist(world, local, cpu_only=args.tp_same_gpu)  # imports torch first when world > 1 (one HIP runtime in the process)

    import crabml_amd as ca
    from crabml_amd import synth

    if args.tp_dry > 1:
        return tp_dry_run(args, ca, synth, local)
    if args.tp and world > 1:
        rc = tp_group_run(args, ca, synth, dist, rank, world, local)
        if rc:
            sys.exit(rc)
        return
    t_build = time.perf_counter()
    dev = ca.HipTensorDevice(device_ordinal=local)
    model = None
    if args.gguf:
        gf = ca.GGUFFile(args.gguf)
        conf = gf.load_config()
        infos = gf.tensor_infos()
        by_name = {t[0]: t for t in infos}
        wtype = by_name["blk.0.attn_q.weight"][2]
        args.wtype = synth.TYPE_NAMES.get(wtype, str(wtype))
        shape = synth.ModelShape(os.path.basename(args.gguf), conf.embedding_dim, conf.hidden_dim, conf.n_layers, conf.n_heads,
                                 conf.n_kv_heads, conf.vocab_size, conf.seq_len, conf.rms_norm_eps, conf.rope_dim)

        def nbytes(t):
            n = 1
            for d in t[1]:
                n *= d
            return n // synth.BLOCK_ELEMS[t[2]] * synth.BLOCK_BYTES[t[2]]

        gemv_bytes = sum(nbytes(t) for t in infos if not t[0].endswith("_norm.weight") and t[0] != "token_embd.weight")
        if "output.weight" not in by_name:  # tied classifier
            gemv_bytes += nbytes(by_name["token_embd.weight"])
        t_upload = time.perf_counter()
        weights = gf.load_weights(conf, dev)
    else:
        shape = synth.SHAPES[args.model]
        k_m = args.wtype.upper() == "Q4_K_M"  # llama.cpp's mix: Q4_K body, attn_v / ffn_down in Q6_K on some layers, Q6_K classifier
        wtype = synth.Q4_K if k_m else synth.TYPE_BY_NAME[args.wtype]
        model = synth.build_model(shape, wtype, seed=8, n_layers=args.layers, k_m_mix=k_m,
                                  output_type=synth.TYPE_BY_NAME[args.output_type] if args.output_type else None)
        t_upload = time.perf_counter()
        conf, weights = synth.to_hip(model, dev)
        gemv_bytes = model.gemv_weight_bytes_per_token()
    dev.sync()
    t_upload = time.perf_counter() - t_upload
    t_build = time.perf_counter() - t_build
    seq_len = max(args.warmup + 2 * args.steps + 16, 144)
    # host placement: run this process's threads on the NUMA node the GPU is attached to (what `numactl --cpunodebind` would do).
    # The fused entry point does not care (one blocking call for K tokens); a host that drives the device token by token -- the
    # reference's runner -- does: launches, the logits in pinned memory and the completion flag cross the socket otherwise
    # (tools/trait_var.py: 670-735 tok/s from the far socket, 760 from the near one, same box).
    host_affinity = None
    if not args.no_affinity:
        try:
            host_affinity = ca.pin_host_to_device_node(dev)
        except Exception as e:  # pragma: no cover
            host_affinity = "unchanged (" + repr(e) + ")"
    trait_seq = max(seq_len, args.warmup + 3 * max(args.steps, 16) + 16)
    trait = ca.Llama2Runner(conf, weights, dev, trait_seq, True)  # f16 KV cache = the CLI default (main.rs:250)
    path = args.path
    fused = None
    if path in ("auto", "fused"):
        try:
            fused = ca.HipLlamaRunner(conf, weights, dev, seq_len, True, True, not args.no_prefetch,
                                      norm_epilogue=not args.no_norm_epilogue, extra_flags=args.flags)
            path = "fused"
        except ca.CrabmlError:
            if path == "fused":
                raise
            path = "trait"

    # What `value` times.  Default: the reference's OWN API -- Llama2Runner<HipTensor>::forward + the host-side greedy sampler per
    # token, unchanged (what `crabml-cli -D hip` runs, patches/0002); its Tensor calls are recorded and served by the fused step
    # (csrc/lazy.hpp).  `--path fused`: crabml_hip_llama_decode_greedy (the step from its hipGraph, arg-max on the device, one blocking
    # call for K tokens) -- reported as `fused_entry_point` next to the headline otherwise.
    headline = "fused" if (args.path == "fused" and path == "fused") else "reference"

    def decode(tok, n, which=None):
        """n greedy decode steps, returns the last sampled token"""
        if (which or headline) == "fused":
            return int(fused.decode_greedy(tok, n)[-1])
        return int(trait.timed_decode(tok, n)[0][-1])

    # ---- warm-up (untimed), then the timed region: EXACTLY K steps between barrier + synchronize ------------------
    # The region is repeated `--repeats` times over the SAME positions (the sequence is rewound: kv length back to 0,
    # W warm-up steps, K timed steps), so the repeats are comparable; `value` is the MEDIAN region, all of them are listed.
    def timed_regions(which):
        nonlocal trait
        out_r = []
        for rep in range(max(1, args.repeats)):
            if which == "fused":
                fused.reset()
            elif rep > 0 or trait.kv_cache_len() > 0:
                trait = ca.Llama2Runner(conf, weights, dev, trait_seq, True)  # (an empty cache: the runner has no rewind)
            tok = decode(1, args.warmup, which) if args.warmup > 0 else 1
            dev.sync()
            dist.barrier()
            t0 = time.perf_counter()
            tok = decode(tok, args.steps, which)
            dev.sync()
            dist.barrier()
            out_r.append(dist.max_sum(time.perf_counter() - t0, args.steps))
        out_r.sort()
        return out_r

    regions = timed_regions(headline)
    elapsed_max, total_tokens = regions[len(regions) // 2]
    fused_regions = timed_regions("fused") if (headline == "reference" and path == "fused") else None

    # ---- instrumented pass: HIP event pairs around every GEMV-stage launch (same process, same weights),
    # taken RIGHT AFTER the timed regions (before the other legs create more contexts and devices) ---
    # Events cannot live inside the captured graph, so the fused step is replayed EAGERLY (identical kernels
    # and launch geometry) with one event pair per GEMV stage, on the backend's own stream.
    roof = None
    if rank == 0:
        STAGES = {0: "matmul_vec (per-op path)", 1: "k_qkv (wq|wk|wv + rope + KV append)", 2: "wo GEMV + residual (+ next rmsnorm/quantize epilogue)",
                  3: "k_gateup_q (gate|up + silu*mul + quantize)", 4: "ffn_down GEMV + residual (+ next rmsnorm/quantize epilogue)",
                  5: "k_gemv (classifier)"}
        n_prof = min(args.steps, 16)
        if path == "fused":
            eager = ca.HipLlamaRunner(conf, weights, dev, n_prof + 8, True, False, not args.no_prefetch,
                                      norm_epilogue=not args.no_norm_epilogue, extra_flags=args.flags)
            eager.decode_greedy(1, 4)  # warm
            dev.sync()
            dev.prof_enable(True)
            eager.decode_greedy(1, n_prof)
        else:
            dev.prof_enable(True)
            trait.timed_decode(1, n_prof)
        recs = [r for r in dev.prof_read() if r["dtype"] == wtype and r["kernel_ms"] > 0]
        dev.prof_enable(False)
        if recs:
            tot_b = sum(r["algo_bytes"] for r in recs)
            tot_ms = sum(r["kernel_ms"] for r in recs)
            launches = sum(r["launches"] for r in recs)
            dom = max(recs, key=lambda r: r["kernel_ms"])  # the kernel most of the GEMV time goes to
            d_bytes = dom["algo_bytes"] / dom["launches"]
            d_us = dom["kernel_ms"] * 1e3 / dom["launches"]
            gbs = d_bytes / (d_us * 1e-6) / 1e9
            pmc = _pmc_traffic()
            pmc_ok = bool(pmc) and not pmc.get("stale") and path == "fused" and args.wtype == "Q4_0" and args.model == "llama3-8b"
            try:  # what a plain streaming-read kernel reaches on THIS box, right now (1 GiB, best of 5)
                ceiling = dev.read_ceiling_gbps(1 << 30, 5)
            except Exception:
                ceiling = None
            roof = {
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4),
                "measured_read_ceiling": round(ceiling, 1) if ceiling else None,
                "frac_of_measured": round(gbs / ceiling, 4) if ceiling else None,
                "traffic": pmc["hbm_bytes_per_launch"] if pmc_ok else None,
                "traffic_source": (pmc["source"] if pmc_ok else ("stale: PMC pass was taken on other kernel code (" + str(pmc.get("measured_on")) + " vs " +
                                                                kernel_code_hash() + ")" if pmc and pmc.get("stale") else None)),
                "kernel_code_hash": kernel_code_hash(),
                "kernel": STAGES.get(dom["stage"], "?"),
                "avg_launch_us": round(d_us, 3),
                # the same fraction from the COMMITTED rocprofv3 --kernel-trace of this command (the tracer adds ~0.3 us per launch):
                # quoted only while the trace was taken on the kernel code this run executes
                "frac_rocprof": (round(d_bytes / (pmc["rocprof_avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                 if pmc_ok and pmc.get("rocprof_avg_launch_us") else None),
                "rocprof_avg_launch_us": pmc.get("rocprof_avg_launch_us") if pmc_ok else None,
                "rocprof_source": pmc.get("rocprof_source") if pmc_ok else None,
                "algo_bytes_per_launch": round(d_bytes, 1),
                "launches_per_token": dom["launches"] / n_prof,
                "all_gemv_stages": {
                    "GBps": round(tot_b / (tot_ms * 1e-3) / 1e9, 1),
                    "frac": round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "launches_per_token": launches / n_prof,
                    "gemv_ms_per_token": round(tot_ms / n_prof, 4),
                },
                "per_stage": {STAGES.get(r["stage"], str(r["stage"])): {
                    "launches_per_token": r["launches"] / n_prof,
                    "avg_us": round(r["kernel_ms"] * 1e3 / r["launches"], 2),
                    "algo_MB_per_launch": round(r["algo_bytes"] / r["launches"] / 1e6, 2),
                    "GBps": round(r["algo_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9, 1)} for r in recs},
                "method": "hipExtLaunchKernelGGL start/stop events (dispatch begin/end timestamps) on the backend's own "
                          "stream around each GEMV-stage launch; the fused step is replayed eagerly for this (graphs "
                          "cannot carry events) right after the timed region; rocprofv3 --kernel-trace durations of the "
                          "same command are committed under profiles/",
            }

    # SURVEY.md config C3 quotes decode over positions 0..127 from an empty cache; the driver's arguments (--steps 20 --warmup 5)
    # time positions 5..24, where attention is nearly free.  Reported next to `value`: 128 steps from position 0, best of 3.
    c3 = None
    if rank == 0 and path == "fused" and not args.no_c3:
        best = None
        for _ in range(3):
            fused.reset()
            dev.sync()
            tc = time.perf_counter()
            fused.decode_greedy(1, 128)
            dev.sync()
            dtc = time.perf_counter() - tc
            best = dtc if best is None else min(best, dtc)
        c3 = {"positions": "0..127", "tokens_per_s": round(128 / best, 2), "ms_per_step": round(best / 128 * 1e3, 4),
              "note": "128 greedy steps from an empty KV cache (SURVEY.md config C3), host clock around one blocking call, best of 3"}

    # The reference's API: Llama2Runner<HipTensor> UNCHANGED, one Tensor call after the other (what `crabml-cli -D hip` runs,
    # patches/0002).  Since ABI version 2 the calls are recorded and a decode token is served by the fused step (csrc/lazy.hpp);
    # the same runner on a device that launches every call immediately (ABI version 1, "per-op") is timed next to it.
    trait_tps = None
    trait_info = None
    if rank == 0 and path == "fused" and not args.no_trait:
        n_t = max(args.steps, 16)
        trait = ca.Llama2Runner(conf, weights, dev, trait_seq, True)  # (an empty cache for this leg)
        st0 = dev.lazy_stats()
        t_tok = int(trait.timed_decode(1, args.warmup if args.warmup > 0 else 2)[0][-1])
        dev.sync()
        best = None
        split = None
        for _ in range(3):
            sa = dev.lazy_stats()
            tt = time.perf_counter()
            ids_t, _sec, samp = trait.timed_decode(t_tok, n_t)
            t_tok = int(ids_t[-1])
            dev.sync()
            dt = time.perf_counter() - tt
            sb = dev.lazy_stats()
            if best is None or dt < best:
                best = dt
                split = {"host_blocked_in_export_ms_per_step": round((sb["wait_ns"] - sa["wait_ns"]) / n_t * 1e-6, 4),
                         "host_argmax_ms_per_step": round(samp / n_t * 1e3, 4)}
        trait_tps = n_t / best
        st1 = dev.lazy_stats()
        trait_info = {"tokens_per_s": round(trait_tps, 2), "ms_per_step": round(best / n_t * 1e3, 4), "steps": n_t,
                      "api": "Llama2Runner<HipTensor>::forward + host arg-max per token (the reference's generic runner, unchanged); "
                             "logits exported every token",
                      "queue": {k: int(st1[k] - st0[k]) for k in st1 if k != "wait_ns"},
                      "split": split,
                      "note": "best of 3 regions; `queue`: Tensor calls recorded / run one launch at a time / tokens served by the fused step"}
        try:
            pdev = ca.HipTensorDevice(local, False, 0, False, "per-op")
            if model is not None:
                pconf, pweights = synth.to_hip(model, pdev)
                ptrait = ca.Llama2Runner(pconf, pweights, pdev, 64, True)
                p_tok = int(ptrait.timed_decode(1, 2)[0][-1])
                pdev.sync()
                tt = time.perf_counter()
                ptrait.timed_decode(p_tok, 8)
                pdev.sync()
                trait_info["per_op_launches_tokens_per_s"] = round(8 / (time.perf_counter() - tt), 2)
                del ptrait, pweights
            del pdev
        except Exception as e:
            trait_info["per_op_launches_tokens_per_s"] = repr(e)

    # ---- batched prefill of a 512-token prompt (crabml_hip_llama_prefill), reported next to the decode number ------
    prefill = None
    if rank == 0 and path == "fused" and not args.no_prefill:
        try:
            n_p = 512
            pr = ca.HipLlamaRunner(conf, weights, dev, n_p + 8, True)
            toks = [(7 * i + 1) % shape.vocab for i in range(n_p)]
            best = None
            for _ in range(2):
                pr.reset()
                dev.sync()
                tp0 = time.perf_counter()
                pr.prefill(toks)  # blocks: returns the last token's logits
                dtp = time.perf_counter() - tp0
                best = dtp if best is None else min(best, dtp)
            prefill = {"prompt_tokens": n_p, "rows_per_pass": 512, "ms": round(best * 1e3, 2),
                       "prompt_tokens_per_s": round(n_p / best, 1),
                       "vs_token_loop": round(n_p / best / (total_tokens / elapsed_max / args.gpus), 2),
                       "note": "llama2.rs:124-129 token loop as (rows, k) passes: weight GEMMs weight-stationary on the f16 matrix cores "
                               "(gemm_f16w.hip, a stated deviation of the fast tier; matmul_vec itself and the strict device keep the bit-exact "
                               "int8 MFMA GEMM) + causal flash attention on the f16 matrix cores; host clock around the blocking call, best of 2"}
            del pr
            # a long prompt: attention is O(n^2) there (the fast step runs it on the f16 matrix cores, k_attn_flash_rows)
            n_l = 4096
            if n_l + 8 <= shape.seq_len:
                pl = ca.HipLlamaRunner(conf, weights, dev, n_l + 8, True)
                ltoks = [(7 * i + 1) % shape.vocab for i in range(n_l)]
                lbest = None
                for _ in range(2):
                    pl.reset()
                    dev.sync()
                    tp0 = time.perf_counter()
                    pl.prefill(ltoks)
                    dtp = time.perf_counter() - tp0
                    lbest = dtp if lbest is None else min(lbest, dtp)
                prefill["long_prompt"] = {"prompt_tokens": n_l, "ms": round(lbest * 1e3, 2), "prompt_tokens_per_s": round(n_l / lbest, 1)}
                del pl
        except Exception as e:
            prefill = {"error": repr(e)}

    # ---- decode at longer contexts: the prompt is prefilled (batched pass), then 32 greedy steps are timed ------------------
    context = None
    if rank == 0 and path == "fused" and not args.no_context and not args.gguf:
        context = {}
        for ctx_len in (256, 1024, 4096):
            try:
                if ctx_len + 64 > shape.seq_len:
                    continue
                cr = ca.HipLlamaRunner(conf, weights, dev, ctx_len + 64, True)
                cr.prefill([(7 * i + 1) % shape.vocab for i in range(ctx_len)])
                ctok = int(cr.decode_greedy(1, 4)[-1])
                dev.sync()
                tc = time.perf_counter()
                cr.decode_greedy(ctok, 32)
                dev.sync()
                context[str(ctx_len)] = round(32 / (time.perf_counter() - tc), 2)
                del cr
            except Exception as e:
                context[str(ctx_len)] = repr(e)

    out = None
    if rank == 0:
        tps = total_tokens / elapsed_max
        out = {
            "metric": f"decode tokens/sec (batch-1 greedy), {shape.name} shape {args.wtype}",
            "value": round(tps, 2), "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "q4_0 x q8_0 -> i32 dot, f32 accumulate" if args.wtype == "Q4_0" else args.wtype,
            "data": "synthetic",
            "config": {
                "workload": f"{shape.name}-shape all-{args.wtype} synthetic GGUF-layout weights, batch-1 greedy decode, "
                            f"f16 KV cache, positions {args.warmup}..{args.warmup + args.steps - 1}",
                "path": ("the reference's API: Llama2Runner<HipTensor>::forward + host arg-max per token, unchanged (crabml-llama2/src/llama2.rs:184-211; "
                         "what crabml-cli -D hip runs) -> recorded Tensor calls -> the fused decode step" if headline == "reference" else
                         "crabml_hip_llama_decode_greedy (the fused decode step from its hipGraph, arg-max on the device)"),
                "n_layers": conf.n_layers, "replicas": args.gpus,
                "gemv_weight_bytes_per_token": gemv_bytes,
            },
            "hbm_roofline_tokens_per_s": round(HBM_PEAK_GBS * 1e9 / gemv_bytes, 1),
            "frac_of_hbm_roofline_tokens": round(tps / args.gpus / (HBM_PEAK_GBS * 1e9 / gemv_bytes), 4),
            "effective_weight_GBps_per_gpu": round(tps / args.gpus * gemv_bytes / 1e9, 1),
            "setup_s": round(t_build, 1), "upload_s": round(t_upload, 2),
            "timed_regions": {"repeats": len(regions), "steps_each": args.steps,
                              "tokens_per_s": [round(u / t, 2) for t, u in regions][::-1],  # fastest first
                              "median": round(tps, 2),
                              "p10": round(regions[min(len(regions) - 1, int(0.9 * len(regions)))][1] / regions[min(len(regions) - 1, int(0.9 * len(regions)))][0], 2),
                              "p90": round(regions[int(0.1 * len(regions))][1] / regions[int(0.1 * len(regions))][0], 2),
                              "note": "every region = the same W warm-up + K timed steps from an empty KV cache; value = the median region "
                                      "(reference API: a fresh runner per region; its first warm-up token is the one the decode context is learned from)"},
        }
        if context:
            out["context"] = {"tokens_per_s_at_position": context,
                              "note": "prompt of that length prefilled in batched passes, then 32 timed greedy steps"}
        if c3:
            out["c3_positions_0_127"] = c3
        out["config"]["host_affinity"] = host_affinity
        out["value_api"] = ("crabml_hip_llama_decode_greedy: the fused step from its hipGraph, arg-max on the device, one blocking call for K tokens"
                            if headline == "fused" else
                            "the reference's own API: Llama2Runner<HipTensor>::forward + host-side greedy sampler per token, unchanged "
                            "(logits exported every token); its recorded Tensor calls are served by the fused decode step (csrc/lazy.hpp)")
        if headline == "reference":
            out["value_through_reference_api"] = round(tps, 2)  # (= value)
        if fused_regions:
            fe, fu = fused_regions[len(fused_regions) // 2]
            out["fused_entry_point"] = {"tokens_per_s": round(fu / fe, 2), "ms_per_step": round(fe / args.steps * 1e3, 4),
                                        "regions_tokens_per_s": [round(u / t, 2) for t, u in fused_regions][::-1],
                                        "api": "crabml_hip_llama_decode_greedy: the same step from its hipGraph, arg-max on the device, one blocking call "
                                               "for K tokens (the headline of rounds 1-4; same positions, same region protocol)"}
        if trait_tps is not None:
            out["trait_path_tokens_per_s"] = round(trait_tps, 2)
            out["trait_path"] = trait_info
        if args.layers is not None:
            out["INVALID"] = "layer count truncated with --layers (debug run)"
        if roof:
            out["roofline"] = roof
        if prefill:
            out["prefill"] = prefill
        if not args.no_gemv_points and args.gpus == 1 and args.wtype in ("Q4_0", "Q8_0", "Q4_1"):
            try:
                out["gemv_points"] = gemv_points(ca, synth, dev, args.wtype)
            except Exception as e:
                out["gemv_points"] = {"error": repr(e)}
        if args.gguf:
            out["data"] = "file: " + args.gguf
            out["config"]["workload"] = f"GGUF file {os.path.basename(args.gguf)} ({args.wtype} body), batch-1 greedy decode, f16 KV cache"
        if not args.no_parity_check and args.gpus == 1 and model is not None and path == "fused":
            try:
                out["parity_check"] = parity_check(ca, synth, model, conf, weights, dev, local)
                if "strict_tokens_per_s" in out["parity_check"]:  # the bit-identical tier as a first-class figure
                    out["value_strict"] = out["parity_check"]["strict_tokens_per_s"]
                    out["value_strict_through_reference_api"] = out["parity_check"].get("strict_reference_api_tokens_per_s")
            except Exception as e:
                out["parity_check"] = {"error": repr(e)}
        if not args.no_cpu_baseline and args.gpus == 1 and model is not None:
            try:
                out["cpu_baseline"] = cpu_baseline(model, args.cpu_seconds)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out), flush=True)
    dist.close()


if __name__ == "__main__":
    main()
