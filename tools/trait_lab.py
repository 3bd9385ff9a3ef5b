"""Where the time of the queue-served trait path goes (8B shape): the fused step from its hipGraph, the same step launched
eagerly, and the reference's unchanged runner through the recorded-op queue (host blocked in export, host arg-max)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama3-8b")
ap.add_argument("--wtype", default="Q4_0")
ap.add_argument("--steps", type=int, default=48)
ap.add_argument("--layers", type=int, default=None)
ap.add_argument("--strict", action="store_true")
args = ap.parse_args()

dev = ca.HipTensorDevice(0, False, 0, args.strict)
if not os.environ.get("NO_PIN"):
    print(ca.pin_host_to_device_node(dev), file=sys.stderr)
model = synth.build_model(synth.SHAPES[args.model], synth.TYPE_BY_NAME[args.wtype], seed=8, n_layers=args.layers)
conf, w = synth.to_hip(model, dev)
out = {"model": args.model, "wtype": args.wtype, "steps": args.steps, "strict": args.strict}
n = args.steps
for name, graph in (("fused_graph", True), ("fused_eager", False)):
    f = ca.HipLlamaRunner(conf, w, dev, 256, True, graph)
    tok = int(f.decode_greedy(1, 8)[-1])
    dev.sync()
    best = None
    for _ in range(3):
        f.reset()
        f.decode_greedy(1, 8)
        dev.sync()
        t0 = time.perf_counter()
        f.decode_greedy(tok, n)
        dev.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out[name] = {"tok_s": round(n / best, 1), "ms": round(best / n * 1e3, 4)}
    del f
r = ca.Llama2Runner(conf, w, dev, 8 + 3 * n + 8, True)
tok = int(r.timed_decode(1, 8)[0][-1])
best = None
for _ in range(3):
    sa = dev.lazy_stats()
    t0 = time.perf_counter()
    ids, sec, samp = r.timed_decode(tok, n)
    dt = time.perf_counter() - t0
    sb = dev.lazy_stats()
    tok = int(ids[-1])
    if best is None or dt < best[0]:
        best = (dt, (sb["wait_ns"] - sa["wait_ns"]) / n * 1e-6, samp / n * 1e3, sb["pinned_exports"] - sa["pinned_exports"],
                sb["fused_tokens"] - sa["fused_tokens"])
out["trait_queue"] = {"tok_s": round(n / best[0], 1), "ms": round(best[0] / n * 1e3, 4), "blocked_in_export_ms": round(best[1], 4),
                      "host_argmax_ms": round(best[2], 4), "pinned_exports": best[3], "fused_tokens": best[4]}
print(json.dumps(out))
