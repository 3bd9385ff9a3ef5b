"""Run-to-run / placement variance of the queue-served trait path: per-region tokens/s, the CPU the thread runs on."""
import ctypes
import os
import sys
import time

_libc = ctypes.CDLL("libc.so.6")


def getcpu():
    return _libc.sched_getcpu()

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crabml_amd as ca  # noqa: E402
from crabml_amd import synth  # noqa: E402

dev = ca.HipTensorDevice(0)
if not os.environ.get("NO_PIN"):
    print(ca.pin_host_to_device_node(dev))
model = synth.build_model(synth.SHAPES["llama3-8b"], synth.Q4_0, seed=8)
conf, w = synth.to_hip(model, dev)
n = 48
r = ca.Llama2Runner(conf, w, dev, 8 + 10 * n + 8, True)
tok = int(r.timed_decode(1, 8)[0][-1])
out = []
for i in range(8):
    cpu0 = getcpu()
    sa = dev.lazy_stats()
    t0 = time.perf_counter()
    ids, sec, samp = r.timed_decode(tok, n)
    dt = time.perf_counter() - t0
    sb = dev.lazy_stats()
    tok = int(ids[-1])
    out.append((round(n / dt, 1), round((sb["wait_ns"] - sa["wait_ns"]) / n * 1e-6, 3), cpu0, getcpu()))
print("affinity", len(os.sched_getaffinity(0)), "regions (tok/s, blocked ms, cpu before, cpu after):", out)
