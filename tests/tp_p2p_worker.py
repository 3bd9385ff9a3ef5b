"""One rank of a tensor-parallel group in its own PROCESS (tests/test_hip_tp_p2p.py launches world of these on one GPU).
The ranks find each other through a shared directory (tp.file_all_gather: the 64-byte inbox handles), connect the one-shot
P2P all-reduce group over hipIpc, decode the given tokens and write their logits; the parent compares them with the
oracle's tensor-parallel restatement.  usage: tp_p2p_worker.py DIR RANK WORLD SHAPE FMT STRICT(0|1) MODE(step|allreduce)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import crabml_amd as ca  # noqa: E402
from crabml_amd import synth, tp as tp_mod  # noqa: E402

TOKS = [1, 365, 400, 282, 7, 9]
synth.SHAPES["tp4"] = synth.ModelShape("tp4", 512, 1024, 2, 8, 4, 1024, 64, 1e-5, None)  # tiny-gqa with 4 kv heads
synth.SHAPES["tp8"] = synth.ModelShape("tp8", 512, 2048, 2, 8, 8, 1024, 64, 1e-5, None)  # 8 kv heads, 256 hidden columns per rank
# BASELINE config C5 at its own widths (dim 8192, hidden 28672, 64 / 8 heads, vocab 128256), two layers: 1024 / 3584 columns per rank at tp = 8
synth.SHAPES["c5-2l"] = synth.ModelShape("Llama-3-70B (2 layers)", 8192, 28672, 2, 64, 8, 128256, 64, 1e-5, None)
C5_TOKS = [1, 365, 9906]


def main():
    d, rank, world, shape, fmt, strict, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6] == "1", sys.argv[7]
    dev = ca.HipTensorDevice(0, False, 0, strict)
    model = synth.build_model(synth.SHAPES[shape], synth.TYPE_BY_NAME[fmt], seed=31)
    toks = C5_TOKS if shape == "c5-2l" else TOKS
    comm = tp_mod.init_tp_p2p(dev, rank, world, model.shape.dim, tp_mod.file_all_gather(d, rank, world))
    if mode == "allreduce":
        out = []
        for it in range(5):  # consecutive collectives alternate the two inbox slots
            x = (np.arange(model.shape.dim, dtype=np.float32) * (rank + 1) + it).astype(np.float32)
            t = ca.HipTensor.from_cpu(x.view(np.uint8), [model.shape.dim], ca.GGMLType.F32, dev)
            comm.all_reduce(t)
            out.append(t.export())
        np.save(os.path.join(d, f"out.{rank}.npy"), np.stack(out))
        return
    conf, w = synth.to_hip(tp_mod.shard_model(model, world, rank, True), dev)
    r = ca.HipLlamaRunner(conf, w, dev, 64, True, True, True, world, rank, comm)
    logits = [r.forward(t, i).copy() for i, t in enumerate(toks)]
    ids = r.decode_greedy(int(np.argmax(logits[-1])), 3)
    np.save(os.path.join(d, f"out.{rank}.npy"), np.stack(logits))
    np.save(os.path.join(d, f"ids.{rank}.npy"), np.asarray(ids))
    del r, comm


if __name__ == "__main__":
    main()
