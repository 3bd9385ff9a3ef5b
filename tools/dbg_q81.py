import sys, numpy as np
sys.path.insert(0, '.')
import crabml_amd as ca
from oracle import oracle as o
sys.path.insert(0, 'tests')
from tests.test_hip_quantize import cases
dev = ca.HipTensorDevice()
n = 256
for name, x in cases(n, n + 4):
    got = ca.HipTensor.new(x, [n], dev).debug_quantize(ca.GGMLType.Q8_1).reshape(-1, 36)
    ref = o.quantize(x, o.Q8_1).reshape(-1, 36)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    print(name, 'bad blocks', bad[:5], 'of', len(ref))
    for b in bad[:2]:
        print('  got d,s', got[b, :4].view(np.float16), 'ref', ref[b, :4].view(np.float16))
        dq = np.nonzero(got[b, 4:] != ref[b, 4:])[0]
        print('  q diff idx', dq[:8], got[b, 4:].view(np.int8)[dq[:8]], ref[b, 4:].view(np.int8)[dq[:8]], 'x', x[b*32:(b+1)*32][dq[:8]])
