"""strive_mlp_fwd on the matrix cores (fp16 x 3 fragments) against the same call with the fragments withheld (fp32 VALU)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from util import product_model
from strive_amd import ops, params, _lib as L
dev = torch.device('cuda:0')
m, sd = product_model(device=dev)
lib = L.get_lib()
mod = m.prior_net.mlp_in if hasattr(m.prior_net, 'mlp_in') else None
name = 'prior_net.mlp_in'
pk = params.pack_mlp({k: v for k, v in m.state_dict().items()}, name)
F, O = pk.struct.dims[0], pk.struct.dims[pk.struct.nlayers]
print('MLP', name, [pk.struct.dims[i] for i in range(pk.struct.nlayers + 1)])
for rows in (8, 512, 4096):
    x = torch.randn((rows, F), device=dev)
    y = [None, None]
    for variant in (0, 1):
        saved = [pk.struct.wf[i] for i in range(L.MAXL)]
        if variant == 0:
            for i in range(L.MAXL):
                pk.struct.wf[i] = None
        out = torch.empty((rows, O), device=dev)
        call = lambda: lib.call('strive_mlp_fwd', pk.ref(), L.ptr(x), rows, L.ptr(out), L.stream_ptr(x))
        for _ in range(5):
            call()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record(); torch.cuda.synchronize()
        y[variant] = out.clone()
        print('  rows %5d  %-22s %7.2f us' % (rows, 'VALU fp32' if variant == 0 else 'MFMA fp16x3', e0.elapsed_time(e1) * 1e3 / 50))
        for i in range(L.MAXL):
            pk.struct.wf[i] = saved[i]
    print('     max |diff| %.3g  (max |y| %.3g)' % (float((y[0] - y[1]).abs().max()), float(y[0].abs().max())))
