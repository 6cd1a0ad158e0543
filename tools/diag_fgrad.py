import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_steps_gpu as T
dev = torch.device('cuda:0')
orig = T._grads_close
def show(named_got, ref64, ref32_list, names, tag=''):
    gmax = max(float(ref64[k].grad.abs().max()) for k in names if ref64[k].grad is not None)
    for k in names:
        if ref64[k].grad is None: continue
        e = float(T._rel_err(named_got[k], ref64[k].grad, gmax).max())
        n = [float(T._rel_err(r[k].grad, ref64[k].grad, gmax).max()) for r in ref32_list]
        own = float(ref64[k].grad.abs().max()) / gmax
        print('%-55s own/gmax %.1e  hip %.2e  fp32 %s' % (k, own, e, ' '.join('%.1e' % v for v in n)))
T._grads_close = show
T.test_feature_gen_gradients_vs_oracle(dev, (4, 'batch_norm'))
