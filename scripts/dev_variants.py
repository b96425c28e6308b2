"""Developer timing: kernel variants on a random 17-species network (9 row classes)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'test_gpu_user_models.py')).read()
ns = {}
exec("import numpy as np\n" + src[src.index("def _random_network"):src.index("@pytest.mark.parametrize('seed,n'")], ns)
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd import _lib
for seed, n in ((3, 17), (2, 11)):
    gm = GeneratedModel(ns['_random_network'](seed, n))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    dm = m.device_model
    V = 4096
    rng = np.random.default_rng(1)
    P = torch.from_numpy(np.exp(rng.uniform(np.log(0.2), np.log(2.0), (V, len(gm.param_order))))).cuda()
    t = torch.linspace(1.0, 20.0, 8, dtype=torch.float64).cuda()
    Y = torch.empty((V, 8, n), dtype=torch.float64, device='cuda')
    S = torch.empty((V, 8, n, gm.n_sens), dtype=torch.float64, device='cuda')
    ns_ = torch.empty(V, dtype=torch.int32, device='cuda')
    for var in ('auto', 'per_wave', 'row_lane', 'row_group'):
        o = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=var)
        dm.sens_dev(P, t, None, o, Y, S, None, ns_, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            dm.sens_dev(P, t, None, o, Y, S, None, ns_, None)
        b.record(); torch.cuda.synchronize()
        print(gm.spec.name, "%-10s %.3f ms  steps %d" % (var, a.elapsed_time(b) / 3, int(ns_.sum())), flush=True)
