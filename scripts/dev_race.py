import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import _lib, models_zoo, build
from sysbio_modeling_amd.symbolic import zoo_model
ctx = _lib.default_context()
gm = zoo_model('cascade20')
hdr = os.path.join(build.MODELS_DIR, 'cascade20.hpp')
plugins = {'nofence': gm.plugin_path(), 'barrier': build.build_plugin('cascade20_barrier', hdr, extra_flags=('-DSBM_RL_BARRIER=1',))}
for V in (64, 256, 1024, 4096):
    _, P = models_zoo.cascade_ensemble(V)
    for tname, t in (('nice', np.concatenate([[0.0], models_zoo.CASCADE_MEASURE_TIMES])),
                     ('grid', np.concatenate([[0.0], np.linspace(0,100,1000)[np.searchsorted(np.linspace(0,100,1000), models_zoo.CASCADE_MEASURE_TIMES)]]))):
        for name, path in plugins.items():
            dm = _lib.LoadedModel(ctx, path)
            Pd, td = torch.from_numpy(P).cuda(), torch.from_numpy(t).cuda()
            for variant in ('row_lane', 'per_wave'):
                bad = []
                for rep in range(3):
                    Y = torch.full((V, len(t), 20), float('nan'), dtype=torch.float64, device='cuda')
                    S = torch.full((V, len(t), 20, 40), float('nan'), dtype=torch.float64, device='cuda')
                    st = torch.full((V,), -1, dtype=torch.int32, device='cuda'); ns = torch.zeros_like(st)
                    dm.sens_dev(Pd, td, None, _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=variant), Y, S, st, ns, None)
                    torch.cuda.synchronize()
                    bad.append(int((st != 0).sum()))
                print("V=%5d t=%s plugin=%-8s variant=%-9s bad per rep %s  steps[0]=%d" % (V, tname, name, variant, bad, int(ns[0])), flush=True)
