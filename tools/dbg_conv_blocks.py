import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from diffdock_pocket_amd import packing as P
from diffdock_pocket_amd.score_model import TensorProductConvLayer
from oracle import thirdparty as tp
from oracle.ref_model import OracleConfig, OracleScoreModel
dev = torch.device("cuda:0")
for ns, nv, layer, E, N in [(60, 10, 3, 200, 23), (60, 10, 1, 129, 5), (60, 10, 2, 64, 9), (16, 4, 1, 63, 10), (24, 6, 3, 65, 7), (60, 10, 0, 129, 5)]:
    torch.manual_seed(ns + layer + E)
    mi, mo = P.irreps_muls(ns, nv, layer), P.irreps_muls(ns, nv, layer + 1)
    spec = P.faster_tp_spec(mi, mo, 3 * ns)
    sg = P.faster_tp_spec(mi, mo, 3 * ns, factorized=True)
    blocks = [(m, d, s) for m, d, s in ((mo[0], 1, True), (mo[1], 3, False), (mo[2], 3, False), (mo[3], 1, False)) if m]
    conv = TensorProductConvLayer(spec, blocks, spec_g=sg)
    with torch.no_grad():
        conv.batch_norm.running_mean.normal_(0, 0.2); conv.batch_norm.running_var.uniform_(0.5, 2)
        conv.batch_norm.weight.uniform_(0.5, 1.5); conv.batch_norm.bias.normal_(0, 0.2)
    x = torch.randn(N, P.irreps_dim(mi))
    ei = torch.stack([torch.randint(0, max(N - 1, 1), (E,)), torch.randint(0, N, (E,))])
    ea = torch.randn(E, 3 * ns)
    sh = tp.spherical_harmonics("1x0e+1x1o", torch.randn(E, 3))
    cfg = OracleConfig(ns=ns, nv=nv)
    sd = {"c." + k: v for k, v in conv.state_dict().items()}
    want = OracleScoreModel(cfg, sd)._conv("c", cfg.irreps(layer), cfg.irreps(layer + 1), x, ei, ea, sh)
    conv = conv.to(dev)
    got = conv(x.to(dev), ei.to(dev), ea.to(dev), sh.to(dev), factorized=True).cpu()
    got_d = conv(x.to(dev), ei.to(dev), ea.to(dev), sh.to(dev), factorized=False).cpu()
    offs = [0, mo[0], mo[0] + 3 * mo[1], mo[0] + 3 * mo[1] + 3 * mo[2], P.irreps_dim(mo)]
    scale = float(want.abs().max())
    print(f"ns={ns} nv={nv} layer={layer} E={E}: roles={sg.roles} nrounds={sg.nrounds}")
    for b in range(4):
        if offs[b + 1] > offs[b]:
            d = (got[:, offs[b]:offs[b + 1]] - want[:, offs[b]:offs[b + 1]]).abs()
            dd = (got_d[:, offs[b]:offs[b + 1]] - want[:, offs[b]:offs[b + 1]]).abs()
            col = d.max(0).values
            print(f"   block {b}: fact err {float(d.max()) / scale:.2e} (direct {float(dd.max()) / scale:.2e}); worst cols {torch.topk(col, min(5, col.numel())).indices.tolist()}  rows with err>1e-4: {int((d.max(1).values > 1e-4 * scale).sum())}/{d.shape[0]}")
