"""Measure HIP-vs-reference gradient / Adam-step errors on the golden fixtures (to set test tolerances)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from test_gpu_e2e import build_model, load_batch
for tag in ('train_32x48_b2', 'train_64x96_b1', 'train_128x128_b2'):
    g = np.load('tests/golden/e2e_%s.npz' % tag)
    model = build_model(True)
    res = model.train_step(load_batch(g))
    pd = dict(model.named_parameters())
    print(tag, 'loss rel', abs(float(res['final_loss']) - float(g['final_loss'])) / abs(float(g['final_loss'])))
    for k in g.files:
        if k.startswith('grad::'):
            n = k[6:]
            ref = torch.from_numpy(g[k]).double()
            mine = pd[n].grad.detach().cpu().double()
            print('  %-60s relL2 %.2e  |ref| %.2e' % (n, ((mine - ref).norm() / ref.norm()).item(), ref.norm().item()))
    names = [str(s) for s in g['grad_names']]
    cs = g['grad_cs']
    worst = []
    for n, c in zip(names, cs):
        if n not in pd or pd[n].grad is None:
            continue
        t = pd[n].grad.detach().double().cpu()
        mine = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])
        rel = abs(mine[1] - c[1]) / max(c[1], 1e-30)
        rel2 = abs(mine[2] - c[2]) / max(c[2], 1e-30)
        worst.append((max(rel, rel2), n, c[1]))
    worst.sort(reverse=True)
    print('  checksum worst:', [(round(w, 5), n) for w, n, _ in worst[:6]])
    print('  checksum median %.2e  n=%d' % (worst[len(worst) // 2][0], len(worst)))
    # Adam-step: parameter checksums after one step
    sd = model.state_dict()
    pn = [str(s) for s in g['post_names']]
    pc = g['post_cs']
    errs = []
    for n, c in zip(pn, pc):
        if n not in sd or not torch.is_floating_point(sd[n]):
            continue
        t = sd[n].detach().double().cpu()
        mine = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])
        errs.append((abs(mine[0] - c[0]) / max(t.numel(), 1), abs(mine[1] - c[1]) / max(t.numel(), 1), n))
    errs.sort(reverse=True)
    print('  post-Adam per-element |d sum|/numel worst:', [(float('%.2e' % a), float('%.2e' % b), n) for a, b, n in errs[:5]])
