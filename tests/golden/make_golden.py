"""Generate golden input/output vectors by RUNNING THE REFERENCE'S OWN PYTHON (CPU).

Run in the build container only (needs /root/reference; it does not exist on the GPU box):
    python tests/golden/make_golden.py
Writes small .npz fixtures (data only: seeded inputs + the reference's outputs) next to this file.
Nothing of the reference's source text is stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def labelmap():
    """draw_labelmap_ori (lib/datasets/util.py:326-363) as the animal `_mt` datasets call it (real_animal_all_mt.py:274-283): for every
    (sigma, type, map size) a set of seeded key points - interior, borders on both sides of the whole-patch-inside rule, negative
    fractions that int32 truncation pulls to 0 - through the reference's own function -> labelmap.npz."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    ref_dutil = load_by_path("ref_dutil", "lib/datasets/util.py")
    rs = np.random.RandomState(5)
    out, cases = {}, []
    for ci, (sigma, typ, res) in enumerate([(1.0, "Gaussian", 64), (2, "Gaussian", 64), (1.0, "Cauchy", 96), (1.0, "Gaussian", 96), (1.5, "Gaussian", 64),
                                            (2, "Cauchy", 64)]):
        K = 40
        pts = rs.uniform(-6, res + 6, size=(K, 3)).astype(np.float32)
        r = int(3 * sigma)
        edge = [(r + 1, r + 1), (r + 0.999, r + 1.5), (r + 1.0, res - r - 1.0), (res - r - 1.0, r + 1), (res - r - 0.001, r + 1), (res - r, res - r),
                (0.5, 10), (r + 1, 0.2), (res - r - 1 + 0.9999, res - r - 1 + 0.5), (res / 2, res / 2)]
        for k, (x, y) in enumerate(edge):
            pts[k, 0], pts[k, 1] = x, y                      # (1-based: the loop below subtracts 1 as the datasets do)
        pts[:, 2] = (rs.rand(K) > 0.2).astype(np.float32)
        gate = rs.rand(K) > 0.15                             # `tpts[i, 1] > 0` before the transform
        target = torch.zeros(K, res, res)
        weight = torch.from_numpy(pts[:, 2].copy()).view(K, 1)
        tp = torch.from_numpy(pts.copy())
        for i in range(K):
            if gate[i]:
                target[i], vis = ref_dutil.draw_labelmap_ori(target[i], tp[i] - 1, sigma, type=typ)
                weight[i, 0] *= vis
        cases.append((float(sigma), typ, res))
        out[f"pts{ci}"], out[f"gate{ci}"], out[f"target{ci}"], out[f"weight{ci}"] = pts, gate, target.numpy(), weight.numpy()
    out["sigmas"] = np.array([c[0] for c in cases])
    out["types"] = np.array([c[1] for c in cases])
    out["sizes"] = np.array([c[2] for c in cases])
    np.savez_compressed(os.path.join(OUT, "labelmap.npz"), **out)
    drawn = [int((out[f"target{ci}"].reshape(40, -1).max(1) > 0).sum()) for ci in range(len(cases))]
    print("labelmap.npz written; stamps drawn per case:", drawn)


def main():
    sys.path.insert(0, REF)
    import utils as ref_utils                      # reference utils.py
    from lib import keypoint_detection as ref_kd   # reference lib/keypoint_detection.py
    ref_loss = load_by_path("ref_loss", "lib/models/loss.py")
    ref_style = load_by_path("ref_style", "lib/models/Style_net.py")
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    ref_dutil = load_by_path("ref_dutil", "lib/datasets/util.py")

    g = torch.Generator().manual_seed(0)
    B, K, H, W = 4, 16, 64, 64

    # ---- losses (A5, A6)
    pred = torch.randn(B, K, H, W, generator=g)
    gt = torch.rand(B, K, H, W, generator=g)
    w = (torch.rand(B, K, 1, generator=g) > 0.3).float()
    mask = torch.rand(B, K, generator=g) > 0.5
    valid = torch.rand(B, H, W, generator=torch.Generator().manual_seed(123)) > 0.4      # (own generator: later draws unchanged)
    np.savez_compressed(
        os.path.join(OUT, "losses.npz"), pred=pred.numpy(), gt=gt.numpy(), w=w.numpy(), mask=mask.numpy(), valid=valid.numpy(),
        cons_valid=ref_loss.ConsLoss()(pred, gt, valid_mask=valid, tea_mask=mask).numpy(),
        cons_valid_only=ref_loss.ConsLoss()(pred, gt, valid_mask=valid).numpy(),
        mse_mean=ref_loss.JointsMSELoss()(pred, gt, w).numpy(),
        mse_mean_now=ref_loss.JointsMSELoss()(pred, gt).numpy(),
        mse_none=ref_loss.JointsMSELoss(reduction="none")(pred, gt, w).numpy(),
        cons_masked=ref_loss.ConsLoss()(pred, gt, tea_mask=mask).numpy(),
        cons_plain=ref_loss.ConsLoss()(pred, gt).numpy())

    # ---- decode / PCK / rectify (A7, A8)
    rs = np.random.RandomState(0)
    kp = rs.uniform(-8, 264, size=(B, K, 2)).astype(np.float32)
    labels = np.stack([ref_dutil.generate_target(kp[b], np.ones((K, 1), np.float32), (W, H), 2, (256, 256))[0]
                       for b in range(B)])
    weights = np.stack([ref_dutil.generate_target(kp[b], np.ones((K, 1), np.float32), (W, H), 2, (256, 256))[1]
                        for b in range(B)])
    noisy = (torch.from_numpy(labels) * 0.8 + 0.15 * torch.randn(B, K, H, W, generator=g)).numpy()
    noisy[0, 0] = -1.0                      # all-negative channel -> coords (0,0)
    noisy[0, 1] = 0.0
    noisy[0, 1, 1, 2] = 3.0
    noisy[0, 1, 3, 3] = 3.0                # tie -> first flat index
    noisy[1, 2, 0, 63] = 9.0               # corner peak (clipped patch)
    p_np, v_np = ref_kd.get_max_preds(noisy)
    acc, avg, cnt, pred_kp = ref_kd.accuracy(noisy, labels)
    p_t, v_t = ref_utils.get_max_preds_torch(torch.from_numpy(noisy))
    rec2 = ref_utils.rectify(torch.from_numpy(noisy), 2)
    rec1 = ref_utils.rectify(torch.from_numpy(noisy), 1.0)
    np.savez_compressed(
        os.path.join(OUT, "decode.npz"), kp=kp, labels=labels, weights=weights, noisy=noisy,
        preds_np=p_np, maxv_np=v_np, acc=acc, avg_acc=np.float64(avg), cnt=np.int64(cnt), pred_kp=pred_kp,
        preds_t=p_t.numpy(), maxv_t=v_t.numpy(), rect_s2=rec2.numpy(), rect_s1=rec1.numpy())

    # ---- EMA (A9)
    src = [torch.randn(7, 5, generator=g), torch.randn(33, generator=g), torch.randn(2, 3, 4, 4, generator=g)]
    tgt = [torch.randn_like(s) for s in src]

    class Holder:
        def __init__(self, ps):
            self.ps = [torch.nn.Parameter(p.clone()) for p in ps]

        def parameters(self):
            return iter(self.ps)

    s_net, t_net = Holder(src), Holder(tgt)
    ema = ref_utils.OldWeightEMA(t_net, s_net, alpha=0.999)
    after_init = [p.detach().clone().numpy() for p in t_net.ps]
    trace = []
    for it in range(3):
        for p in s_net.ps:
            p.data.add_(0.01 * torch.randn(p.shape, generator=g))
        trace.append([p.detach().clone().numpy() for p in s_net.ps])
        ema.step()
    np.savez_compressed(
        os.path.join(OUT, "ema.npz"),
        **{f"src{i}": s.numpy() for i, s in enumerate(src)},
        **{f"init{i}": a for i, a in enumerate(after_init)},
        **{f"stu_it{it}_{i}": trace[it][i] for it in range(3) for i in range(3)},
        **{f"final{i}": p.detach().numpy() for i, p in enumerate(t_net.ps)})

    # ---- AdaIN statistics + whole style net at reduced size (A10, A11, A12), seeded random weights
    c = torch.randn(2, 512, 8, 8, generator=g) * 2 + 0.5
    s = torch.randn(2, 512, 8, 8, generator=g) * 0.7 - 1
    m, sd = ref_style.calc_mean_std(c)
    ad = ref_style.adain(c, s)
    sys.path.insert(0, OUT)
    from seeded import fill_style_weights
    fill_style_weights(ref_style.vgg, 11)
    fill_style_weights(ref_style.decoder, 12)
    vgg31 = torch.nn.Sequential(*list(ref_style.vgg.children())[:31])
    net = ref_style.Net(vgg31, ref_style.decoder).eval()
    content = torch.randn(2, 3, 64, 64, generator=g)
    style = torch.randn(2, 3, 64, 64, generator=g) * 1.3 + 0.2
    with torch.no_grad():
        lc, ls, g_t = net(content, style, 0.6)
        feat = vgg31(content)
    np.savez_compressed(
        os.path.join(OUT, "style.npz"), c=c.numpy(), s=s.numpy(), mean=m.numpy(), std=sd.numpy(), adain=ad.numpy(),
        content=content.numpy(), style=style.numpy(), alpha=np.float64(0.6), g_t=g_t.numpy(), feat=feat.numpy(),
        loss_c=lc.numpy(), loss_s=ls.numpy(), gram=ref_style.gram_matrix(feat).numpy())

    # ---- reference-owned Upsampling + head wrapper (A3, A4) under a throw-away torchvision stand-in:
    # only the reference's own pose_resnet.py code (deconv stack, head, init, state_dict names) is exercised.
    import torch.nn as nn
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvu = types.ModuleType("torchvision.models.utils")
    tvr = types.ModuleType("torchvision.models.resnet")

    class _RN(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.fc = nn.Linear(2048, 1000)

    tvm.ResNet = _RN
    tvu.load_state_dict_from_url = lambda *a, **k: {}
    tvr.BasicBlock = tvr.Bottleneck = object
    tvr.model_urls = {}
    tv.models = tvm
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvu,
                        "torchvision.models.resnet": tvr})
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [os.path.join(REF, "lib/models")]
    sys.modules["refmodels"] = pkg
    spec = importlib.util.spec_from_file_location("refmodels.resnet", os.path.join(REF, "lib/models/resnet.py"))
    rmod = importlib.util.module_from_spec(spec); sys.modules["refmodels.resnet"] = rmod; spec.loader.exec_module(rmod)
    spec = importlib.util.spec_from_file_location("refmodels.pose_resnet", os.path.join(REF, "lib/models/pose_resnet.py"))
    pmod = importlib.util.module_from_spec(spec); sys.modules["refmodels.pose_resnet"] = pmod; spec.loader.exec_module(pmod)
    torch.manual_seed(2)
    up = pmod.Upsampling(64, hidden_dims=(32, 32, 32))
    head = nn.Conv2d(32, 5, 1)
    x = torch.randn(2, 64, 4, 4, generator=g)
    up.train()
    y = head(up(x))
    np.savez_compressed(
        os.path.join(OUT, "upsampling.npz"), x=x.numpy(), y=y.detach().numpy(),
        **{f"up_{k}": v.numpy() for k, v in up.state_dict().items()},
        head_w=head.weight.detach().numpy(), head_b=head.bias.detach().numpy())
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "labelmap":      # (round 4: only the new fixture; the others stay byte for byte)
        labelmap()
    else:
        main()
        labelmap()
